// BPR training kernels for gfx950 (CDNA4).  Wave = 64 lanes.
//
// Layout of one triplet (u, i, j) inside a wave: lane l holds elements 64*r + l of P[u], Q[i]
// and Q[j], so every load / store / atomic wave-instruction covers 256 contiguous bytes of one
// row -- the shape that runs at the full float-atomic and plain-store rates on MI355X.  The
// k-length dots are reduced with a 64-lane butterfly (DPP-fused adds), which is the summation
// order oracle/bpr_oracle.c:dot64 restates.
//
// Arithmetic follows recommender/cf/BPR.py:50-57 of the reference: margin in fp32, sigmoid in
// double on it (tool/qmath.py:115-116), coefficient rounded to fp32 once, every multiply and
// add rounded separately (the file is compiled with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace yue {

constexpr int kNllSlots = 1024;
constexpr int kMaxAttempts = 64;

struct TrainArgs {
    float *P, *Q, *dP, *dQ;
    float *stage;                // staged item-row differences of the running round: one k-float row per touch
#ifdef YUE_STAMPS
    unsigned long long *stamps;  // diagnostic build only (make stamps): 8 phase time stamps per update wave
#endif
    const int32_t *ev_u, *ev_i;
    int32_t *ev_j;
    const int64_t *indptr;
    const int32_t *indices;
    double *nll_slots;
    int64_t m, n;
    int k;
    float ru, ri;
    double lr;
    uint64_t seed;
    uint32_t epoch;
    int32_t neg_lo, neg_range;
};

// One round of the S-round schedule.
struct RoundArgs {
    int64_t e_begin, e_end;      // events updated by this launch
    int64_t n_begin, n_end;      // events of the NEXT round: negatives drawn and row touches counted here
    // Item-row touch counters, one 64-bit word per row: high half = touches in the round (fixed while
    // the round runs: decides in-place store vs atomic path), low half = touches not yet retired
    // (decremented by the last-arriver protocol).  A late wave must never see a decremented total.
    unsigned long long *cnt_cur;     // round [e_begin, e_end)   (filled by the previous launch)
    unsigned long long *cnt_next;    // round [n_begin, n_end)
    uint32_t *cntp_cur;          // user-row flushes (runs of equal users inside a wave's batch) in this round
    uint32_t *cntp_next;         // ... in the next round
    // Slot table, kStageMax words per item row: the staging slots (2*(event - first event of the round)
    // + 0 for the positive / 1 for the negative) of the row's first kStageMax touches, in ticket order.
    uint32_t *tab_cur, *tab_next;
    int staged;                  // 1: rows with 2..kStageMax touches go through the staging rows instead of float atomics
    int apply_p;                 // 1: user rows are finished in this launch; 0: dP is left for the all-reduce
    int debug_skip;              // TIMING EXPERIMENTS ONLY (results wrong): bit 0 = no retire phase, bit 1 = no tickets, bit 2 = no drain wait
};

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31; return z;
}

// Counter-based draw for (seed, epoch, event, attempt): same function as the oracle's ctr_draw.
__device__ __forceinline__ int32_t ctr_draw(uint64_t seed, uint32_t epoch, uint64_t e, uint32_t a, int32_t lo, int32_t n_range) {
    uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ull * (e + 1));
    z = mix64(z ^ (0xD1B54A32D192ED03ull * (uint64_t)(epoch + 1) + 0x8CB92BA72F3D8DD7ull * (uint64_t)a));
    return lo + (int32_t)(((z >> 32) * (uint64_t)(uint32_t)n_range) >> 32);
}

__device__ __forceinline__ bool csr_contains(const int64_t *indptr, const int32_t *indices, int64_t row, int32_t x) {
    int64_t lo = indptr[row], hi = indptr[row + 1];
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        int32_t v = indices[mid];
        if (v < x) lo = mid + 1; else if (v > x) hi = mid; else return true;
    }
    return false;
}

// BPR.py:46-48 with a counter-based stream: first attempt outside the user's listened row.
__device__ __forceinline__ int32_t sample_negative(const TrainArgs &a, int32_t u, int64_t e) {
    for (uint32_t t = 0; t < (uint32_t)kMaxAttempts; ++t) {
        int32_t c = ctr_draw(a.seed, a.epoch, (uint64_t)e, t, a.neg_lo, a.neg_range);
        if (!csr_contains(a.indptr, a.indices, u, c)) return c;
    }
    return -1;
}

__global__ void __launch_bounds__(256) k_sample(TrainArgs a, int64_t E) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) a.ev_j[e] = sample_negative(a, a.ev_u[e], e);
}

__device__ __forceinline__ float rdlane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes in the canonical order of oracle/bpr_oracle.c:dot64 -- butterfly with
// partner lane^1, ^2, ^4, ^8, ^16, then (lanes 0-31) + (lanes 32-63).  The first four steps are
// DPP-fused adds (quad_perm, quad_perm, row_half_mirror, row_mirror: same operands as the xor
// partners because the value is already constant over the smaller groups), xor 16 is a ds_swizzle.
// Returns the total as a wave-uniform value.
__device__ __forceinline__ float wave_sum(float v) {
    v = v + dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v = v + dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v = v + dpp_mov<0x141>(v);      // row_half_mirror
    v = v + dpp_mov<0x140>(v);      // row_mirror
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // swap 16
    return rdlane(v, 0) + rdlane(v, 32);
}

// Reference triplet update on one element (BPR.py:51-57), every product and sum rounded separately.
struct Elem { float p2, qi2, qj2; };
__device__ __forceinline__ Elem bpr_elem(float p, float qi, float qj, float c, float ru, float ri) {
    const float d = qi - qj;
    const float td = c * d;
    const float p1 = p + td;            // :51
    const float tq = c * p1;
    const float qi1 = qi + tq;          // :52 (updated P[u])
    const float qj1 = qj - tq;          // :53
    const float rp = ru * p1;
    const float ra = ri * qi1;
    const float rb = ri * qj1;
    Elem o;
    o.p2 = p1 - rp;                     // :55
    o.qi2 = qi1 - ra;                   // :56
    o.qj2 = qj1 - rb;                   // :57
    return o;
}

// ------------------------------------------------------------------------------------------
// Wave layout: lane l holds elements 64*r + l (r < KR) of P[u], Q[i] and Q[j]; every load, store
// and atomic wave-instruction covers 256 contiguous bytes of one row.
//
// Exact replay: one launch = one dependency level (pairwise row-disjoint triplets), rows are
// rewritten in place.  One wave walks `tpw` consecutive triplets of the level.
// ------------------------------------------------------------------------------------------
template <int KR>
__global__ void __launch_bounds__(256) k_bpr_level(TrainArgs a, int64_t e_begin, int64_t e_end, int tpw) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t base = e_begin + wave * tpw;
    if (base >= e_end) return;
    const int cnt = (int)((e_end - base) < (int64_t)tpw ? (e_end - base) : (int64_t)tpw);
    int u = 0, i = 0, j = 0;
    if (lane < cnt) { u = a.ev_u[base + lane]; i = a.ev_i[base + lane]; j = a.ev_j[base + lane]; }
    const int k = a.k;
    double nll = 0.0;
    for (int t = 0; t < cnt; ++t) {
        const int64_t tu = __builtin_amdgcn_readlane(u, t);
        const int64_t ti = __builtin_amdgcn_readlane(i, t);
        const int64_t tj = __builtin_amdgcn_readlane(j, t);
        float p[KR], qi[KR], qj[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = 64 * r + lane;
            p[r] = e < k ? a.P[tu * k + e] : 0.0f;
            qi[r] = e < k ? a.Q[ti * k + e] : 0.0f;
            qj[r] = e < k ? a.Q[tj * k + e] : 0.0f;
        }
        float ai = 0.0f, aj = 0.0f;
#pragma unroll
        for (int r = 0; r < KR; ++r) { const float a1 = p[r] * qi[r]; ai = ai + a1; const float a2 = p[r] * qj[r]; aj = aj + a2; }
        const float x = wave_sum(ai) - wave_sum(aj);               // BPR.py:50, fp32 margin
        const double s = 1.0 / (1.0 + exp(-(double)x));            // qmath.py:115-116
        const float c = (float)(a.lr * (1.0 - s));
        nll += -log(s);                                            // BPR.py:58
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = 64 * r + lane;
            const Elem o = bpr_elem(p[r], qi[r], qj[r], c, a.ru, a.ri);
            if (e < k) { a.P[tu * k + e] = o.p2; a.Q[ti * k + e] = o.qi2; a.Q[tj * k + e] = o.qj2; }
        }
    }
    if (lane == 0) atomicAdd(a.nll_slots + (wave & (kNllSlots - 1)), nll);
}

// ------------------------------------------------------------------------------------------
// S-round launch.  Every wave first takes the tickets of TPW events of the NEXT round (touch count +
// staging-slot table of their two item rows, user-row flush counts; lanes TPW .. 2 TPW - 1), then
// updates TPW events of THIS round -- no separate workgroups for the counting, so every resident
// workgroup carries 4 * TPW events of the round (the negatives come from k_sample, drawn up front): one wave takes
// TPW consecutive events, requests all their rows up front (straight-line code: the compiler's
// counted waits keep every gather in flight), evaluates the TPW sigmoids in one double-precision
// pass (lane t holds triplet t), then writes:
//   * an item row touched exactly once in the round (cnt == 1): the new row, in place, plain
//     stores -- nobody else reads or writes it this round;
//   * a row with 2..kStageMax touches: (new - old) written to the touch's own staging row (behind the
//     item rows in the Q allocation) with
//     write-through (sc1) stores; the last toucher of the row (touch count reaching zero: every other
//     toucher has drained its stores before its decrement) reads the staged rows back with sc1 loads,
//     adds them in event order -- the order of the oracle's sum -- and rewrites the row once.
//     Scattered 512-byte float-atomic rows run at ~2.4e9/s on MI355X, stores and loads at ~1.1e10/s
//     (tools/experiments/rowops.hip), and most contended rows of a round have 2-3 touches;
//   * a hotter row: (new - old) added into dQ with float atomics; the last toucher swaps the sum out
//     of dQ (leaving it zeroed) and rewrites the row.
// P[u] differences are summed in registers over a run of equal users and flushed into dP; user
// rows are finished by the same last-arriver rule (or, on a communicator, by the all-reduce).
// ------------------------------------------------------------------------------------------
// Raw buffer access: one 128-bit descriptor per matrix in SGPRs, the row's byte offset in an SGPR
// (soffset), the lane's element offset in one VGPR (voffset) -- no per-access address arithmetic in
// the vector ALU.  A lane whose element index is >= k gets a voffset beyond num_records: the
// hardware returns 0 for its loads and drops its stores and atomics.
#define YUE_BLOAD(rs, vo, so) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32((rs), (vo), (so), 0))
#define YUE_BSTORE(val, rs, vo, so) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (val)), (rs), (vo), (so), 0)
#ifdef YUE_INPLACE_AUX
#define YUE_BSTORE_INPLACE(val, rs, vo, so) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (val)), (rs), (vo), (so), YUE_INPLACE_AUX)
#else
#define YUE_BSTORE_INPLACE(val, rs, vo, so) YUE_BSTORE(val, rs, vo, so)
#endif
#define YUE_BLOAD_SC1(rs, vo, so) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32((rs), (vo), (so), 16))
#define YUE_BSTORE_SC1(val, rs, vo, so) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (val)), (rs), (vo), (so), 16)
#define YUE_BATOMIC(val, rs, vo, so) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32((val), (rs), (vo), (so), 0)
#ifdef YUE_STAMPS
#define YUE_STAMP(ix, waits) do { asm volatile(waits ::: "memory"); if (a.stamps && lane == 0) a.stamps[(size_t)wave * 8 + (ix)] = wall_clock64(); } while (0)
#else
#define YUE_STAMP(ix, waits) do {} while (0)
#endif
constexpr unsigned kOobOffset = 0x80000000u;
constexpr unsigned long long kTouch = 0x100000001ull;      // +1 touch in both halves of a counter word
constexpr int kRsrcFlags = 0x00020000;
constexpr unsigned kStageMax = 4;                          // touches per row served by the staging rows

// evu / evi / evj: the event arrays again, as restrict-qualified read-only views of THIS round's
// range, so that the wave-uniform reads of a batch's (u, i, j) become scalar loads (s_load): they do
// not queue behind the vector-memory traffic of the other waves.
template <int KR, int TPW>
__global__ void __launch_bounds__(256) k_round(TrainArgs a, RoundArgs ra, const int32_t *__restrict__ evu,
                                               const int32_t *__restrict__ evi, const int32_t *__restrict__ evj) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Tickets of the NEXT round: lanes TPW .. 2 TPW - 1 of wave w take them for events n_begin + w * TPW + (lane - TPW)
    // -- one returning atomic per item row (the low half before the add is the touch's ticket on its row), one
    // flush count per run of equal users.  An updating wave issues them behind its row stores, so that they
    // return inside the drain wait it needs anyway (in front of the gathers they would hold the gathers back:
    // returns are in order and atomics are the slowest); the slot-table entries follow the wait.
    uint32_t tk_i = 0xffffffffu, tk_j = 0xffffffffu, nx_slot = 0u;
    int32_t nx_i = 0, nx_j = -1;
    // the ids of the next round's event are fetched with the header, so that the tickets can leave right behind the stores
    {
        const int64_t ne = ra.n_begin + wave * TPW + (lane - TPW);
        if (lane >= TPW && lane < 2 * TPW && ne < ra.n_end) { nx_i = a.ev_i[ne]; nx_j = a.ev_j[ne]; }
    }
    auto take_tickets = [&]() {
        const int64_t ne = ra.n_begin + wave * TPW + (lane - TPW);
        if (lane >= TPW && lane < 2 * TPW && ne < ra.n_end) {
            nx_slot = 2u * (uint32_t)(ne - ra.n_begin);
            if (nx_j >= 0) { tk_i = (uint32_t)atomicAdd(ra.cnt_next + nx_i, kTouch); tk_j = (uint32_t)atomicAdd(ra.cnt_next + nx_j, kTouch); }
            if (ra.apply_p) {      // one flush per run of equal users inside a TPW-aligned batch
                const int32_t u = a.ev_u[ne];
                if (lane == TPW || a.ev_u[ne - 1] != u) atomicAdd(ra.cntp_next + u, 1u);
            }
        }
    };
    auto publish_slots = [&]() {
        if (ra.staged && nx_j >= 0) {
            if (tk_i < kStageMax) ra.tab_next[(size_t)nx_i * kStageMax + tk_i] = nx_slot;
            if (tk_j < kStageMax) ra.tab_next[(size_t)nx_j * kStageMax + tk_j] = nx_slot + 1u;
        }
    };
    const int64_t base = ra.e_begin + wave * TPW;
    if (base >= ra.e_end) { take_tickets(); publish_slots(); return; }
    YUE_STAMP(0, "");

    // batch header through the scalar unit: (u, i, j) of the TPW events
    int hu[TPW], hi_[TPW], hj[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const bool ex = base + t < ra.e_end;
        const int64_t ix = ex ? base + t : ra.e_end - 1;
        hu[t] = evu[ix]; hi_[t] = evi[ix]; hj[t] = ex ? evj[ix] : -1;
    }
    // lane t keeps event t (the retire phase and the loss work per lane)
    int i = 0, j = -1;
#pragma unroll
    for (int t = 0; t < TPW; ++t) if (lane == t) { i = hi_[t]; j = hj[t]; }
    YUE_STAMP(1, "s_waitcnt lgkmcnt(0)");
    uint32_t ci = 0, cj = 0;             // touches of my rows in this round (the immutable half)
    if (lane < TPW && j >= 0) { ci = (uint32_t)(ra.cnt_cur[i] >> 32); cj = (uint32_t)(ra.cnt_cur[j] >> 32); }
    const unsigned k = (unsigned)a.k;
    const unsigned row_bytes = k * 4u;
    unsigned vo[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; vo[r] = e < k ? e * 4u : kOobOffset; }

    // users of one batch are neighbours (user-major events): P / dP are addressed relative to the
    // batch's smallest user id, so the 31-bit byte offsets hold for any number of users
    unsigned u0 = 0xffffffffu;
#pragma unroll
    for (int t = 0; t < TPW; ++t) if (base + t < ra.e_end && (unsigned)hu[t] < u0) u0 = (unsigned)hu[t];
    const uint64_t qbytes = (uint64_t)a.n * row_bytes, pbytes = (uint64_t)(a.m - u0) * row_bytes;
    const int qrec = (int)(qbytes < 0x7fffffffull ? qbytes : 0x7fffffffull);
    const int prec = (int)(pbytes < 0x7fffffffull ? pbytes : 0x7fffffffull);
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(a.Q, 0, qrec, kRsrcFlags);
    const auto rsdQ = __builtin_amdgcn_make_buffer_rsrc(a.dQ, 0, qrec, kRsrcFlags);
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc(a.P + (uint64_t)u0 * k, 0, prec, kRsrcFlags);
    const auto rsdP = __builtin_amdgcn_make_buffer_rsrc(a.dP + (uint64_t)u0 * k, 0, prec, kRsrcFlags);
    // staging rows of this round: 2 per event (the host keeps 2 * events * row_bytes below 2^31 when staged)
    const auto rsS = __builtin_amdgcn_make_buffer_rsrc(a.stage, 0, ra.staged ? (int)(2u * (unsigned)(ra.e_end - ra.e_begin) * row_bytes) : 0, kRsrcFlags);

    unsigned oi[TPW], oj[TPW], ou[TPW], ru_[TPW], ri_[TPW], rj_[TPW];
    bool ok[TPW];
    float qi[TPW][KR], qj[TPW][KR], p[TPW][KR];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        ru_[t] = (unsigned)hu[t];
        ri_[t] = (unsigned)hi_[t];
        const int tj = hj[t];
        ok[t] = tj >= 0;                                 // no event / sampler gave up: nothing is written
        rj_[t] = ok[t] ? (unsigned)tj : 0u;
        oi[t] = ri_[t] * row_bytes; oj[t] = rj_[t] * row_bytes;
        ou[t] = (base + t < ra.e_end ? ru_[t] - u0 : 0u) * row_bytes;
#pragma unroll
        for (int r = 0; r < KR; ++r) { qi[t][r] = YUE_BLOAD(rsQ, vo[r], oi[t]); qj[t][r] = YUE_BLOAD(rsQ, vo[r], oj[t]); }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < KR; ++r) p[t][r] = YUE_BLOAD(rsP, vo[r], ou[t]);

    YUE_STAMP(2, "s_waitcnt vmcnt(0)");
    float xs = 0.0f;                                     // lane t will hold the margin of triplet t
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        float ai = 0.0f, aj = 0.0f;
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const float a1 = p[t][r] * qi[t][r]; ai = ai + a1;
            const float a2 = p[t][r] * qj[t][r]; aj = aj + a2;
        }
        const float x = wave_sum(ai) - wave_sum(aj);               // BPR.py:50, fp32 margin
        xs = lane == t ? x : xs;
    }
    __builtin_amdgcn_sched_barrier(0);
    const double s = 1.0 / (1.0 + exp(-(double)xs));               // qmath.py:115-116
    const float cs = (float)(a.lr * (1.0 - s));
    double nll = (lane < TPW && j >= 0) ? -log(s) : 0.0;           // BPR.py:58
    __builtin_amdgcn_sched_barrier(0);

    YUE_STAMP(3, "");
    float dp[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) dp[r] = 0.0f;
    bool run_ok = false;                                 // some triplet of the current user run wrote
    int nruns = 0;                                       // runs of equal users in this batch (wave-uniform)
    unsigned run_u = 0;                                  // lane q holds the user of run q
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const float c = rdlane(cs, t);
        const unsigned cti = (unsigned)__builtin_amdgcn_readlane(ci, t), ctj = (unsigned)__builtin_amdgcn_readlane(cj, t);
        const bool uniq_i = cti == 1u, uniq_j = ctj == 1u;
        const bool stg_i = ra.staged && cti <= kStageMax, stg_j = ra.staged && ctj <= kStageMax;
        const unsigned ss = 2u * ((unsigned)wave * TPW + t) * row_bytes;     // my two staging rows
        if (ok[t]) {                                     // wave-uniform
            run_ok = true;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const Elem o = bpr_elem(p[t][r], qi[t][r], qj[t][r], c, a.ru, a.ri);
                if (uniq_i) YUE_BSTORE_INPLACE(o.qi2, rsQ, vo[r], oi[t]);
                else if (stg_i) YUE_BSTORE_SC1(o.qi2 - qi[t][r], rsS, vo[r], ss);
                else YUE_BATOMIC(o.qi2 - qi[t][r], rsdQ, vo[r], oi[t]);
                if (uniq_j) YUE_BSTORE_INPLACE(o.qj2, rsQ, vo[r], oj[t]);
                else if (stg_j) YUE_BSTORE_SC1(o.qj2 - qj[t][r], rsS, vo[r], ss + row_bytes);
                else YUE_BATOMIC(o.qj2 - qj[t][r], rsdQ, vo[r], oj[t]);
                dp[r] += o.p2 - p[t][r];
            }
            if (lane == 0) {                             // sole toucher: reset the counter here
                if (uniq_i) ra.cnt_cur[ri_[t]] = 0ull;
                if (uniq_j) ra.cnt_cur[rj_[t]] = 0ull;
            }
        }
        // end of a run of equal users (or of the batch): flush the summed P[u] differences
        const bool exists = base + t < ra.e_end;
        const bool last = exists && ((t == TPW - 1) || base + t + 1 >= ra.e_end || ru_[t + 1 < TPW ? t + 1 : t] != ru_[t]);
        if (last) {
            if (run_ok) {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_BATOMIC(dp[r], rsdP, vo[r], ou[t]);
            }
            run_u = lane == nruns ? ru_[t] : run_u;
            ++nruns;
#pragma unroll
            for (int r = 0; r < KR; ++r) dp[r] = 0.0f;
            run_ok = false;
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // Retire this batch's touches of contended rows.  Every toucher decrements the row's count once
    // its own adds are acknowledged; whoever takes the count to zero knows every add of the round
    // has been performed and every toucher has read the row: it swaps the sum out of dQ (returning
    // atomic: coherent at the memory side, leaves dQ zeroed) and rewrites the row.
    YUE_STAMP(4, "");
    // staging slots of my contended rows (written by the previous launch): fetched behind the drain of
    // my stores, only where a last toucher could need them (rows with 2..kStageMax touches)
    uint32_t si[kStageMax], sj[kStageMax];
#pragma unroll
    for (unsigned q = 0; q < kStageMax; ++q) si[q] = sj[q] = 0xffffffffu;
    if (ra.staged && lane < TPW && j >= 0) {
        if (ci != 1u && ci <= kStageMax) {
            const uint4 wi = *reinterpret_cast<const uint4 *>(ra.tab_cur + (size_t)i * kStageMax);
            si[0] = wi.x; si[1] = wi.y; si[2] = wi.z; si[3] = wi.w;
        }
        if (cj != 1u && cj <= kStageMax) {
            const uint4 wj = *reinterpret_cast<const uint4 *>(ra.tab_cur + (size_t)j * kStageMax);
            sj[0] = wj.x; sj[1] = wj.y; sj[2] = wj.z; sj[3] = wj.w;
        }
    }
    if (!(ra.debug_skip & 2)) take_tickets();
    if (!(ra.debug_skip & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    YUE_STAMP(5, "");
    if (!(ra.debug_skip & 2)) publish_slots();
    // entries beyond the row's touch count are leftovers of earlier rounds: drop them, then put the
    // slots in ascending order = event order, the order in which the oracle sums a row's differences
    static_assert(kStageMax == 4, "the sorting network below is written for four slots");
#pragma unroll
    for (unsigned q = 0; q < kStageMax; ++q) { if (q >= ci) si[q] = 0xffffffffu; if (q >= cj) sj[q] = 0xffffffffu; }
#define YUE_CSWAP(x, y) { const uint32_t lo_ = min(x, y), hi2_ = max(x, y); x = lo_; y = hi2_; }
    YUE_CSWAP(si[0], si[1]) YUE_CSWAP(si[2], si[3]) YUE_CSWAP(si[0], si[2]) YUE_CSWAP(si[1], si[3]) YUE_CSWAP(si[1], si[2])
    YUE_CSWAP(sj[0], sj[1]) YUE_CSWAP(sj[2], sj[3]) YUE_CSWAP(sj[0], sj[2]) YUE_CSWAP(sj[1], sj[3]) YUE_CSWAP(sj[1], sj[2])
#undef YUE_CSWAP

    bool last_i = false, last_j = false, last_p = false;
    if ((ra.debug_skip & 1) && lane < TPW && j >= 0) { ra.cnt_cur[i] = 0ull; ra.cnt_cur[j] = 0ull; }     // timing experiment: keep the counters usable
    if (!(ra.debug_skip & 1) && lane < TPW && j >= 0) {
        if (ci != 1u) last_i = (uint32_t)atomicAdd(ra.cnt_cur + i, ~0ull) == 1u;     // -1 on the low half
        if (cj != 1u) last_j = (uint32_t)atomicAdd(ra.cnt_cur + j, ~0ull) == 1u;
    }
    if (ra.apply_p && lane < nruns) last_p = atomicSub(ra.cntp_cur + run_u, 1u) == 1u;
    // winners: bits [0,TPW) = item row i of that lane, [TPW,2*TPW) = row j, [2*TPW,3*TPW) = user run
    unsigned long long win = (__ballot(last_i) & ((1ull << TPW) - 1)) | ((__ballot(last_j) & ((1ull << TPW) - 1)) << TPW) |
                             ((__ballot(last_p) & ((1ull << TPW) - 1)) << (2 * TPW));
    YUE_STAMP(6, "s_waitcnt vmcnt(0)");
    while (win) {
        // up to four rows per pass: all swaps and row loads are issued before the first store
        float *xp[4], *dx[4];
        bool act[4];
        unsigned nst[4], so[4][kStageMax];               // staged rows to add (0: the row went through dQ / dP)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            act[sl] = win != 0;
            const int b = act[sl] ? __ffsll((long long)win) - 1 : 0;
            if (act[sl]) win &= win - 1;
            const int src = b % TPW;
            const unsigned row = b < TPW ? (unsigned)__builtin_amdgcn_readlane(i, src)
                                 : b < 2 * TPW ? (unsigned)__builtin_amdgcn_readlane(j, src)
                                               : (unsigned)__builtin_amdgcn_readlane((int)run_u, src);
            const unsigned touches = b < TPW ? (unsigned)__builtin_amdgcn_readlane(ci, src)
                                     : b < 2 * TPW ? (unsigned)__builtin_amdgcn_readlane(cj, src) : 0u;
            nst[sl] = act[sl] && ra.staged && b < 2 * TPW && touches <= kStageMax ? touches : 0u;
#pragma unroll
            for (unsigned q = 0; q < kStageMax; ++q)
                so[sl][q] = (b < TPW ? (unsigned)__builtin_amdgcn_readlane(si[q], src) : (unsigned)__builtin_amdgcn_readlane(sj[q], src)) * row_bytes;
            const uint64_t o = (uint64_t)row * k;
            xp[sl] = (b < 2 * TPW ? a.Q : a.P) + o;
            dx[sl] = (b < 2 * TPW ? a.dQ : a.dP) + o;
            if (act[sl] && b < 2 * TPW && lane == 0) ra.cnt_cur[row] = 0ull;      // every touch retired: clear the word
        }
        float d[4][KR], x[4][KR];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
            if (act[sl]) {
                if (nst[sl]) {
                    // every load of the staged bytes is an sc1 load issued after this wave's own
                    // decrement returned; slots beyond the count get an out-of-range offset (reads 0)
                    float st[kStageMax][KR];
#pragma unroll
                    for (unsigned q = 0; q < kStageMax; ++q)
#pragma unroll
                        for (int r = 0; r < KR; ++r)
                            st[q][r] = YUE_BLOAD_SC1(rsS, q < nst[sl] ? vo[r] : kOobOffset, q < nst[sl] ? so[sl][q] : 0u);
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const unsigned e = 64u * r + lane;
                        if (e < k) x[sl][r] = xp[sl][e];
                        float acc = st[0][r];
#pragma unroll
                        for (unsigned q = 1; q < kStageMax; ++q) acc = acc + st[q][r];
                        d[sl][r] = acc;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const unsigned e = 64u * r + lane;
                        if (e < k) { d[sl][r] = atomicExch(dx[sl] + e, 0.0f); x[sl][r] = xp[sl][e]; }
                    }
                }
            }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
            if (act[sl]) {
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const unsigned e = 64u * r + lane;
                    if (e < k) xp[sl][e] = x[sl][r] + d[sl][r];
                }
            }
    }
    YUE_STAMP(7, "s_waitcnt vmcnt(0)");
#pragma unroll
    for (int off = 1; off < TPW; off <<= 1) nll += __shfl_xor(nll, off);
    if (lane == 0 && nll != 0.0) atomicAdd(a.nll_slots + (wave & (kNllSlots - 1)), nll);
}

// ------------------------------------------------------------------------------------------
// S-round launch, pipelined form (the default).  Same round semantics and the same touch protocol as
// k_round above, restructured so that one wave streams a LONG run of the round's events instead of one batch:
//   * a wave takes T (<= 64) consecutive events; lane l keeps event l's (u, i, j), its two touch counts,
//     and later its tickets, decrements and wins -- the per-event bookkeeping is lane-parallel, once per wave;
//   * the events are walked in groups of G with two register buffers: the row gathers of group g+1 are issued
//     before group g is evaluated and stored, so a wave always has a group's gathers in flight behind the
//     stores of the previous one (a group that does not exist is requested through a zero-record descriptor:
//     the range check drops the loads, the instruction stream and the counted waits stay the same);
//   * the drain, the next round's tickets, the count decrements and the last-toucher rewrites are paid once
//     per T events instead of once per 8.
// ------------------------------------------------------------------------------------------
template <int KR, int G>
__global__ void __launch_bounds__(256) k_round2(TrainArgs a, RoundArgs ra, int T) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Tickets of the NEXT round: lane l < T of wave w takes them for event n_begin + w * T + l (see k_round).
    uint32_t tk_i = 0xffffffffu, tk_j = 0xffffffffu, nx_slot = 0u;
    int32_t nx_i = 0, nx_j = -1;
    auto take_tickets = [&]() {
        const int64_t ne = ra.n_begin + wave * T + lane;
        if (lane < T && ne < ra.n_end) {
            nx_i = a.ev_i[ne]; nx_j = a.ev_j[ne];
            nx_slot = 2u * (uint32_t)(ne - ra.n_begin);
            if (nx_j >= 0) { tk_i = (uint32_t)atomicAdd(ra.cnt_next + nx_i, kTouch); tk_j = (uint32_t)atomicAdd(ra.cnt_next + nx_j, kTouch); }
            if (ra.apply_p) {      // one flush per run of equal users inside a wave's T events
                const int32_t un = a.ev_u[ne];
                if (lane == 0 || a.ev_u[ne - 1] != un) atomicAdd(ra.cntp_next + un, 1u);
            }
        }
    };
    auto publish_slots = [&]() {
        if (ra.staged && nx_j >= 0) {
            if (tk_i < kStageMax) ra.tab_next[(size_t)nx_i * kStageMax + tk_i] = nx_slot;
            if (tk_j < kStageMax) ra.tab_next[(size_t)nx_j * kStageMax + tk_j] = nx_slot + 1u;
        }
    };
    const int64_t base = ra.e_begin + wave * T;
    if (base >= ra.e_end) { take_tickets(); publish_slots(); return; }
    YUE_STAMP(0, "");
    const int cnt = (int)((ra.e_end - base) < (int64_t)T ? (ra.e_end - base) : (int64_t)T);     // wave-uniform

    // lane l keeps event l: ids, touches of its two item rows in this round (the immutable half of the counter word)
    int u = 0, i = 0, j = -1;
    if (lane < cnt) { u = a.ev_u[base + lane]; i = a.ev_i[base + lane]; j = a.ev_j[base + lane]; }
    uint32_t ci = 0, cj = 0;
    if (lane < cnt && j >= 0) { ci = (uint32_t)(ra.cnt_cur[i] >> 32); cj = (uint32_t)(ra.cnt_cur[j] >> 32); }
    const unsigned k = (unsigned)a.k;
    const unsigned row_bytes = k * 4u;
    unsigned vo[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; vo[r] = e < k ? e * 4u : kOobOffset; }

    // P / dP are addressed relative to the wave's smallest user id (31-bit byte offsets for any number of users)
    unsigned u0 = lane < cnt ? (unsigned)u : 0xffffffffu;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned o = (unsigned)__shfl_xor((int)u0, off); u0 = o < u0 ? o : u0; }
    u0 = (unsigned)__builtin_amdgcn_readfirstlane((int)u0);
    const uint64_t qbytes = (uint64_t)a.n * row_bytes, pbytes = (uint64_t)(a.m - u0) * row_bytes;
    const int qrec = (int)(qbytes < 0x7fffffffull ? qbytes : 0x7fffffffull);
    const int prec = (int)(pbytes < 0x7fffffffull ? pbytes : 0x7fffffffull);
    float *const Pb = a.P + (uint64_t)u0 * k;
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(a.Q, 0, qrec, kRsrcFlags);
    const auto rsdQ = __builtin_amdgcn_make_buffer_rsrc(a.dQ, 0, qrec, kRsrcFlags);
    const auto rsdP = __builtin_amdgcn_make_buffer_rsrc(a.dP + (uint64_t)u0 * k, 0, prec, kRsrcFlags);
    const auto rsS = __builtin_amdgcn_make_buffer_rsrc(a.stage, 0, ra.staged ? (int)(2u * (unsigned)(ra.e_end - ra.e_begin) * row_bytes) : 0, kRsrcFlags);
    const int ngroups = (cnt + G - 1) / G;
    const unsigned slot0 = 2u * (unsigned)((int)wave * T);          // staging slot of my first event's positive row

    // gathers of one group into a register buffer; live == false: the group does not exist (zero-record descriptors)
    auto gather = [&](float (&qi)[G][KR], float (&qj)[G][KR], float (&p)[G][KR], int g, bool live) {
        const auto rQ = __builtin_amdgcn_make_buffer_rsrc(a.Q, 0, live ? qrec : 0, kRsrcFlags);
        const auto rP = __builtin_amdgcn_make_buffer_rsrc(Pb, 0, live ? prec : 0, kRsrcFlags);
#pragma unroll
        for (int t = 0; t < G; ++t) {
            int idx = G * g + t;
            idx = idx < cnt ? idx : cnt - 1;             // events beyond the wave's range re-read the last one; nothing of them is written
            const unsigned ti = (unsigned)__builtin_amdgcn_readlane(i, idx);
            const int tjs = __builtin_amdgcn_readlane(j, idx);
            const unsigned tj = tjs >= 0 ? (unsigned)tjs : 0u;
            const unsigned oi = ti * row_bytes, oj = tj * row_bytes;
#pragma unroll
            for (int r = 0; r < KR; ++r) { qi[t][r] = YUE_BLOAD(rQ, vo[r], oi); qj[t][r] = YUE_BLOAD(rQ, vo[r], oj); }
        }
#pragma unroll
        for (int t = 0; t < G; ++t) {
            int idx = G * g + t;
            idx = idx < cnt ? idx : cnt - 1;
            const unsigned ou = ((unsigned)__builtin_amdgcn_readlane(u, idx) - u0) * row_bytes;
#pragma unroll
            for (int r = 0; r < KR; ++r) p[t][r] = YUE_BLOAD(rP, vo[r], ou);
        }
    };

    double nll = 0.0;
    float dp[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) dp[r] = 0.0f;
    bool run_ok = false;                                 // some triplet of the current user run wrote
    int nruns = 0;                                       // runs of equal users in this wave's events (wave-uniform)
    unsigned run_u = 0;                                  // lane q holds the user of run q

    auto process = [&](float (&qi)[G][KR], float (&qj)[G][KR], float (&p)[G][KR], int g) {
        float xs = 0.0f;                                 // lane G*g + t will hold the margin of the group's triplet t
#pragma unroll
        for (int t = 0; t < G; ++t) {
            float ai = 0.0f, aj = 0.0f;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const float a1 = p[t][r] * qi[t][r]; ai = ai + a1;
                const float a2 = p[t][r] * qj[t][r]; aj = aj + a2;
            }
            const float x = wave_sum(ai) - wave_sum(aj);               // BPR.py:50, fp32 margin
            xs = lane == G * g + t ? x : xs;
        }
        __builtin_amdgcn_sched_barrier(0);
        const double s = 1.0 / (1.0 + exp(-(double)xs));               // qmath.py:115-116
        const float cs = (float)(a.lr * (1.0 - s));
        if (lane >= G * g && lane < G * g + G && lane < cnt && j >= 0) nll += -log(s);      // BPR.py:58
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < G; ++t) {
            const int idx = G * g + t;
            const bool exists = idx < cnt;
            const int li = exists ? idx : cnt - 1;
            const float c = rdlane(cs, li);
            const unsigned tu = (unsigned)__builtin_amdgcn_readlane(u, li), ti = (unsigned)__builtin_amdgcn_readlane(i, li);
            const int tjs = __builtin_amdgcn_readlane(j, li);
            const bool okt = exists && tjs >= 0;              // no event / sampler gave up: nothing is written
            const unsigned tj = tjs >= 0 ? (unsigned)tjs : 0u;
            const unsigned oi = ti * row_bytes, oj = tj * row_bytes, ou = (tu - u0) * row_bytes;
            const unsigned cti = (unsigned)__builtin_amdgcn_readlane(ci, li), ctj = (unsigned)__builtin_amdgcn_readlane(cj, li);
            const bool uniq_i = cti == 1u, uniq_j = ctj == 1u;
            const bool stg_i = ra.staged && cti <= kStageMax, stg_j = ra.staged && ctj <= kStageMax;
            const unsigned ss = (slot0 + 2u * (unsigned)idx) * row_bytes;        // my two staging rows
            if (okt) {                                       // wave-uniform
                run_ok = true;
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const Elem o = bpr_elem(p[t][r], qi[t][r], qj[t][r], c, a.ru, a.ri);
                    if (uniq_i) YUE_BSTORE(o.qi2, rsQ, vo[r], oi);
                    else if (stg_i) YUE_BSTORE_SC1(o.qi2 - qi[t][r], rsS, vo[r], ss);
                    else YUE_BATOMIC(o.qi2 - qi[t][r], rsdQ, vo[r], oi);
                    if (uniq_j) YUE_BSTORE(o.qj2, rsQ, vo[r], oj);
                    else if (stg_j) YUE_BSTORE_SC1(o.qj2 - qj[t][r], rsS, vo[r], ss + row_bytes);
                    else YUE_BATOMIC(o.qj2 - qj[t][r], rsdQ, vo[r], oj);
                    dp[r] += o.p2 - p[t][r];
                }
            }
            // end of a run of equal users (or of the wave's events): flush the summed P[u] differences
            const int ln = li + 1 < cnt ? li + 1 : li;
            const bool last = exists && (idx == cnt - 1 || (unsigned)__builtin_amdgcn_readlane(u, ln) != tu);
            if (last) {
                if (run_ok) {
#pragma unroll
                    for (int r = 0; r < KR; ++r) YUE_BATOMIC(dp[r], rsdP, vo[r], ou);
                }
                run_u = lane == nruns ? tu : run_u;
                ++nruns;
#pragma unroll
                for (int r = 0; r < KR; ++r) dp[r] = 0.0f;
                run_ok = false;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    float qiA[G][KR], qjA[G][KR], pA[G][KR], qiB[G][KR], qjB[G][KR], pB[G][KR];
    gather(qiA, qjA, pA, 0, true);
    YUE_STAMP(1, "");
    for (int g = 0; g < ngroups; g += 2) {
        gather(qiB, qjB, pB, g + 1, g + 1 < ngroups);
        process(qiA, qjA, pA, g);
#ifdef YUE_STAMPS
        if (g == 0) YUE_STAMP(2, "");
#endif
        gather(qiA, qjA, pA, g + 2, g + 2 < ngroups);
        if (g + 1 < ngroups) process(qiB, qjB, pB, g + 1);
#ifdef YUE_STAMPS
        if (g == 0) YUE_STAMP(3, "");
#endif
    }
    YUE_STAMP(4, "");

    // sole touchers reset their rows' counter words (nobody else reads them in this round)
    if (lane < cnt && j >= 0) {
        if (ci == 1u) ra.cnt_cur[i] = 0ull;
        if (cj == 1u) ra.cnt_cur[j] = 0ull;
    }
    // staging slots of my contended rows (written by the previous launch), where a last toucher could need them
    uint32_t si[kStageMax], sj[kStageMax];
#pragma unroll
    for (unsigned q = 0; q < kStageMax; ++q) si[q] = sj[q] = 0xffffffffu;
    if (ra.staged && lane < cnt && j >= 0) {
        if (ci != 1u && ci <= kStageMax) {
            const uint4 wi = *reinterpret_cast<const uint4 *>(ra.tab_cur + (size_t)i * kStageMax);
            si[0] = wi.x; si[1] = wi.y; si[2] = wi.z; si[3] = wi.w;
        }
        if (cj != 1u && cj <= kStageMax) {
            const uint4 wj = *reinterpret_cast<const uint4 *>(ra.tab_cur + (size_t)j * kStageMax);
            sj[0] = wj.x; sj[1] = wj.y; sj[2] = wj.z; sj[3] = wj.w;
        }
    }
    take_tickets();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    YUE_STAMP(5, "");
    publish_slots();
    static_assert(kStageMax == 4, "the sorting network below is written for four slots");
#pragma unroll
    for (unsigned q = 0; q < kStageMax; ++q) { if (q >= ci) si[q] = 0xffffffffu; if (q >= cj) sj[q] = 0xffffffffu; }
#define YUE_CSWAP(x, y) { const uint32_t lo_ = min(x, y), hi2_ = max(x, y); x = lo_; y = hi2_; }
    YUE_CSWAP(si[0], si[1]) YUE_CSWAP(si[2], si[3]) YUE_CSWAP(si[0], si[2]) YUE_CSWAP(si[1], si[3]) YUE_CSWAP(si[1], si[2])
    YUE_CSWAP(sj[0], sj[1]) YUE_CSWAP(sj[2], sj[3]) YUE_CSWAP(sj[0], sj[2]) YUE_CSWAP(sj[1], sj[3]) YUE_CSWAP(sj[1], sj[2])
#undef YUE_CSWAP

    bool last_i = false, last_j = false, last_p = false;
    if (lane < cnt && j >= 0) {
        if (ci != 1u) last_i = (uint32_t)atomicAdd(ra.cnt_cur + i, ~0ull) == 1u;     // -1 on the low half
        if (cj != 1u) last_j = (uint32_t)atomicAdd(ra.cnt_cur + j, ~0ull) == 1u;
    }
    if (ra.apply_p && lane < nruns) last_p = atomicSub(ra.cntp_cur + run_u, 1u) == 1u;
    unsigned long long wi_ = __ballot(last_i), wj_ = __ballot(last_j), wp_ = __ballot(last_p);
    YUE_STAMP(6, "s_waitcnt vmcnt(0)");
    while (wi_ | wj_ | wp_) {
        // up to four rows per pass: all swaps and row loads are issued before the first store
        float *xp[4], *dx[4];
        bool act[4];
        unsigned nst[4], so[4][kStageMax];               // staged rows to add (0: the row went through dQ / dP)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            int kind = 3, src = 0;
            if (wi_) { kind = 0; src = __ffsll((long long)wi_) - 1; wi_ &= wi_ - 1; }
            else if (wj_) { kind = 1; src = __ffsll((long long)wj_) - 1; wj_ &= wj_ - 1; }
            else if (wp_) { kind = 2; src = __ffsll((long long)wp_) - 1; wp_ &= wp_ - 1; }
            act[sl] = kind != 3;
            const unsigned row = kind == 0 ? (unsigned)__builtin_amdgcn_readlane(i, src)
                                 : kind == 1 ? (unsigned)__builtin_amdgcn_readlane(j, src)
                                             : (unsigned)__builtin_amdgcn_readlane((int)run_u, src);
            const unsigned touches = kind == 0 ? (unsigned)__builtin_amdgcn_readlane(ci, src)
                                     : kind == 1 ? (unsigned)__builtin_amdgcn_readlane(cj, src) : 0u;
            nst[sl] = act[sl] && ra.staged && kind < 2 && touches <= kStageMax ? touches : 0u;
#pragma unroll
            for (unsigned q = 0; q < kStageMax; ++q)
                so[sl][q] = (kind == 0 ? (unsigned)__builtin_amdgcn_readlane(si[q], src) : (unsigned)__builtin_amdgcn_readlane(sj[q], src)) * row_bytes;
            const uint64_t o = (uint64_t)row * k;
            xp[sl] = (kind < 2 ? a.Q : a.P) + o;
            dx[sl] = (kind < 2 ? a.dQ : a.dP) + o;
            if (act[sl] && kind < 2 && lane == 0) ra.cnt_cur[row] = 0ull;      // every touch retired: clear the word
        }
        float d[4][KR], x[4][KR];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
            if (act[sl]) {
                if (nst[sl]) {
                    float st[kStageMax][KR];
#pragma unroll
                    for (unsigned q = 0; q < kStageMax; ++q)
#pragma unroll
                        for (int r = 0; r < KR; ++r)
                            st[q][r] = YUE_BLOAD_SC1(rsS, q < nst[sl] ? vo[r] : kOobOffset, q < nst[sl] ? so[sl][q] : 0u);
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const unsigned e = 64u * r + lane;
                        if (e < k) x[sl][r] = xp[sl][e];
                        float acc = st[0][r];
#pragma unroll
                        for (unsigned q = 1; q < kStageMax; ++q) acc = acc + st[q][r];
                        d[sl][r] = acc;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const unsigned e = 64u * r + lane;
                        if (e < k) { d[sl][r] = atomicExch(dx[sl] + e, 0.0f); x[sl][r] = xp[sl][e]; }
                    }
                }
            }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
            if (act[sl]) {
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const unsigned e = 64u * r + lane;
                    if (e < k) xp[sl][e] = x[sl][r] + d[sl][r];
                }
            }
    }
    YUE_STAMP(7, "s_waitcnt vmcnt(0)");
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) nll += __shfl_xor(nll, off);
    if (lane == 0 && nll != 0.0) atomicAdd(a.nll_slots + (wave & (kNllSlots - 1)), nll);
}

// ------------------------------------------------------------------------------------------
// S-round launch without a retire protocol (the default of yue_bpr_epoch).  Same round semantics as k_round;
// what changes is how the differences of a contended item row meet:
//   * a row with 2..kTabMax touches: every toucher but the LAST ONE IN EVENT ORDER writes (new - old) to its own
//     staging row as 8-byte {value, round tag} granules (one write-through store per granule: a reader sees a
//     granule whole or not at all) and is done -- no drain, no counter.  The last toucher in event order (it finds
//     itself in the row's slot table, written with the previous launch's tickets) keeps its own difference and the
//     old row in registers, reads the other touchers' staging rows until every granule carries this round's tag,
//     adds them in event order (the oracle's order) and rewrites the row once.  It only ever waits for EARLIER
//     events, i.e. for waves of its own or of lower-numbered workgroups.
//   * a HOT row (the host marks the rows expected to collect more than a few touches per round; ev_h carries the
//     marks per event) is never rewritten inside a round: its differences are added with float atomics into a
//     small side buffer (three buffers in rotation: read = last round's sum, acc = this round's, zero = the one
//     cleared for the next round); a reader takes old row + read buffer.  A few extra workgroups per launch carry
//     every hot row's pending sum over (acc += read; zero = 0).  yue_bpr_epoch folds the sums into the rows at the
//     end of a group of rounds (k_hot_fold).
//   * any other row with more than kTabMax touches (rare: a cold row collecting that many random negatives)
//     goes through dQ and the counter protocol of k_round.
// ------------------------------------------------------------------------------------------
constexpr unsigned kTabMax = 8;                            // slot-table words per item row (k_round3)
constexpr unsigned kPollLimit = 1u << 22;                  // give up (error flag) instead of spinning forever

struct Round3Args {
    int64_t e_begin, e_end;      // events updated by this launch
    int64_t n_begin, n_end;      // events of the NEXT round: row touches counted here
    unsigned long long *cnt_cur, *cnt_next;
    uint32_t *tab_cur, *tab_next;        // kTabMax words per item row
    uint32_t tag;                        // this round's tag (never 0, never repeats while a slot can hold it)
    const int32_t *ev_h;                 // per event: (hot index of i + 1) | (hot index of j + 1) << 16, 0 = not hot
    float *hq_read, *hq_acc, *hq_zero;   // [nhot][k] pending sums of the hot rows (rotation, see above)
    int32_t nhot;
    int32_t event_blocks;                // blocks [0, event_blocks) take events; the rest carry the hot rows over
    int32_t *err_flag;                   // set when a poll gave up
};

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define YUE_TLOAD(rs, vo, so) __builtin_amdgcn_raw_buffer_load_b64((rs), (vo), (so), 16)
#define YUE_TSTORE(val, tg, rs, vo, so) __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, (val)), (tg)}, (rs), (vo), (so), 16)

// scalar sorting network for the eight slot-table words of one row (wave-uniform values: scalar ALU)
#define YUE_SSWAP(x, y) { const uint32_t lo_ = x < y ? x : y, hi2_ = x < y ? y : x; x = lo_; y = hi2_; }
__device__ __forceinline__ void sort8_uniform(uint32_t (&v)[8]) {
    YUE_SSWAP(v[0], v[1]) YUE_SSWAP(v[2], v[3]) YUE_SSWAP(v[4], v[5]) YUE_SSWAP(v[6], v[7])
    YUE_SSWAP(v[0], v[2]) YUE_SSWAP(v[1], v[3]) YUE_SSWAP(v[4], v[6]) YUE_SSWAP(v[5], v[7])
    YUE_SSWAP(v[1], v[2]) YUE_SSWAP(v[5], v[6])
    YUE_SSWAP(v[0], v[4]) YUE_SSWAP(v[1], v[5]) YUE_SSWAP(v[2], v[6]) YUE_SSWAP(v[3], v[7])
    YUE_SSWAP(v[2], v[4]) YUE_SSWAP(v[3], v[5])
    YUE_SSWAP(v[1], v[2]) YUE_SSWAP(v[3], v[4]) YUE_SSWAP(v[5], v[6])
}
#undef YUE_SSWAP

#ifndef YUE_KEEP
#define YUE_KEEP 2
#endif
#ifndef YUE_HOTREGS
#define YUE_HOTREGS 2
#endif
constexpr int kKeep = YUE_KEEP;                            // last-toucher rows a wave keeps in registers
constexpr int kHotRegs = YUE_HOTREGS;                      // hot-row pending sums a wave fetches beside its gathers

template <int KR, int TPW>
__global__ void __launch_bounds__(256) k_round3(TrainArgs a, Round3Args ra, const int32_t *__restrict__ evu,
                                                const int32_t *__restrict__ evi, const int32_t *__restrict__ evj,
                                                const int32_t *__restrict__ evh) {
    const int lane = threadIdx.x & 63;
    const unsigned k = (unsigned)a.k;
    if ((int)blockIdx.x >= ra.event_blocks) {
        // carry the hot rows' pending sums over: acc += read, zero = 0 (every hot row, every round)
        const int nw = ((int)gridDim.x - ra.event_blocks) * 4;
        const int w0 = ((int)blockIdx.x - ra.event_blocks) * 4 + (int)(threadIdx.x >> 6);
        for (int h = w0; h < ra.nhot; h += nw)
            for (unsigned e = lane; e < k; e += 64) {
                const size_t o = (size_t)h * k + e;
                const float d = ra.hq_read[o];
                if (d != 0.0f) atomicAdd(ra.hq_acc + o, d);
                ra.hq_zero[o] = 0.0f;
            }
        return;
    }
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // tickets of the NEXT round (lanes TPW .. 2 TPW - 1, as in k_round); hot rows take none
    uint32_t tk_i = 0xffffffffu, tk_j = 0xffffffffu;
    auto take_tickets = [&]() {
        const int64_t ne = ra.n_begin + wave * TPW + (lane - TPW);
        if (lane >= TPW && lane < 2 * TPW && ne < ra.n_end) {
            const int32_t nx_i = a.ev_i[ne], nx_j = a.ev_j[ne];
            const uint32_t nh = (uint32_t)ra.ev_h[ne];
            if (nx_j >= 0) {
                if (!(nh & 0xffffu)) tk_i = (uint32_t)atomicAdd(ra.cnt_next + nx_i, kTouch);
                if (!(nh >> 16)) tk_j = (uint32_t)atomicAdd(ra.cnt_next + nx_j, kTouch);
            }
        }
    };
    auto publish_slots = [&]() {
        const int64_t ne = ra.n_begin + wave * TPW + (lane - TPW);
        if (lane >= TPW && lane < 2 * TPW && ne < ra.n_end) {
            const uint32_t nx_slot = 2u * (uint32_t)(ne - ra.n_begin);
            if (tk_i < kTabMax) ra.tab_next[(size_t)a.ev_i[ne] * kTabMax + tk_i] = nx_slot;
            if (tk_j < kTabMax) ra.tab_next[(size_t)a.ev_j[ne] * kTabMax + tk_j] = nx_slot + 1u;
        }
    };
    const int64_t base = ra.e_begin + wave * TPW;
    if (base >= ra.e_end) { take_tickets(); publish_slots(); return; }
    YUE_STAMP(0, "");

    // batch header through the scalar unit: (u, i, j, hot marks) of the TPW events
    int hu[TPW], hi_[TPW], hj[TPW], hh[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const bool ex = base + t < ra.e_end;
        const int64_t ix = ex ? base + t : ra.e_end - 1;
        hu[t] = evu[ix]; hi_[t] = evi[ix]; hj[t] = ex ? evj[ix] : -1; hh[t] = evh[ix];
    }
    // lane t keeps event t: ids, hot marks, touch counts of its two rows and whether it is their last toucher
    int i = 0, j = -1;
    uint32_t hm = 0u;
#pragma unroll
    for (int t = 0; t < TPW; ++t) if (lane == t) { i = hi_[t]; j = hj[t]; hm = (uint32_t)hh[t]; }
    YUE_STAMP(1, "s_waitcnt lgkmcnt(0)");
    const bool mine = lane < TPW && j >= 0;
    const bool hot_i = (hm & 0xffffu) != 0u, hot_j = (hm >> 16) != 0u;
    const uint32_t my_slot = 2u * ((uint32_t)wave * TPW + (uint32_t)lane);
    // class of a touch: 0 hot, 1 sole toucher, 2 staged (not the last one), 3 the last toucher of a staged row
    // (the largest slot among the row's touches = the last one in event order), 4 through dQ + counters
    uint32_t ci = 0, cj = 0, cls_i = 0u, cls_j = 0u;
    if (mine && !hot_i) {
        ci = (uint32_t)(ra.cnt_cur[i] >> 32);
        const uint4 w0 = *reinterpret_cast<const uint4 *>(ra.tab_cur + (size_t)i * kTabMax), w1 = *reinterpret_cast<const uint4 *>(ra.tab_cur + (size_t)i * kTabMax + 4);
        const uint32_t wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        uint32_t mx = wv[0];
#pragma unroll
        for (unsigned q = 1; q < kTabMax; ++q) if (q < ci && wv[q] > mx) mx = wv[q];
        cls_i = ci == 1u ? 1u : ci > kTabMax ? 4u : mx == my_slot ? 3u : 2u;
    }
    if (mine && !hot_j) {
        cj = (uint32_t)(ra.cnt_cur[j] >> 32);
        const uint4 w0 = *reinterpret_cast<const uint4 *>(ra.tab_cur + (size_t)j * kTabMax), w1 = *reinterpret_cast<const uint4 *>(ra.tab_cur + (size_t)j * kTabMax + 4);
        const uint32_t wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        uint32_t mx = wv[0];
#pragma unroll
        for (unsigned q = 1; q < kTabMax; ++q) if (q < cj && wv[q] > mx) mx = wv[q];
        cls_j = cj == 1u ? 1u : cj > kTabMax ? 4u : mx == my_slot + 1u ? 3u : 2u;
    }
    const unsigned row_bytes = k * 4u;
    unsigned vo[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; vo[r] = e < k ? e * 4u : kOobOffset; }

    unsigned u0 = 0xffffffffu;
#pragma unroll
    for (int t = 0; t < TPW; ++t) if (base + t < ra.e_end && (unsigned)hu[t] < u0) u0 = (unsigned)hu[t];
    const uint64_t qbytes = (uint64_t)a.n * row_bytes, pbytes = (uint64_t)(a.m - u0) * row_bytes;
    const int qrec = (int)(qbytes < 0x7fffffffull ? qbytes : 0x7fffffffull);
    const int prec = (int)(pbytes < 0x7fffffffull ? pbytes : 0x7fffffffull);
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(a.Q, 0, qrec, kRsrcFlags);
    const auto rsdQ = __builtin_amdgcn_make_buffer_rsrc(a.dQ, 0, qrec, kRsrcFlags);
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc(a.P + (uint64_t)u0 * k, 0, prec, kRsrcFlags);
    const auto rsdP = __builtin_amdgcn_make_buffer_rsrc(a.dP + (uint64_t)u0 * k, 0, prec, kRsrcFlags);
    // tagged staging rows of this round: 2 per event, 8 bytes per element (the host keeps the size below 2^31)
    const auto rsT = __builtin_amdgcn_make_buffer_rsrc(a.stage, 0, (int)(2u * (unsigned)(ra.e_end - ra.e_begin) * 2u * row_bytes), kRsrcFlags);

    unsigned oi[TPW], oj[TPW], ou[TPW], ru_[TPW];
    bool ok[TPW];
    float qi[TPW][KR], qj[TPW][KR], p[TPW][KR];
    // the first kHotRegs hot touches of the batch fetch their pending sums beside the gathers
    float hq[kHotRegs][KR];
    int nhq = 0;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        ru_[t] = (unsigned)hu[t];
        const int tj = hj[t];
        ok[t] = tj >= 0;                                 // no event / sampler gave up: nothing is written
        oi[t] = (unsigned)hi_[t] * row_bytes; oj[t] = (ok[t] ? (unsigned)tj : 0u) * row_bytes;
        ou[t] = (base + t < ra.e_end ? ru_[t] - u0 : 0u) * row_bytes;
#pragma unroll
        for (int r = 0; r < KR; ++r) { qi[t][r] = YUE_BLOAD(rsQ, vo[r], oi[t]); qj[t][r] = YUE_BLOAD(rsQ, vo[r], oj[t]); }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < KR; ++r) p[t][r] = YUE_BLOAD(rsP, vo[r], ou[t]);
    {
        // scalar search for the first kHotRegs hot touches: (event, role) pairs in order
        unsigned hrow[kHotRegs];
#pragma unroll
        for (int s2 = 0; s2 < kHotRegs; ++s2) hrow[s2] = 0xffffffffu;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const unsigned hmt = ok[t] ? (unsigned)hh[t] : 0u;
            if (hmt & 0xffffu) { if (nhq < kHotRegs) hrow[nhq < kHotRegs ? nhq : 0] = (hmt & 0xffffu) - 1u; ++nhq; }
            if (hmt >> 16) { if (nhq < kHotRegs) hrow[nhq < kHotRegs ? nhq : 0] = (hmt >> 16) - 1u; ++nhq; }
        }
#pragma unroll
        for (int s2 = 0; s2 < kHotRegs; ++s2) {
            const bool want = hrow[s2] != 0xffffffffu;
            const auto rsH = __builtin_amdgcn_make_buffer_rsrc(ra.hq_read + (size_t)(want ? hrow[s2] : 0u) * k, 0, want ? (int)row_bytes : 0, kRsrcFlags);
#pragma unroll
            for (int r = 0; r < KR; ++r) hq[s2][r] = YUE_BLOAD(rsH, vo[r], 0u);
        }
    }

    YUE_STAMP(2, "s_waitcnt vmcnt(0)");
    // pending sums of the hot rows are part of the round-start row
    {
        int seen = 0;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const unsigned hmt = ok[t] ? (unsigned)hh[t] : 0u;
            if (hmt & 0xffffu) {                             // wave-uniform
                if (seen < kHotRegs) {
#pragma unroll
                    for (int s2 = 0; s2 < kHotRegs; ++s2) if (s2 == seen) {
#pragma unroll
                        for (int r = 0; r < KR; ++r) qi[t][r] = qi[t][r] + hq[s2][r];
                    }
                } else {
                    const float *src = ra.hq_read + (size_t)((hmt & 0xffffu) - 1u) * k;
#pragma unroll
                    for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; if (e < k) qi[t][r] = qi[t][r] + src[e]; }
                }
                ++seen;
            }
            if (hmt >> 16) {
                if (seen < kHotRegs) {
#pragma unroll
                    for (int s2 = 0; s2 < kHotRegs; ++s2) if (s2 == seen) {
#pragma unroll
                        for (int r = 0; r < KR; ++r) qj[t][r] = qj[t][r] + hq[s2][r];
                    }
                } else {
                    const float *src = ra.hq_read + (size_t)((hmt >> 16) - 1u) * k;
#pragma unroll
                    for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; if (e < k) qj[t][r] = qj[t][r] + src[e]; }
                }
                ++seen;
            }
        }
    }
    // the next round's tickets are issued here, behind every gather: they return while this round is evaluated and stored
    take_tickets();
    float xs = 0.0f;                                     // lane t will hold the margin of triplet t
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        float ai = 0.0f, aj = 0.0f;
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const float a1 = p[t][r] * qi[t][r]; ai = ai + a1;
            const float a2 = p[t][r] * qj[t][r]; aj = aj + a2;
        }
        const float x = wave_sum(ai) - wave_sum(aj);               // BPR.py:50, fp32 margin
        xs = lane == t ? x : xs;
    }
    __builtin_amdgcn_sched_barrier(0);
    const double s = 1.0 / (1.0 + exp(-(double)xs));               // qmath.py:115-116
    const float cs = (float)(a.lr * (1.0 - s));
    double nll = mine ? -log(s) : 0.0;                             // BPR.py:58
    __builtin_amdgcn_sched_barrier(0);

    YUE_STAMP(3, "");
    // rows whose rewrite is mine: up to kKeep of them keep the old row and my difference in registers
    float kb[kKeep][KR], kd[kKeep][KR];
    int kt[kKeep], kq[kKeep];                            // event and role (0 positive, 1 negative) of the kept rows
#pragma unroll
    for (int sl = 0; sl < kKeep; ++sl) { kt[sl] = 0; kq[sl] = 0; }
    int nkept = 0;
    bool any_slow = false;
    float dp[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) dp[r] = 0.0f;
    bool run_ok = false;
    const unsigned tstride = 2u * row_bytes;             // bytes of a tagged staging row
    auto vo2 = [&](int r) -> unsigned { return vo[r] == kOobOffset ? kOobOffset : 2u * vo[r]; };       // granule offset inside a tagged row
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const float c = rdlane(cs, t);
        const unsigned cli = (unsigned)__builtin_amdgcn_readlane(cls_i, t), clj = (unsigned)__builtin_amdgcn_readlane(cls_j, t);
        const unsigned cti = (unsigned)__builtin_amdgcn_readlane(ci, t), ctj = (unsigned)__builtin_amdgcn_readlane(cj, t);
        const unsigned hmt = (unsigned)hh[t];
        const unsigned ss = 2u * ((unsigned)wave * TPW + t) * tstride;     // my two tagged staging rows
        if (ok[t]) {                                     // wave-uniform
            run_ok = true;
            // does a row of this event stay in registers for its rewrite?  (rows with more than 4 touches take the wide path)
            const bool keep_i = cli == 3u && cti <= 4u && nkept < kKeep;
            const int slot_i = nkept;
            if (keep_i) {
#pragma unroll
                for (int sl = 0; sl < kKeep; ++sl) if (sl == nkept) { kt[sl] = t; kq[sl] = 0; }
                ++nkept;
            }
            const bool keep_j = clj == 3u && ctj <= 4u && nkept < kKeep;
            const int slot_j = nkept;
            if (keep_j) {
#pragma unroll
                for (int sl = 0; sl < kKeep; ++sl) if (sl == nkept) { kt[sl] = t; kq[sl] = 1; }
                ++nkept;
            }
            if (cli == 4u || clj == 4u) any_slow = true;
            const auto rsHi = __builtin_amdgcn_make_buffer_rsrc(ra.hq_acc + (size_t)(cli == 0u ? (hmt & 0xffffu) - 1u : 0u) * k, 0, (int)row_bytes, kRsrcFlags);
            const auto rsHj = __builtin_amdgcn_make_buffer_rsrc(ra.hq_acc + (size_t)(clj == 0u ? (hmt >> 16) - 1u : 0u) * k, 0, (int)row_bytes, kRsrcFlags);
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const Elem o = bpr_elem(p[t][r], qi[t][r], qj[t][r], c, a.ru, a.ri);
                const float di = o.qi2 - qi[t][r], dj = o.qj2 - qj[t][r];
                if (cli == 1u) YUE_BSTORE(o.qi2, rsQ, vo[r], oi[t]);
                else if (cli == 0u) YUE_BATOMIC(di, rsHi, vo[r], 0u);
                else if (cli == 4u) YUE_BATOMIC(di, rsdQ, vo[r], oi[t]);
                else if (keep_i) {
#pragma unroll
                    for (int sl = 0; sl < kKeep; ++sl) if (sl == slot_i) { kb[sl][r] = qi[t][r]; kd[sl][r] = di; }
                } else YUE_TSTORE(di, ra.tag, rsT, vo2(r), ss);
                if (clj == 1u) YUE_BSTORE(o.qj2, rsQ, vo[r], oj[t]);
                else if (clj == 0u) YUE_BATOMIC(dj, rsHj, vo[r], 0u);
                else if (clj == 4u) YUE_BATOMIC(dj, rsdQ, vo[r], oj[t]);
                else if (keep_j) {
#pragma unroll
                    for (int sl = 0; sl < kKeep; ++sl) if (sl == slot_j) { kb[sl][r] = qj[t][r]; kd[sl][r] = dj; }
                } else YUE_TSTORE(dj, ra.tag, rsT, vo2(r), ss + tstride);
                dp[r] += o.p2 - p[t][r];
            }
        }
        // end of a run of equal users (or of the batch): flush the summed P[u] differences
        const bool exists = base + t < ra.e_end;
        const bool last = exists && ((t == TPW - 1) || base + t + 1 >= ra.e_end || ru_[t + 1 < TPW ? t + 1 : t] != ru_[t]);
        if (last) {
            if (run_ok) {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_BATOMIC(dp[r], rsdP, vo[r], ou[t]);
            }
#pragma unroll
            for (int r = 0; r < KR; ++r) dp[r] = 0.0f;
            run_ok = false;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    YUE_STAMP(4, "");

    // sole touchers reset their rows' counter words here; a last toucher does it once it has seen every other
    // toucher's staging row (a toucher reads the word in its header, before it writes anything)
    if (mine) {
        if (cls_i == 1u) ra.cnt_cur[i] = 0ull;
        if (cls_j == 1u) ra.cnt_cur[j] = 0ull;
    }

    // ---- rewrites by the last touchers -------------------------------------------------------------
    // the row's slot words come through the scalar unit, sorted ascending = event order (the oracle's order of the sum)
    auto row_slots = [&](unsigned row, unsigned touches, uint32_t (&w)[8]) {
        // one vector load (lane q reads word q; the scalar cache is not refreshed between launches), then scalars
        const uint32_t v = lane < (int)kTabMax ? ra.tab_cur[(size_t)row * kTabMax + lane] : 0xffffffffu;
#pragma unroll
        for (unsigned q = 0; q < kTabMax; ++q) { const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)q); w[q] = q < touches ? x : 0xffffffffu; }
        sort8_uniform(w);
    };
    // pass A: the rows kept in registers (at most three other touchers each): every staging row of every kept row is
    // requested before the first one is looked at
    if (nkept > 0) {
        unsigned so[kKeep][3], np[kKeep], orow[kKeep];
#pragma unroll
        for (int sl = 0; sl < kKeep; ++sl) {
            const bool act = sl < nkept;
            const int t = kt[sl];
            const bool neg = kq[sl] != 0;
            const unsigned touches = act ? (neg ? (unsigned)__builtin_amdgcn_readlane(cj, t) : (unsigned)__builtin_amdgcn_readlane(ci, t)) : 1u;
            const unsigned row = neg ? (unsigned)__builtin_amdgcn_readlane(j, t) : (unsigned)__builtin_amdgcn_readlane(i, t);
            uint32_t w[8];
            row_slots(row, touches, w);
            np[sl] = touches - 1u;                       // the other touchers: the first entries of the sorted table
#pragma unroll
            for (unsigned q = 0; q < 3; ++q) so[sl][q] = w[q] * tstride;
            orow[sl] = row * row_bytes;
        }
        unsigned spins = 0;
        for (;;) {
            u32x2 st[kKeep][3][KR];
#pragma unroll
            for (int sl = 0; sl < kKeep; ++sl)
#pragma unroll
                for (unsigned q = 0; q < 3; ++q)
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const bool want = sl < nkept && q < np[sl];
                        st[sl][q][r] = YUE_TLOAD(rsT, want ? vo2(r) : kOobOffset, want ? so[sl][q] : 0u);
                    }
            bool stale = false;
#pragma unroll
            for (int sl = 0; sl < kKeep; ++sl)
#pragma unroll
                for (unsigned q = 0; q < 3; ++q)
#pragma unroll
                    for (int r = 0; r < KR; ++r)
                        if (sl < nkept && q < np[sl] && 64u * r + lane < k && st[sl][q][r][1] != ra.tag) stale = true;
            if (__ballot(stale) == 0ull) {
#pragma unroll
                for (int sl = 0; sl < kKeep; ++sl)
                    if (sl < nkept) {
#pragma unroll
                        for (int r = 0; r < KR; ++r) {
                            // event order: the other touchers first (ascending), my own difference last
                            float acc = __builtin_bit_cast(float, st[sl][0][r][0]);
#pragma unroll
                            for (unsigned q = 1; q < 3; ++q) if (q < np[sl]) acc = acc + __builtin_bit_cast(float, st[sl][q][r][0]);
                            acc = acc + kd[sl][r];
                            YUE_BSTORE(kb[sl][r] + acc, rsQ, vo[r], orow[sl]);
                        }
                        if (lane == 0) ra.cnt_cur[orow[sl] / row_bytes] = 0ull;
                    }
                break;
            }
            if (++spins > kPollLimit) { if (lane == 0) atomicOr(ra.err_flag, 1); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    // pass B: last-toucher rows that did not fit pass A (a further row of the wave, or a row with 5..kTabMax touches):
    // one row at a time, all of its staging rows (my own among them) and the old row are read back
    {
        unsigned long long wi_ = __ballot(cls_i == 3u), wj_ = __ballot(cls_j == 3u);
        // drop the rows pass A took
#pragma unroll
        for (int sl = 0; sl < kKeep; ++sl) if (sl < nkept) { if (kq[sl]) wj_ &= ~(1ull << kt[sl]); else wi_ &= ~(1ull << kt[sl]); }
        while (wi_ | wj_) {
            const bool neg = wi_ == 0ull;
            const int t = neg ? __ffsll((long long)wj_) - 1 : __ffsll((long long)wi_) - 1;
            if (neg) wj_ &= wj_ - 1; else wi_ &= wi_ - 1;
            const unsigned touches = neg ? (unsigned)__builtin_amdgcn_readlane(cj, t) : (unsigned)__builtin_amdgcn_readlane(ci, t);
            const unsigned row = neg ? (unsigned)__builtin_amdgcn_readlane(j, t) : (unsigned)__builtin_amdgcn_readlane(i, t);
            uint32_t w[8];
            row_slots(row, touches, w);
            const unsigned orow = row * row_bytes;
            unsigned spins = 0;
            for (;;) {
                u32x2 st[kTabMax][KR];
#pragma unroll
                for (unsigned q = 0; q < kTabMax; ++q)
#pragma unroll
                    for (int r = 0; r < KR; ++r) st[q][r] = YUE_TLOAD(rsT, q < touches ? vo2(r) : kOobOffset, q < touches ? w[q] * tstride : 0u);
                float x[KR];
#pragma unroll
                for (int r = 0; r < KR; ++r) x[r] = YUE_BLOAD(rsQ, vo[r], orow);
                bool stale = false;
#pragma unroll
                for (unsigned q = 0; q < kTabMax; ++q)
#pragma unroll
                    for (int r = 0; r < KR; ++r)
                        if (q < touches && 64u * r + lane < k && st[q][r][1] != ra.tag) stale = true;
                if (__ballot(stale) == 0ull) {
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        float acc = __builtin_bit_cast(float, st[0][r][0]);
#pragma unroll
                        for (unsigned q = 1; q < kTabMax; ++q) if (q < touches) acc = acc + __builtin_bit_cast(float, st[q][r][0]);
                        YUE_BSTORE(x[r] + acc, rsQ, vo[r], orow);
                    }
                    if (lane == 0) ra.cnt_cur[row] = 0ull;
                    break;
                }
                if (++spins > kPollLimit) { if (lane == 0) atomicOr(ra.err_flag, 1); break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
    }
    YUE_STAMP(5, "");

    // ---- the rare rows with more than kTabMax touches outside the hot set: dQ + counters, as in k_round ----
    if (any_slow) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool last_i = false, last_j = false;
        if (mine) {
            if (cls_i == 4u) last_i = (uint32_t)atomicAdd(ra.cnt_cur + i, ~0ull) == 1u;     // -1 on the low half
            if (cls_j == 4u) last_j = (uint32_t)atomicAdd(ra.cnt_cur + j, ~0ull) == 1u;
        }
        unsigned long long win = (__ballot(last_i) & ((1ull << TPW) - 1)) | ((__ballot(last_j) & ((1ull << TPW) - 1)) << TPW);
        while (win) {
            const int b = __ffsll((long long)win) - 1;
            win &= win - 1;
            const int src = b % TPW;
            const unsigned row = b < TPW ? (unsigned)__builtin_amdgcn_readlane(i, src) : (unsigned)__builtin_amdgcn_readlane(j, src);
            float *xp = a.Q + (uint64_t)row * k, *dx = a.dQ + (uint64_t)row * k;
            if (lane == 0) ra.cnt_cur[row] = 0ull;
            for (unsigned e = lane; e < k; e += 64) { const float d = atomicExch(dx + e, 0.0f); xp[e] = xp[e] + d; }
        }
    }
    YUE_STAMP(6, "");
    publish_slots();                                     // waits for the tickets taken behind the gathers
    YUE_STAMP(7, "");
#pragma unroll
    for (int off = 1; off < TPW; off <<= 1) nll += __shfl_xor(nll, off);
    if (lane == 0 && nll != 0.0) atomicAdd(a.nll_slots + (wave & (kNllSlots - 1)), nll);
}

// folds the hot rows' pending sums into the rows and clears the three buffers (end of a group of rounds / of the epoch)
__global__ void __launch_bounds__(256) k_hot_fold(float *Q, const int32_t *hot_rows, int nhot, int k, float *cur, float *b1, float *b2) {
    const int lane = threadIdx.x & 63;
    const int nw = (int)gridDim.x * 4;
    for (int h = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); h < nhot; h += nw) {
        float *row = Q + (size_t)hot_rows[h] * k;
        for (int e = lane; e < k; e += 64) {
            const size_t o = (size_t)h * k + e;
            row[e] = row[e] + cur[o];
            cur[o] = 0.0f; b1[o] = 0.0f; b2[o] = 0.0f;
        }
    }
}

// per event: (hot index of i + 1) | (hot index of j + 1) << 16 from the per-item table (0 = not hot)
__global__ void __launch_bounds__(256) k_hot_marks(const int32_t *ev_i, const int32_t *ev_j, const uint16_t *hot_of, int64_t E, int32_t *ev_h) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int32_t j = ev_j[e];
    ev_h[e] = (int32_t)((uint32_t)hot_of[ev_i[e]] | ((j >= 0 ? (uint32_t)hot_of[j] : 0u) << 16));
}

// Multi-GPU: after the all-reduce of dP[first .. first+count) the same range is applied everywhere.
__global__ void __launch_bounds__(256) k_apply_range(float *X, float *dX, int64_t first_elem, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
        const int64_t o = first_elem + t;
        X[o] += dX[o];
        dX[o] = 0.0f;
    }
}

// BPR.py:59 -- sum of fp32 squares, accumulated in double.
__global__ void __launch_bounds__(256) k_sumsq(const float *X, int64_t count, double *out) {
    __shared__ double part[4];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
        const float sq = X[t] * X[t];
        s += (double)sq;
    }
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

__global__ void __launch_bounds__(64) k_sum_slots(const double *slots, int nslots, double *out) {
    double s = 0.0;
    for (int t = threadIdx.x; t < nslots; t += 64) s += slots[t];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) *out = s;
}

}  // namespace yue
