// Scoring + selection kernels for gfx950 (CDNA4).
//
// predict()  (recommender/cf/BPR.py:131-134): scores = Q . P[u]
// evalRanking's selection (base/IterativeRecommender.py:98-145): mask, seed with the first N
// candidates (stable descending sort), then the overwrite-scan over every candidate in id order.
//
// Numerics: a score is the k-ascending fp32 fused-multiply-add chain
//     acc = fma(P[u][e], Q[i][e], acc),  e = 0..k-1,  acc0 = 0
// which is what v_mfma_f32_32x32x2_f32 computes (one rounding per product, no wider
// accumulation) and what oracle/bpr_oracle.c:score_chain restates.  The selection compares
// fp32 scores, so identical scores give identical integer lists.
//
// k_topn_scan: a workgroup = 4 waves = 128 users; each wave keeps its 32 users' factor rows in
// registers as MFMA A-operands and sweeps all items in tiles of 32 (B-operand tile shared by the
// 4 waves through LDS, double buffered, fetched one tile ahead).  After the 32x32 tile's MFMA
// chain the scores are tested against the users' running thresholds in the accumulator registers;
// the survivors (a few per tile) are parked in LDS with a per-user column mask, and lane r (< 32)
// feeds user r's survivors in ascending item id through the reference's state machine, whose N
// slots live in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace yue {

constexpr int kWorkSlots = 1024;
__device__ __forceinline__ unsigned long long *work_slot(unsigned long long *work, unsigned wave_id) { return work + 4u * (wave_id & (kWorkSlots - 1)); }

struct ScanArgs {
    const float *P, *Q;
    int64_t n;
    int k;
    const int32_t *users;
    int64_t nu;
    int N;
    const int64_t *mask_ptr;
    const int32_t *mask_idx;
    int mask_by_user;      // 1: mask row = user id (training CSR); 0: mask row = position in users[]
    int32_t *out_ids;
    float *out_scores;
    int32_t *flags;        // [0] some user had < N candidates
    unsigned long long *work; // kWorkSlots x {[0] 32-user x 32-item tiles actually scored (the bf16 kernel skips tiles that cannot matter),
                              // [1] state-machine events, [2] exact re-scores (bf16 path), [3] unused}: a wave adds to the slot of its
                              // number (one counter for a million waves is 12 ns per add, in series); 64-bit: 1.6e9 re-scores on a full config-5 scan
    const float *tile_norm_max; // max ||Q[i]||_2 over each tile of 32 items (bf16 pre-filter margin); unused by the f32 kernel
    const float *tile_norm_sufmax; // max of tile_norm_max over this and all later tiles (early exit of the bf16 kernel)
    int true_topn;             // 0: the reference's overwrite-scan (default); 1: a real top-N (ties: lower id first)
};

// Per-user selection state: N slots in LDS + the running threshold (the reference's state machine,
// base/IterativeRecommender.py:107-145).  Shared by the exact-f32 and the bf16-pre-filter kernels.
struct ScanState {
    float *st_a;           // LDS: scores, sorted descending
    int32_t *st_id;        // LDS: item ids
    float *g_sc;           // global scratch for the unsorted seeds = this user's output row
    int32_t *g_id;
    int cnt;
    float thr;
    int events;
};

template <bool TRUE_TOPN>
__device__ __forceinline__ void scan_push_t(ScanState &S, int N, float s, int32_t item) {
    ++S.events;
    if (S.cnt < N) {
        // seed phase (:107-112): keep id order in global scratch
        if (!TRUE_TOPN) { S.g_sc[S.cnt] = s; S.g_id[S.cnt] = item; }
        // stable descending insertion (:114): later ids go after equal scores
        int p = S.cnt;
        while (p > 0 && S.st_a[p - 1] < s) { S.st_a[p] = S.st_a[p - 1]; S.st_id[p] = S.st_id[p - 1]; --p; }
        S.st_a[p] = s; S.st_id[p] = item;
        ++S.cnt;
        if (S.cnt == N) {
            if (!TRUE_TOPN) {
                // the scan (:121-144) starts over from the first candidate
                for (int q = 0; q < N; ++q) {
                    const float sq = S.g_sc[q];
                    if (S.st_a[N - 1] < sq) {
                        int p2 = N - 1;
                        while (p2 > 0 && S.st_a[p2 - 1] < sq) --p2;
                        S.st_a[p2] = sq; S.st_id[p2] = S.g_id[q];
                    }
                }
            }
            S.thr = S.st_a[N - 1];
        }
    } else if (TRUE_TOPN) {
        int p = N - 1;                         // real top-N: insert with shift, the smallest entry drops out
        while (p > 0 && S.st_a[p - 1] < s) { S.st_a[p] = S.st_a[p - 1]; S.st_id[p] = S.st_id[p - 1]; --p; }
        S.st_a[p] = s; S.st_id[p] = item;
        S.thr = S.st_a[N - 1];
    } else {
        // first slot strictly below s; a[] is sorted and a survivor usually beats only the tail,
        // so the search walks up from the last slot (same slot as the reference's binary search)
        int p = N - 1;
        while (p > 0 && S.st_a[p - 1] < s) --p;
        S.st_a[p] = s; S.st_id[p] = item;      // overwrite, no shift (:142-144)
        S.thr = S.st_a[N - 1];
    }
}

__device__ __forceinline__ void scan_push(ScanState &S, int N, float s, int32_t item, int true_topn) {
    if (true_topn) scan_push_t<true>(S, N, s, item); else scan_push_t<false>(S, N, s, item);
}

__device__ __forceinline__ void scan_finish(const ScanState &S, int N, int32_t *flags, unsigned long long *work) {      // work: the wave's slot
    if (S.cnt < N) {
        atomicOr(flags, 1);
        for (int q = 0; q < N; ++q) { S.g_sc[q] = -INFINITY; S.g_id[q] = -1; }
    } else {
        for (int q = 0; q < N; ++q) { S.g_sc[q] = S.st_a[q]; S.g_id[q] = S.st_id[q]; }
    }
    atomicAdd(work + 1, (unsigned long long)S.events);
}

// max ||Q[i]||_2 over each tile of 32 consecutive items: one wave per tile, lane pair (r, h) sums half
// of item r's squares
__global__ void __launch_bounds__(256) k_tile_norm_max(const float *Q, int64_t n, int k, float *out) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int64_t tl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tl * 32 >= n) return;
    const int64_t it = tl * 32 + r;
    float s = 0.0f;
    if (it < n) for (int e = h; e < k; e += 2) s = __builtin_fmaf(Q[it * k + e], Q[it * k + e], s);
    s += __shfl_xor(s, 32);
    for (int off = 16; off >= 1; off >>= 1) s = fmaxf(s, __shfl_xor(s, off));
    if (lane == 0) out[tl] = __builtin_sqrtf(s) * 1.0001f;      // the two half sums round differently: keep it an upper bound
}

// sufmax[t] = max(norm_max[t], norm_max[t+1], ...): one wave, chunks of 64 tiles from the back
__global__ void __launch_bounds__(64) k_tile_norm_sufmax(const float *norm_max, int64_t ntile, float *sufmax) {
    const int lane = threadIdx.x;
    float carry = 0.0f;
    for (int64_t hi = ntile; hi > 0; hi -= 64) {
        const int64_t t = hi - 64 + lane;
        float v = t >= 0 ? norm_max[t] : 0.0f;
        // inclusive max-scan towards lower lanes: lane l gets max over lanes l..63
        for (int off = 1; off < 64; off <<= 1) { const float o = __shfl_down(v, off); if (lane + off < 64) v = fmaxf(v, o); }
        v = fmaxf(v, carry);
        if (t >= 0) sufmax[t] = v;
        carry = __shfl(v, 0);
    }
}

__global__ void __launch_bounds__(256) k_scores_one(const float *pu, const float *Q, int64_t n, int k, float *out) {
    const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= n) return;
    const float *q = Q + it * k;
    float acc = 0.0f;
    for (int e = 0; e < k; ++e) acc = __builtin_fmaf(pu[e], q[e], acc);
    out[it] = acc;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kScanTile = 32;        // items per tile
constexpr int kScanWaves = 4;
constexpr int kScLd = 33;            // parked-tile row stride (floats)

__host__ __device__ inline int scan_ns(int N) { return N | 1; }            // odd slot stride
__host__ __device__ inline size_t scan_lds_bytes(int K2, int N) {
    const size_t tile = 2u * kScanTile * (2 * K2 + 1) * sizeof(float);
    const size_t park = (size_t)kScanWaves * 32 * kScLd * sizeof(float);
    const size_t state = (size_t)kScanWaves * 32 * scan_ns(N) * (sizeof(float) + sizeof(int32_t));
    const size_t sel = (size_t)kScanWaves * 32 * (sizeof(float) + sizeof(uint32_t));     // thresholds + candidate masks
    return tile + park + state + sel;
}

template <int K2>
__global__ void __launch_bounds__(256) k_topn_scan(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int LD = 2 * K2 + 1;
    const int N = a.N, NS = scan_ns(N), k = a.k;
    float *tile = reinterpret_cast<float *>(lds_raw);                       // [2][32][LD]
    float *park_all = tile + 2 * kScanTile * LD;                            // [4][32][kScLd]
    float *st_a_all = park_all + kScanWaves * 32 * kScLd;                   // [4][32][NS]
    int32_t *st_id_all = reinterpret_cast<int32_t *>(st_a_all + kScanWaves * 32 * NS);
    float *thr_all = reinterpret_cast<float *>(st_id_all + kScanWaves * 32 * NS);         // [4][32]
    uint32_t *pm_all = reinterpret_cast<uint32_t *>(thr_all + kScanWaves * 32);           // [4][32]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    float *park = park_all + w * 32 * kScLd;
    float *thr_w = thr_all + w * 32;
    uint32_t *pm_w = pm_all + w * 32;
    float *st_a = st_a_all + (w * 32 + r) * NS;       // lane r's user (used by lanes < 32)
    int32_t *st_id = st_id_all + (w * 32 + r) * NS;

    const int64_t upos = (int64_t)blockIdx.x * (kScanWaves * 32) + w * 32 + r;   // position in users[]
    const bool uvalid = upos < a.nu;
    const int32_t uid = a.users[uvalid ? upos : 0];

    // A operands: this lane supplies P[user r][2s + h] at step s (zero beyond k)
    float af[K2];
#pragma unroll
    for (int s = 0; s < K2; ++s) { const int e = 2 * s + h; af[s] = e < k ? a.P[(int64_t)uid * k + e] : 0.0f; }

    // zero the padding columns of both tile buffers once
    for (int idx = tid; idx < 2 * kScanTile * LD; idx += 256) tile[idx] = 0.0f;
    __syncthreads();

    // Item tiles travel global -> registers (issued one tile ahead, before the MFMA chain) -> LDS
    // (written after it), so the global latency hides under the previous tile's arithmetic.
    constexpr int PF = (kScanTile * 2 * K2 + 255) / 256;
    const int tile_elems = kScanTile * k;
    int lds_off[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) { const int idx = tid + 256 * q; const int row = idx / k; lds_off[q] = row * LD + (idx - row * k); }
    float pre[PF];
    auto fetch = [&](int64_t it0) {
        const int64_t limit = (a.n - it0) * k;                      // elements of Q left from this tile on
        const float *src = a.Q + it0 * k;
#pragma unroll
        for (int q = 0; q < PF; ++q) { const int idx = tid + 256 * q; pre[q] = (idx < tile_elems && idx < limit) ? src[idx] : 0.0f; }
    };
    auto commit = [&](int buf) {
        float *dst = tile + buf * kScanTile * LD;
#pragma unroll
        for (int q = 0; q < PF; ++q) if (tid + 256 * q < tile_elems) dst[lds_off[q]] = pre[q];
    };

    // per-user scan state (lanes < 32)
    ScanState S;
    S.st_a = st_a; S.st_id = st_id; S.cnt = 0; S.thr = -INFINITY; S.events = 0;
    S.g_sc = a.out_scores + (uvalid ? upos : 0) * N;      // also the unsorted-seed scratch
    S.g_id = a.out_ids + (uvalid ? upos : 0) * N;
    int64_t mcur = 0, mend = 0;
    int32_t mnext = 0x7fffffff;
    if (h == 0 && uvalid) {
        const int64_t mrow = a.mask_by_user ? (int64_t)uid : upos;
        mcur = a.mask_ptr[mrow]; mend = a.mask_ptr[mrow + 1];
        if (mcur < mend) mnext = a.mask_idx[mcur];
    }
    if (h == 0) thr_w[r] = -INFINITY;

    const int64_t ntiles = (a.n + kScanTile - 1) / kScanTile;
    fetch(0);
    commit(0);
    __syncthreads();

    for (int64_t t = 0; t < ntiles; ++t) {
        const int cur = (int)(t & 1);
        const int64_t it0 = t * kScanTile;
        const float *bt = tile + cur * kScanTile * LD + r * LD + h;
        if (t + 1 < ntiles) fetch(it0 + kScanTile);

        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
#pragma unroll
        for (int s = 0; s < K2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bt[2 * s], acc, 0, 0, 0);

        // Threshold test in the accumulator registers: acc[q] is user row (q&3)+8*(q>>2)+4*h, item
        // column r.  A stale (lower) threshold only lets more candidates through; the state
        // machine below re-checks exactly.  Per row the passing columns become a 32-bit mask.
        const bool colok = it0 + r < a.n;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
            const bool pass = colok && (thr_w[row] < acc[q]);
            const unsigned long long b = __ballot(pass);
            if (pass) park[row * kScLd + r] = acc[q];
            if (r == 0) pm_w[row] = h ? (uint32_t)(b >> 32) : (uint32_t)b;
        }
        if (t + 1 < ntiles) commit(cur ^ 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        if (h == 0 && uvalid) {
            // masked columns of this tile for my user
            uint32_t mb = 0u;
            while (mnext < it0 + kScanTile) {
                mb |= 1u << (uint32_t)(mnext - it0);
                ++mcur;
                mnext = mcur < mend ? a.mask_idx[mcur] : 0x7fffffff;
            }
            uint32_t cand = pm_w[r] & ~mb;
            const float *row = park + r * kScLd;
            while (cand) {
                const int c = __ffs(cand) - 1;
                cand &= cand - 1;
                const float s = row[c];
                if (S.cnt == N && !(S.thr < s)) continue;
                scan_push(S, N, s, (int32_t)(it0 + c), a.true_topn);
            }
            thr_w[r] = S.thr;
        }
        __syncthreads();
    }

    unsigned long long *wk = work_slot(a.work, blockIdx.x * kScanWaves + w);
    if (h == 0 && uvalid) scan_finish(S, N, a.flags, wk);
    if (lane == 0) atomicAdd(wk, (unsigned long long)ntiles);
}

// ------------------------------------------------------------------------------------------
// bf16 pre-filter in front of the exact path (k = 16*K16 <= 128).
// The 32x32 tile of scores comes from v_mfma_f32_32x32x16_bf16 on bf16-rounded factors (1/16 of the
// f32-MFMA time).  Rounding a factor to bf16 (8 significant bits, to nearest) moves it by at most 2^-8 of
// its size, a product of two rounded factors by at most (2^-7 + 2^-16) |p_e q_e|; the fp32 accumulation of the
// MFMA and of the exact chain add a few 2^-17 sum|p_e q_e|.  Hence |bf16 score - exact score| <= 1.005 * 2^-7 *
// sum|p_e q_e| <= 1.005 * 2^-7 ||P_u|| ||Q_i||, and a pair can only matter to the state machine if
// bf16 score + 1.01 * 2^-7 ||P_u|| max||Q_tile|| > threshold_u.  Those few survivors are re-scored exactly -- the same k-ascending
// fp32 fma chain as k_topn_scan / the oracle, P row in registers, Q row from the fp32 LDS tile --
// and only the exact score enters the state machine: results are identical to the f32 kernel.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// T = tiles whose survivors are re-scored together (the fp32 tiles live in a ring of T + 1 LDS buffers), WAVES = waves
// (of 32 users) per workgroup.  Per phase of the re-score pipeline a wave runs ONE chain for every user that still has a
// survivor: the phases of a batch number max over the wave's users of their survivors (+ 1); with the survivors of T
// tiles pooled that maximum grows far slower than T (measured on C5, random factors: ~3 phases per tile at T = 1).
// Exchange between the two lanes (r, 0) / (r, 1) = lanes r and r + 32 of a user: v_permlane32_swap (gfx950) instead of a
// ds_bpermute round trip through the LDS.  lo: every lane gets the value of lane (lane & 31); hi: of lane 32 + (lane & 31).
__device__ __forceinline__ void half_bcast(uint32_t v, uint32_t &lo, uint32_t &hi) {
    // (inline assembly: see wave_sum in train_kernels.hpp about the builtin)
    lo = v; hi = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
}
__device__ __forceinline__ uint32_t lo_bcast(uint32_t v) { uint32_t lo, hi; half_bcast(v, lo, hi); return lo; }
__device__ __forceinline__ uint32_t hi_bcast(uint32_t v) { uint32_t lo, hi; half_bcast(v, lo, hi); return hi; }
__device__ __forceinline__ float lo_bcast(float v) { return __builtin_bit_cast(float, lo_bcast(__builtin_bit_cast(uint32_t, v))); }
__device__ __forceinline__ float hi_bcast(float v) { return __builtin_bit_cast(float, hi_bcast(__builtin_bit_cast(uint32_t, v))); }

constexpr int kScanBfPad = 8;        // bf16 elements of padding per tile row: rows 16 lanes apart fall on different banks
__host__ __device__ inline size_t scan_bf16_lds_bytes(int k, int N, int T, int waves) {
    const size_t tile = (size_t)(T + 1) * kScanTile * (k + 4) * sizeof(float) + 2u * kScanTile * (k + kScanBfPad) * 2u;
    const size_t state = (size_t)waves * 32 * scan_ns(N) * (sizeof(float) + sizeof(int32_t));
    return tile + state;
}

template <int K16, int T, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 512 / (64 * WAVES)) k_topn_scan_bf16(ScanArgs a) {
    static_assert(T == 1 || T == 2 || T == 4, "survivor words: one 32-bit mask per tile, at most 128 bits");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int K = 16 * K16, LD = K + 4, R = T + 1, NT = 64 * WAVES;
    const int N = a.N, NS = scan_ns(N);
    // Two copies of an item tile: fp32 rows in a ring of R buffers (read by the exact re-score chains of the batch), and the
    // same rows rounded to bf16 in a double buffer -- converted ONCE per workgroup when the tile is committed; every wave
    // reads its MFMA operands from there (half the LDS bytes of the fp32 rows, no conversion in the loop).
    constexpr int LDB = K + kScanBfPad;
    float *tile = reinterpret_cast<float *>(lds_raw);                       // [R][32][LD] fp32
    __bf16 *btile = reinterpret_cast<__bf16 *>(tile + R * kScanTile * LD);  // [2][32][LDB] bf16
    float *st_a_all = reinterpret_cast<float *>(btile + 2 * kScanTile * LDB);
    int32_t *st_id_all = reinterpret_cast<int32_t *>(st_a_all + WAVES * 32 * NS);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    const int64_t upos = (int64_t)blockIdx.x * (WAVES * 32) + w * 32 + r;
    const bool uvalid = upos < a.nu;
    const int32_t uid = a.users[uvalid ? upos : 0];
    const float *prow = a.P + (int64_t)uid * K;

    // exact fp32 row of user r for the re-score chain, split over its two lanes: lane (r,h) keeps
    // elements [h*K/2, (h+1)*K/2).  bf16 A fragments: lane (r,h) supplies A[row r][16*s + 8*h + j].
    constexpr int KH = K / 2;
    float mu;                                                  // margin factor of this lane's user
    float pf[KH];
#pragma unroll
    for (int e = 0; e < KH; e += 4) { const f32x4 v = *reinterpret_cast<const f32x4 *>(prow + h * KH + e); pf[e] = v[0]; pf[e + 1] = v[1]; pf[e + 2] = v[2]; pf[e + 3] = v[3]; }
    bf16x8 af[K16];
#pragma unroll
    for (int s = 0; s < K16; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) af[s][jj] = (__bf16)prow[16 * s + 8 * h + jj];
    {
        float ss = 0.0f;
#pragma unroll
        for (int e = 0; e < KH; ++e) ss = __builtin_fmaf(pf[e], pf[e], ss);
        ss += __shfl_xor(ss, 32);
        mu = __builtin_sqrtf(ss) * (1.01f / 128.0f);           // 2^-7 ||P_u||, 1 % slack for the norm roundings
    }
    // Cauchy-Schwarz: no exact score of a tile exceeds ||P_u|| * max ||Q_i|| (both rounded up: the chain's own rounding,
    // k * 2^-24 relative, is far inside the 1e-4 slacks).  A full list whose threshold is at or above that bound cannot
    // change in the tile: the wave skips the tile when that holds for all its users, the workgroup stops when it holds
    // for all its users against the largest norm of ALL remaining tiles.
    const float pn = mu * (128.0f * 1.0001f / 1.01f);

    // item tiles: global -> registers (one tile ahead) -> LDS, float4 granularity
    constexpr int PF4 = (kScanTile * K / 4 + NT - 1) / NT;
    int lds_off[PF4], ldb_off[PF4];
#pragma unroll
    for (int q = 0; q < PF4; ++q) {
        const int el = (tid + NT * q) * 4; const int row = el / K;
        lds_off[q] = row * LD + (el - row * K); ldb_off[q] = row * LDB + (el - row * K);
    }
    f32x4 pre[PF4];
    float nu_next = 0.0f;
    auto fetch = [&](int64_t it0) {
        const int64_t limit = (a.n - it0) * K;
        const float *src = a.Q + it0 * K;
#pragma unroll
        for (int q = 0; q < PF4; ++q) {
            const int el = (tid + NT * q) * 4;
            pre[q] = (el < kScanTile * K && el < limit) ? *reinterpret_cast<const f32x4 *>(src + el) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        nu_next = a.tile_norm_max[it0 / kScanTile];             // largest ||Q_i|| of the tile (wave-uniform)
    };
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    auto commit = [&](int buf, int bbuf) {
        float *dst = tile + buf * kScanTile * LD;
        __bf16 *bdst = btile + bbuf * kScanTile * LDB;
#pragma unroll
        for (int q = 0; q < PF4; ++q)
            if ((tid + NT * q) * 4 < kScanTile * K) {
                *reinterpret_cast<f32x4 *>(dst + lds_off[q]) = pre[q];
                bf16x4 b;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) b[jj] = (__bf16)pre[q][jj];
                *reinterpret_cast<bf16x4 *>(bdst + ldb_off[q]) = b;
            }
    };

    ScanState S;
    S.st_a = st_a_all + (w * 32 + r) * NS; S.st_id = st_id_all + (w * 32 + r) * NS;
    S.cnt = 0; S.thr = -INFINITY; S.events = 0;
    S.g_sc = a.out_scores + (uvalid ? upos : 0) * N;
    S.g_id = a.out_ids + (uvalid ? upos : 0) * N;
    int rescored = 0;
    unsigned tiles_done = 0;                                   // tiles this wave scored (wave-uniform)
    float thr_lane = -INFINITY;
    int64_t mcur = 0, mend = 0;
    int32_t mnext = 0x7fffffff;
    if (h == 0 && uvalid) {
        const int64_t mrow = a.mask_by_user ? (int64_t)uid : upos;
        mcur = a.mask_ptr[mrow]; mend = a.mask_ptr[mrow + 1];
        if (mcur < mend) mnext = a.mask_idx[mcur];
    }
    // survivors of the running batch of T tiles (lanes (r,0)): bit 32 * (tile in batch) + column
    unsigned long long cand_lo = 0ull, cand_hi = 0ull;

    const int64_t ntiles = (a.n + kScanTile - 1) / kScanTile;
    fetch(0);
    commit(0, 0);
    float nu = nu_next;
    __syncthreads();

    for (int64_t t = 0; t < ntiles; ++t) {
        const int tb_in = (int)(t % T);                        // position of this tile in its batch
        const int64_t it0 = t * kScanTile;
        const __bf16 *tbb = btile + (int)(t & 1) * kScanTile * LDB;
        if (t + 1 < ntiles) fetch(it0 + kScanTile);
        // can any user of this wave still change in this tile?  (mask cursors of skipped tiles catch up in the next scanned one)
        const bool settled = !(h == 0 && uvalid) || (S.cnt == N && pn * nu <= S.thr);
        if (__ballot(!settled) != 0ull) {
        ++tiles_done;

        // bf16 scores with ITEMS as rows and USERS as columns: lane (r,h) supplies A[item r][16*s+8*h+j]
        // from item r's fp32 row in the tile and B[16*s+8*h+j][user r] from its user's fragments, and
        // receives D[item (q&3)+8*(q>>2)+4*h][user r]: all 32 scores of user r sit on lanes r and r+32.
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
        const __bf16 *irow = tbb + r * LDB + 8 * h;
#pragma unroll
        for (int s = 0; s < K16; ++s) {
            const bf16x8 itf = *reinterpret_cast<const bf16x8 *>(irow + 16 * s);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(itf, af[s], acc, 0, 0, 0);
        }

        // pre-filter in registers: the user's threshold minus the margin is one per-lane value
        const float bar = thr_lane - mu * nu;
        uint32_t pmask = 0u;
#pragma unroll
        for (int q = 0; q < 16; ++q) pmask |= (bar < acc[q] ? 1u : 0u) << ((q & 3) + 8 * (q >> 2));
        pmask <<= 4 * h;                                       // rows of lane (r,1) are shifted by 4
        { uint32_t plo, phi; half_bcast(pmask, plo, phi); pmask = plo | phi; }    // both lanes of a user now hold all 32 columns
        const int64_t left = a.n - it0;
        if (left < kScanTile) pmask &= (1u << (uint32_t)left) - 1u;

        if (h == 0 && uvalid) {
            uint32_t mb = 0u;
            while (mnext < it0 + kScanTile) {
                if (mnext >= it0) mb |= 1u << (uint32_t)(mnext - it0);      // masked items of skipped tiles just pass by
                ++mcur;
                mnext = mcur < mend ? a.mask_idx[mcur] : 0x7fffffff;
            }
#if defined(YUE_SCAN_ABL) && YUE_SCAN_ABL == 1       // timing-only ablation of a diagnostic build: no survivors after the seeds
            const unsigned long long cd = S.cnt < N ? (unsigned long long)(pmask & ~mb) : 0ull;
#else
            const unsigned long long cd = (unsigned long long)(pmask & ~mb);
#endif
            if (tb_in < 2) cand_lo |= cd << (32 * (tb_in & 1)); else cand_hi |= cd << (32 * (tb_in & 1));
        }
        }
        // Survivors of the batch, once its last tile is in (or the scan ends).  Exact score = k-ascending fp32 fma chain
        // (== v_mfma_f32_32x32x2_f32, == oracle), run as a two-stage pipeline over the user's two lanes: in every phase
        // lane (r,0) takes the user's next survivor through elements [0,K/2) while lane (r,1) finishes the previous one
        // through [K/2,K) from the partial handed over -- one chain execution per phase for the whole wave, survivors
        // complete in ascending item order.  Until then the thresholds of the filter are those of the previous batch: a
        // lower threshold only lets more candidates through, the state machine re-checks with the exact score.
        if (tb_in == T - 1 || t + 1 == ntiles) {
            const int64_t tfirst = t - tb_in;                  // first tile of the batch
            const int64_t it_batch = tfirst * kScanTile;
            int c_prev = -1;
            float s_prev = 0.0f;
            for (;;) {
                int c_pop = -1;
                if (h == 0) { if (cand_lo) c_pop = __ffsll((long long)cand_lo) - 1; else if (cand_hi) c_pop = 64 + __ffsll((long long)cand_hi) - 1; }
                const int c_from = (int)lo_bcast((uint32_t)c_prev);
                const float s_from = lo_bcast(s_prev);
                const int c_cur = h ? c_from : c_pop;
                float sc = h ? s_from : 0.0f;
                if (__ballot(c_cur >= 0) == 0ull) break;
                if (c_pop >= 0) { if (c_pop < 64) cand_lo &= cand_lo - 1; else cand_hi &= cand_hi - 1; }
                if (c_cur >= 0) {
                    const int slot = (int)((tfirst + (c_cur >> 5)) % R);
                    const float *qrow = tile + (slot * kScanTile + (c_cur & 31)) * LD + h * KH;
#pragma unroll
                    for (int e = 0; e < KH; e += 4) {
                        const f32x4 qv = *reinterpret_cast<const f32x4 *>(qrow + e);
                        sc = __builtin_fmaf(pf[e], qv[0], sc); sc = __builtin_fmaf(pf[e + 1], qv[1], sc);
                        sc = __builtin_fmaf(pf[e + 2], qv[2], sc); sc = __builtin_fmaf(pf[e + 3], qv[3], sc);
                    }
                }
                const float s_done = hi_bcast(sc);                // finished scores travel back to lane (r,0)
                const int c_done = (int)hi_bcast((uint32_t)c_cur);
                if (h == 0 && c_done >= 0) {
                    ++rescored;
                    if (!(S.cnt == N && !(S.thr < s_done))) scan_push(S, N, s_done, (int32_t)(it_batch + c_done), a.true_topn);
                }
                c_prev = c_cur;
                s_prev = sc;
            }
            thr_lane = lo_bcast(S.thr);                            // lane (r,1) filters with its user's threshold too
        }
        if (t + 1 < ntiles) commit((int)((t + 1) % R), (int)((t + 1) & 1));
        nu = nu_next;
        if ((t & 15) == 15 && t + 1 < ntiles) {
            // every 16 tiles (a batch boundary): does any user of the workgroup still have a chance in ANY remaining tile?
            const bool done = !(h == 0 && uvalid) || (S.cnt == N && pn * a.tile_norm_sufmax[t + 1] <= S.thr);
            if (!__syncthreads_or(!done)) break;
        } else {
            __syncthreads();
        }
    }

    unsigned long long *wk = work_slot(a.work, blockIdx.x * WAVES + w);
    if (h == 0 && uvalid) { scan_finish(S, N, a.flags, wk); atomicAdd(wk + 2, (unsigned long long)rescored); }
    if (lane == 0) atomicAdd(wk, (unsigned long long)tiles_done);
}

// Variant with TP tiles per loop iteration (one barrier per TP tiles): the MFMA chains of the iteration's tiles are
// independent, so the pre-filter arithmetic of one tile runs while the matrix pipe works on the next; survivors of the TP
// tiles are re-scored together at the end of the iteration.  LDS: two stages of TP tiles, fp32 and bf16 copies.
__host__ __device__ inline size_t scan_bf16p_lds_bytes(int k, int N, int TP, int waves) {
    const size_t tile = (size_t)(2 * TP) * kScanTile * ((k + 4) * sizeof(float) + (k + kScanBfPad) * 2u);
    const size_t state = (size_t)waves * 32 * scan_ns(N) * (sizeof(float) + sizeof(int32_t));
    return tile + state;
}

template <int K16, int TP, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 512 / (64 * WAVES)) k_topn_scan_bf16p(ScanArgs a) {
    static_assert(TP == 2, "survivor word: 64 bits");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int K = 16 * K16, LD = K + 4, LDB = K + kScanBfPad, NT = 64 * WAVES;
    const int N = a.N, NS = scan_ns(N);
    float *tile = reinterpret_cast<float *>(lds_raw);                                // [2][TP][32][LD] fp32
    __bf16 *btile = reinterpret_cast<__bf16 *>(tile + 2 * TP * kScanTile * LD);      // [2][TP][32][LDB] bf16
    float *st_a_all = reinterpret_cast<float *>(btile + 2 * TP * kScanTile * LDB);
    int32_t *st_id_all = reinterpret_cast<int32_t *>(st_a_all + WAVES * 32 * NS);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t upos = (int64_t)blockIdx.x * (WAVES * 32) + w * 32 + r;
    const bool uvalid = upos < a.nu;
    const int32_t uid = a.users[uvalid ? upos : 0];
    const float *prow = a.P + (int64_t)uid * K;

    constexpr int KH = K / 2;
    float mu;
    float pf[KH];
#pragma unroll
    for (int e = 0; e < KH; e += 4) { const f32x4 v = *reinterpret_cast<const f32x4 *>(prow + h * KH + e); pf[e] = v[0]; pf[e + 1] = v[1]; pf[e + 2] = v[2]; pf[e + 3] = v[3]; }
    bf16x8 af[K16];
#pragma unroll
    for (int s = 0; s < K16; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) af[s][jj] = (__bf16)prow[16 * s + 8 * h + jj];
    {
        float ss = 0.0f;
#pragma unroll
        for (int e = 0; e < KH; ++e) ss = __builtin_fmaf(pf[e], pf[e], ss);
        ss = lo_bcast(ss) + hi_bcast(ss);                      // sum of the two halves on both lanes
        mu = __builtin_sqrtf(ss) * (1.01f / 128.0f);
    }
    const float pn = mu * (128.0f * 1.0001f / 1.01f);

    // one stage = TP consecutive tiles = TP * 32 rows of Q, contiguous in memory
    constexpr int ROWS = TP * kScanTile;
    constexpr int PF4 = (ROWS * K / 4 + NT - 1) / NT;
    int lds_off[PF4], ldb_off[PF4];
#pragma unroll
    for (int q = 0; q < PF4; ++q) {
        const int el = (tid + NT * q) * 4; const int row = el / K;
        lds_off[q] = row * LD + (el - row * K); ldb_off[q] = row * LDB + (el - row * K);
    }
    f32x4 pre[PF4];
    float nu_next[TP];
    const int64_t ntiles = (a.n + kScanTile - 1) / kScanTile;
    constexpr bool kWhole = (ROWS * K / 4) % NT == 0;          // every thread moves whole float4s of a stage (K >= 32)
    auto fetch = [&](int64_t it0) {
        const float *src = a.Q + it0 * K;
        if (kWhole && it0 + ROWS <= a.n) {                     // a whole stage: no per-element bounds (all but the last iteration)
#pragma unroll
            for (int q = 0; q < PF4; ++q) pre[q] = *reinterpret_cast<const f32x4 *>(src + (tid + NT * q) * 4);
#pragma unroll
            for (int q = 0; q < TP; ++q) nu_next[q] = a.tile_norm_max[it0 / kScanTile + q];
        } else {
            const int64_t limit = (a.n - it0) * K;
#pragma unroll
            for (int q = 0; q < PF4; ++q) {
                const int el = (tid + NT * q) * 4;
                pre[q] = (el < ROWS * K && el < limit) ? *reinterpret_cast<const f32x4 *>(src + el) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int q = 0; q < TP; ++q) { const int64_t tl = it0 / kScanTile + q; nu_next[q] = tl < ntiles ? a.tile_norm_max[tl] : 0.0f; }
        }
    };
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    auto commit = [&](int stage) {
        float *dst = tile + stage * ROWS * LD;
        __bf16 *bdst = btile + stage * ROWS * LDB;
#pragma unroll
        for (int q = 0; q < PF4; ++q)
            if (kWhole || (tid + NT * q) * 4 < ROWS * K) {
                *reinterpret_cast<f32x4 *>(dst + lds_off[q]) = pre[q];
                bf16x4 b;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) b[jj] = (__bf16)pre[q][jj];
                *reinterpret_cast<bf16x4 *>(bdst + ldb_off[q]) = b;
            }
    };

    ScanState S;
    S.st_a = st_a_all + (w * 32 + r) * NS; S.st_id = st_id_all + (w * 32 + r) * NS;
    S.cnt = 0; S.thr = -INFINITY; S.events = 0;
    S.g_sc = a.out_scores + (uvalid ? upos : 0) * N;
    S.g_id = a.out_ids + (uvalid ? upos : 0) * N;
    int rescored = 0;
    unsigned tiles_done = 0;
    float thr_lane = -INFINITY;
    int64_t mcur = 0, mend = 0;
    int32_t mnext = 0x7fffffff;
    if (h == 0 && uvalid) {
        const int64_t mrow = a.mask_by_user ? (int64_t)uid : upos;
        mcur = a.mask_ptr[mrow]; mend = a.mask_ptr[mrow + 1];
        if (mcur < mend) mnext = a.mask_idx[mcur];
    }

    const int64_t niter = (ntiles + TP - 1) / TP;
    fetch(0);
    commit(0);
    float nu[TP];
#pragma unroll
    for (int q = 0; q < TP; ++q) nu[q] = nu_next[q];
    __syncthreads();

    for (int64_t it = 0; it < niter; ++it) {
        const int stage = (int)(it & 1);
        const int64_t it0 = it * ROWS;                                 // first item of the iteration
        const float *tb = tile + stage * ROWS * LD;
        const __bf16 *tbb = btile + stage * ROWS * LDB;
#if defined(YUE_SCAN_ABL) && (YUE_SCAN_ABL & 2)     // timing-only ablation: no global loads of item tiles after the first
        (void)0;
#else
        if (it + 1 < niter) fetch(it0 + ROWS);
#endif
        float numax = nu[0];
#pragma unroll
        for (int q = 1; q < TP; ++q) numax = fmaxf(numax, nu[q]);
        const bool settled = !(h == 0 && uvalid) || (S.cnt == N && pn * numax <= S.thr);
        unsigned long long cand = 0ull;
        if (__ballot(!settled) != 0ull) {
            tiles_done += (unsigned)min((int64_t)TP, ntiles - it * TP);
            f32x16 acc[TP];
#pragma unroll
            for (int q = 0; q < TP; ++q)
#pragma unroll
                for (int z = 0; z < 16; ++z) acc[q][z] = 0.0f;
#pragma unroll
            for (int q = 0; q < TP; ++q) {
                const __bf16 *irow = tbb + (q * kScanTile + r) * LDB + 8 * h;
#pragma unroll
                for (int s = 0; s < K16; ++s) {
                    const bf16x8 itf = *reinterpret_cast<const bf16x8 *>(irow + 16 * s);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(itf, af[s], acc[q], 0, 0, 0);
                }
            }
            uint32_t pm[TP];
#pragma unroll
            for (int q = 0; q < TP; ++q) {
                // bar < score  <=>  (bar - score) carries the sign bit (a difference of two different floats is never +0):
                // the 16 sign bits are shifted into one word (v_sub + v_alignbit per score, no condition codes), bit z = acc[z],
                // then the four nibbles (rows 8g .. 8g+3 of the 32x32 block, + 4h) are spread to the item columns
                const float bar = thr_lane - mu * nu[q];
                uint32_t bits = 0u;
#pragma unroll
                for (int z = 15; z >= 0; --z) bits = __builtin_amdgcn_alignbit(bits, __builtin_bit_cast(uint32_t, bar - acc[q][z]), 31);
                const uint32_t pmask = (bits & 0xFu) | ((bits & 0xF0u) << 4) | ((bits & 0xF00u) << 8) | ((bits & 0xF000u) << 12);
                pm[q] = pmask << (4 * h);
            }
            {   // both lanes of a user get all 64 columns: one swap per 32-bit word
                uint32_t l0, h0, l1, h1;
                half_bcast(pm[0], l0, h0); half_bcast(pm[1], l1, h1);
                pm[0] = l0 | h0; pm[1] = l1 | h1;
            }
            const int64_t left = a.n - it0;
            unsigned long long pm64 = (unsigned long long)pm[0] | ((unsigned long long)pm[1] << 32);
            if (left < ROWS) pm64 &= (1ull << (unsigned)left) - 1ull;
            if (h == 0 && uvalid) {
                unsigned long long mb = 0ull;
                while (mnext < it0 + ROWS) {
                    if (mnext >= it0) mb |= 1ull << (unsigned)(mnext - it0);      // masked items of skipped iterations just pass by
                    ++mcur;
                    mnext = mcur < mend ? a.mask_idx[mcur] : 0x7fffffff;
                }
#if defined(YUE_SCAN_ABL) && (YUE_SCAN_ABL & 1)     // timing-only ablation: no survivors after the seeds
                cand = S.cnt < N ? (pm64 & ~mb) : 0ull;
#else
                cand = pm64 & ~mb;
#endif
            }
            // survivors of the iteration: the two-lane re-score pipeline of k_topn_scan_bf16 (not entered when no user of
            // the wave has one: the common case on trained factors)
            int c_prev = -1;
            float s_prev = 0.0f;
            if (__ballot(cand != 0ull) != 0ull) {
            for (;;) {
                const int c_pop = (h == 0 && cand) ? __ffsll((long long)cand) - 1 : -1;
                const int c_from = (int)lo_bcast((uint32_t)c_prev);
                const float s_from = lo_bcast(s_prev);
                const int c_cur = h ? c_from : c_pop;
                float sc = h ? s_from : 0.0f;
                if (__ballot(c_cur >= 0) == 0ull) break;
                if (c_pop >= 0) cand &= cand - 1;
                if (c_cur >= 0) {
                    const float *qrow = tb + c_cur * LD + h * KH;
#pragma unroll
                    for (int e = 0; e < KH; e += 4) {
                        const f32x4 qv = *reinterpret_cast<const f32x4 *>(qrow + e);
                        sc = __builtin_fmaf(pf[e], qv[0], sc); sc = __builtin_fmaf(pf[e + 1], qv[1], sc);
                        sc = __builtin_fmaf(pf[e + 2], qv[2], sc); sc = __builtin_fmaf(pf[e + 3], qv[3], sc);
                    }
                }
                const float s_done = hi_bcast(sc);
                const int c_done = (int)hi_bcast((uint32_t)c_cur);
                if (h == 0 && c_done >= 0) {
                    ++rescored;
                    if (!(S.cnt == N && !(S.thr < s_done))) scan_push(S, N, s_done, (int32_t)(it0 + c_done), a.true_topn);
                }
                c_prev = c_cur;
                s_prev = sc;
            }
            thr_lane = lo_bcast(S.thr);
            }
        }
        if (it + 1 < niter) commit(stage ^ 1);
#pragma unroll
        for (int q = 0; q < TP; ++q) nu[q] = nu_next[q];
        if ((it & 7) == 7 && it + 1 < niter) {
            const bool done = !(h == 0 && uvalid) || (S.cnt == N && pn * a.tile_norm_sufmax[(it + 1) * TP] <= S.thr);
            if (!__syncthreads_or(!done)) break;
        } else {
#if defined(YUE_SCAN_ABL) && (YUE_SCAN_ABL & 4)     // timing-only ablation: no barrier between the iterations
            __builtin_amdgcn_wave_barrier();
#else
            __syncthreads();
#endif
        }
    }

    unsigned long long *wk = work_slot(a.work, blockIdx.x * WAVES + w);
    if (h == 0 && uvalid) { scan_finish(S, N, a.flags, wk); atomicAdd(wk + 2, (unsigned long long)rescored); }
    if (lane == 0) atomicAdd(wk, (unsigned long long)tiles_done);
}

template <int K16, int TP, int WAVES>
inline void launch_scan_bf16p(const ScanArgs &a, hipStream_t stream) {
    const size_t lds = scan_bf16p_lds_bytes(a.k, a.N, TP, WAVES);
    const dim3 grid((unsigned)((a.nu + WAVES * 32 - 1) / (WAVES * 32)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_topn_scan_bf16p<K16, TP, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_topn_scan_bf16p<K16, TP, WAVES>), grid, dim3(64 * WAVES), lds, stream, a);
}

template <int K2>
inline void launch_scan_f32(const ScanArgs &a, hipStream_t stream, dim3 grid, size_t lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_topn_scan<K2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_topn_scan<K2>, grid, dim3(256), lds, stream, a);
}
template <int K16, int T, int WAVES>
inline void launch_scan_bf16(const ScanArgs &a, hipStream_t stream) {
    const size_t lds = scan_bf16_lds_bytes(a.k, a.N, T, WAVES);
    const dim3 grid((unsigned)((a.nu + WAVES * 32 - 1) / (WAVES * 32)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_topn_scan_bf16<K16, T, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_topn_scan_bf16<K16, T, WAVES>), grid, dim3(64 * WAVES), lds, stream, a);
}
template <int K16>
inline void launch_scan_bf16_cfg(const ScanArgs &a, hipStream_t stream, int batch) {
    // batch 0 (default): k_topn_scan_bf16p, two tiles per iteration, 8 waves of 32 users per workgroup -- if the N slots
    // of 256 users fit beside the tiles in the CU's 160 KB; batch 1 (and the fallback): k_topn_scan_bf16 with one tile
    // per iteration and 4 waves per workgroup.  (Measured on C5, 1M users, random factors: 251 ms against 325 ms; batches
    // of 2 / 4 tiles in the one-tile kernel 288 / 267 ms -- profiles/README.md.)
    if (batch != 1 && scan_bf16p_lds_bytes(a.k, a.N, 2, 8) <= 160u * 1024u) launch_scan_bf16p<K16, 2, 8>(a, stream);
    else launch_scan_bf16<K16, 1, 4>(a, stream);
}

// force_f32 != 0: always the exact-f32-MFMA kernel.  Otherwise k in {16,32,64,128} takes the bf16
// pre-filter kernel (identical results), anything else the f32 kernel.  Returns 1 if bf16 was used.
inline int launch_scan(const ScanArgs &a, hipStream_t stream, int force_f32, int batch = 0) {
    if (a.k > 256 || a.N > 100) return -1;
    const dim3 grid((unsigned)((a.nu + kScanWaves * 32 - 1) / (kScanWaves * 32)));
    if (!force_f32 && (a.k == 16 || a.k == 32 || a.k == 64 || a.k == 128)) {
        switch (a.k) {
            case 16: launch_scan_bf16_cfg<1>(a, stream, batch); break;
            case 32: launch_scan_bf16_cfg<2>(a, stream, batch); break;
            case 64: launch_scan_bf16_cfg<4>(a, stream, batch); break;
            default: launch_scan_bf16_cfg<8>(a, stream, batch); break;
        }
        return 1;
    }
    const int kp = a.k + (a.k & 1);
    const int K2 = kp / 2 <= 8 ? 8 : kp / 2 <= 16 ? 16 : kp / 2 <= 32 ? 32 : kp / 2 <= 64 ? 64 : 128;      // k <= 256, as the training kernels
    const size_t lds = scan_lds_bytes(K2, a.N);
    if (lds > 160u * 1024u) return -1;                     // 128 < k <= 256 together with N > 66: the N slots of 128 users no longer fit beside the tiles
    switch (K2) {
        case 8: launch_scan_f32<8>(a, stream, grid, lds); break;
        case 16: launch_scan_f32<16>(a, stream, grid, lds); break;
        case 32: launch_scan_f32<32>(a, stream, grid, lds); break;
        case 64: launch_scan_f32<64>(a, stream, grid, lds); break;
        default: launch_scan_f32<128>(a, stream, grid, lds); break;
    }
    return 0;
}

}  // namespace yue
