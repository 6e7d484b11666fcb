// Exact sequential semantics of the reference's epoch loop (recommender/cf/BPR.py:40-62) as a DATAFLOW kernel.
//
// The reference applies the triplets strictly one after the other.  Two triplets commute unless they share a row (P[u],
// Q[i] or Q[j]), so the loop's result is fixed by the order in which every single ROW sees its touches -- the stream order.
// k_bpr_chain runs exactly that partial order and nothing more:
//   * a pre-pass (chain_host.hip) gives every touch of an item row its ORDINAL: the number of earlier touches of that row
//     in the stream (a stable sort of the 2T touches by row);
//   * item rows live in a versioned copy Qv made of naturally aligned GRANULES {values, version}: version = number of
//     touches the row has received so far.  A granule is 8 bytes {element, version}, written by ONE sc1 (write-through)
//     store and read by sc1 loads: the hand-off needs no flag, no fence and no drain -- a reader that finds version == its
//     ordinal on all granules of a row holds exactly the row its predecessor wrote (MI355X_MICROARCH.md, price list:
//     handoff-1to1, "data-tagged granules"; ~1-3 us per hop).  (A 16-byte granule {value, value, version} was tried: its
//     8-byte halves do tear -- one wrong row in ~1e5 showed up in the loss.)
//   * one wave walks one RUN of consecutive triplets with the same user (user-major events: a run = a user), P[u] stays in
//     registers for the whole run; per triplet it waits until both item rows carry its ordinals, applies the reference's
//     update (bpr_device.hpp: same arithmetic as every other training kernel), and stores the two rows with version + 1;
//   * runs are claimed IN STREAM ORDER from one device counter by resident waves, so the earliest unfinished run never
//     waits for an unclaimed one: every wait ends (the predecessor of a touch belongs to an earlier run, which is held by
//     a running wave, or to the same run).
// The critical path is the hottest row's chain of hand-offs (BASELINE config 3: item 0 takes 1/447 of 50M positives =
// 112K hops), not a launch or a grid barrier per dependency level.
//
// General streams (yue_bpr_replay: any (u, i, j) order) version the user rows too (Pv, ordinal per run).
#pragma once
#include "bpr_device.hpp"

namespace yue {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define YUE_GLOAD(rs, vo, so) __builtin_amdgcn_raw_buffer_load_b64((rs), (vo), (so), 16)            /* sc1 */
#define YUE_GSTORE(val, rs, vo, so) __builtin_amdgcn_raw_buffer_store_b64((val), (rs), (vo), (so), 16)

struct ChainArgs {
    float *P;                    // user rows, plain fp32 (ord_u == nullptr: every user has at most one run)
    unsigned *Pv;                // user rows as granules (general streams)
    unsigned *Qv;                // item rows as granules {values, version}
    const int64_t *run_ptr;      // [R + 1] triplet offsets of the runs
    const int32_t *run_u;        // [R] user of a run; nullptr: run r is user r (the uploaded events, user-major)
    const uint32_t *ord_u;       // [R] runs of the same user before this one; nullptr with plain P
    const int32_t *ev_i, *ev_j;  // [T]; ev_j < 0: the sampler gave up, the triplet is skipped
    const uint32_t *ord_i, *ord_j;   // [T] touches of the row before this one
    int64_t R;
    unsigned long long *claim;   // next unclaimed run
    uint32_t *status;            // [0] != 0: some wave gave up waiting (host reports an error)
    double *nll_slots;
    int64_t m, n;
    int k;
    float ru, ri;
    double lr;
    uint32_t spin_limit;         // polls a wave spends on ONE wait before it gives up (guards against a hung GPU)
#ifdef YUE_CHAIN_STATS
    unsigned long long *stats;   // diagnostic build only (make chainstats): [0] cycles in steps without a wait, [1] such steps,
                                 // [2] cycles in steps that waited, [3] such steps, [4] cycles per run outside the steps, [5] runs
#endif
};
#ifdef YUE_CHAIN_STATS
#define YUE_CS(...) __VA_ARGS__
#else
#define YUE_CS(...)
#endif

// Rows (fp32, k elements) -> granule rows of 64 * KR granules {element e, version 0} (value 0 for e >= k), and back.
__host__ __device__ inline int chain_granules_per_row(int k) { return k <= 64 ? 64 : k <= 128 ? 128 : 256; }
__global__ void __launch_bounds__(256) k_chain_pack(const float *X, unsigned *Xv, int64_t rows, int k) {
    const int gpr = chain_granules_per_row(k);
    const int64_t total = rows * gpr, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t row = t / gpr; const int e = (int)(t - row * gpr);
        u32x2 g; g.x = e < k ? __builtin_bit_cast(unsigned, X[row * k + e]) : 0u; g.y = 0u;
        reinterpret_cast<u32x2 *>(Xv)[t] = g;
    }
}
__global__ void __launch_bounds__(256) k_chain_unpack(const unsigned *Xv, float *X, int64_t rows, int k) {
    const int gpr = chain_granules_per_row(k);
    const int64_t total = rows * gpr, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t row = t / gpr; const int e = (int)(t - row * gpr);
        if (e < k) X[row * k + e] = __builtin_bit_cast(float, Xv[2 * t]);
    }
}

// Ordinals from the stably sorted touches: sorted position p holds (row key[p], touch code val[p] = 2 * triplet + side).
//   k_chain_seg: first position of every row's segment;  k_chain_ord: ordinal = position - segment start.
__global__ void __launch_bounds__(256) k_chain_seg(const uint32_t *key, int64_t count, uint32_t nrows, uint32_t *seg_start) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    const uint32_t r = key[p];
    if (r < nrows && (p == 0 || key[p - 1] != r)) seg_start[r] = (uint32_t)p;
}
__global__ void __launch_bounds__(256) k_chain_ord(const uint32_t *key, const uint32_t *val, int64_t count, uint32_t nrows,
                                                  const uint32_t *seg_start, uint32_t *ord_a, uint32_t *ord_b) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    const uint32_t r = key[p];
    if (r >= nrows) return;                              // touches of skipped triplets sort behind all rows
    const uint32_t code = val[p], o = (uint32_t)p - seg_start[r];
    if (ord_b == nullptr) ord_a[code] = o;               // one touch per unit (runs of a user)
    else ((code & 1u) ? ord_b : ord_a)[code >> 1] = o;
}
// Touch keys of a triplet stream: key[2t] = i, key[2t+1] = j (or nrows for a skipped triplet), val = position.
// flags[0] |= 1 for an id out of range, |= 2 for i == j.
__global__ void __launch_bounds__(256) k_chain_keys(const int32_t *ev_u, const int32_t *ev_i, const int32_t *ev_j, int64_t T, int64_t m, uint32_t nrows,
                                                   uint32_t *key, uint32_t *val, uint32_t *flags) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int32_t i = ev_i[t], j = ev_j[t];
    uint32_t bad = 0u;
    if (i < 0 || (uint32_t)i >= nrows || j >= (int32_t)nrows || (ev_u && (ev_u[t] < 0 || ev_u[t] >= m))) bad |= 1u;
    if (j >= 0 && i == j) bad |= 2u;
    if (bad) atomicOr(flags, bad);
    const bool live = j >= 0 && !bad;
    key[2 * t] = live ? (uint32_t)i : nrows; key[2 * t + 1] = live ? (uint32_t)j : nrows;
    val[2 * t] = (uint32_t)(2 * t); val[2 * t + 1] = (uint32_t)(2 * t + 1);
}
// Runs of equal consecutive users: head[t] = 1 where a run starts (scanned by the host code into run numbers).
__global__ void __launch_bounds__(256) k_chain_heads(const int32_t *ev_u, int64_t T, uint32_t *head) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) head[t] = (t == 0 || ev_u[t] != ev_u[t - 1]) ? 1u : 0u;
}
// incl[t] = inclusive scan of head: run number + 1.  Writes run_ptr / run_u / the run keys for the ordinal sort.
__global__ void __launch_bounds__(256) k_chain_runs(const int32_t *ev_u, const uint32_t *head, const uint32_t *incl, int64_t T,
                                                   int64_t *run_ptr, int32_t *run_u, uint32_t *rkey, uint32_t *rval, uint32_t m) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    if (head[t]) { const uint32_t r = incl[t] - 1u; run_ptr[r] = t; run_u[r] = ev_u[t]; rkey[r] = (uint32_t)ev_u[t] < m ? (uint32_t)ev_u[t] : m; rval[r] = r; }
    if (t == T - 1) run_ptr[incl[t]] = T;
}

// exp(y) in double for |y| <= 700, as a SHORT dependency chain: this value sits on the critical path of every triplet (the
// epoch's time is its longest chain of dependent triplets times the latency of one).  Cody-Waite reduction y = n ln2 + r,
// |r| <= ln2 / 2, the degree-13 Taylor polynomial of exp(r) by Estrin's scheme (4 levels of independent fused multiply-adds
// instead of 13 dependent ones), ldexp.  Error about 1 ulp -- like the libm value the reference's math.exp returns, it is the
// fp32 rounding of lr * (1 - s) that enters the factors (a 1-ulp difference in exp moves that rounding once in ~1e8 triplets).
__device__ __forceinline__ double chain_exp(double y) {
    const double n = __builtin_rint(y * 1.4426950408889634074);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, y);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    const double r2 = r * r;
    const double a0 = __builtin_fma(r, 1.0, 1.0), a1 = __builtin_fma(r, 1.0 / 6.0, 0.5), a2 = __builtin_fma(r, 1.0 / 120.0, 1.0 / 24.0),
                 a3 = __builtin_fma(r, 1.0 / 5040.0, 1.0 / 720.0), a4 = __builtin_fma(r, 1.0 / 362880.0, 1.0 / 40320.0),
                 a5 = __builtin_fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0), a6 = __builtin_fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
    const double r4 = r2 * r2;
    const double b0 = __builtin_fma(a1, r2, a0), b1 = __builtin_fma(a3, r2, a2), b2 = __builtin_fma(a5, r2, a4);
    const double r8 = r4 * r4;
    const double c0 = __builtin_fma(b1, r4, b0), c1 = __builtin_fma(a6, r4, b2);
    return __builtin_ldexp(__builtin_fma(c1, r8, c0), (int)n);
}
// 1 / d for a finite d >= 1: hardware reciprocal, two Newton steps, one residual correction (faithful; the division the
// reference performs is correctly rounded: the two agree except in rare last-bit cases, see chain_exp).
__device__ __forceinline__ double chain_rcp(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(e, y, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(e, y, y);
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(e, y, y);
}

// One wave per run.  Lane l holds elements 64 r + l (r < KR) of the three rows, as everywhere in the training kernels.
//
// The wave walks its run as a software pipeline over a ring of G slots: while triplet t is computed, the item rows of
// triplets t+1 .. t+G-1 are already in flight (a prefetched row is used only if all its granules carry the triplet's
// ordinal -- the tag makes speculation free; otherwise the row is polled again).  The (i, j, ordinals) headers of up to 64
// triplets sit in the lanes of four registers (one vector load each per segment of 64, a v_readlane per use).  What remains
// between two dependent triplets of a run is arithmetic.
//
// The ring's loads and stores are inline assembly with COUNTED waits: the compiler's own wait insertion drains the whole
// queue (vmcnt(0)) wherever a polling loop joins the straight-line path, which would put a full memory latency back between
// any two triplets.  Every step of the pipeline issues exactly 2 GR stores and 2 GR loads (out-of-range dummies where an
// event or a refill does not exist), so the loads of a slot always have (G - 1) * 4 GR younger operations behind them when
// the slot's turn comes: s_waitcnt vmcnt((G - 1) * 4 GR) is exact.  Anything issued in between (re-polls of the slow path)
// only adds younger operations or drains: the count stays a lower bound.
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct Gran {
    typedef u32x2 reg; typedef float val;
    static __device__ __forceinline__ void load(reg &d, unsigned vo, const i32x4 &rs, unsigned so) { asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen sc1" : "=v"(d) : "v"(vo), "s"(rs), "s"(so) : "memory"); }
    static __device__ __forceinline__ void store(const reg &d, unsigned vo, const i32x4 &rs, unsigned so) { asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen sc1" :: "v"(d), "v"(vo), "s"(rs), "s"(so) : "memory"); }
    static __device__ __forceinline__ val value(const reg &d) { return __builtin_bit_cast(float, d.x); }
    static __device__ __forceinline__ bool is(const reg &d, unsigned want) { return d.y == want; }
    static __device__ __forceinline__ unsigned version(const reg &d) { return d.y; }
    static __device__ __forceinline__ reg make(val v, unsigned ver) { reg d; d.x = __builtin_bit_cast(unsigned, v); d.y = ver; return d; }
};

template <int GR, int N, typename R>
__device__ __forceinline__ void ring_wait(R (&gi)[GR], R (&gj)[GR]) {
    static_assert(N <= 63, "vmcnt is a 6-bit field");
    if constexpr (GR == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(gi[0]), "+v"(gj[0]) : "n"(N) : "memory");
    else if constexpr (GR == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(gi[0]), "+v"(gi[1]), "+v"(gj[0]), "+v"(gj[1]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%8)" : "+v"(gi[0]), "+v"(gi[1]), "+v"(gi[2]), "+v"(gi[3]), "+v"(gj[0]), "+v"(gj[1]), "+v"(gj[2]), "+v"(gj[3]) : "n"(N) : "memory");
}
template <int GR, typename R>
__device__ __forceinline__ void row_wait_all(R (&g)[GR]) {
    if constexpr (GR == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]) :: "memory");
    else if constexpr (GR == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]) :: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) :: "memory");
}

template <int KR, bool PVER, int G>
__global__ void __launch_bounds__(256) k_bpr_chain(ChainArgs a, const int32_t *__restrict__ evi, const int32_t *__restrict__ evj,
                                                   const uint32_t *__restrict__ ordi, const uint32_t *__restrict__ ordj) {
    constexpr int GR = KR, GB = 8;                       // granules per lane and row, granule bytes
    typedef Gran GT;
    typedef typename GT::reg greg;
    typedef typename GT::val gval;
    const int lane = threadIdx.x & 63;
    const unsigned k = (unsigned)a.k;
    const unsigned row_bytes = (unsigned)GR * 64u * GB;  // granule rows
    unsigned vo[GR];
#pragma unroll
    for (int g = 0; g < GR; ++g) vo[g] = (64u * g + lane) * GB;      // (elements beyond k are zero-valued granules of the copy: no masking here)
    const unsigned v_oob = kOobOffset;
    const uint64_t qbytes = (uint64_t)a.n * row_bytes;
    const int qrec = (int)(qbytes < 0x7fffffffull ? qbytes : 0x7fffffffull);
    i32x4 rq;                                            // buffer descriptor of Qv as plain words, for the assembly operands
    { const uint64_t qa = (uint64_t)a.Qv; rq.x = (int)(uint32_t)qa; rq.y = (int)((uint32_t)(qa >> 32) & 0xffffu); rq.z = qrec; rq.w = kRsrcFlags; }
    double nl = 0.0;                                     // per-lane partial of sum -log(s)
    double sv = 1.0;                                     // lane q keeps the sigmoid of the q-th triplet since the last flush
    unsigned nsv = 0;                                    // (the logs are taken 64 at a time, off the dependency chain)
    uint64_t wave_slot = 0;
    bool dead = false;
    YUE_CS(unsigned long long cs_fast = 0, cs_nfast = 0, cs_slow = 0, cs_nslow = 0, cs_run = 0, cs_nrun = 0;)

    auto all_mine = [&](uint32_t want, const greg (&g)[GR]) -> bool {
        bool mine = true;
#pragma unroll
        for (int q = 0; q < GR; ++q) mine = mine && GT::is(g[q], want);
        return __builtin_amdgcn_ballot_w64(!mine) == 0ull;
    };
    // Slow path of a wait: polls until all granules of the row carry `want`; false if the wave gave up (status set).
    // Far from its turn (the row's version says how far) a wave sleeps in proportion and polls ONE granule; only the next
    // toucher re-reads the whole row.  (Its loads are assembly with their own full waits as well: a compiler-tracked load
    // here would make the compiler drain the queue where this path joins the straight-line one.)
    auto acquire_slow = [&](const i32x4 &rs, unsigned so, uint32_t want, greg (&g)[GR]) -> bool {
        uint32_t polls = 0;
        bool fresh = false;                                      // g is the prefetch of several triplets ago: its version says nothing yet
        for (;;) {
            // how far away is my turn?  (granule 0 of the row; a row in the middle of a rewrite reads as distance 0)
            uint32_t dist = fresh ? want - (uint32_t)__builtin_amdgcn_readfirstlane((int)GT::version(g[0])) : 0u;
            fresh = true;
            while ((int32_t)dist > 1) {
                const uint32_t naps = dist < 16u ? dist : 16u;      // ~0.5 us per touch in front of me, capped
                for (uint32_t q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(20);
                greg one[1];
                const unsigned v1 = lane == 0 ? 0u : kOobOffset;
                GT::load(one[0], v1, rs, so);
                row_wait_all<1>(one);
                dist = want - (uint32_t)__builtin_amdgcn_readfirstlane((int)GT::version(one[0]));
                if (++polls > a.spin_limit || (int32_t)dist < 0) break;
                if ((polls & 63u) == 0u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            }
            if ((int32_t)dist < 0 || ++polls > a.spin_limit) { if (lane == 0) atomicOr(a.status, (int32_t)dist < 0 ? 2u : 1u); return false; }
            if ((polls & 255u) == 0u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
#pragma unroll
            for (int q = 0; q < GR; ++q) GT::load(g[q], vo[q], rs, so);
            row_wait_all<GR>(g);
            if (all_mine(want, g)) return true;
        }
    };
    auto flush_logs = [&]() {
        if ((unsigned)lane < nsv) nl += -log(sv);                    // BPR.py:58
        nsv = 0;
    };

    while (!dead) {
        unsigned long long run = 0;
        if (lane == 0) run = atomicAdd(a.claim, 1ull);
        run = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(run >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)run);
        if ((int64_t)run >= a.R) break;
        YUE_CS(const unsigned long long cs_r0 = __builtin_readcyclecounter(); unsigned long long cs_steps = 0;)
        wave_slot = run;
        const int64_t e0 = a.run_ptr[run], e1 = a.run_ptr[run + 1];
        if (e1 <= e0) continue;
        const int64_t u = a.run_u ? (int64_t)a.run_u[run] : (int64_t)run;
        const unsigned len = (unsigned)(e1 - e0 < 0x7fffffff ? e1 - e0 : 0x7fffffff);

        gval p[GR];
        uint32_t pver = 0u;
        i32x4 rp;
        if (PVER) {
            // user rows as granules, addressed through a descriptor based at the row (any number of users)
            const uint64_t pa = (uint64_t)(a.Pv + (uint64_t)u * (row_bytes / 4u));
            rp.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)pa); rp.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(pa >> 32) & 0xffffu));
            rp.z = (int)row_bytes; rp.w = kRsrcFlags;
            // v_readfirstlane has just written two words of the descriptor: a vector-memory instruction may read an SGPR a
            // VALU instruction wrote only 5 wait states later, and the compiler does not see into the assembly below
            asm volatile("s_nop 4" : "+s"(rp));
            pver = a.ord_u[run];
            greg g[GR];
#pragma unroll
            for (int q = 0; q < GR; ++q) GT::load(g[q], vo[q], rp, 0u);
            row_wait_all<GR>(g);
            if (!all_mine(pver, g) && !acquire_slow(rp, 0u, pver, g)) { dead = true; break; }
#pragma unroll
            for (int q = 0; q < GR; ++q) p[q] = GT::value(g[q]);
        } else {
            const float *prow = a.P + (uint64_t)u * k;
#pragma unroll
            for (int q = 0; q < GR; ++q) { const unsigned e = 64u * q + lane; p[q] = e < k ? prow[e] : 0.0f; }
        }

        // The run in segments of 64 triplets (one segment for the usual run): the segment's headers sit in the lanes of four
        // registers, the ring is filled at its start and drained at its end.
        for (unsigned seg = 0; seg < len && !dead; seg += 64u) {
            int hAi, hAj;
            uint32_t hAwi, hAwj;
            {
                const bool ex = seg + (unsigned)lane < len;
                const int64_t e = e0 + seg + (ex ? lane : 0);
                hAi = evi[e]; hAj = evj[e]; hAwi = ordi[e]; hAwj = ordj[e];
                if (!ex) { hAi = 0; hAj = -1; }
            }
            greg gi[G][GR], gj[G][GR];
            // loads of one slot: 2 GR ring loads, real ones for an event with a negative, out-of-range dummies otherwise
            auto fill = [&](int s, int i_, int j_) {
                const bool live = j_ >= 0;
                const unsigned oi = live ? (unsigned)i_ * row_bytes : 0u, oj = live ? (unsigned)j_ * row_bytes : 0u;
#pragma unroll
                for (int q = 0; q < GR; ++q) {
                    const unsigned v = live ? vo[q] : v_oob;
                    GT::load(gj[s][q], v, rq, oj); GT::load(gi[s][q], v, rq, oi);
                }
            };
            // (empty statements that read what the compiler's own loads above produced: the compiler places ITS waits for them
            // here, not in front of every step's arithmetic, where they would drain the ring)
#pragma unroll
            for (int q = 0; q < GR; ++q) asm volatile("" : "+v"(p[q]));
            asm volatile("" : "+v"(hAi), "+v"(hAj), "+v"(hAwi), "+v"(hAwj));
#pragma unroll
            for (int s = 0; s < G; ++s) fill(s, __builtin_amdgcn_readlane(hAi, s), __builtin_amdgcn_readlane(hAj, s));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the ring is full: from here on the counted waits hold

            const unsigned seg_len = len - seg < 64u ? len - seg : 64u;
            for (unsigned ol = 0; ol < seg_len && !dead; ol += G) {
                const bool more = ol + G < 64u;                      // refills stay inside the segment
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    YUE_CS(const unsigned long long cs_t0 = __builtin_readcyclecounter(); bool cs_waited = false;)
                    ring_wait<GR, (G - 1) * 4 * GR>(gi[s], gj[s]);
                    const int ti = __builtin_amdgcn_readlane(hAi, ol + s), tj = __builtin_amdgcn_readlane(hAj, ol + s);
                    bool done = false;
                    if (tj >= 0 && !dead) {                          // (wave-uniform) an event of the run with a negative
                        const uint32_t wi = (uint32_t)__builtin_amdgcn_readlane((int)hAwi, ol + s), wj = (uint32_t)__builtin_amdgcn_readlane((int)hAwj, ol + s);
                        const unsigned oi = (unsigned)ti * row_bytes, oj = (unsigned)tj * row_bytes;
                        bool ok = true;
                        if (!all_mine(wj, gj[s])) { ok = acquire_slow(rq, oj, wj, gj[s]); YUE_CS(cs_waited = true;) }
                        if (ok && !all_mine(wi, gi[s])) { ok = acquire_slow(rq, oi, wi, gi[s]); YUE_CS(cs_waited = true;) }
                        if (!ok) dead = true;
                        else {
                            gval qi[GR], qj[GR];
#pragma unroll
                            for (int q = 0; q < GR; ++q) { qi[q] = GT::value(gi[s][q]); qj[q] = GT::value(gj[s][q]); }
                            // per-lane partials in element order 64 r + l, r ascending (oracle/bpr_oracle.c: dot64)
                            float ai = 0.0f, aj = 0.0f;
#pragma unroll
                            for (int q = 0; q < GR; ++q) { const gval mi = p[q] * qi[q], mj = p[q] * qj[q]; ai = ai + mi; aj = aj + mj; }
                            const float x = wave_sum(ai) - wave_sum(aj);             // BPR.py:50, fp32 margin
                            const double xd = (double)x;
                            double sg;
                            if (__builtin_fabs(xd) <= 700.0) sg = chain_rcp(1.0 + chain_exp(-xd)); // qmath.py:115-116
                            else sg = 1.0 / (1.0 + exp(-xd));
                            const float c = (float)(a.lr * (1.0 - sg));
#pragma unroll
                            for (int q = 0; q < GR; ++q) {
                                // BPR.py:51-57 (as bpr_elem)
                                const gval d = qi[q] - qj[q];
                                const gval td = c * d;
                                const gval p1 = p[q] + td;
                                const gval tq = c * p1;
                                const gval qi1 = qi[q] + tq, qj1 = qj[q] - tq;
                                const gval rpp = a.ru * p1, ra = a.ri * qi1, rb = a.ri * qj1;
                                p[q] = p1 - rpp;
                                GT::store(GT::make(qi1 - ra, wi + 1u), vo[q], rq, oi);   // the positive's row first: the hotter of the two
                                GT::store(GT::make(qj1 - rb, wj + 1u), vo[q], rq, oj);
                            }
                            sv = (unsigned)lane == nsv ? sg : sv;
                            if (++nsv == 64u) flush_logs();
                            done = true;
                        }
                    }
                    if (!done) {                                     // no event in this slot: the step's stores as dummies
                        greg z = GT::make(0.0f, 0u);
#pragma unroll
                        for (int q = 0; q < GR; ++q) { GT::store(z, v_oob, rq, 0u); GT::store(z, v_oob, rq, 0u); }
                    }
                    // refill with the same slot of the next group (dummies past the segment's last group)
                    fill(s, more ? __builtin_amdgcn_readlane(hAi, (ol + G + s) & 63u) : 0, more ? __builtin_amdgcn_readlane(hAj, (ol + G + s) & 63u) : -1);
                    YUE_CS(if (done) { const unsigned long long dt = __builtin_readcyclecounter() - cs_t0; cs_steps += dt; if (cs_waited) { cs_slow += dt; ++cs_nslow; } else { cs_fast += dt; ++cs_nfast; } })
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // nothing of this segment's ring is in flight when the next one refills it
        }
        if (dead) break;

        if (PVER) {
#pragma unroll
            for (int q = 0; q < GR; ++q) GT::store(GT::make(p[q], pver + 1u), vo[q], rp, 0u);
        } else {
            float *prow = a.P + (uint64_t)u * k;
#pragma unroll
            for (int q = 0; q < GR; ++q) { const unsigned e = 64u * q + lane; if (e < k) prow[e] = p[q]; }
        }
        YUE_CS(cs_run += __builtin_readcyclecounter() - cs_r0 - cs_steps; ++cs_nrun;)
    }
    YUE_CS(if (lane == 0) { atomicAdd(a.stats + 0, cs_fast); atomicAdd(a.stats + 1, cs_nfast); atomicAdd(a.stats + 2, cs_slow); atomicAdd(a.stats + 3, cs_nslow); atomicAdd(a.stats + 4, cs_run); atomicAdd(a.stats + 5, cs_nrun); })
    flush_logs();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) nl += __shfl_xor(nl, off);
    if (lane == 0 && nl != 0.0) atomicAdd(a.nll_slots + (wave_slot & (kNllSlots - 1)), nl);
}

// ------------------------------------------------------------------------------------------------------------------------
// The same dataflow with the work of a run split over TWO waves (a pair on two SIMDs of one CU).  A lone wave issues its
// instructions one after the other, so the time of a step is the number of its instructions: ~200 in k_bpr_chain, of which
// the next triplet of the run only needs the margin, the sigmoid and the new user row (~85).
//   wave M ("memory"): everything of k_bpr_chain except the margin -- claim, headers, the ring of prefetched rows with its
//       counted waits, version checks and polling, the item-row updates and their granule stores, the user row at the end.
//       It PUBLISHES the two rows of the next triplet in LDS before it waits for the coefficient of the current one;
//   wave C ("chain"): reads published rows from LDS, computes the two dots, the sigmoid and c = fp32(lr (1 - s)), hands c back
//       through LDS, updates its copy of the user row, keeps the loss.  It never touches global memory.
// Both waves keep the user row and update it with the same instructions, so only c crosses back.  Packets M -> C carry a
// sequence number that grows over the whole launch (slot = number mod kPairRing): a stale slot never matches.  LDS serves a
// wave's operations in order, so "rows, then tag" written by M and "tag, then rows" read by C need no fence.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kPairRing = 8;
constexpr unsigned kPktEvent = 0u, kPktStart = 1u, kPktExit = 2u;

template <int KR>
struct PairBox {
    float rows[kPairRing][2 * KR][64];     // event: qi[0..KR), qj[0..KR); start: the user row in [0..KR)
    unsigned tag[kPairRing];               // (sequence number << 2) | packet type
    unsigned cbits[kPairRing];             // the coefficient of the slot's triplet
    unsigned cseq[kPairRing];              // sequence number it belongs to
};

// LDS words shared by the two waves of a pair: relaxed workgroup-scope atomics (plain ds_read / ds_write, never hoisted or
// merged by the compiler, no waits on the vector-memory queue) between compiler barriers; the LDS keeps a wave's operations in order.
__device__ __forceinline__ unsigned lds_get(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_put(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ float lds_getf(const float *p) { return __builtin_bit_cast(float, lds_get(reinterpret_cast<const unsigned *>(p))); }
__device__ __forceinline__ void lds_putf(float *p, float v) { lds_put(reinterpret_cast<unsigned *>(p), __builtin_bit_cast(unsigned, v)); }

template <int KR, bool PVER, int G>
__global__ void __launch_bounds__(256) k_bpr_chain2(ChainArgs a, const int32_t *__restrict__ evi, const int32_t *__restrict__ evj,
                                                    const uint32_t *__restrict__ ordi, const uint32_t *__restrict__ ordj) {
    constexpr int GR = KR, GB = 8;
    typedef Gran GT;
    typedef typename GT::reg greg;
    __shared__ PairBox<KR> boxes[2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    PairBox<KR> &box = boxes[wave >> 1];
    if ((wave & 1) == 0 && lane < kPairRing) { box.tag[lane] = 0u; box.cseq[lane] = 0u; }
    __syncthreads();

    if ((wave & 1) == 0) {
        // ---------------------------------------------------------------- wave C: margin, sigmoid, coefficient, user row, loss
        float p[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) p[r] = 0.0f;
        double nl = 0.0, sv = 1.0;
        unsigned nsv = 0, seq = 1;
        YUE_CS(unsigned long long cs_work = 0, cs_wait = 0, cs_n = 0;)
        for (;;) {
            const unsigned slot = seq & (kPairRing - 1);
            unsigned tg;
            YUE_CS(const unsigned long long cs_t0 = __builtin_readcyclecounter();)
            while (((tg = lds_get(&box.tag[slot])) >> 2) != seq) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            YUE_CS(const unsigned long long cs_t1 = __builtin_readcyclecounter(); cs_wait += cs_t1 - cs_t0;)
            const unsigned type = tg & 3u;
            if (type == kPktExit) break;
            float d[2 * KR];
#pragma unroll
            for (int r = 0; r < 2 * KR; ++r) d[r] = lds_getf(&box.rows[slot][r][lane]);
            if (type == kPktStart) {
#pragma unroll
                for (int r = 0; r < KR; ++r) p[r] = d[r];
                ++seq;
                continue;
            }
            float ai = 0.0f, aj = 0.0f;
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float mi = p[r] * d[r], mj = p[r] * d[KR + r]; ai = ai + mi; aj = aj + mj; }
            const float x = wave_sum(ai) - wave_sum(aj);                 // BPR.py:50, fp32 margin
            const double xd = (double)x;
            double sg;
            if (__builtin_fabs(xd) <= 700.0) sg = chain_rcp(1.0 + chain_exp(-xd));     // qmath.py:115-116
            else sg = 1.0 / (1.0 + exp(-xd));
            const float c = (float)(a.lr * (1.0 - sg));
            if (lane == 0) { lds_put(&box.cbits[slot], __builtin_bit_cast(unsigned, c)); asm volatile("" ::: "memory"); lds_put(&box.cseq[slot], seq); }
#pragma unroll
            for (int r = 0; r < KR; ++r) {                               // BPR.py:51, :55 on the user row (as bpr_elem)
                const float dd = d[r] - d[KR + r];
                const float td = c * dd;
                const float p1 = p[r] + td;
                const float rpp = a.ru * p1;
                p[r] = p1 - rpp;
            }
            sv = (unsigned)lane == nsv ? sg : sv;
            if (++nsv == 64u) { nl += -log(sv); nsv = 0; }              // BPR.py:58, 64 logs at a time
            ++seq;
            YUE_CS(cs_work += __builtin_readcyclecounter() - cs_t1; ++cs_n;)
        }
        YUE_CS(if (lane == 0) { atomicAdd(a.stats + 0, cs_work); atomicAdd(a.stats + 1, cs_n); atomicAdd(a.stats + 2, cs_wait); atomicAdd(a.stats + 3, cs_n); })
        if ((unsigned)lane < nsv) nl += -log(sv);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) nl += __shfl_xor(nl, off);
        if (lane == 0 && nl != 0.0) atomicAdd(a.nll_slots + ((blockIdx.x * 2 + (wave >> 1)) & (kNllSlots - 1)), nl);
        return;
    }

    // -------------------------------------------------------------------- wave M: memory side
    const unsigned k = (unsigned)a.k;
    const unsigned row_bytes = (unsigned)GR * 64u * GB;
    unsigned vo[GR];
#pragma unroll
    for (int g = 0; g < GR; ++g) vo[g] = (64u * g + lane) * GB;
    const unsigned v_oob = kOobOffset;
    const uint64_t qbytes = (uint64_t)a.n * row_bytes;
    const int qrec = (int)(qbytes < 0x7fffffffull ? qbytes : 0x7fffffffull);
    i32x4 rq;
    { const uint64_t qa = (uint64_t)a.Qv; rq.x = (int)(uint32_t)qa; rq.y = (int)((uint32_t)(qa >> 32) & 0xffffu); rq.z = qrec; rq.w = kRsrcFlags; }
    unsigned seq = 1;                                    // next packet number
    bool dead = false;
    YUE_CS(unsigned long long cs_mwait = 0, cs_mn = 0;)

    auto all_mine = [&](uint32_t want, const greg (&g)[GR]) -> bool {
        bool mine = true;
#pragma unroll
        for (int q = 0; q < GR; ++q) mine = mine && GT::is(g[q], want);
        return __builtin_amdgcn_ballot_w64(!mine) == 0ull;
    };
    auto acquire_slow = [&](const i32x4 &rs, unsigned so, uint32_t want, greg (&g)[GR]) -> bool {
        uint32_t polls = 0;
        bool fresh = false;
        for (;;) {
            uint32_t dist = fresh ? want - (uint32_t)__builtin_amdgcn_readfirstlane((int)GT::version(g[0])) : 0u;
            fresh = true;
            while ((int32_t)dist > 1) {
                const uint32_t naps = dist < 16u ? dist : 16u;
                for (uint32_t q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(20);
                greg one[1];
                const unsigned v1 = lane == 0 ? 0u : kOobOffset;
                GT::load(one[0], v1, rs, so);
                row_wait_all<1>(one);
                dist = want - (uint32_t)__builtin_amdgcn_readfirstlane((int)GT::version(one[0]));
                if (++polls > a.spin_limit || (int32_t)dist < 0) break;
                if ((polls & 63u) == 0u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            }
            if ((int32_t)dist < 0 || ++polls > a.spin_limit) { if (lane == 0) atomicOr(a.status, (int32_t)dist < 0 ? 2u : 1u); return false; }
            if ((polls & 255u) == 0u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
#pragma unroll
            for (int q = 0; q < GR; ++q) GT::load(g[q], vo[q], rs, so);
            row_wait_all<GR>(g);
            if (all_mine(want, g)) return true;
        }
    };
    // a packet to wave C: 2 KR values per lane (or KR for the user row), then the tag
    auto publish = [&](unsigned type, const float (&v)[2 * KR], int count) {
        const unsigned slot = seq & (kPairRing - 1);
#pragma unroll
        for (int r = 0; r < 2 * KR; ++r) if (r < count) lds_putf(&box.rows[slot][r][lane], v[r]);
        asm volatile("" ::: "memory");
        if (lane == 0) lds_put(&box.tag[slot], (seq << 2) | type);
        ++seq;
    };

    while (!dead) {
        unsigned long long run = 0;
        if (lane == 0) run = atomicAdd(a.claim, 1ull);
        run = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(run >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)run);
        if ((int64_t)run >= a.R) break;
        const int64_t e0 = a.run_ptr[run], e1 = a.run_ptr[run + 1];
        if (e1 <= e0) continue;
        const int64_t u = a.run_u ? (int64_t)a.run_u[run] : (int64_t)run;
        const unsigned len = (unsigned)(e1 - e0 < 0x7fffffff ? e1 - e0 : 0x7fffffff);

        float p[GR];
        uint32_t pver = 0u;
        i32x4 rp;
        if (PVER) {
            const uint64_t pa = (uint64_t)(a.Pv + (uint64_t)u * (row_bytes / 4u));
            rp.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)pa); rp.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(pa >> 32) & 0xffffu));
            rp.z = (int)row_bytes; rp.w = kRsrcFlags;
            asm volatile("s_nop 4" : "+s"(rp));                       // VALU-written SGPRs, vector memory in assembly: see k_bpr_chain
            pver = a.ord_u[run];
            greg g[GR];
#pragma unroll
            for (int q = 0; q < GR; ++q) GT::load(g[q], vo[q], rp, 0u);
            row_wait_all<GR>(g);
            if (!all_mine(pver, g) && !acquire_slow(rp, 0u, pver, g)) { dead = true; break; }
#pragma unroll
            for (int q = 0; q < GR; ++q) p[q] = GT::value(g[q]);
        } else {
            const float *prow = a.P + (uint64_t)u * k;
#pragma unroll
            for (int q = 0; q < GR; ++q) { const unsigned e = 64u * q + lane; p[q] = e < k ? prow[e] : 0.0f; }
        }
        {   // the user row to wave C
            float v[2 * KR];
#pragma unroll
            for (int r = 0; r < KR; ++r) { v[r] = p[r]; v[KR + r] = 0.0f; }
            publish(kPktStart, v, KR);
        }

        for (unsigned seg = 0; seg < len && !dead; seg += 64u) {
            int hAi, hAj;
            uint32_t hAwi, hAwj;
            {
                const bool ex = seg + (unsigned)lane < len;
                const int64_t e = e0 + seg + (ex ? lane : 0);
                hAi = evi[e]; hAj = evj[e]; hAwi = ordi[e]; hAwj = ordj[e];
                if (!ex) { hAi = 0; hAj = -1; }
            }
            greg gi[G][GR], gj[G][GR];
            auto fill = [&](int s, int i_, int j_) {
                const bool live = j_ >= 0;
                const unsigned oi = live ? (unsigned)i_ * row_bytes : 0u, oj = live ? (unsigned)j_ * row_bytes : 0u;
#pragma unroll
                for (int q = 0; q < GR; ++q) {
                    const unsigned v = live ? vo[q] : v_oob;
                    GT::load(gj[s][q], v, rq, oj); GT::load(gi[s][q], v, rq, oi);
                }
            };
#pragma unroll
            for (int q = 0; q < GR; ++q) asm volatile("" : "+v"(p[q]));
            asm volatile("" : "+v"(hAi), "+v"(hAj), "+v"(hAwi), "+v"(hAwj));
#pragma unroll
            for (int s = 0; s < G; ++s) fill(s, __builtin_amdgcn_readlane(hAi, s), __builtin_amdgcn_readlane(hAj, s));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

            const unsigned seg_len = len - seg < 64u ? len - seg : 64u;
            // Counted waits as in k_bpr_chain: every step issues 2 GR stores and 2 GR loads; a slot's loads have at least
            // (G - 2) whole steps behind them when the step BEFORE its own looks ahead at it.
            // Step t: [rows of triplet t + 1: wait, check, publish (postponed if a row still waits for MY store of triplet t)]
            // [coefficient of t from wave C] [update and store t] [postponed publish] [refill slot t with triplet t + G].
            // Published-but-unanswered triplets: at most two, the ring of kPairRing slots never wraps onto them.
            bool have = false;                                       // rows of the current triplet are published
            unsigned cur_seq = 0;
            // try to publish the rows of slot s (already waited for); false: a row does not carry its ordinal yet
            auto try_publish = [&](int s, uint32_t wi, uint32_t wj) -> bool {
                if (!all_mine(wj, gj[s]) || !all_mine(wi, gi[s])) return false;
                float v[2 * KR];
#pragma unroll
                for (int q = 0; q < GR; ++q) { v[q] = GT::value(gi[s][q]); v[KR + q] = GT::value(gj[s][q]); }
                publish(kPktEvent, v, 2 * KR);
                return true;
            };
            for (unsigned ol = 0; ol < seg_len && !dead; ol += G) {
                const bool more = ol + G < 64u;
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    const int ti = __builtin_amdgcn_readlane(hAi, ol + s), tj = __builtin_amdgcn_readlane(hAj, ol + s);
                    const uint32_t wi = (uint32_t)__builtin_amdgcn_readlane((int)hAwi, ol + s), wj = (uint32_t)__builtin_amdgcn_readlane((int)hAwj, ol + s);
                    const unsigned oi = (unsigned)(ti < 0 ? 0 : ti) * row_bytes, oj = (unsigned)(tj < 0 ? 0 : tj) * row_bytes;
                    const bool live = tj >= 0 && !dead;
                    // the NEXT live triplet's slot and header (inside this group: slot s + 1; the first slot of the next group
                    // is handled at that group's first step, after its own wait)
                    if (live && !have) {                             // first triplet of a group / after a gap: publish it now
                        ring_wait<GR, (G - 2) * 4 * GR>(gi[s], gj[s]);
                        bool ok = try_publish(s, wi, wj);
                        if (!ok) {
                            ok = (all_mine(wj, gj[s]) || acquire_slow(rq, oj, wj, gj[s])) && (all_mine(wi, gi[s]) || acquire_slow(rq, oi, wi, gi[s]));
                            if (ok) ok = try_publish(s, wi, wj);
                        }
                        if (!ok) dead = true;
                        cur_seq = seq - 1u;
                        have = ok;
                    }
                    bool done = false;
                    if (live && !dead) {
                        // look ahead: rows of the next triplet of this group
                        bool next_pub = false, next_live = false;
                        int nti = 0, ntj = -1; uint32_t nwi = 0u, nwj = 0u;
                        if (s + 1 < G) {
                            nti = __builtin_amdgcn_readlane(hAi, (ol + s + 1) & 63u); ntj = (ol + s + 1 < 64u) ? __builtin_amdgcn_readlane(hAj, (ol + s + 1) & 63u) : -1;
                            nwi = (uint32_t)__builtin_amdgcn_readlane((int)hAwi, (ol + s + 1) & 63u); nwj = (uint32_t)__builtin_amdgcn_readlane((int)hAwj, (ol + s + 1) & 63u);
                            next_live = ntj >= 0;
                            if (next_live) {
                                ring_wait<GR, (G - 2) * 4 * GR>(gi[(s + 1) % G], gj[(s + 1) % G]);
                                next_pub = try_publish((s + 1) % G, nwi, nwj);
                            }
                        }
                        // the coefficient of this triplet
                        const unsigned cslot = cur_seq & (kPairRing - 1);
                        YUE_CS(const unsigned long long cs_m0 = __builtin_readcyclecounter();)
                        while (lds_get(&box.cseq[cslot]) != cur_seq) __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");
                        YUE_CS(cs_mwait += __builtin_readcyclecounter() - cs_m0; ++cs_mn;)
                        const float c = __builtin_bit_cast(float, lds_get(&box.cbits[cslot]));
#pragma unroll
                        for (int q = 0; q < GR; ++q) {
                            const float qi = GT::value(gi[s][q]), qj = GT::value(gj[s][q]);
                            const Elem o = bpr_elem(p[q], qi, qj, c, a.ru, a.ri);
                            p[q] = o.p2;
                            GT::store(GT::make(o.qi2, wi + 1u), vo[q], rq, oi);
                            GT::store(GT::make(o.qj2, wj + 1u), vo[q], rq, oj);
                        }
                        done = true;
                        have = false;
                        if (next_live) {
                            if (!next_pub) {                         // its row waited for the stores just issued (or for another wave)
                                const unsigned noi = (unsigned)nti * row_bytes, noj = (unsigned)ntj * row_bytes;
                                bool ok = (all_mine(nwj, gj[(s + 1) % G]) || acquire_slow(rq, noj, nwj, gj[(s + 1) % G])) &&
                                          (all_mine(nwi, gi[(s + 1) % G]) || acquire_slow(rq, noi, nwi, gi[(s + 1) % G]));
                                if (ok) ok = try_publish((s + 1) % G, nwi, nwj);
                                if (!ok) dead = true;
                            }
                            cur_seq = seq - 1u;
                            have = !dead;
                        }
                    }
                    if (!done) {
                        greg z = GT::make(0.0f, 0u);
#pragma unroll
                        for (int q = 0; q < GR; ++q) { GT::store(z, v_oob, rq, 0u); GT::store(z, v_oob, rq, 0u); }
                    }
                    fill(s, more ? __builtin_amdgcn_readlane(hAi, (ol + G + s) & 63u) : 0, more ? __builtin_amdgcn_readlane(hAj, (ol + G + s) & 63u) : -1);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (dead) break;
        if (PVER) {
#pragma unroll
            for (int q = 0; q < GR; ++q) GT::store(GT::make(p[q], pver + 1u), vo[q], rp, 0u);
        } else {
            float *prow = a.P + (uint64_t)u * k;
#pragma unroll
            for (int q = 0; q < GR; ++q) { const unsigned e = 64u * q + lane; if (e < k) prow[e] = p[q]; }
        }
    }
    YUE_CS(if (lane == 0) { atomicAdd(a.stats + 4, cs_mwait); atomicAdd(a.stats + 5, cs_mn); })
    {   // wave C leaves
        float v[2 * KR];
#pragma unroll
        for (int r = 0; r < 2 * KR; ++r) v[r] = 0.0f;
        publish(kPktExit, v, 0);
    }
}

}  // namespace yue
