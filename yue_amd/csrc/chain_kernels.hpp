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

#include <type_traits>

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
    uint32_t xcd;                // k_bpr_chain3<.., ONE_XCD>: the XCD whose workgroups do the work (the others leave at once)
    unsigned long long *groups;  // ... and how many workgroups stayed
#ifdef YUE_CHAIN_STATS
    unsigned long long *stats;   // diagnostic build only (make chainstats): [0] cycles in steps without a wait, [1] such steps,
                                 // [2] cycles in steps that waited, [3] such steps, [4] cycles per run outside the steps, [5] runs
#endif
};
// timing-only ablations of k_bpr_chain3 (make EXTRA=-DYUE_ABL=bits OBJDIR=build_ablN LIB=libyue_hip_ablN.so; results WRONG by
// construction, never loaded by the package): 1 no version check in wave L, 2 no row writes to LDS, 4 no header words, 8 no
// refill loads, 16 no granule stores in wave S (implied by 4), 32 no arithmetic in wave S, 64 wave L's per-step headers from the
// packet number instead of v_readlane
#ifndef YUE_ABL
#define YUE_ABL 0
#endif
#ifdef YUE_CHAIN_STATS
#define YUE_CS(...) __VA_ARGS__
#else
#define YUE_CS(...)
#endif

// Rows (fp32, k elements) -> granule rows of 64 * KR granules {element, version 0} (value 0 for elements >= k), and back.
// A lane holds elements 64 r + l (r < KR) of a row, as everywhere in the training kernels.  For k > 64 its granules of an
// even / odd pair of r lie NEXT TO EACH OTHER in the granule row (slot (r / 2) * 128 + 2 l + (r & 1)), so that ONE 16-byte
// memory instruction moves two granules of a lane -- each still validated by its own version: a 16-byte access that tears
// into its 8-byte halves (observed, round 3) leaves one half stale and recognisable, never a wrong value.
__host__ __device__ inline int chain_granules_per_row(int k) { return k <= 64 ? 64 : k <= 128 ? 128 : 256; }
__host__ __device__ inline int chain_element_of_slot(int gpr, int t) {      // which element of the row granule slot t holds
    if (gpr == 64) return t;
    const int q = t >> 7, l = (t & 127) >> 1, r = 2 * q + (t & 1);
    return 64 * r + l;
}
__global__ void __launch_bounds__(256) k_chain_pack(const float *X, unsigned *Xv, int64_t rows, int k) {
    const int gpr = chain_granules_per_row(k);
    const int64_t total = rows * gpr, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t row = t / gpr; const int e = chain_element_of_slot(gpr, (int)(t - row * gpr));
        u32x2 g; g.x = e < k ? __builtin_bit_cast(unsigned, X[row * k + e]) : 0u; g.y = 0u;
        reinterpret_cast<u32x2 *>(Xv)[t] = g;
    }
}
__global__ void __launch_bounds__(256) k_chain_unpack(const unsigned *Xv, float *X, int64_t rows, int k) {
    const int gpr = chain_granules_per_row(k);
    const int64_t total = rows * gpr, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t row = t / gpr; const int e = chain_element_of_slot(gpr, (int)(t - row * gpr));
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

// An SGPR written by the vector ALU (v_readlane, v_readfirstlane) may be read by a vector-memory instruction only 5 wait
// states later.  The compiler counts them for its own instructions but does not see into inline assembly: every scalar
// operand of the assembly loads / stores below (descriptors, row offsets) that can come out of the vector ALU passes through
// here first.  (tests/test_asm_hazards.py checks the listing for it.)
__device__ __forceinline__ void vmem_sgpr_guard(unsigned &s0) { asm volatile("s_nop 4" : "+s"(s0)); }
__device__ __forceinline__ void vmem_sgpr_guard(unsigned &s0, unsigned &s1) { asm volatile("s_nop 4" : "+s"(s0), "+s"(s1)); }

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
__device__ __forceinline__ void vmem_sgpr_guard(i32x4 &rs) { asm volatile("s_nop 4" : "+s"(rs)); }

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
// The granules of ONE row in a lane: KR = 1 -- one granule, 8-byte instructions; KR >= 2 -- KR / 2 pairs {value, version, value,
// version} of elements 64 (2 q) + l and 64 (2 q + 1) + l, 16-byte instructions.
template <int KR>
struct GranRow {
    static constexpr int NL = KR == 1 ? 1 : KR / 2;                      // memory instructions per row and lane
    typedef typename std::conditional<KR == 1, u32x2, u32x4v>::type reg;
    static __device__ __forceinline__ unsigned voffset(int q, int lane) { return KR == 1 ? (unsigned)lane * 8u : (unsigned)q * 1024u + (unsigned)lane * 16u; }
    static __device__ __forceinline__ void load(reg &d, unsigned vo, const i32x4 &rs, unsigned so) {
        if constexpr (KR == 1) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen sc1" : "=v"(d) : "v"(vo), "s"(rs), "s"(so) : "memory");
        else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen sc1" : "=v"(d) : "v"(vo), "s"(rs), "s"(so) : "memory");
    }
    static __device__ __forceinline__ void store(const reg &d, unsigned vo, const i32x4 &rs, unsigned so) {
        if constexpr (KR == 1) asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen sc1" :: "v"(d), "v"(vo), "s"(rs), "s"(so) : "memory");
        // (a vector-memory store of more than 64 bits reads its data registers late: a VALU write of them needs 2 wait states
        // behind it -- the compiler counts those for its own stores, not for this one, hence the s_nop inside the statement)
        else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 1" :: "v"(d), "v"(vo), "s"(rs), "s"(so) : "memory");
    }
    // the same store left in the XCD's L2 (no write-through): only for a launch whose waves ALL sit behind one L2 (k_bpr_chain3<.., ONE_XCD>)
    static __device__ __forceinline__ void store_l2(const reg &d, unsigned vo, const i32x4 &rs, unsigned so) {
        if constexpr (KR == 1) asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" :: "v"(d), "v"(vo), "s"(rs), "s"(so) : "memory");
        else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(d), "v"(vo), "s"(rs), "s"(so) : "memory");
    }
    // element r (0 .. KR - 1) of the row
    static __device__ __forceinline__ float value(const reg (&g)[NL], int r) {
        if constexpr (KR == 1) return __builtin_bit_cast(float, g[0].x);
        else return __builtin_bit_cast(float, (r & 1) ? g[r >> 1].z : g[r >> 1].x);
    }
    // 0 iff every granule of the lane carries version `want`
    static __device__ __forceinline__ unsigned bad(const reg (&g)[NL], unsigned want) {
        unsigned b = 0u;
#pragma unroll
        for (int q = 0; q < NL; ++q) { b |= g[q].y ^ want; if constexpr (KR != 1) b |= g[q].w ^ want; }
        return b;
    }
    static __device__ __forceinline__ unsigned version0(const reg &d) { return d.y; }
    static __device__ __forceinline__ void set(reg (&g)[NL], int r, float v, unsigned ver) {
        if constexpr (KR == 1) { g[0].x = __builtin_bit_cast(unsigned, v); g[0].y = ver; }
        else if (r & 1) { g[r >> 1].z = __builtin_bit_cast(unsigned, v); g[r >> 1].w = ver; }
        else { g[r >> 1].x = __builtin_bit_cast(unsigned, v); g[r >> 1].y = ver; }
    }
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

template <int KR, bool PVER, int G, bool FAST>
__global__ void __launch_bounds__(256) k_bpr_chain(ChainArgs a, const int32_t *__restrict__ evi, const int32_t *__restrict__ evj,
                                                   const uint32_t *__restrict__ ordi, const uint32_t *__restrict__ ordj) {
    typedef GranRow<KR> GT;
    constexpr int GR = GT::NL, GB = 8;                   // memory instructions per row and lane, granule bytes
    typedef typename GT::reg greg;
    typedef float gval;
    const int lane = threadIdx.x & 63;
    const unsigned k = (unsigned)a.k;
    const unsigned row_bytes = (unsigned)KR * 64u * GB;  // granule rows
    unsigned vo[GR];
#pragma unroll
    for (int g = 0; g < GR; ++g) vo[g] = GT::voffset(g, lane);       // (elements beyond k are zero-valued granules of the copy: no masking here)
    const unsigned v_oob = kOobOffset;
    const uint64_t qbytes = (uint64_t)a.n * row_bytes;
    const int qrec = (int)(qbytes < 0x7fffffffull ? qbytes : 0x7fffffffull);
    i32x4 rq;                                            // buffer descriptor of Qv as plain words, for the assembly operands
    { const uint64_t qa = (uint64_t)a.Qv; rq.x = (int)(uint32_t)qa; rq.y = (int)((uint32_t)(qa >> 32) & 0xffffu); rq.z = qrec; rq.w = kRsrcFlags; }
    double nl = 0.0;                                     // per-lane partial of sum -log(s)
    float xs = 0.0f;                                     // lane q keeps the margin of the q-th triplet since the last flush
    unsigned nsv = 0;                                    // (sigmoid and log of the loss are taken 64 at a time, off the dependency chain)
    uint64_t wave_slot = 0;
    bool dead = false;
    YUE_CS(unsigned long long cs_fast = 0, cs_nfast = 0, cs_slow = 0, cs_nslow = 0, cs_run = 0, cs_nrun = 0;)

    auto all_mine = [&](uint32_t want, const greg (&g)[GR]) -> bool {
        return __builtin_amdgcn_ballot_w64(GT::bad(g, want) != 0u) == 0ull;
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
            uint32_t dist = fresh ? want - (uint32_t)__builtin_amdgcn_readfirstlane((int)GT::version0(g[0])) : 0u;
            fresh = true;
            while ((int32_t)dist > 1) {
                const uint32_t naps = dist < 16u ? dist : 16u;      // ~0.5 us per touch in front of me, capped
                for (uint32_t q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(20);
                greg one[1];
                const unsigned v1 = lane == 0 ? 0u : kOobOffset;
                GT::load(one[0], v1, rs, so);
                row_wait_all<1>(one);
                dist = want - (uint32_t)__builtin_amdgcn_readfirstlane((int)GT::version0(one[0]));
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
        if ((unsigned)lane < nsv) nl += -log(chain_sigmoid(xs));     // BPR.py:58
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

        gval p[KR];
        uint32_t pver = 0u;
        i32x4 rp;
        if (PVER) {
            // user rows as granules, addressed through a descriptor based at the row (any number of users)
            const uint64_t pa = (uint64_t)(a.Pv + (uint64_t)u * (row_bytes / 4u));
            rp.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)pa); rp.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(pa >> 32) & 0xffffu));
            rp.z = (int)row_bytes; rp.w = kRsrcFlags;
            vmem_sgpr_guard(rp);                                     // v_readfirstlane has just written two words of the descriptor
            pver = a.ord_u[run];
            greg g[GR];
#pragma unroll
            for (int q = 0; q < GR; ++q) GT::load(g[q], vo[q], rp, 0u);
            row_wait_all<GR>(g);
            if (!all_mine(pver, g) && !acquire_slow(rp, 0u, pver, g)) { dead = true; break; }
#pragma unroll
            for (int r = 0; r < KR; ++r) p[r] = GT::value(g, r);
        } else {
            const float *prow = a.P + (uint64_t)u * k;
#pragma unroll
            for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; p[r] = e < k ? prow[e] : 0.0f; }
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
                unsigned oi = live ? (unsigned)i_ * row_bytes : 0u, oj = live ? (unsigned)j_ * row_bytes : 0u;
                vmem_sgpr_guard(oi, oj);
#pragma unroll
                for (int q = 0; q < GR; ++q) {
                    const unsigned v = live ? vo[q] : v_oob;
                    GT::load(gj[s][q], v, rq, oj); GT::load(gi[s][q], v, rq, oi);
                }
            };
            // (empty statements that read what the compiler's own loads above produced: the compiler places ITS waits for them
            // here, not in front of every step's arithmetic, where they would drain the ring)
#pragma unroll
            for (int r = 0; r < KR; ++r) asm volatile("" : "+v"(p[r]));
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
                        unsigned oi = (unsigned)ti * row_bytes, oj = (unsigned)tj * row_bytes;
                        vmem_sgpr_guard(oi, oj);
                        bool ok = true;
                        if (!all_mine(wj, gj[s])) { ok = acquire_slow(rq, oj, wj, gj[s]); YUE_CS(cs_waited = true;) }
                        if (ok && !all_mine(wi, gi[s])) { ok = acquire_slow(rq, oi, wi, gi[s]); YUE_CS(cs_waited = true;) }
                        if (!ok) dead = true;
                        else {
                            gval qi[KR], qj[KR];
#pragma unroll
                            for (int r = 0; r < KR; ++r) { qi[r] = GT::value(gi[s], r); qj[r] = GT::value(gj[s], r); }
                            float x;
                            if (FAST) {
                                // option chain_fast (within 1e-5, not bit-equal): one 64-lane sum of p . (qi - qj)
                                float acc = 0.0f;
#pragma unroll
                                for (int q = 0; q < KR; ++q) { const gval dq = qi[q] - qj[q]; acc = q == 0 ? p[0] * dq : __builtin_fmaf(p[q], dq, acc); }
                                x = wave_sum_any(acc);
                            } else {
                                // per-lane partials in element order 64 r + l, r ascending (oracle/bpr_oracle.c: dot64)
                                float ai = 0.0f, aj = 0.0f;
#pragma unroll
                                for (int q = 0; q < KR; ++q) { const gval mi = p[q] * qi[q], mj = p[q] * qj[q]; ai = ai + mi; aj = aj + mj; }
                                x = wave_sum(ai) - wave_sum(aj);                     // BPR.py:50, fp32 margin
                            }
                            const float c = FAST ? chain_coef_fast(x, (float)a.lr) : (float)(a.lr * (1.0 - chain_sigmoid(x)));   // qmath.py:115-116
                            greg oi_[GR], oj_[GR];
#pragma unroll
                            for (int q = 0; q < KR; ++q) {
                                // BPR.py:51-57 (as bpr_elem)
                                const gval d = qi[q] - qj[q];
                                const gval td = c * d;
                                const gval p1 = p[q] + td;
                                const gval tq = c * p1;
                                const gval qi1 = qi[q] + tq, qj1 = qj[q] - tq;
                                const gval rpp = a.ru * p1, ra = a.ri * qi1, rb = a.ri * qj1;
                                p[q] = p1 - rpp;
                                GT::set(oi_, q, qi1 - ra, wi + 1u);
                                GT::set(oj_, q, qj1 - rb, wj + 1u);
                            }
#pragma unroll
                            for (int q = 0; q < GR; ++q) {
                                GT::store(oi_[q], vo[q], rq, oi);                    // the positive's row first: the hotter of the two
                                GT::store(oj_[q], vo[q], rq, oj);
                            }
                            xs = (unsigned)lane == nsv ? x : xs;
                            if (++nsv == 64u) flush_logs();
                            done = true;
                        }
                    }
                    if (!done) {                                     // no event in this slot: the step's stores as dummies
                        greg z[GR];
#pragma unroll
                        for (int r = 0; r < KR; ++r) GT::set(z, r, 0.0f, 0u);
#pragma unroll
                        for (int q = 0; q < GR; ++q) { GT::store(z[q], v_oob, rq, 0u); GT::store(z[q], v_oob, rq, 0u); }
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
            greg pg[GR];
#pragma unroll
            for (int r = 0; r < KR; ++r) GT::set(pg, r, p[r], pver + 1u);
#pragma unroll
            for (int q = 0; q < GR; ++q) GT::store(pg[q], vo[q], rp, 0u);
        } else {
            float *prow = a.P + (uint64_t)u * k;
#pragma unroll
            for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; if (e < k) prow[e] = p[r]; }
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
// The same dataflow with the work of a run split over the waves of one workgroup (a CU's four SIMDs).  A lone wave issues its
// instructions one after the other, so the latency of a step of k_bpr_chain is the number of its instructions (~250), of which
// the next triplet of the run needs only the margin, the sigmoid and the new user row.  Here
//   wave C ("chain"):   reads published rows from LDS, computes the margin, the sigmoid and c = fp32(lr (1 - s)), hands c on
//       through LDS, updates its copy of the user row, keeps the loss.  No global memory, no headers, no version checks: the
//       dependency chain of a run and nothing else;
//   waves L0, L1 ("loads"): claim runs (L0; L1 follows through a mailbox), keep the headers, a ring of prefetched granule rows
//       each with its counted waits, the version checks and the polling; L0 takes the even triplets of a segment, L1 the odd
//       ones; a row pair that carries its ordinals is PUBLISHED to the other waves through LDS under the triplet's packet
//       number (start packet of the run + live triplets before it: both waves count the same way);
//   waves Si, Sj ("stores"): take the published rows and c, apply the reference's update to the positive's (Si) or the
//       negative's (Sj) row and to their own copy of the user row (same instructions as wave C, same values), store the
//       granules with version + 1; Si stores the user row as soon as the run's end packet arrives.
// (Round 3's two-wave split left the memory side as long as the chain side: no gain.  Measured per triplet on one wave alone:
// chain 265 cycles with the short coefficient (600 in double precision), loads 490, stores 370 -- hence two of each.)
// Packets carry a sequence number that grows over the whole launch (slot = number mod kTrioRing): a stale slot never matches.
// Every word of the LDS mailboxes exists once per lane (lane l reads and writes word l): no exec masking, no broadcast; LDS
// serves a wave's operations in order, so "rows, then tag" written by L and "tag, then rows" read by C and S need no fence, and
// a wave that has been answered (coefficient from C, progress words from S) knows the other side's reads of the slot are done.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kTrioRing = 16;
constexpr unsigned kPktEvent = 0u, kPktStart = 1u, kPktExit = 2u, kPktEnd = 3u;

template <int KR>
struct TrioBox {
    // Every record is 8 bytes per lane, written and read as ONE 64-bit LDS operation per lane (half the LDS instructions of
    // single words, and a record's two halves always belong together):
    unsigned long long rows[kTrioRing][KR][64];   // the 2 KR values of a packet in pairs: event qi[0..KR), qj[0..KR); start: the user row first
    unsigned long long hdr[kTrioRing][64];        // {header word (low), tag (high)} -- written after the rows, read before them
    unsigned long long cw[kTrioRing][64];         // from wave C: {coefficient of the slot's triplet (low), sequence number it answers (high)}
    unsigned sdone[2][64];                 // from the two S waves: last packet each is done with (a slot is reused kTrioRing packets later)
    unsigned runbox[kTrioRing][64];        // from wave L0 to wave L1: the runs it has claimed (lanes 0..4: first event lo / hi, length, number of the start packet, run counter)
};
static_assert(sizeof(TrioBox<4>) <= 64 * 1024, "static LDS");

// LDS words shared by the waves of a group: relaxed workgroup-scope atomics (plain ds_read / ds_write, never hoisted or
// merged by the compiler, no waits on the vector-memory queue) between compiler barriers; the LDS keeps a wave's operations in order.
__device__ __forceinline__ unsigned lds_get(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_put(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
struct Pair { unsigned lo, hi; };
__device__ __forceinline__ Pair lds_get2(const unsigned long long *p) {
    const unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    Pair r; r.lo = (unsigned)v; r.hi = (unsigned)(v >> 32); return r;
}
__device__ __forceinline__ void lds_put2(unsigned long long *p, unsigned lo, unsigned hi) {
    __hip_atomic_store(p, ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float bitsf(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ bool all_lanes(bool b) { return __builtin_amdgcn_ballot_w64(!b) == 0ull; }
// v with lane L replaced by the wave-uniform s (v_writelane_b32; this compiler has no builtin for it)
template <int L> __device__ __forceinline__ unsigned write_lane(unsigned v, unsigned s) { asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(L)); return v; }

// ONE_XCD: every hand-off of a row between two waves costs a round trip to the memory side (a write-through store drops the
// line from L2, the reader's L1-bypassing load then misses L2): ~1.2 us per hop, and as much for every prefetched row that
// turns out stale.  The L2 of ONE XCD is coherent for the 32 CUs in front of it, so a launch whose working
// waves all sit on one XCD can hand rows over through that L2: plain stores (the line stays in L2), sc1 loads (past the
// reader's L1, served by L2).  The grid is 8 x the wanted groups; a workgroup reads its XCD from the hardware register and
// leaves at once unless it is the chosen one (workgroups are dealt round-robin over the XCDs; nothing depends on that being
// exact: whoever stays works, the host checks that somebody did).
template <int KR, bool PVER, int G, bool FAST, bool ONE_XCD>
__global__ void __launch_bounds__(320) k_bpr_chain3(ChainArgs a, const int32_t *__restrict__ evi, const int32_t *__restrict__ evj,
                                                    const uint32_t *__restrict__ ordi, const uint32_t *__restrict__ ordj) {
    typedef GranRow<KR> GT;
    constexpr int GR = GT::NL, GB = 8;                   // memory instructions per row and lane, granule bytes
    typedef typename GT::reg greg;
    typedef TrioBox<KR> Box;
    __shared__ Box box;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (ONE_XCD) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((xcc & 15u) != a.xcd) return;                // (the whole workgroup: it sits on one XCD)
        if (threadIdx.x == 0) atomicAdd(a.groups, 1ull);
    }
    for (int t = threadIdx.x; t < kTrioRing * 64; t += 320) { box.hdr[t >> 6][t & 63] = 0ull; box.cw[t >> 6][t & 63] = 0ull; box.runbox[t >> 6][t & 63] = 0u; }
    if (threadIdx.x < 128) box.sdone[threadIdx.x >> 6][threadIdx.x & 63] = 0u;
    __syncthreads();
    const unsigned k = (unsigned)a.k;
    const unsigned row_bytes = (unsigned)KR * 64u * GB;  // granule rows
    // every wait of waves C and S ends when wave L publishes; wave L's own waits are bounded by the spin limit.  The bound
    // below only guards against a protocol error: a wave that has slept this often sets the status word and leaves.
    constexpr uint32_t kIdleLimit = 1u << 30;
    uint32_t idle = 0;
    auto nap = [&]() -> bool {                           // false: give up (status set by somebody, or the idle bound)
        __builtin_amdgcn_s_sleep(1);                     // (spinning without the sleep changes neither the step nor the hop)
        if (__builtin_expect((++idle & 0xfffu) == 0u, 0)) {
            if (idle >= kIdleLimit) { if (lane == 0) atomicOr(a.status, 4u); return false; }
            if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        }
        return true;
    };

    if (wave == 1) {
        // ---------------------------------------------------------------- wave C: margin, sigmoid, coefficient, user row, loss
        float p[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) p[r] = 0.0f;
        double nl = 0.0;
        float xs = 0.0f;                                 // lane q keeps the margin of the q-th triplet since the last flush
        unsigned nsv = 0, seq = 1;
        unsigned tg;
        float d[2 * KR];
        const float lrf = (float)a.lr;
        auto fetch = [&](unsigned q) {                   // tag first, then the rows: a matching tag vouches for the rows read behind it
            const unsigned slot = q & (kTrioRing - 1);
            tg = lds_get2(&box.hdr[slot][lane]).hi;
            asm volatile("" ::: "memory");
#pragma unroll
            for (int w = 0; w < KR; ++w) { const Pair v = lds_get2(&box.rows[slot][w][lane]); d[2 * w] = bitsf(v.lo); d[2 * w + 1] = bitsf(v.hi); }
            asm volatile("" ::: "memory");
        };
        auto flush = [&]() {                             // BPR.py:58, 64 logs at a time, off the chain
            if ((unsigned)lane < nsv) nl += -log(chain_sigmoid(xs));
            nsv = 0;
        };
        YUE_CS(unsigned long long cs_work = 0, cs_wait = 0, cs_n = 0; unsigned long long cs_t0 = __builtin_readcyclecounter();)
        fetch(seq);
        for (;;) {
            if (__builtin_expect(!all_lanes(tg == (seq << 2)), 0)) {     // not (yet) the event with this number
                if (!all_lanes((tg >> 2) == seq)) {                      // not published yet
                    if (!nap()) break;
                    fetch(seq);
                    continue;
                }
                idle = 0;
                const unsigned type = (unsigned)__builtin_amdgcn_readfirstlane((int)tg) & 3u;
                if (type == kPktExit) break;
                if (type == kPktStart) {                                 // start of a run: its user row (end of a run: nothing to do here)
#pragma unroll
                    for (int r = 0; r < KR; ++r) p[r] = d[r];
                }
                lds_put2(&box.cw[seq & (kTrioRing - 1)][lane], 0u, seq);  // acknowledged: the S waves do not run ahead of this wave
                ++seq;
                fetch(seq);
                YUE_CS(cs_t0 = __builtin_readcyclecounter();)
                continue;
            }
            idle = 0;
            YUE_CS(const unsigned long long cs_t1 = __builtin_readcyclecounter(); cs_wait += cs_t1 - cs_t0;)
            const unsigned slot = seq & (kTrioRing - 1);
            float dd[KR], x;
            if (FAST) {
                // within north_star's 1e-5, not bit-equal to the oracle: ONE 64-lane sum of p . (qi - qj), fp32 coefficient
                float acc = 0.0f;
#pragma unroll
                for (int r = 0; r < KR; ++r) { dd[r] = d[r] - d[KR + r]; acc = r == 0 ? p[0] * dd[0] : __builtin_fmaf(p[r], dd[r], acc); }
                x = wave_sum_any(acc);
            } else {
                // per-lane partials in element order 64 r + l, r ascending (oracle/bpr_oracle.c: dot64)
                float ai = 0.0f, aj = 0.0f;
#pragma unroll
                for (int r = 0; r < KR; ++r) { const float mi = p[r] * d[r], mj = p[r] * d[KR + r]; ai = ai + mi; aj = aj + mj; dd[r] = d[r] - d[KR + r]; }
                x = wave_sum(ai) - wave_sum(aj);                         // BPR.py:50, fp32 margin
            }
            fetch(seq + 1u);                                             // the next packet's reads fly under the sigmoid
            const float c = FAST ? chain_coef_fast(x, lrf) : (float)(a.lr * (1.0 - chain_sigmoid(x)));
            lds_put2(&box.cw[slot][lane], fbits(c), seq);
#pragma unroll
            for (int r = 0; r < KR; ++r) {                               // BPR.py:51, :55 on the user row (as bpr_elem)
                const float td = c * dd[r];
                const float p1 = p[r] + td;
                const float rpp = a.ru * p1;
                p[r] = p1 - rpp;
            }
            xs = (unsigned)lane == nsv ? x : xs;
            if (__builtin_expect(++nsv == 64u, 0)) flush();
            ++seq;
            YUE_CS(cs_t0 = __builtin_readcyclecounter(); cs_work += cs_t0 - cs_t1; ++cs_n;)
        }
        YUE_CS(if (lane == 0) { atomicAdd(a.stats + 0, cs_work); atomicAdd(a.stats + 1, cs_n); atomicAdd(a.stats + 2, cs_wait); })
        flush();
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) nl += __shfl_xor(nl, off);
        if (lane == 0 && nl != 0.0) atomicAdd(a.nll_slots + (blockIdx.x & (kNllSlots - 1)), nl);
        return;
    }

    unsigned vo[GR];
#pragma unroll
    for (int g = 0; g < GR; ++g) vo[g] = GT::voffset(g, lane);       // (elements beyond k are zero-valued granules of the copy: no masking here)
    const uint64_t qbytes = (uint64_t)a.n * row_bytes;
    const int qrec = (int)(qbytes < 0x7fffffffull ? qbytes : 0x7fffffffull);
    i32x4 rq;                                            // buffer descriptor of Qv as plain words, for the assembly operands
    { const uint64_t qa = (uint64_t)a.Qv; rq.x = (int)(uint32_t)qa; rq.y = (int)((uint32_t)(qa >> 32) & 0xffffu); rq.z = qrec; rq.w = kRsrcFlags; }
    auto user_rsrc = [&](int64_t u) -> i32x4 {          // user rows as granules: a descriptor based at the row (any number of users)
        const uint64_t pa = (uint64_t)(a.Pv + (uint64_t)u * (row_bytes / 4u));
        i32x4 rp;
        rp.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)pa); rp.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(pa >> 32) & 0xffffu));
        rp.z = (int)row_bytes; rp.w = kRsrcFlags;
        vmem_sgpr_guard(rp);
        return rp;
    };

    if (wave == 2 || wave == 3) {
        // ---------------------------------------------------------------- waves Si (2) / Sj (3): one item row's update and stores each
        // (two instances of the loop: which row a wave updates is fixed at compile time, so the other row's arithmetic falls away)
        auto stores = [&](auto which) {
            constexpr bool neg = decltype(which)::value;
        float p[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) p[r] = 0.0f;
        int64_t u_run = 0;
        uint32_t pver1 = 0u;
        unsigned seq = 1;
        YUE_CS(unsigned long long cs_swait = 0, cs_sn = 0;)
        auto store_user_row = [&]() {
            if (PVER) {
                const i32x4 rp = user_rsrc(u_run);
                greg pg[GR];
#pragma unroll
                for (int r = 0; r < KR; ++r) GT::set(pg, r, p[r], pver1);
#pragma unroll
                for (int q = 0; q < GR; ++q) { if (ONE_XCD) GT::store_l2(pg[q], vo[q], rp, 0u); else GT::store(pg[q], vo[q], rp, 0u); }
            } else {
                float *prow = a.P + (uint64_t)u_run * k;
#pragma unroll
                for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; if (e < k) prow[e] = p[r]; }
            }
        };
        // One packet at a time, but never one LDS round trip after the other: the next packet's tag, rows, header and wave C's
        // answer to it are read while this one is computed.  The hot loop takes events whose answer is there; everything else
        // (not published yet, no answer yet, start / end / exit packets) is sorted out behind it.
        unsigned tg, mvn, sq;
        float dn[2 * KR], cn;
        auto fetch = [&](unsigned q) {                   // tag before the rows, wave C's sequence number before its coefficient:
            const unsigned slot = q & (kTrioRing - 1);   // a matching tag / number vouches for what is read behind it
            { const Pair h = lds_get2(&box.hdr[slot][lane]); mvn = h.lo; tg = h.hi; }
            { const Pair w = lds_get2(&box.cw[slot][lane]); cn = bitsf(w.lo); sq = w.hi; }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int w = 0; w < KR; ++w) { const Pair v = lds_get2(&box.rows[slot][w][lane]); dn[2 * w] = bitsf(v.lo); dn[2 * w + 1] = bitsf(v.hi); }
            asm volatile("" ::: "memory");
        };
        fetch(seq);
        for (;;) {
            for (;;) {
                unsigned bad = (tg ^ (seq << 2)) | (sq ^ seq);
                asm volatile("" : "+v"(bad));
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad != 0u) != 0ull, 0)) break;
                // an event and its coefficient
                lds_put(&box.sdone[neg][lane], seq);     // behind this wave's reads of the slot (LDS order): the L waves may reuse it
                float d[2 * KR];
#pragma unroll
                for (int r = 0; r < 2 * KR; ++r) d[r] = dn[r];
                const unsigned mv = mvn;
                const float c = cn;
                ++seq;
                fetch(seq);                              // the next packet's reads fly under this one's arithmetic
                // lanes 0..3 of the header word: positive, negative, their ordinals -- this wave's row and the version it leaves with
                unsigned orow = (unsigned)(neg ? __builtin_amdgcn_readlane((int)mv, 1) : __builtin_amdgcn_readlane((int)mv, 0)) * row_bytes;
                const uint32_t w1 = (uint32_t)(neg ? __builtin_amdgcn_readlane((int)mv, 3) : __builtin_amdgcn_readlane((int)mv, 2)) + 1u;
                vmem_sgpr_guard(orow);
                greg og[GR];
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    Elem o;
                    if (YUE_ABL & 32) { o.p2 = p[r]; o.qi2 = d[r]; o.qj2 = c; } else o = bpr_elem(p[r], d[r], d[KR + r], c, a.ru, a.ri);    // BPR.py:51-57
                    p[r] = o.p2;
                    GT::set(og, r, neg ? o.qj2 : o.qi2, w1);
                }
#pragma unroll
                for (int q = 0; q < GR; ++q) {
                    if (YUE_ABL & (16 | 4)) { asm volatile("" :: "v"(og[q]), "s"(orow)); }
                    else if (ONE_XCD) GT::store_l2(og[q], vo[q], rq, orow);
                    else GT::store(og[q], vo[q], rq, orow);
                }
                YUE_CS(++cs_sn;)
            }
            YUE_CS(const unsigned long long cs_s0 = __builtin_readcyclecounter();)
            const unsigned type = (unsigned)__builtin_amdgcn_readfirstlane((int)tg) & 3u;
            const bool here = all_lanes((tg >> 2) == seq);
            if (here && type == kPktExit) break;
            if (!here || !all_lanes(sq == seq) || type == kPktEvent) {   // not published yet, or wave C has not answered yet (events: the hot loop's business)
                if (!nap()) break;
                fetch(seq);
                YUE_CS(cs_swait += __builtin_readcyclecounter() - cs_s0;)
                continue;
            }
            idle = 0;
            lds_put(&box.sdone[neg][lane], seq);
            if (type == kPktStart) {
#pragma unroll
                for (int r = 0; r < KR; ++r) p[r] = dn[r];
                u_run = (int64_t)__builtin_amdgcn_readlane((int)mvn, 0);
                pver1 = (uint32_t)__builtin_amdgcn_readlane((int)mvn, 1);
            } else if (type == kPktEnd && !neg) store_user_row();    // the user row leaves NOW (a later run of the user, any group's, waits for it)
            ++seq;
            fetch(seq);
        }
        YUE_CS(if (lane == 0) { atomicAdd(a.stats + 5, cs_swait); atomicAdd(a.stats + 6, cs_sn); })
        };
        if (wave == 2) stores(std::false_type{}); else stores(std::true_type{});
        return;
    }

    // -------------------------------------------------------------------- waves L0 (0) / L1 (4): claims, headers, prefetch rings, version checks
    // (A workgroup's waves are dealt to the CU's four SIMDs in turn, so waves 0 and 4 share one: the two loading waves, each of
    // which works on every other triplet only -- measured per triplet: L 2 x 167 cycles, S 2 x 250, C 265 with the short coefficient.)
    // L0 takes the even triplets of a segment of 64, L1 the odd ones, each with a ring of its own.  A triplet's packet number is
    // the segment's first number plus the live triplets in front of it (from the ballot of the segment's headers: both waves
    // count alike), so the two waves publish independently and waves C / S still see the run's triplets in order.
    const unsigned par = wave == 0 ? 0u : 1u;            // 0: L0 (also: claims, start / end / exit packets), 1: L1
    unsigned sdone_seen = 0;                             // min over the two S waves, as last read
    unsigned next_seq = 1;                               // number of the next run's start packet
    YUE_CS(unsigned long long cs_lfast = 0, cs_lnfast = 0, cs_lring = 0;)

    auto all_mine = [&](uint32_t want, const greg (&g)[GR]) -> bool {
        return all_lanes(GT::bad(g, want) == 0u);
    };
    // Slow path of a wait, as in k_bpr_chain: far from its turn a wave sleeps in proportion and polls ONE granule.
    auto acquire_slow = [&](const i32x4 &rs, unsigned so, uint32_t want, greg (&g)[GR]) -> bool {
        uint32_t polls = 0;
        bool fresh = false;
        for (;;) {
            uint32_t dist = fresh ? want - (uint32_t)__builtin_amdgcn_readfirstlane((int)GT::version0(g[0])) : 0u;
            fresh = true;
            while ((int32_t)dist > 1) {
                const uint32_t naps = dist < 16u ? dist : 16u;
                for (uint32_t q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(20);
                greg one[1];
                const unsigned v1 = lane == 0 ? 0u : kOobOffset;
                GT::load(one[0], v1, rs, so);
                row_wait_all<1>(one);
                dist = want - (uint32_t)__builtin_amdgcn_readfirstlane((int)GT::version0(one[0]));
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
    // the slot of packet `seq` is free once both S waves are done with packet seq - kTrioRing.  false: gave up
    auto reserve_slow = [&](unsigned seq) -> bool {
        YUE_CS(const unsigned long long cs_r0 = __builtin_readcyclecounter();)
        for (;;) {
            const unsigned d0 = lds_get(&box.sdone[0][lane]), d1 = lds_get(&box.sdone[1][lane]);
            sdone_seen = (unsigned)__builtin_amdgcn_readfirstlane((int)((int)(d0 - d1) < 0 ? d0 : d1));
            if ((int)(seq - sdone_seen) <= kTrioRing) { idle = 0; YUE_CS(cs_lring += __builtin_readcyclecounter() - cs_r0;) return true; }
            if (!nap()) return false;
        }
    };
    auto publish = [&](unsigned seq, unsigned type, const float (&v)[2 * KR], int count, unsigned mv) {
        const unsigned slot = seq & (kTrioRing - 1);
#pragma unroll
        for (int w = 0; w < KR; ++w) if (2 * w < count && !((YUE_ABL & 2) && type == kPktEvent)) lds_put2(&box.rows[slot][w][lane], fbits(v[2 * w]), fbits(v[2 * w + 1]));
        asm volatile("" ::: "memory");
        lds_put2(&box.hdr[slot][lane], mv, (seq << 2) | type);
    };
    float zeros[2 * KR];
#pragma unroll
    for (int r = 0; r < 2 * KR; ++r) zeros[r] = 0.0f;

    for (unsigned rc = 1;; ++rc) {                       // rc: the group's run counter (slot of the mailbox)
        int64_t e0;
        unsigned len, seq0;                              // seq0: number of the run's start packet
        if (par == 0) {
            unsigned long long run = 0;
            if (lane == 0) run = atomicAdd(a.claim, 1ull);
            run = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(run >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)run);
            int64_t e1 = 0;
            e0 = 0;
            const bool more = (int64_t)run < a.R;
            if (more) { e0 = a.run_ptr[run]; e1 = a.run_ptr[run + 1]; }
            len = (unsigned)(e1 - e0 < 0x7fffffff ? e1 - e0 : 0x7fffffff);
            if (more && e1 <= e0) { --rc; continue; }                   // an empty run: nothing to tell anybody
            seq0 = next_seq;
            {   // the mailbox for L1 (a ring as long as the packet ring: L1 cannot be that many runs behind, see reserve_slow)
                unsigned mb = rc;
                mb = write_lane<0>(mb, (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)e0));
                mb = write_lane<1>(mb, (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)e0 >> 32)));
                mb = write_lane<2>(mb, more ? len : 0xffffffffu);       // (no more runs: L1 leaves)
                mb = write_lane<3>(mb, seq0);
                lds_put(&box.runbox[rc & (kTrioRing - 1)][lane], mb);
            }
            if (!more) break;
            const int64_t u = a.run_u ? (int64_t)a.run_u[run] : (int64_t)run;
            // the user row to waves C and S
            float v[2 * KR];
            uint32_t pver = 0u;
#pragma unroll
            for (int r = 0; r < 2 * KR; ++r) v[r] = 0.0f;
            if (PVER) {
                const i32x4 rp = user_rsrc(u);
                pver = a.ord_u[run];
                greg g[GR];
#pragma unroll
                for (int q = 0; q < GR; ++q) GT::load(g[q], vo[q], rp, 0u);
                row_wait_all<GR>(g);
                if (!all_mine(pver, g) && !acquire_slow(rp, 0u, pver, g)) goto bail;
#pragma unroll
                for (int r = 0; r < KR; ++r) v[r] = GT::value(g, r);
            } else {
                const float *prow = a.P + (uint64_t)u * k;
#pragma unroll
                for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; v[r] = e < k ? prow[e] : 0.0f; }
            }
            unsigned mv = pver + 1u;                                     // lane 0: the user, lane 1: the version the row leaves with
            mv = write_lane<0>(mv, (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)u));
            if ((int)(seq0 - sdone_seen) > kTrioRing && !reserve_slow(seq0)) goto bail;
            publish(seq0, kPktStart, v, KR, mv);
        } else {
            unsigned mb;
            for (;;) {
                mb = lds_get(&box.runbox[rc & (kTrioRing - 1)][lane]);
                if ((unsigned)__builtin_amdgcn_readlane((int)mb, 4) == rc) break;
                if (!nap()) goto bail;
            }
            idle = 0;
            len = (unsigned)__builtin_amdgcn_readlane((int)mb, 2);
            if (len == 0xffffffffu) break;
            e0 = (int64_t)(((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)mb, 1) << 32) | (unsigned)__builtin_amdgcn_readlane((int)mb, 0));
            seq0 = (unsigned)__builtin_amdgcn_readlane((int)mb, 3);
        }

        unsigned seg_seq = seq0 + 1u;                    // number of the segment's first live triplet
        for (unsigned seg = 0; seg < len; seg += 64u) {
            int hAi, hAj;
            uint32_t hAwi, hAwj;
            {
                const bool ex = seg + (unsigned)lane < len;
                const int64_t e = e0 + seg + (ex ? lane : 0);
                hAi = evi[e]; hAj = evj[e]; hAwi = ordi[e]; hAwj = ordj[e];
                if (!ex) { hAi = 0; hAj = -1; }
            }
            greg gi[G][GR], gj[G][GR];
            // loads of one slot: always 2 GR ring loads; a slot without an event (a skipped triplet, the lanes past the run's end,
            // the last group's refills) loads row 0 instead -- never looked at, it only keeps the count of the waits exact
            auto fill = [&](int s, int i_, int j_) {
                unsigned oi = (unsigned)i_ * row_bytes, oj = (unsigned)(j_ < 0 ? 0 : j_) * row_bytes;
                vmem_sgpr_guard(oi, oj);
#pragma unroll
                for (int q = 0; q < GR; ++q) { GT::load(gj[s][q], vo[q], rq, oj); GT::load(gi[s][q], vo[q], rq, oi); }
            };
            // (the compiler's own loads above are waited for HERE, not in front of a step, where the wait would drain the ring)
            asm volatile("" : "+v"(hAi), "+v"(hAj), "+v"(hAwi), "+v"(hAwj));
            const unsigned long long live = __builtin_amdgcn_ballot_w64(hAj >= 0);
#pragma unroll
            for (int s = 0; s < G; ++s) fill(s, __builtin_amdgcn_readlane(hAi, 2 * s + par), __builtin_amdgcn_readlane(hAj, 2 * s + par));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the ring is full: from here on the counted waits hold

            const unsigned seg_len = len - seg < 64u ? len - seg : 64u;
            // a wave issues loads only: every step exactly 2 GR of them, so a slot's loads have (G - 1) * 2 GR younger ones
            // behind them when its turn comes (refills stay inside the segment of 64: the last group's reload the group itself)
            for (unsigned ol = 0; ol < seg_len; ol += 2 * G) {
                const unsigned nxt = ol + 2 * G < 64u ? ol + 2 * G : ol;
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    YUE_CS(const unsigned long long cs_t0 = __builtin_readcyclecounter(); bool cs_waited = false;)
                    const unsigned t = ol + 2 * s + par;             // the triplet of this step (index in the segment)
                    if (!(YUE_ABL & 8)) ring_wait<GR, (G - 1) * 2 * GR>(gi[s], gj[s]);
                    if ((live >> t) & 1ull) {                        // (wave-uniform) an event of the run with a negative
                        const unsigned seq = seg_seq + (unsigned)__builtin_popcountll(live & ((1ull << t) - 1ull));
                        const int tj = __builtin_amdgcn_readlane(hAj, t);
                        const uint32_t wi = (uint32_t)__builtin_amdgcn_readlane((int)hAwi, t), wj = (uint32_t)__builtin_amdgcn_readlane((int)hAwj, t);
                        const int ti = __builtin_amdgcn_readlane(hAi, t);
                        // all granules carry their ordinals and the ring of packets has room: ONE test (xor / or in the vector ALU;
                        // the empty statement keeps the compiler from turning it back into one compare and branch per granule)
                        unsigned bad = (int)(seq - sdone_seen) > kTrioRing ? 1u : 0u;
                        if (!(YUE_ABL & 1)) {
                            bad |= GT::bad(gi[s], wi) | GT::bad(gj[s], wj);
                        }
                        asm volatile("" : "+v"(bad));
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad != 0u) != 0ull, 0)) {
                            // a row does not carry its ordinal yet, or the ring of packets is full
                            unsigned oi = (unsigned)ti * row_bytes, oj = (unsigned)tj * row_bytes;
                            vmem_sgpr_guard(oi, oj);
                            if (!(YUE_ABL & 1)) {
                                if (!all_mine(wj, gj[s]) && !acquire_slow(rq, oj, wj, gj[s])) goto bail;
                                if (!all_mine(wi, gi[s]) && !acquire_slow(rq, oi, wi, gi[s])) goto bail;
                            }
                            if ((int)(seq - sdone_seen) > kTrioRing && !reserve_slow(seq)) goto bail;
                            YUE_CS(cs_waited = true;)
                        }
                        float v[2 * KR];
#pragma unroll
                        for (int r = 0; r < KR; ++r) { v[r] = GT::value(gi[s], r); v[KR + r] = GT::value(gj[s], r); }
                        unsigned mv = wj;                            // lanes 0..3: the two items, the two ordinals (the S waves add the rest)
                        if (!(YUE_ABL & 4)) {
                            mv = write_lane<0>(mv, (unsigned)ti);
                            mv = write_lane<1>(mv, (unsigned)tj);
                            mv = write_lane<2>(mv, wi);
                        }
                        publish(seq, kPktEvent, v, 2 * KR, mv);
                        YUE_CS(if (!cs_waited) { cs_lfast += __builtin_readcyclecounter() - cs_t0; ++cs_lnfast; })
                    }
                    // refill with the same slot of the next group
                    if (!(YUE_ABL & 8)) fill(s, __builtin_amdgcn_readlane(hAi, nxt + 2 * s + par), __builtin_amdgcn_readlane(hAj, nxt + 2 * s + par));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // nothing of this segment's ring is in flight when the next one refills it
            seg_seq += (unsigned)__builtin_popcountll(live);
        }
        next_seq = seg_seq + 1u;                         // (seg_seq: the run's end packet)
        if (par == 0) {                                  // end of the run: wave Si stores the user row
            if ((int)(seg_seq - sdone_seen) > kTrioRing && !reserve_slow(seg_seq)) goto bail;
            publish(seg_seq, kPktEnd, zeros, 0, 0u);
        }
    }
bail:
    YUE_CS(if (lane == 0) { atomicAdd(a.stats + 3, cs_lfast); atomicAdd(a.stats + 4, cs_lnfast); atomicAdd(a.stats + 7, cs_lring); })
    // the other waves leave (a full ring drains first; when a wave gave up, the status word ends everybody's waits)
    if (par == 0 && ((int)(next_seq - sdone_seen) <= kTrioRing || reserve_slow(next_seq))) publish(next_seq, kPktExit, zeros, 0, 0u);
}

}  // namespace yue
