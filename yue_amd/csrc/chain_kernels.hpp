// Exact sequential semantics of the reference's epoch loop (recommender/cf/BPR.py:40-62) as a DATAFLOW kernel.
//
// The reference applies the triplets strictly one after the other.  Two triplets commute unless they share a row (P[u],
// Q[i] or Q[j]), so the loop's result is fixed by the order in which every single ROW sees its touches -- the stream order.
// k_bpr_chain runs exactly that partial order and nothing more:
//   * a pre-pass (chain_host.hip) gives every touch of an item row its ORDINAL: the number of earlier touches of that row
//     in the stream (a stable sort of the 2T touches by row);
//   * item rows live in a versioned copy Qv: every element is one naturally aligned 8-byte granule {fp32 value, version},
//     version = number of touches the row has received so far.  A granule is written by ONE sc1 (write-through) store
//     and read by sc1 loads: the hand-off needs no flag, no fence and no drain -- a reader that finds version == its
//     ordinal on all k granules holds exactly the row its predecessor wrote (MI355X_MICROARCH.md, price list: handoff-1to1,
//     "data-tagged granules"; ~1-3 us per hop);
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
    u32x2 *Pv;                   // user rows as granules (general streams)
    u32x2 *Qv;                   // item rows as granules {value bits, version}
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
};

// Q (fp32 rows) -> granules with version 0, and back.
__global__ void __launch_bounds__(256) k_chain_pack(const float *X, u32x2 *Xv, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
        u32x2 g; g.x = __builtin_bit_cast(unsigned, X[t]); g.y = 0u;
        Xv[t] = g;
    }
}
__global__ void __launch_bounds__(256) k_chain_unpack(const u32x2 *Xv, float *X, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) X[t] = __builtin_bit_cast(float, Xv[t].x);
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

// One wave per run.  Lane l holds elements 64 r + l (r < KR) of the three rows, as everywhere in the training kernels; a
// granule load / store wave-instruction covers 512 contiguous bytes of one row.
template <int KR, bool PVER>
__global__ void __launch_bounds__(256) k_bpr_chain(ChainArgs a) {
    const int lane = threadIdx.x & 63;
    const unsigned k = (unsigned)a.k;
    const unsigned row_bytes = k * 8u;                   // granule rows
    unsigned vo[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; vo[r] = e < k ? e * 8u : kOobOffset; }
    const uint64_t qbytes = (uint64_t)a.n * row_bytes;
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(a.Qv, 0, (int)(qbytes < 0x7fffffffull ? qbytes : 0x7fffffffull), kRsrcFlags);
    double nll = 0.0;
    uint64_t wave_slot = 0;

    // Waits until all k granules of a row carry `want`; returns false if the wave gave up (status set).
    // The row is in g[] afterwards.  Far from its turn (the row's version says how far) a wave sleeps in proportion and
    // polls ONE granule; only the next toucher re-reads the whole row.
    auto acquire = [&](const decltype(rsQ) &rs, unsigned so, uint32_t want, u32x2 (&g)[KR]) -> bool {
        uint32_t polls = 0;
        for (;;) {
#pragma unroll
            for (int r = 0; r < KR; ++r) g[r] = YUE_GLOAD(rs, vo[r], so);
            bool mine = true;
#pragma unroll
            for (int r = 0; r < KR; ++r) mine = mine && (vo[r] == kOobOffset || g[r].y == want);
            if (__builtin_amdgcn_ballot_w64(!mine) == 0ull) return true;
            // how far away is my turn?  (granule 0 of the row; a row in the middle of a rewrite reads as distance 0)
            uint32_t dist = want - (uint32_t)__builtin_amdgcn_readfirstlane((int)g[0].y);
            while ((int32_t)dist > 1) {
                // ~0.8 us per touch in front of me, capped (the sleeps below add up to at most ~14 us)
                const uint32_t naps = dist < 16u ? dist : 16u;
                for (uint32_t q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(32);
                const u32x2 one = YUE_GLOAD(rs, lane == 0 ? 0u : kOobOffset, so);
                dist = want - (uint32_t)__builtin_amdgcn_readfirstlane((int)one.y);
                if (++polls > a.spin_limit || (int32_t)dist < 0) { if (lane == 0) atomicOr(a.status, (int32_t)dist < 0 ? 2u : 1u); return false; }
                if ((polls & 63u) == 0u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            }
            if ((int32_t)dist < 0) { if (lane == 0) atomicOr(a.status, 2u); return false; }
            __builtin_amdgcn_s_sleep(2);
            if (++polls > a.spin_limit) { if (lane == 0) atomicOr(a.status, 1u); return false; }
            if ((polls & 255u) == 0u && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        }
    };

    for (;;) {
        unsigned long long run = 0;
        if (lane == 0) run = atomicAdd(a.claim, 1ull);
        run = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(run >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)run);
        if ((int64_t)run >= a.R) break;
        wave_slot = run;
        const int64_t e0 = a.run_ptr[run], e1 = a.run_ptr[run + 1];
        if (e1 <= e0) continue;
        const int64_t u = a.run_u ? (int64_t)a.run_u[run] : (int64_t)run;

        float p[KR];
        uint32_t pver = 0u;
        if (PVER) {
            // user rows as granules, addressed through a descriptor based at the row (any number of users)
            const auto rsP = __builtin_amdgcn_make_buffer_rsrc(a.Pv + (uint64_t)u * k, 0, (int)row_bytes, kRsrcFlags);
            pver = a.ord_u[run];
            u32x2 g[KR];
            if (!acquire(rsP, 0u, pver, g)) return;
#pragma unroll
            for (int r = 0; r < KR; ++r) p[r] = __builtin_bit_cast(float, g[r].x);
        } else {
            const float *prow = a.P + (uint64_t)u * k;
#pragma unroll
            for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; p[r] = e < k ? prow[e] : 0.0f; }
        }

        for (int64_t e = e0; e < e1; ++e) {
            const int32_t i = a.ev_i[e], j = a.ev_j[e];
            if (j < 0) continue;                                     // the sampler gave up on this event (BPR.py:47 would spin)
            const uint32_t wi = a.ord_i[e], wj = a.ord_j[e];
            const unsigned oi = (unsigned)i * row_bytes, oj = (unsigned)j * row_bytes;
            u32x2 gi[KR], gj[KR];
            if (!acquire(rsQ, oj, wj, gj)) return;                   // the negative is almost always a cold row: first
            if (!acquire(rsQ, oi, wi, gi)) return;
            float qi[KR], qj[KR];
#pragma unroll
            for (int r = 0; r < KR; ++r) { qi[r] = __builtin_bit_cast(float, gi[r].x); qj[r] = __builtin_bit_cast(float, gj[r].x); }
            float ai = 0.0f, aj = 0.0f;
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float a1 = p[r] * qi[r]; ai = ai + a1; const float a2 = p[r] * qj[r]; aj = aj + a2; }
            const float x = wave_sum(ai) - wave_sum(aj);             // BPR.py:50, fp32 margin
            const double s = 1.0 / (1.0 + exp(-(double)x));          // qmath.py:115-116
            const float c = (float)(a.lr * (1.0 - s));
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const Elem o = bpr_elem(p[r], qi[r], qj[r], c, a.ru, a.ri);
                p[r] = o.p2;
                u32x2 ni, nj;
                ni.x = __builtin_bit_cast(unsigned, o.qi2); ni.y = wi + 1u;
                nj.x = __builtin_bit_cast(unsigned, o.qj2); nj.y = wj + 1u;
                YUE_GSTORE(ni, rsQ, vo[r], oi);                      // the hot row first: its next toucher is waiting
                YUE_GSTORE(nj, rsQ, vo[r], oj);
            }
            nll += -log(s);                                          // BPR.py:58 (behind the stores: off the hand-off path)
        }

        if (PVER) {
            const auto rsP = __builtin_amdgcn_make_buffer_rsrc(a.Pv + (uint64_t)u * k, 0, (int)row_bytes, kRsrcFlags);
#pragma unroll
            for (int r = 0; r < KR; ++r) { u32x2 g; g.x = __builtin_bit_cast(unsigned, p[r]); g.y = pver + 1u; YUE_GSTORE(g, rsP, vo[r], 0u); }
        } else {
            float *prow = a.P + (uint64_t)u * k;
#pragma unroll
            for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; if (e < k) prow[e] = p[r]; }
        }
    }
    if (lane == 0 && nll != 0.0) atomicAdd(a.nll_slots + (wave_slot & (kNllSlots - 1)), nll);
}

}  // namespace yue
