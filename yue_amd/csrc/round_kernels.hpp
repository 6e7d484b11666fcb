// The epoch path's S-round kernels (round 2).  Same semantics as k_round (train_kernels.hpp) -- what
// changed is who counts the touches of a round and who finishes its contended rows:
//   k_round        every event takes 2 returning 64-bit device atomics for the NEXT round's tickets, publishes its staging
//                  slot in a per-item table and resets its counter word; a contended row is finished inside the launch by
//                  its last toucher (drain own stores, count down, read the table, sort the slots, rewrite): the waves of
//                  a launch wait for each other, a launch is one wave generation;
//   k_round_meta   ONE pre-pass per epoch over all rounds: per (round, item range) an LDS histogram of the round's touches,
//                  a prefix sum that lays the contended rows' staging blocks out behind each other, 8 bytes of metadata per
//                  event and the list of the round's contended rows -- no global atomics;
//   k_round_m      reads the metadata, gathers, updates, stores (in place / to the touch's staging row base + ticket / float
//                  atomics for hot rows) and ends: no wave waits for another, a launch may be several wave generations;
//   k_round_fold   the next launch on the stream -- the kernel boundary is the synchronisation -- rewrites the round's
//                  contended rows from the list: row += its block of staged differences in ticket order.
#pragma once
#include "train_kernels.hpp"

// timing-only ablations of diagnostic builds (results wrong by construction; never defined in the product build)
#ifdef YUE_ABL_NO_HOT
#define YUE_M_HOT(val, rs, vo, so) asm volatile("" :: "v"(val))
#else
#define YUE_M_HOT(val, rs, vo, so) YUE_BATOMIC(val, rs, vo, so)
#endif
#ifdef YUE_ABL_NO_DP
#define YUE_M_DP(val, rs, vo, so) asm volatile("" :: "v"(val))
#else
#define YUE_M_DP(val, rs, vo, so) YUE_BATOMIC(val, rs, vo, so)
#endif

#define YUE_BLOAD_NT(rs, vo, so) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32((rs), (vo), (so), 2))
// the gathers of NEGATIVE item rows (uniform over the catalogue: no re-use) are non-temporal; positives (Zipf head: an XCD's L2
// serves some of the repeats) are cached.  Measured on C3: both non-temporal 32.4 against 33.0 ms/epoch both cached (round 2);
// positives cached 28.45 against 28.8 both non-temporal (round 3, two runs each on one box; C2 unchanged).  The cache space
// is worth more to the staged rows, which the fold launch reads back: non-temporal STORES of those cost 1.6 ms.
#if defined(YUE_EXP_LOAD_CACHED)
#define YUE_M_LOAD_J(rs, vo, so) YUE_BLOAD(rs, vo, so)
#define YUE_M_LOAD_I(rs, vo, so) YUE_BLOAD(rs, vo, so)
#elif defined(YUE_EXP_LOAD_NT)
#define YUE_M_LOAD_J(rs, vo, so) YUE_BLOAD_NT(rs, vo, so)
#define YUE_M_LOAD_I(rs, vo, so) YUE_BLOAD_NT(rs, vo, so)
#else
#define YUE_M_LOAD_J(rs, vo, so) YUE_BLOAD_NT(rs, vo, so)
#define YUE_M_LOAD_I(rs, vo, so) YUE_BLOAD(rs, vo, so)
#endif

// the fold launch reads every staged row exactly once: non-temporal (32.3 against 32.6 ms/epoch)
#define YUE_FOLD_LD(p) __builtin_nontemporal_load(p)
#define YUE_FOLD_LDROW(p) (*(p))

namespace yue {

// Metadata word of one touch (an event's positive or negative item row) in its round:
//   bits 0..1   kind: 0 no touch (the sampler gave up on the event), 1 the round's only touch of the row (stored in place),
//               2 staged (the row has 2..stage_max touches), 3 hot (hotter: float atomics into dQ)
//   bits 2..31  staged: THIS touch's staging row (first row of the row's block + the touch's ticket)
// Fold-list word of a contended row:
//   bits 25..31 staged: the row's touches = rows of its block (2..kMetaStageMax); 0: hot
//   bits 0..24  staged: first staging row of the block
constexpr uint32_t kMetaStageMax = 64u;  // largest block (7-bit count field)
constexpr uint32_t kMetaStageDefault = 4u;   // measured on C3 (k = 128): staging rows with 5..8 touches too is slower (34.0 vs 33.2 ms/epoch)
constexpr uint32_t kMetaUnique = 1u, kMetaStaged = 2u, kMetaHot = 3u;
constexpr int kMetaUnroll = 8;       // events per thread and step of k_round_meta's two sweeps (their loads are in flight together)
__host__ __device__ inline uint32_t meta_kind(uint32_t w) { return w & 3u; }
__host__ __device__ inline uint32_t meta_slot(uint32_t w) { return w >> 2; }
__host__ __device__ inline uint32_t fold_count(uint32_t w) { return w >> 25; }
__host__ __device__ inline uint32_t fold_block(uint32_t w) { return w & 0x1ffffffu; }

struct MetaArgs {
    const int32_t *ev_i, *ev_j;
    const int64_t *bounds;       // [R + 1] event offsets of the rounds
    int64_t R;
    int32_t n;                   // item rows
    int32_t range;               // item rows per work item (one LDS word each)
    int32_t G;                   // ranges: ceil(n / range)
    int32_t chunk;               // LDS words per thread in the block scan (odd: conflict-free strides)
    uint32_t stage_max;          // rows with 2..stage_max touches are staged (<= kMetaStageMax; 1: no staging rows in this call)
    uint32_t *meta_i, *meta_j;   // [events]
    // [R] per round, zeroed before the launch: low half = staging rows handed out, high half = contended rows listed
    unsigned long long *round_rows;
    // Fold list: the contended rows of round r are fold[fold_base(bounds[r], r) ...), (round_rows[r] >> 32) of them --
    // {item row, fold-list word} each; a round has at most as many contended rows as events.
    uint2 *fold;
    // Large catalogues (more ranges than a work item should re-read its round for): the round's touches, bucketed by item
    // range by k_round_bucket -- {item row, 2 * (event - first event of the round) + (0 positive | 1 negative)} each, the
    // touches of round r in bk_touch[2 * bounds[r] ...), those of range g from bk_ptr[r * (G + 1) + g] on (offsets relative
    // to the round's first touch slot).  Null: every work item sweeps the round's events itself.
    const uint2 *bk_touch;
    const uint32_t *bk_ptr;
};

// First entry of round r's list (the round starts at event e0): 16-byte aligned, lists do not overlap.
__host__ __device__ inline int64_t fold_base(int64_t e0, int64_t r) { return (e0 + 4 * r + 3) & ~(int64_t)3; }

// One work item = (round, item range): count the round's touches of the range's rows in LDS, lay the contended rows'
// staging blocks out behind each other (prefix sum over the range, the range's base from one atomic per work item), list
// the contended rows, then hand every touch its class / ticket / block.  A work item reads the round's (i, j) twice; ranges
// of one round run on different CUs.  Tickets follow the order in which the LDS serves the touches, not the event order:
// the sum of a row's staged differences is taken in ticket order (any order is within the fp32 tolerance of the oracle's).
template <bool BUCKETED>
__global__ void __launch_bounds__(1024) k_round_meta(MetaArgs a) {
    extern __shared__ uint32_t slots[];
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long range_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // Work items are dealt out per XCD (workgroup b runs on XCD b % 8; the grid is a multiple of 8): an XCD takes the
    // rounds r = x (mod 8) with all their ranges, its workgroups walk them range-minor -- the ranges of one round run
    // side by side behind ONE L2, which then serves all but the first reader of the round's events.
    const int64_t xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int64_t rounds_here = (a.R - xcd + 7) >> 3;                      // rounds of this XCD
    for (int64_t sq = local; sq < rounds_here * a.G; sq += per_xcd) {
        const int64_t r = (sq / a.G) * 8 + xcd;
        const int32_t g = (int32_t)(sq % a.G);
        const int32_t lo = g * a.range;
        const int32_t width = min(a.range, a.n - lo);
        const int64_t e0 = a.bounds[r], e1 = a.bounds[r + 1];
        for (int s = tid; s < a.range; s += 1024) slots[s] = 0u;
        __syncthreads();
        const uint2 *mine_t = nullptr;                      // BUCKETED: this work item's touches
        uint32_t mine_n = 0u;
        if (BUCKETED) {
            const uint32_t *bp = a.bk_ptr + r * (a.G + 1) + g;
            mine_t = a.bk_touch + 2 * e0 + bp[0];
            mine_n = bp[1] - bp[0];
        }
        if (BUCKETED) {
            for (uint32_t t = tid; t < mine_n; t += 1024) atomicAdd(&slots[mine_t[t].x - (uint32_t)lo], 1u);
        } else {
        for (int64_t eb = e0 + tid; eb < e1; eb += 1024 * kMetaUnroll) {
            int32_t vi[kMetaUnroll], vj[kMetaUnroll];
#pragma unroll
            for (int q = 0; q < kMetaUnroll; ++q) {
                const int64_t e = eb + 1024 * q;
                vj[q] = e < e1 ? a.ev_j[e] : -1; vi[q] = e < e1 ? a.ev_i[e] : 0;
            }
#pragma unroll
            for (int q = 0; q < kMetaUnroll; ++q)
                if (vj[q] >= 0) {
                    const uint32_t si = (uint32_t)(vi[q] - lo), sj = (uint32_t)(vj[q] - lo);
                    if (si < (uint32_t)width) atomicAdd(&slots[si], 1u);
                    if (sj < (uint32_t)width) atomicAdd(&slots[sj], 1u);
                }
        }
        }
        __syncthreads();
        // staging rows of this range: the sum of the touch counts of its rows with 2..stage_max touches (low half);
        // its contended rows (high half)
        const int s0 = tid * a.chunk;
        unsigned long long mine = 0ull;
        for (int q = 0; q < a.chunk; ++q) {
            const int s = s0 + q;
            if (s < width) { const uint32_t c = slots[s]; if (c >= 2u) mine += (1ull << 32) + (c <= a.stage_max ? c : 0u); }
        }
        unsigned long long incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const unsigned long long up = __shfl_up(incl, off); if (lane >= off) incl += up; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        if (tid == 0) {
            unsigned long long total = 0ull;
            for (int q = 0; q < 16; ++q) total += wsum[q];
            range_base = total ? atomicAdd(a.round_rows + r, total) : 0ull;
        }
        __syncthreads();
        unsigned long long run = range_base + incl - mine;
        for (int q = 0; q < wv; ++q) run += wsum[q];
        uint2 *fold = a.fold + fold_base(e0, r);
        for (int q = 0; q < a.chunk; ++q) {
            const int s = s0 + q;
            if (s < width) {
                const uint32_t c = slots[s];
                if (c >= 2u) {
                    const bool st = c <= a.stage_max;
                    slots[s] = st ? (kMetaStaged | ((uint32_t)run << 2)) : kMetaHot;
                    fold[run >> 32] = make_uint2((uint32_t)(lo + s), st ? ((c << 25) | ((uint32_t)run & 0x1ffffffu)) : 0u);
                    run += (1ull << 32) + (st ? c : 0u);
                }
            }
        }
        __syncthreads();
        if (BUCKETED) {
            // (events the sampler gave up on carry no touches: k_round_bucket wrote their two zero words)
            for (uint32_t t = tid; t < mine_n; t += 1024) {
                const uint2 tc = mine_t[t];
                const uint32_t sl = tc.x - (uint32_t)lo;
                const uint32_t wd = slots[sl];
                const uint32_t mw = meta_kind(wd) == kMetaStaged ? atomicAdd(&slots[sl], 4u) : wd;
                ((tc.y & 1u) ? a.meta_j : a.meta_i)[e0 + (tc.y >> 1)] = mw;
            }
        } else
        for (int64_t eb = e0 + tid; eb < e1; eb += 1024 * kMetaUnroll) {
            int32_t vi[kMetaUnroll], vj[kMetaUnroll];
#pragma unroll
            for (int q = 0; q < kMetaUnroll; ++q) {
                const int64_t e = eb + 1024 * q;
                vj[q] = e < e1 ? a.ev_j[e] : -1; vi[q] = e < e1 ? a.ev_i[e] : 0;
            }
#pragma unroll
            for (int q = 0; q < kMetaUnroll; ++q) {
                const int64_t e = eb + 1024 * q;
                if (vj[q] >= 0) {
                    const uint32_t si = (uint32_t)(vi[q] - lo), sj = (uint32_t)(vj[q] - lo);
                    // a staged row's word hands out the block's rows: every touch takes the next one
                    if (si < (uint32_t)width) {
                        const uint32_t wd = slots[si];
                        a.meta_i[e] = meta_kind(wd) == kMetaStaged ? atomicAdd(&slots[si], 4u) : wd;
                    }
                    if (sj < (uint32_t)width) {
                        const uint32_t wd = slots[sj];
                        a.meta_j[e] = meta_kind(wd) == kMetaStaged ? atomicAdd(&slots[sj], 4u) : wd;
                    }
                } else if (g == 0 && e < e1) {
                    a.meta_i[e] = 0u; a.meta_j[e] = 0u;
                }
            }
        }
        __syncthreads();
    }
}

// Large catalogues: one workgroup per round sorts the round's touches into their item ranges (range = 2^kBucketShift rows:
// a counting pass and a scatter pass over the round's events, bucket offsets from an LDS prefix sum), so that a work item of
// k_round_meta<true> reads its own touches only.  Also writes the two zero metadata words of events without a negative.
constexpr int kBucketShift = 15;
constexpr int kBucketRangesMax = 256;
struct BucketArgs {
    const int32_t *ev_i, *ev_j;
    const int64_t *bounds;
    int64_t R;
    int32_t G;
    uint2 *bk_touch;
    uint32_t *bk_ptr;            // [R * (G + 1)]
    uint32_t *meta_i, *meta_j;
};

__global__ void __launch_bounds__(1024) k_round_bucket(BucketArgs a) {
    __shared__ uint32_t cnt[kBucketRangesMax], off[kBucketRangesMax + 1];
    const int tid = threadIdx.x;
    for (int64_t r = blockIdx.x; r < a.R; r += gridDim.x) {
        const int64_t e0 = a.bounds[r], e1 = a.bounds[r + 1];
        for (int g = tid; g < a.G; g += 1024) cnt[g] = 0u;
        __syncthreads();
        for (int64_t eb = e0 + tid; eb < e1; eb += 1024 * kMetaUnroll) {
            int32_t vi[kMetaUnroll], vj[kMetaUnroll];
#pragma unroll
            for (int q = 0; q < kMetaUnroll; ++q) {
                const int64_t e = eb + 1024 * q;
                vj[q] = e < e1 ? a.ev_j[e] : 0; vi[q] = e < e1 ? a.ev_i[e] : -1;       // (vi < 0: no such event)
            }
#pragma unroll
            for (int q = 0; q < kMetaUnroll; ++q) {
                if (vi[q] < 0) continue;
                if (vj[q] >= 0) { atomicAdd(&cnt[vi[q] >> kBucketShift], 1u); atomicAdd(&cnt[vj[q] >> kBucketShift], 1u); }
                else { a.meta_i[eb + 1024 * q] = 0u; a.meta_j[eb + 1024 * q] = 0u; }
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0u;
            for (int g = 0; g < a.G; ++g) { off[g] = run; run += cnt[g]; cnt[g] = 0u; }
            off[a.G] = run;
        }
        __syncthreads();
        for (int g = tid; g <= a.G; g += 1024) a.bk_ptr[r * (a.G + 1) + g] = off[g];
        uint2 *dst = a.bk_touch + 2 * e0;
        for (int64_t eb = e0 + tid; eb < e1; eb += 1024 * kMetaUnroll) {
            int32_t vi[kMetaUnroll], vj[kMetaUnroll];
#pragma unroll
            for (int q = 0; q < kMetaUnroll; ++q) {
                const int64_t e = eb + 1024 * q;
                vj[q] = e < e1 ? a.ev_j[e] : -1; vi[q] = e < e1 ? a.ev_i[e] : 0;
            }
#pragma unroll
            for (int q = 0; q < kMetaUnroll; ++q)
                if (vj[q] >= 0) {
                    const uint32_t code = 2u * (uint32_t)(eb + 1024 * q - e0);
                    const int gi = vi[q] >> kBucketShift, gj = vj[q] >> kBucketShift;
                    dst[off[gi] + atomicAdd(&cnt[gi], 1u)] = make_uint2((uint32_t)vi[q], code);
                    dst[off[gj] + atomicAdd(&cnt[gj], 1u)] = make_uint2((uint32_t)vj[q], code | 1u);
                }
        }
        __syncthreads();
    }
}


struct RoundMArgs {
    int64_t e_begin, e_end;      // events updated by this launch
    int staged;                  // 1: the staging rows are in use (the metadata was made with stage_max > 1)
};

// The S-round update launch on pre-pass metadata.  Update semantics, arithmetic and the three kinds of row stores are those
// of k_round; user rows always stay in dP (the epoch path applies them per group of rounds).  The launch ends behind its
// stores: the round's contended rows are rewritten by k_round_fold.
// At most 96 SGPRs: a CU then holds 7 workgroups of this kernel (floor(800 / (96 + 16)); with the 106 the compiler would
// take it is 6, whatever the register-file arithmetic says -- DESIGN.md section 5); the excess lives in VGPR lanes.
// BIGQ: item matrices (+ staging rows) of 2 GiB and more -- the item rows, the staging rows and dQ are addressed through 64-bit
// row pointers (scalar base + lane offset; lanes beyond k masked by hand) instead of one buffer descriptor with 32-bit row offsets.
template <int KR, int TPW, bool BIGQ = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96))) k_round_m(TrainArgs a, RoundMArgs ra, const int32_t *__restrict__ evu,
                                                 const int32_t *__restrict__ evi, const int32_t *__restrict__ evj,
                                                 const uint32_t *__restrict__ mti, const uint32_t *__restrict__ mtj) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t base = ra.e_begin + wave * TPW;
    if (base >= ra.e_end) return;
    YUE_STAMP(0, "");

    // batch header through the scalar unit: (u, i, j) and the two metadata words of the TPW events, ONE block load per array
    // (the arrays carry kHeaderSlack words of slack behind their last event; with per-event clamped indices the compiler
    // issued the 5 * TPW words as a chain of ten dependent groups of single-word loads: 5 us of a wave's 16)
    int hu[TPW], hi_[TPW], hj[TPW];
    uint32_t hmi[TPW], hmj[TPW];
    {
        const HeaderBlock<TPW> bu = *reinterpret_cast<const HeaderBlock<TPW> *>(evu + base);
        const HeaderBlock<TPW> bi = *reinterpret_cast<const HeaderBlock<TPW> *>(evi + base);
        const HeaderBlock<TPW> bj = *reinterpret_cast<const HeaderBlock<TPW> *>(evj + base);
        const HeaderBlock<TPW> bmi = *reinterpret_cast<const HeaderBlock<TPW> *>(mti + base);
        const HeaderBlock<TPW> bmj = *reinterpret_cast<const HeaderBlock<TPW> *>(mtj + base);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const bool ex = base + t < ra.e_end;
            hu[t] = ex ? bu.w[t] : bu.w[0]; hi_[t] = ex ? bi.w[t] : bi.w[0]; hj[t] = ex ? bj.w[t] : -1;
            hmi[t] = ex ? (uint32_t)bmi.w[t] : 0u; hmj[t] = ex ? (uint32_t)bmj.w[t] : 0u;
        }
    }
    int j = -1;                                          // lane t keeps the negative of event t (the loss works per lane)
#pragma unroll
    for (int t = 0; t < TPW; ++t) if (lane == t) j = hj[t];
    YUE_STAMP(1, "s_waitcnt lgkmcnt(0)");
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
    // the staging rows lie behind the n item rows in the same allocation (TrainArgs.stage == Q + n * k): one descriptor
    // serves gathers, in-place stores and staged stores -- a store's destination is a scalar offset, not a branch
    const unsigned stage0 = (unsigned)a.n * row_bytes;
    const uint64_t qsbytes = qbytes + (ra.staged ? 2ull * (uint64_t)(ra.e_end - ra.e_begin) * row_bytes : 0ull);
    const int qsrec = (int)(qsbytes < 0x7fffffffull ? qsbytes : 0x7fffffffull);
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(a.Q, 0, qsrec, kRsrcFlags);
    const auto rsdQ = __builtin_amdgcn_make_buffer_rsrc(a.dQ, 0, qrec, kRsrcFlags);
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc(a.P + (uint64_t)u0 * k, 0, prec, kRsrcFlags);
    const auto rsdP = __builtin_amdgcn_make_buffer_rsrc(a.dP + (uint64_t)u0 * k, 0, prec, kRsrcFlags);

    unsigned oi[TPW], oj[TPW], ou[TPW], ru_[TPW];
    bool ok[TPW];
    float qi[TPW][KR], qj[TPW][KR], p[TPW][KR];
    unsigned el[KR];                                     // BIGQ: the lane's element of a row (clamped), and whether it exists
    bool ev[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; ev[r] = e < k; el[r] = ev[r] ? e : 0u; }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        ru_[t] = (unsigned)hu[t];
        const int tj = hj[t];
        ok[t] = tj >= 0;                                 // no event / sampler gave up: nothing is written
        oi[t] = (unsigned)hi_[t] * row_bytes; oj[t] = (ok[t] ? (unsigned)tj : 0u) * row_bytes;
        ou[t] = (base + t < ra.e_end ? ru_[t] - u0 : 0u) * row_bytes;
        if (BIGQ) {
            const float *rowi = a.Q + (uint64_t)(unsigned)hi_[t] * k, *rowj = a.Q + (uint64_t)(ok[t] ? (unsigned)tj : 0u) * k;
#pragma unroll
            for (int r = 0; r < KR; ++r) { qi[t][r] = ev[r] ? rowi[el[r]] : 0.0f; qj[t][r] = ev[r] ? __builtin_nontemporal_load(rowj + el[r]) : 0.0f; }
        } else {
#pragma unroll
            for (int r = 0; r < KR; ++r) { qi[t][r] = YUE_M_LOAD_I(rsQ, vo[r], oi[t]); qj[t][r] = YUE_M_LOAD_J(rsQ, vo[r], oj[t]); }
        }
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
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const float c = rdlane(cs, t);
        const unsigned cli = meta_kind(hmi[t]), clj = meta_kind(hmj[t]);
        const bool uniq_i = cli == kMetaUnique, uniq_j = clj == kMetaUnique;
        const bool hot_i = cli == kMetaHot, hot_j = clj == kMetaHot;
        // where the row's store goes: the row itself (only touch of the round) or the touch's staging row
        const unsigned wi = uniq_i ? oi[t] : stage0 + meta_slot(hmi[t]) * row_bytes;
        const unsigned wj = uniq_j ? oj[t] : stage0 + meta_slot(hmj[t]) * row_bytes;
        if (ok[t]) {                                     // wave-uniform
            run_ok = true;
            Elem o[KR];
#pragma unroll
            for (int r = 0; r < KR; ++r) { o[r] = bpr_elem(p[t][r], qi[t][r], qj[t][r], c, a.ru, a.ri); dp[r] += o[r].p2 - p[t][r]; }
            // the new row in place, or (new - old) to the staging row: one store, value and offset selected; a hot row's
            // difference goes into dQ with float atomics (one wave-uniform branch per touch)
            if (BIGQ) {
                const uint64_t ri_ = (uint64_t)(unsigned)hi_[t] * k, rj_ = (uint64_t)(unsigned)hj[t] * k;
                float *di = hot_i ? a.dQ + ri_ : uniq_i ? a.Q + ri_ : a.stage + (uint64_t)meta_slot(hmi[t]) * k;
                float *dj = hot_j ? a.dQ + rj_ : uniq_j ? a.Q + rj_ : a.stage + (uint64_t)meta_slot(hmj[t]) * k;
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    if (ev[r]) {
                        if (hot_i) unsafeAtomicAdd(di + el[r], o[r].qi2 - qi[t][r]); else di[el[r]] = uniq_i ? o[r].qi2 : o[r].qi2 - qi[t][r];
                        if (hot_j) unsafeAtomicAdd(dj + el[r], o[r].qj2 - qj[t][r]); else dj[el[r]] = uniq_j ? o[r].qj2 : o[r].qj2 - qj[t][r];
                    }
                }
            } else {
            if (!hot_i) {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_BSTORE(uniq_i ? o[r].qi2 : o[r].qi2 - qi[t][r], rsQ, vo[r], wi);
            } else {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_M_HOT(o[r].qi2 - qi[t][r], rsdQ, vo[r], oi[t]);
            }
            if (!hot_j) {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_BSTORE(uniq_j ? o[r].qj2 : o[r].qj2 - qj[t][r], rsQ, vo[r], wj);
            } else {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_M_HOT(o[r].qj2 - qj[t][r], rsdQ, vo[r], oj[t]);
            }
            }
        }
        // end of a run of equal users (or of the batch): flush the summed P[u] differences
        const bool exists = base + t < ra.e_end;
        const bool last = exists && ((t == TPW - 1) || base + t + 1 >= ra.e_end || ru_[t + 1 < TPW ? t + 1 : t] != ru_[t]);
        if (last) {
            if (run_ok) {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_M_DP(dp[r], rsdP, vo[r], ou[t]);
            }
#pragma unroll
            for (int r = 0; r < KR; ++r) dp[r] = 0.0f;
            run_ok = false;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    YUE_STAMP(4, ""); YUE_STAMP(5, ""); YUE_STAMP(6, "");      // (no retire phases in this kernel: the diagnostic tool's columns 5, 6 stay empty)
#pragma unroll
    for (int off = 1; off < TPW; off <<= 1) nll += __shfl_xor(nll, off);
    if (lane == 0 && nll != 0.0) atomicAdd(a.nll_slots + (wave & (kNllSlots - 1)), nll);
    YUE_STAMP(7, "s_waitcnt vmcnt(0)");
}

// The S-round update launch with SEQUENTIAL USER ROWS (round 4; one GPU: a user's events all lie in one round and on one rank).
// k_round_m evaluates every triplet of a round on the round-start factors, so a user's ~50 events of an epoch take ONE summed
// step of P[u] -- that, not the staleness of the item rows, is most of the distance between S-round and the reference's loop
// (the P column does not depend on W).  Here one wave owns one user of the round and walks the user's
// events in order with P[u] in registers, exactly as the reference does (BPR.py:50-51, :55); the item rows keep round semantics
// (read as the round started; in place / staging row / float atomics by the pre-pass's metadata, finished by k_round_fold).
// P[u] is stored in place when the user is done: no dP, no float atomics on user rows, no k_apply_range.
// Oracle: oracle/bpr_oracle.c: orc_bpr_rounds_seq_user (a round of one event = the reference's loop).
// FAST: the step's coefficient in single precision (chain_coef_fast: five instructions instead of ~35 double-precision ones on
// all lanes -- k_round_m computes eight sigmoids at once in the lanes, a chain of dependent events cannot).  The loss is summed
// from double-precision sigmoids of the events' margins, which this kernel leaves in a buffer for k_loss_margins (one pass per epoch).
struct RoundUArgs {
    int64_t u_begin, u_end;      // users of the round: one wave each
    const int64_t *ev_ptr;       // [m + 1] first event of a user
    int64_t e_begin, e_end;      // events of the round (size of the staging area)
    int staged;
    float *margins;              // [E] the events' margins x (BPR.py:50), for the loss (k_loss_margins)
};

template <int KR, bool FAST>
__global__ void __launch_bounds__(256) k_round_u(TrainArgs a, RoundUArgs ra, const int32_t *__restrict__ evi, const int32_t *__restrict__ evj,
                                                 const uint32_t *__restrict__ mti, const uint32_t *__restrict__ mtj) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t u = ra.u_begin + wave;
    if (u >= ra.u_end) return;
    const int64_t e0 = ra.ev_ptr[u], e1 = ra.ev_ptr[u + 1];
    if (e1 <= e0) return;
    const unsigned k = (unsigned)a.k;
    const unsigned row_bytes = k * 4u;
    unsigned vo[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; vo[r] = e < k ? e * 4u : kOobOffset; }
    const uint64_t qbytes = (uint64_t)a.n * row_bytes;
    const int qrec = (int)(qbytes < 0x7fffffffull ? qbytes : 0x7fffffffull);
    const unsigned stage0 = (unsigned)a.n * row_bytes;   // the staging rows lie behind the n item rows in the same allocation
    const uint64_t qsbytes = qbytes + (ra.staged ? 2ull * (uint64_t)(ra.e_end - ra.e_begin) * row_bytes : 0ull);
    const int qsrec = (int)(qsbytes < 0x7fffffffull ? qsbytes : 0x7fffffffull);
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(a.Q, 0, qsrec, kRsrcFlags);
    const auto rsdQ = __builtin_amdgcn_make_buffer_rsrc(a.dQ, 0, qrec, kRsrcFlags);
    float *prow = a.P + (uint64_t)u * k;
    float p[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; p[r] = e < k ? prow[e] : 0.0f; }
    const float lrf = (float)a.lr;
    const unsigned len = (unsigned)(e1 - e0 < 0x7fffffff ? e1 - e0 : 0x7fffffff);
    for (unsigned seg = 0; seg < len; seg += 64u) {
        // the segment's headers in the lanes of four registers (one vector load each), a v_readlane per use
        int hI, hJ;
        uint32_t hMI, hMJ;
        {
            const bool ex = seg + (unsigned)lane < len;
            const int64_t e = e0 + seg + (ex ? lane : 0);
            hI = evi[e]; hJ = evj[e]; hMI = mti[e]; hMJ = mtj[e];
            if (!ex) { hI = 0; hJ = -1; hMI = 0u; hMJ = 0u; }
        }
        const unsigned seg_len = len - seg < 64u ? len - seg : 64u;
        float xs = 0.0f;                                 // lane t keeps the margin of the segment's event t (the loss is summed from them by k_loss_margins: no double-precision code in this kernel)
        // the two item rows of an event, gathered D events ahead of the arithmetic
        auto gather = [&](float (&qi)[KR], float (&qj)[KR], unsigned t) {
            const int ti = __builtin_amdgcn_readlane(hI, t), tj = __builtin_amdgcn_readlane(hJ, t);
            const unsigned oi = (unsigned)ti * row_bytes, oj = (unsigned)(tj < 0 ? 0 : tj) * row_bytes;
#pragma unroll
            for (int r = 0; r < KR; ++r) { qi[r] = YUE_M_LOAD_I(rsQ, vo[r], oi); qj[r] = YUE_M_LOAD_J(rsQ, vo[r], oj); }
        };
        auto step = [&](const float (&qi)[KR], const float (&qj)[KR], unsigned t) {
            const int tj = __builtin_amdgcn_readlane(hJ, t);
            if (tj < 0) return;                          // (wave-uniform) the sampler gave up on the event: nothing is written
            const int ti = __builtin_amdgcn_readlane(hI, t);
            const uint32_t mi = (uint32_t)__builtin_amdgcn_readlane((int)hMI, t), mj = (uint32_t)__builtin_amdgcn_readlane((int)hMJ, t);
            // per-lane partials in element order 64 r + l, r ascending (oracle/bpr_oracle.c: dot64)
            float ai = 0.0f, aj = 0.0f;
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float a1 = p[r] * qi[r]; ai = ai + a1; const float a2 = p[r] * qj[r]; aj = aj + a2; }
            const float x = wave_sum(ai) - wave_sum(aj);                 // BPR.py:50, fp32 margin
            const float c = FAST ? chain_coef_fast(x, lrf) : (float)(a.lr * (1.0 - chain_sigmoid(x)));
            xs = (unsigned)lane == t ? x : xs;
            const unsigned cli = meta_kind(mi), clj = meta_kind(mj);
            const bool uniq_i = cli == kMetaUnique, uniq_j = clj == kMetaUnique;
            const unsigned oi = (unsigned)ti * row_bytes, oj = (unsigned)tj * row_bytes;
            // where the row's store goes: the row itself (only touch of the round) or the touch's staging row
            const unsigned wi = uniq_i ? oi : stage0 + meta_slot(mi) * row_bytes;
            const unsigned wj = uniq_j ? oj : stage0 + meta_slot(mj) * row_bytes;
            Elem o[KR];
#pragma unroll
            for (int r = 0; r < KR; ++r) { o[r] = bpr_elem(p[r], qi[r], qj[r], c, a.ru, a.ri); p[r] = o[r].p2; }   // BPR.py:51-57; the user row moves on
            if (cli != kMetaHot) {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_BSTORE(uniq_i ? o[r].qi2 : o[r].qi2 - qi[r], rsQ, vo[r], wi);
            } else {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_M_HOT(o[r].qi2 - qi[r], rsdQ, vo[r], oi);
            }
            if (clj != kMetaHot) {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_BSTORE(uniq_j ? o[r].qj2 : o[r].qj2 - qj[r], rsQ, vo[r], wj);
            } else {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_M_HOT(o[r].qj2 - qj[r], rsdQ, vo[r], oj);
            }
        };
        // a ring of D events: the rows of event t + D are requested when event t has been taken (lanes past the segment's end
        // hold item 0 / no negative: their gathers read row 0 and are never looked at)
        constexpr int D = KR == 1 ? 8 : 4;               // events in flight per wave (k <= 64, rows of 256 bytes: 8 -- config 2 2.31e9 -> 2.43e9 triplets/s; 16 there, or 8 at k = 128, change nothing)
        float qi[D][KR], qj[D][KR];
#pragma unroll
        for (int s_ = 0; s_ < D; ++s_) gather(qi[s_], qj[s_], (unsigned)s_);
        for (unsigned t = 0; t < seg_len; t += D) {
#pragma unroll
            for (int s_ = 0; s_ < D; ++s_) {
                step(qi[s_], qj[s_], t + s_);
                gather(qi[s_], qj[s_], (t + s_ + D) & 63u);
            }
        }
        if ((unsigned)lane < seg_len) ra.margins[e0 + seg + lane] = xs;
    }
#pragma unroll
    for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; if (e < k) prow[e] = p[r]; }
}

// sum over the events with a negative of -log(sigmoid(margin)) (BPR.py:58), double precision, into the loss slots
__global__ void __launch_bounds__(256) k_loss_margins(const float *__restrict__ x, const int32_t *__restrict__ evj, int64_t E, double *nll_slots) {
    double nl = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride)
        if (evj[e] >= 0) nl += -log(chain_sigmoid(x[e]));
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) nl += __shfl_xor(nl, off);
    if ((threadIdx.x & 63) == 0 && nl != 0.0) atomicAdd(nll_slots + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (kNllSlots - 1)), nl);
}

// Rewrites the contended item rows of one round from the pre-pass's fold list: row += the block of staged differences in
// ticket order, or += the row of dQ (zeroed again) for a hot row.  A wave takes EPG consecutive list entries at a time
// (rows of neighbouring items: their staging blocks lie behind each other), all loads before the first store.
struct FoldArgs {
    const uint2 *fold;                     // this round's entries (16-byte aligned)
    const unsigned long long *round_rows;  // this round's counters (high half: entries)
    float *Q, *dQ;
    const float *stage;
    int k;
    uint32_t capacity;                     // entries the round's list area has room for (a multiple of 4)
};

template <int KR, int EPG>
__device__ __forceinline__ void fold_group(const FoldArgs &f, int lane, unsigned k, uint32_t nact, const uint2 (&ent)[4]) {
    // straight-line loads (no wait between the entries): the first two rows of a block always (a hot row reads its row of
    // dQ), rows 2..3 behind a wave-uniform branch on the row's touch count; lanes beyond k read element k - 1.  Larger
    // blocks (rows 4..) are summed behind the stores of the group, four rows at a time.
    float x[EPG][KR], st[EPG][4][KR];
    unsigned el[KR];
    bool big = false;
#pragma unroll
    for (int r = 0; r < KR; ++r) el[r] = min(64u * r + lane, k - 1u);
#pragma unroll
    for (int sl = 0; sl < EPG; ++sl) {
        if ((uint32_t)sl >= nact) continue;
        const uint32_t c = fold_count(ent[sl].y);
        const bool hot = c == 0u;
        const float *row = f.Q + (uint64_t)ent[sl].x * k;
        const float *src = hot ? f.dQ + (uint64_t)ent[sl].x * k : f.stage + (uint64_t)fold_block(ent[sl].y) * k;
#pragma unroll
        for (int r = 0; r < KR; ++r) { st[sl][0][r] = YUE_FOLD_LD(src + el[r]); if (!hot) st[sl][1][r] = YUE_FOLD_LD(src + (uint64_t)k + el[r]); }
        if (c > 2u) {
#pragma unroll
            for (unsigned q = 2; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < KR; ++r) st[sl][q][r] = YUE_FOLD_LD(src + (uint64_t)(q < c ? q : 0u) * k + el[r]);
        }
        big = big || c > 4u;
#pragma unroll
        for (int r = 0; r < KR; ++r) x[sl][r] = YUE_FOLD_LDROW(row + el[r]);
    }
#pragma unroll
    for (int sl = 0; sl < EPG; ++sl) {
        if ((uint32_t)sl >= nact) continue;
        const uint32_t c = fold_count(ent[sl].y);
        const bool hot = c == 0u;
        float *row = f.Q + (uint64_t)ent[sl].x * k;
        float *dr = f.dQ + (uint64_t)ent[sl].x * k;
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const unsigned e = 64u * r + lane;
            float acc = st[sl][0][r];
#pragma unroll
            for (unsigned q = 1; q < 4; ++q) { const float v = q < c ? st[sl][q][r] : 0.0f; acc = acc + v; }
            st[sl][0][r] = acc;
            if (c <= 4u && e < k) { row[e] = x[sl][r] + acc; if (hot) dr[e] = 0.0f; }
        }
    }
    if (big) {                                           // wave-uniform; the rest of a large block in ticket order
#pragma unroll
        for (int sl = 0; sl < EPG; ++sl) {
            if ((uint32_t)sl >= nact) continue;
            const uint32_t c = fold_count(ent[sl].y);
            if (c <= 4u) continue;
            const float *src = f.stage + (uint64_t)fold_block(ent[sl].y) * k;
            float *row = f.Q + (uint64_t)ent[sl].x * k;
            float acc[KR];
#pragma unroll
            for (int r = 0; r < KR; ++r) acc[r] = st[sl][0][r];
            constexpr unsigned STEP = KR == 1 ? 8u : 4u;      // rows in flight per step (k <= 64: blocks are long, rows short)
            for (uint32_t q0 = 4u; q0 < c; q0 += STEP) {
                float v[STEP][KR];
#pragma unroll
                for (unsigned q = 0; q < STEP; ++q)
#pragma unroll
                    for (int r = 0; r < KR; ++r) v[q][r] = YUE_FOLD_LD(src + (uint64_t)(q0 + q < c ? q0 + q : q0) * k + el[r]);
#pragma unroll
                for (unsigned q = 0; q < STEP; ++q)
#pragma unroll
                    for (int r = 0; r < KR; ++r) acc[r] = acc[r] + (q0 + q < c ? v[q][r] : 0.0f);
            }
#pragma unroll
            for (int r = 0; r < KR; ++r) { const unsigned e = 64u * r + lane; if (e < k) row[e] = x[sl][r] + acc[r]; }
        }
    }
}

template <int KR>
__global__ void __launch_bounds__(256) k_round_fold(FoldArgs f) {
    constexpr int EPG = KR == 4 ? 2 : 4;                // list entries per group (register budget: EPG * 5 * KR row registers)
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nwaves = gridDim.x * 4u;
    const unsigned k = (unsigned)f.k;
    // The first group's entries are requested together with the round's entry count (one round trip less on a kernel that
    // starts with a chain of dependent loads): the list area of a round has room for f.capacity entries, whatever they
    // hold is only used below the count.
    const uint32_t b0 = wave * EPG;
    uint2 ent[4];
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    u32x8 ev;
    unsigned long long cntw;
    const bool spec = b0 + 4u <= f.capacity;              // else: too close to the end of the list area, the group is loaded below
    const uint2 *first = f.fold + (spec ? b0 : 0u);
    // (written out: the compiler would load the count, branch on it, and only then ask for the entries)
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(ev), "=&s"(cntw) : "s"(first), "s"(f.round_rows) : "memory");
    const uint32_t cnt = (uint32_t)(cntw >> 32);
    ent[0] = make_uint2(ev[0], ev[1]); ent[1] = make_uint2(ev[2], ev[3]); ent[2] = make_uint2(ev[4], ev[5]); ent[3] = make_uint2(ev[6], ev[7]);
    if (b0 < cnt) {
        if (!spec) {
#pragma unroll
            for (int sl = 0; sl < EPG; ++sl) ent[sl] = f.fold[b0 + sl < cnt ? b0 + sl : b0];
        }
        fold_group<KR, EPG>(f, lane, k, min((uint32_t)EPG, cnt - b0), ent);
    }
    for (uint32_t b = b0 + nwaves * EPG; b < cnt; b += nwaves * EPG) {
#pragma unroll
        for (int sl = 0; sl < EPG; ++sl) ent[sl] = f.fold[b + sl < cnt ? b + sl : b];
        fold_group<KR, EPG>(f, lane, k, min((uint32_t)EPG, cnt - b), ent);
    }
}

}  // namespace yue
