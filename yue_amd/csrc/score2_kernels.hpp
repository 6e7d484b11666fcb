// Two-phase scoring for long item catalogues (k in {16, 32, 64, 128}, N <= 64): the same lists and scores as k_topn_scan*,
// with the MFMA work and the sequential selection in separate launches.
//
// evalRanking's selection (base/IterativeRecommender.py:121-144) only ever acts on a candidate whose score exceeds the
// user's current threshold a[N-1], and that threshold never falls.  So for a CHUNK of items scanned with the thresholds
// the users had when the chunk started, a candidate whose exact score cannot exceed that older threshold can be dropped
// without changing anything; everything else is looked at exactly, in ascending item order, as the reference does.
//   k_scan_filter   (per chunk) bf16 MFMA scores of all users against the chunk's items; a pair survives if
//                   bf16 score + 1.01 * 2^-7 ||P_u|| max||Q_tile|| > threshold_u (the error bound of score_kernels.hpp);
//                   one 32-bit word per (user, tile of 32 items) -- nothing else: no state, no exact arithmetic, no
//                   survivor handling in the matrix loop;
//   k_scan_select   (per chunk) one wave per user: walks the user's words, gathers the surviving item rows 64 at a time
//                   (coalesced row loads through a small LDS staging area, one survivor per lane), computes the exact
//                   k-ascending fp32 fma chain per lane -- every lane busy -- and feeds the exact scores in ascending item
//                   order through the reference's state machine, whose N slots live one per lane (an insert position is a
//                   ballot + popcount, an overwrite one predicated move);
// chunks grow geometrically (512, 512, 1024, ...), so a threshold is never staler than a factor two in items seen.
// The first chunk runs through k_topn_scan_bf16p, which also seeds the lists (base/IterativeRecommender.py:107-116).
#pragma once
#include "score_kernels.hpp"

namespace yue {

// timing-only ablations of k_scan_filter (make EXTRA=-DYUE_FILTER_ABL=bits OBJDIR=build_fablN LIB=libyue_hip_fablN.so; lists WRONG by
// construction): 1 no barrier between stages, 2 no sifting of the scores, 8 no item rows fetched / committed
#ifndef YUE_FILTER_ABL
#define YUE_FILTER_ABL 0
#endif

struct FilterArgs {
    const float *P;
    const __bf16 *Qb;            // item factors rounded to bf16, rows padded with zero rows to a multiple of 64
    const int32_t *users;
    int64_t nu, n;               // users of this launch, items of the catalogue
    int N;
    const float *thr_rows;       // [nu][N] the lists' scores so far: threshold of user t = thr_rows[t * N + N - 1]
    const float *tile_norm_max;  // max ||Q_i|| per tile of 32 items (global tile index)
    const float *tile_norm_sufmax; // max of tile_norm_max over this and all later tiles (a workgroup whose users are all settled against it stops)
    int64_t item0, item1;        // the chunk; item0 is a multiple of 64
    int64_t iters_per_block;     // stages of 64 items one workgroup takes (blockIdx.y-th share of the chunk)
    uint32_t *masks;             // [nu][mask_stride]: word w of a user = tile item0 / 32 + w; only words with survivors are written
    int64_t mask_stride;
    uint32_t *summary;           // [nu][sum_stride]: bit b of word s = mask word 32 s + b was written (every word of the chunk's
    int64_t sum_stride;          // range is written: no clearing between chunks)
    unsigned long long *work;    // slots as ScanArgs.work: [0] += tiles scored
};

__global__ void __launch_bounds__(256) k_q_to_bf16(const float *Q, __bf16 *Qb, int64_t count, int64_t padded) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < padded; t += stride) Qb[t] = t < count ? (__bf16)Q[t] : (__bf16)0.0f;
}

// Which regime is the scan in?  Every `step`-th user, a wave per 16 of them: is the user settled against all items from `item`
// on (||P_u|| * max later ||Q_i|| <= the threshold its first chunk left)?  counts[0] += sampled users, counts[1] += settled ones
// (one pair of atomics per wave: tens of thousands of atomics on one word would take longer than the scan's first chunk).
__global__ void __launch_bounds__(256) k_scan_settled_sample(const float *P, const int32_t *users, int64_t nu, int k, int N, const float *thr_rows,
                                                             const float *tile_norm_sufmax, int64_t item, int step, unsigned *counts) {
    const int lane = threadIdx.x & 63;
    const int64_t first = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    const float sm = tile_norm_sufmax[item / kScanTile];
    unsigned sampled = 0u, settled = 0u;
    for (int q = 0; q < 16; ++q) {
        const int64_t upos = (first + q) * step;
        if (upos >= nu) break;
        const float *prow = P + (int64_t)users[upos] * k;
        float ss = 0.0f;
        for (int e = lane; e < k; e += 64) ss = __builtin_fmaf(prow[e], prow[e], ss);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        sampled += 1u;
        settled += __builtin_sqrtf(ss) * 1.0001f * sm <= thr_rows[upos * N + N - 1] ? 1u : 0u;
    }
    if (lane == 0 && sampled) { atomicAdd(counts, sampled); atomicAdd(counts + 1, settled); }
}

// Users whose first chunk held fewer than N unmasked candidates (the fused kernel left id -1 in their rows): their positions and
// ids into a list for a fused scan over ALL items of just these users (yue_topn_scan), and a threshold of +infinity into their
// rows so that the filter / select pair finds nothing to do for them meanwhile.
__global__ void __launch_bounds__(256) k_scan_collect_few(const int32_t *ids, float *scores, const int32_t *users, int64_t nu, int N,
                                                          int32_t *few_pos, int32_t *few_users, unsigned *count) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nu || ids[t * N] >= 0) return;
    const unsigned q = atomicAdd(count, 1u);
    few_pos[q] = (int32_t)t; few_users[q] = users[t];
    scores[t * N + N - 1] = INFINITY;
}
// rows of the few-candidates users' lists back to their places
__global__ void __launch_bounds__(256) k_scan_scatter_rows(const int32_t *src_ids, const float *src_scores, const int32_t *pos, int64_t count, int N,
                                                           int32_t *ids, float *scores) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * N) return;
    const int64_t row = t / N, col = t - row * N;
    ids[(int64_t)pos[row] * N + col] = src_ids[t]; scores[(int64_t)pos[row] * N + col] = src_scores[t];
}

// A wave takes UB blocks of 32 users: one item-tile fragment read from LDS feeds UB MFMAs.  The library's form is UB = 2 in
// workgroups of four waves (the same 256 users per workgroup as UB = 1 with eight): 168 VGPRs = three waves per SIMD, one from
// each of three workgroups, half the LDS reads and half the barriers per MFMA (config 5, factors of 25 epochs: 55 against 69 ms
// per scan on the same box, the matrix pipe busy 64 % of the filter's cycles against 54 %; the shader clock falls to 1.7 GHz).
//
// DMA (UB = 2, four waves, k = 64 or 128): the item rows go from global memory straight into LDS (global_load_lds_dwordx4: no
// staging registers, no ds_write; one wave-instruction fills 1 KB = 64 consecutive 16-byte pieces, so the rows lie unpadded and
// the 16-byte column of a piece is XORed with a function of its row -- on the SOURCE address and again in the fragment reads --
// which keeps ds_read_b128 free of bank conflicts).  (Tried with the freed registers: a whole tile's fragments resident and the
// two user blocks one after the other, each block's scores sifted under the other's MFMAs -- 128 registers of fragments and
// accumulators are live at the sift then, the compiler spills the user fragments inside the stage loop.)
__host__ __device__ constexpr size_t filter_tile_bytes(int k) { return 2u * 64u * (size_t)(k + kScanBfPad) * 2u; }
__host__ __device__ constexpr size_t filter_lds_bytes(int k) { return filter_tile_bytes(k) + 16u; }
template <int K16, int WAVES, int UB, bool SETTLE, bool DMA = false>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(UB == 2 ? 3 : 2))) k_scan_filter(FilterArgs a) {
    static_assert(!DMA || (UB == 2 && WAVES == 4 && (K16 == 4 || K16 == 8)), "the DMA form: two user blocks, four waves, k = 64 or 128");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int K = 16 * K16, LDB = K + kScanBfPad, NT = 64 * WAVES, ROWS = 64;
    __bf16 *btile = reinterpret_cast<__bf16 *>(lds_raw);                 // [2][64][LDB]   (DMA: [2][64][K], swizzled)
    // waves none of whose users can take a later item: a word behind the tiles (a static __shared__ word would move the tiles off
    // offset 0, and every fragment read of the DMA layout would pay an add beside its XOR)
    unsigned &settled_waves = *reinterpret_cast<unsigned *>(lds_raw + filter_tile_bytes(K));
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    bool wave_settled = false;
    if (tid == 0) settled_waves = 0u;
    if (DMA && (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds_raw != 0u) __builtin_trap();     // (see frag() below)
    int64_t upos[UB];
    bool uvalid[UB];
    bf16x8 af[UB][K16];
    float mu[UB], pn[UB], thr[UB];
#pragma unroll
    for (int b = 0; b < UB; ++b) {
        upos[b] = ((int64_t)blockIdx.x * WAVES + w) * (32 * UB) + 32 * b + r;
        uvalid[b] = upos[b] < a.nu;
        const int32_t uid = a.users[uvalid[b] ? upos[b] : 0];
        const float *prow = a.P + (int64_t)uid * K;
#pragma unroll
        for (int s = 0; s < K16; ++s)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) af[b][s][jj] = (__bf16)prow[16 * s + 8 * h + jj];
        float ss = 0.0f;
        for (int e = h * (K / 2); e < (h + 1) * (K / 2); ++e) ss = __builtin_fmaf(prow[e], prow[e], ss);
        ss = lo_bcast(ss) + hi_bcast(ss);
        mu[b] = __builtin_sqrtf(ss) * (1.01f / 128.0f);        // the margin factors of k_topn_scan_bf16p
        pn[b] = mu[b] * (128.0f * 1.0001f / 1.01f);
        thr[b] = uvalid[b] ? a.thr_rows[upos[b] * a.N + a.N - 1] : INFINITY;
    }

    // one stage = 64 consecutive rows of Qb = ROWS * K * 2 bytes, contiguous; 16-byte pieces
    constexpr int PIECES = ROWS * K * 2 / 16, PF = DMA ? 1 : (PIECES + NT - 1) / NT;
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    u32x4s pre[PF];
    int ldo[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) { const int pc = tid + NT * q; const int row = pc / (K / 8); ldo[q] = row * LDB + (pc - row * (K / 8)) * 8; }
    auto fetch = [&](int64_t it0) {
        const u32x4s *src = reinterpret_cast<const u32x4s *>(a.Qb + it0 * K);
#pragma unroll
        for (int q = 0; q < PF; ++q) if (PIECES % NT == 0 || tid + NT * q < PIECES) pre[q] = src[tid + NT * q];
    };
    auto commit = [&](int stage) {
        __bf16 *dst = btile + stage * ROWS * LDB;
#pragma unroll
        for (int q = 0; q < PF; ++q) if (PIECES % NT == 0 || tid + NT * q < PIECES) *reinterpret_cast<u32x4s *>(dst + ldo[q]) = pre[q];
    };
    // DMA: RB bytes per row, PR pieces per row; one instruction of a wave = 64 pieces = RPI rows; wave w fills rows 16 w .. 16 w + 15 of
    // a stage with NI instructions.  Piece c of row y lies at column c ^ swz(y), swz(y) = y & 15 (k = 128: a row is the 256 bytes all
    // 64 banks span) or (y >> 1) & 7 (k = 64: two rows per span): the 16 lanes ds_read_b128 serves together (rows 0-3, 12-15 and 20-27
    // of a tile, one column) then fall on 16 different 16-byte slots of the span.
    constexpr int RB = K * 2, PR = K / 8, RPI = 64 / PR, NI = 16 / RPI;
    auto swz = [](int y) { return K16 == 8 ? (y & 15) : ((y >> 1) & 7); };
    // (byte offset of lane's piece in instruction i: row 16 w + RPI i + lane / PR, column (lane % PR) ^ swz(row); swz(row) = 4 i | lane / 16
    // in both layouts, so instruction i differs from instruction 0 by an XOR of 64 i in the column field and 1024 i bytes of rows)
    const unsigned dma_off0 = DMA ? (unsigned)((16 * w + lane / PR) * RB + (((lane % PR) ^ (lane >> 4)) << 4)) : 0u;
    // (buffer_load ... lds rather than global_load_lds: behind a FLAT-encoded load that may touch LDS the compiler's wait insertion
    // turns every later lgkmcnt wait into lgkmcnt(0) until the load itself has been waited for -- the ring below would lose its depth)
    const auto rsQb = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(a.Qb), 0, (int)(((a.n + 63) / 64 * 64) * K * 2), 0x00020000);
    auto dma = [&](int64_t it0, int stage) {
        unsigned off0 = dma_off0;
        asm volatile("" : "+v"(off0));                         // (recomputed per stage: hoisted per-instruction offsets would cost registers)
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQb, (__attribute__((address_space(3))) void *)(lds_raw + (stage * ROWS + 16 * w + RPI * i) * RB), 16,
                                                     (int)((off0 ^ (64u * i)) + 1024u * i), (int)(it0 * K * 2), 0, 0);
    };

    const int64_t niter_all = (a.item1 - a.item0 + ROWS - 1) / ROWS;
    const int64_t it_begin = (int64_t)blockIdx.y * a.iters_per_block;
    const int64_t niter = it_begin + a.iters_per_block < niter_all ? it_begin + a.iters_per_block : niter_all;
    unsigned tiles_done = 0;
    uint32_t summ[UB];                                       // lanes (r, 0): which of the running 32 mask words have survivors
#pragma unroll
    for (int b = 0; b < UB; ++b) summ[b] = 0u;
    if (it_begin >= niter) return;
    float sm_next = SETTLE ? a.tile_norm_sufmax[(a.item0 + it_begin * ROWS) / kScanTile] : 0.0f;
    float nu0_next, nu1_next;
    {
        const int64_t i0 = a.item0 + it_begin * ROWS;
        nu0_next = a.tile_norm_max[i0 / kScanTile]; nu1_next = (i0 + kScanTile < a.n) ? a.tile_norm_max[i0 / kScanTile + 1] : 0.0f;
    }
    if (DMA) dma(a.item0 + it_begin * ROWS, (int)(it_begin & 1));
    else { fetch(a.item0 + it_begin * ROWS); commit((int)(it_begin & 1)); }
    __syncthreads();                                         // (with a DMA in flight the compiler's barrier waits for it: vmcnt(0))
    for (int64_t it = it_begin; it < niter; ++it) {
        const int stage = (int)(it & 1);
        const int64_t it0 = a.item0 + it * ROWS;
        if (SETTLE && ((it - it_begin) & 63) == 0) {
            // every 64 stages (four summary words): can any user of this wave still take an item from here on?  (Cauchy-Schwarz
            // against the largest norm of ALL later items -- trained factors: the catalogue's tail cannot reach the lists.)  A wave
            // that cannot says so once in LDS; when all waves have, the rest of the workgroup's share has no survivors: zero
            // summary words, nothing else to write or read.  No barrier: a wave that reads the count too early leaves one check
            // later (its stages in between score nothing), the barriers of the stage loop only count waves that still run.
            if (!wave_settled) {
                const float sm = sm_next;
                {   // the next check's value now: its load latency stays out of the check
                    const int64_t itn = it + 64 < niter ? it + 64 : it;
                    sm_next = a.tile_norm_sufmax[(a.item0 + itn * ROWS) / kScanTile];
                }
                bool open_ = false;
#pragma unroll
                for (int b = 0; b < UB; ++b) open_ = open_ || (uvalid[b] && !(pn[b] * sm <= thr[b]));
                wave_settled = __ballot(open_) == 0ull;
                if (wave_settled && lane == 0) atomicAdd(&settled_waves, 1u);
            }
            if (wave_settled && __hip_atomic_load(&settled_waves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)WAVES) {
#pragma unroll
                for (int b = 0; b < UB; ++b)
                    if (h == 0 && uvalid[b])
                        for (int64_t g = it >> 4; g <= (niter - 1) >> 4; ++g) a.summary[upos[b] * a.sum_stride + g] = 0u;
                break;
            }
        }
        const __bf16 *tbb = btile + stage * ROWS * LDB;
        // the norms of the NEXT stage's tiles are asked for here, in front of the next stage's rows: their latency (and the
        // vmcnt(0) the compiler puts in front of their first use) would otherwise open every stage
        const float nu0 = nu0_next, nu1 = nu1_next;
        if (it + 1 < niter) {
            const int64_t tn = (it0 + ROWS) / kScanTile;
            nu0_next = a.tile_norm_max[tn]; nu1_next = (it0 + ROWS + kScanTile < a.n) ? a.tile_norm_max[tn + 1] : 0.0f;
            if (!(YUE_FILTER_ABL & 8)) { if (DMA) dma(it0 + ROWS, stage ^ 1); else fetch(it0 + ROWS); }
        }
        bool settled = true;                                 // Cauchy-Schwarz: no exact score of these tiles reaches the threshold
#pragma unroll
        for (int b = 0; b < UB; ++b) settled = settled && (!uvalid[b] || pn[b] * fmaxf(nu0, nu1) <= thr[b]);
        if (__ballot(!settled) != 0ull) {
            tiles_done += 2 * UB;
            uint32_t pm[UB][2];
            auto sift = [&](const f32x16 &sc, int b, int q) {
                if (YUE_FILTER_ABL & 2) { asm volatile("" :: "v"(sc)); pm[b][q] = 0u; return; }
                const float bar = thr[b] - mu[b] * (q ? nu1 : nu0);
                // most tiles hold no survivor for any user of the wave: one max over the lane's 16 scores decides that
                float mx = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
                for (int z = 3; z < 15; z += 2) mx = fmaxf(fmaxf(mx, sc[z]), sc[z + 1]);
                mx = fmaxf(mx, sc[15]);
                pm[b][q] = 0u;
                if (__ballot(bar < mx) != 0ull) {
                    uint32_t bits = 0u;
#pragma unroll
                    for (int z = 15; z >= 0; --z) bits = __builtin_amdgcn_alignbit(bits, __builtin_bit_cast(uint32_t, bar - sc[z]), 31);
                    const uint32_t pmask = (bits & 0xFu) | ((bits & 0xF0u) << 4) | ((bits & 0xF00u) << 8) | ((bits & 0xF000u) << 12);
                    pm[b][q] = pmask << (4 * h);
                }
            };
            // The stage's 2 K16 item fragments as ONE stream through a ring of RD registers, each ds_read_b128 RD - 1 k-steps
            // ahead of its UB MFMAs and across the seam between the two tiles (the scheduling barriers pin that order: left to
            // itself the compiler issues two reads and waits for them in front of two MFMAs).  UB = 1: the second tile
            // accumulates beside the first, whose survivors are sifted under the second's MFMAs; UB = 2: one generation of
            // accumulators (af[2][K16] + two generations would not leave three waves per SIMD).
            constexpr int J = 2 * K16, RD = J < 4 ? J : 4, SIFT0 = UB == 1 ? (K16 + 2 < J ? K16 + 2 : J - 1) : K16 - 1;
            const __bf16 *irow = tbb + r * LDB + 8 * h;
            // DMA layout: column 2 s + h of row r lies at ((2 s) ^ h ^ swz(r)) * 16 -- stage, row and column are separate bit fields of
            // the offset: one XOR per k-step and stage (K16 addresses, the second tile's reads add 32 rows in the instruction).  The
            // offset is an LDS address as it stands (the tiles begin at LDS address 0: no static __shared__ in this kernel) and passes
            // through an empty statement once per stage (addresses hoisted out of the stage loop for both stages would cost registers).
            typedef const __attribute__((address_space(3))) bf16x8 *lds_frag_ptr;
            unsigned frag0 = (unsigned)(stage * ROWS * RB + r * RB) | (unsigned)((h ^ swz(r)) << 4);
            if (DMA) asm volatile("" : "+v"(frag0));
            auto frag = [&](int j) {
                if constexpr (DMA) return *reinterpret_cast<lds_frag_ptr>((frag0 ^ (32u * (j % K16))) + (j / K16) * kScanTile * RB);
                else return *reinterpret_cast<const bf16x8 *>(irow + (j / K16) * kScanTile * LDB + 16 * (j % K16));
            };
            bf16x8 ring[RD];
#pragma unroll
            for (int j = 0; j < RD - 1; ++j) ring[j] = frag(j);
            f32x16 acc[2][UB];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int b = 0; b < UB; ++b)
#pragma unroll
                    for (int z = 0; z < 16; ++z) acc[q][b][z] = 0.0f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                if (j + RD - 1 < J) ring[(j + RD - 1) % RD] = frag(j + RD - 1);
#pragma unroll
                for (int b = 0; b < UB; ++b) acc[j / K16][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[j % RD], af[b][j % K16], acc[j / K16][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (j == SIFT0) {
#pragma unroll
                    for (int b = 0; b < UB; ++b) sift(acc[0][b], b, 0);
                }
            }
#pragma unroll
            for (int b = 0; b < UB; ++b) sift(acc[1][b], b, 1);
            const int64_t left = a.n - it0;                      // items of the catalogue from this stage on
#pragma unroll
            for (int b = 0; b < UB; ++b) {
                if (__ballot((pm[b][0] | pm[b][1]) != 0u) == 0ull) continue;
                uint32_t l0, h0, l1, h1;
                half_bcast(pm[b][0], l0, h0); half_bcast(pm[b][1], l1, h1);
                uint32_t p0 = l0 | h0, p1 = l1 | h1;
                if (left < 64) { if (left <= 32) { p1 = 0u; if (left < 32) p0 &= (1u << (uint32_t)left) - 1u; } else p1 &= (1u << (uint32_t)(left - 32)) - 1u; }
                if (h == 0 && uvalid[b] && (p0 | p1) != 0u) {
                    uint32_t *dst = a.masks + upos[b] * a.mask_stride + (it0 - a.item0) / kScanTile;
                    dst[0] = p0; dst[1] = p1;
                    summ[b] |= ((p0 != 0u ? 1u : 0u) | (p1 != 0u ? 2u : 0u)) << (2 * (unsigned)(it & 15));
                }
            }
        }
        if ((it & 15) == 15 || it + 1 == niter) {                // 16 stages = 32 mask words = one summary word (a block's share starts at a multiple of 16)
#pragma unroll
            for (int b = 0; b < UB; ++b) {
                if (h == 0 && uvalid[b]) a.summary[upos[b] * a.sum_stride + (it >> 4)] = summ[b];
                summ[b] = 0u;
            }
        }
        if (!DMA && it + 1 < niter && !(YUE_FILTER_ABL & 8)) commit(stage ^ 1);
        if (!(YUE_FILTER_ABL & 1)) __syncthreads();
    }
    if (lane == 0) atomicAdd(work_slot(a.work, (blockIdx.y * gridDim.x + blockIdx.x) * WAVES + w), (unsigned long long)tiles_done);
}

// ---------------------------------------------------------------------------------------------------------------------------
struct SelectArgs {
    const float *P, *Q;
    int64_t n;
    int k;
    const int32_t *users;
    int64_t nu;
    int N;
    const int64_t *mask_ptr;
    const int32_t *mask_idx;
    int mask_by_user;
    int32_t *ids;                // [nu][N] in: the lists so far (complete: N entries, scores descending); out: after the chunk
    float *scores;
    const uint32_t *masks;       // survivor words of the chunk, as k_scan_filter wrote them
    int64_t mask_stride;
    const uint32_t *summary;     // which words hold survivors
    int64_t sum_stride;
    int64_t item0, item1;
    unsigned long long *work;    // slots as ScanArgs.work: [1] += state-machine events, [2] += exact scores computed
    int true_topn;
};

constexpr int kSelKC = 32;           // floats of a row staged per pass
constexpr int kSelLd = kSelKC + 4;   // staging row stride (floats): 16 lanes x 16 bytes fall on 64 different banks
constexpr int kSelWin = 32;          // mask words whose survivors join the pending list at a time
constexpr int kSelList = 64 + 32 * kSelWin;   // pending survivors: a rest below 64 plus one window's worth
constexpr int kSelMaskCap = 256;     // masked (training) items of a user kept in LDS (longer rows: binary search in global memory)

__host__ __device__ inline size_t select_wave_bytes() {
    return 64 * kSelLd * sizeof(float) + 256 * sizeof(float) + kSelList * sizeof(int32_t) + kSelMaskCap * sizeof(int32_t);
}
__host__ __device__ inline size_t select_lds_bytes(int waves) { return (size_t)waves * select_wave_bytes(); }

// One wave per user.  Any k <= 256 (rows are read in passes of kSelKC elements, zero beyond k).
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_scan_select(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned char *base = lds_raw + (size_t)w * select_wave_bytes();
    float *stage = reinterpret_cast<float *>(base);                                  // [64][kSelLd]
    float *prow_l = stage + 64 * kSelLd;                                             // [256] the user's row (zero beyond k)
    int32_t *list = reinterpret_cast<int32_t *>(prow_l + 256);                       // [kSelList] pending survivors, ascending
    int32_t *mrow = list + kSelList;                                                 // [kSelMaskCap]
    const int64_t upos = (int64_t)blockIdx.x * WAVES + w;
    if (upos >= a.nu) return;
    const int k = a.k, N = a.N;
    const int32_t uid = a.users[upos];
    const float *prow = a.P + (int64_t)uid * k;
    for (int e = lane; e < 256; e += 64) prow_l[e] = e < k ? prow[e] : 0.0f;
    const int64_t mrow_id = a.mask_by_user ? (int64_t)uid : upos;
    const int64_t m0 = a.mask_ptr[mrow_id], m1 = a.mask_ptr[mrow_id + 1];
    const bool mask_in_lds = (m1 - m0) <= kSelMaskCap;
    const int mcount = mask_in_lds ? (int)(m1 - m0) : 0;
    for (int e = lane; e < mcount; e += 64) mrow[e] = a.mask_idx[m0 + e];
    // the list, one slot per lane (N <= 64), scores descending
    float sa = lane < N ? a.scores[upos * N + lane] : -INFINITY;
    int32_t si = lane < N ? a.ids[upos * N + lane] : -1;
    float thr = __shfl(sa, N - 1);
    unsigned long long events = 0, exact = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    auto is_masked = [&](int32_t item) -> bool {
        if (mask_in_lds) {
            int lo = 0, hi = mcount;
            while (lo < hi) { const int mid = (lo + hi) >> 1; const int32_t v = mrow[mid]; if (v < item) lo = mid + 1; else hi = mid; }
            return lo < mcount && mrow[lo] == item;
        }
        int64_t lo = m0, hi = m1;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; const int32_t v = a.mask_idx[mid]; if (v < item) lo = mid + 1; else hi = mid; }
        return lo < m1 && a.mask_idx[lo] == item;
    };
    const bool vec_rows = (k & 3) == 0;
    // elements [e0, e0 + 32) of the batch's rows: 8 lanes x 16 bytes per row, 8 rows per load instruction
    auto gather = [&](const int32_t (&ritem)[8], int e0, f32x4 (&v)[8]) {
        const int e = e0 + 4 * (lane & 7);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            v[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ritem[g] >= 0) {
                const float *src = a.Q + (int64_t)ritem[g] * k + e;
                if (vec_rows && e + 3 < k) v[g] = *reinterpret_cast<const f32x4 *>(src);
                else { if (e < k) v[g][0] = src[0]; if (e + 1 < k) v[g][1] = src[1]; if (e + 2 < k) v[g][2] = src[2]; if (e + 3 < k) v[g][3] = src[3]; }
            }
        }
    };
    // exact scores of list[b0 .. b0 + 64) and the state machine on them
    auto batch = [&](int b0, int count) {
        const bool have = lane < count;
        const int32_t item = have ? list[b0 + lane] : -1;
        const bool live = have && !is_masked(item);
        int32_t ritem[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) { const int row = 8 * g + (lane >> 3); ritem[g] = row < count ? list[b0 + row] : -1; }
        // acc = fma(P[e], Q[item][e], acc), e ascending (== score_chain of the oracle, == k_scores_one); the rows of the
        // next pass are in flight while this pass is summed
        float acc = 0.0f;
        f32x4 v[8];
        gather(ritem, 0, v);
        for (int e0 = 0; e0 < k; e0 += kSelKC) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < 8; ++g) *reinterpret_cast<f32x4 *>(stage + (8 * g + (lane >> 3)) * kSelLd + 4 * (lane & 7)) = v[g];
            if (e0 + kSelKC < k) gather(ritem, e0 + kSelKC, v);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const float *qr = stage + lane * kSelLd;
#pragma unroll
            for (int e = 0; e < kSelKC; e += 4) {
                const f32x4 pv = *reinterpret_cast<const f32x4 *>(prow_l + e0 + e);
                const f32x4 qv = *reinterpret_cast<const f32x4 *>(qr + e);
                acc = __builtin_fmaf(pv[0], qv[0], acc); acc = __builtin_fmaf(pv[1], qv[1], acc);
                acc = __builtin_fmaf(pv[2], qv[2], acc); acc = __builtin_fmaf(pv[3], qv[3], acc);
            }
        }
        exact += (unsigned long long)__popcll(__ballot(live));
        // the state machine, candidates in ascending item order = ascending lane
        unsigned long long cand = __ballot(live && thr < acc);
        while (cand) {
            const int l = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            // (l is wave-uniform: v_readlane, a few cycles, instead of a trip through the LDS crossbar per value)
            const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), l));
            if (!(thr < s)) continue;                                  // the threshold has risen meanwhile
            const int32_t it = __builtin_amdgcn_readlane(item, l);
            ++events;
            // first slot strictly below s (slots with a[q] >= s come first: the list is sorted descending)
            const int p = __popcll(__ballot(lane < N && sa >= s));
            if (a.true_topn) {
                // real top-N: insert with shift, the smallest entry drops out
                const float up_a = __shfl_up(sa, 1);
                const int32_t up_i = __shfl_up(si, 1);
                if (lane > p && lane < N) { sa = up_a; si = up_i; }
            }
            if (lane == p) { sa = s; si = it; }
            thr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sa), N - 1));
        }
    };

    const int64_t words = (a.item1 - a.item0 + kScanTile - 1) / kScanTile;
    const int64_t swords = (words + 31) / 32;
    const uint32_t *mw = a.masks + upos * a.mask_stride;
    const uint32_t *sw = a.summary + upos * a.sum_stride;
    int fill = 0;                                                      // pending survivors in list[0 .. fill)
    // summary words 64 at a time (one per lane = 2,048 mask words = 65,536 items); the mask words with survivors are then
    // fetched one per lane, 64 at a time, in ascending order
    for (int64_t s0 = 0; s0 < swords; s0 += 64) {
        uint32_t sword = (s0 + lane < swords) ? sw[s0 + lane] : 0u;
        if (__ballot(sword != 0u) == 0ull) continue;
        // number the set bits of the 64 summary words in ascending (lane, bit) order
        const int scnt = __popc(sword);
        int sincl = scnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(sincl, off); if (lane >= off) sincl += up; }
        const int stotal = __shfl(sincl, 63);
        const int sbase = sincl - scnt;
        for (int t0 = 0; t0 < stotal; t0 += 64) {
            // lane l takes the (t0 + l)-th word with survivors: find its owner lane (the one whose [sbase, sbase + scnt) holds it)
            const int want = t0 + lane;
            int owner = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) { const int cand_l = owner + step; const int b = __shfl(sbase, cand_l & 63); if (cand_l < 64 && b <= want) owner = cand_l; }
            const uint32_t osw = __shfl(sword, owner);
            const int obase = __shfl(sbase, owner);
            uint32_t word = 0u;
            int64_t widx = 0;
            if (want < stotal) {
                // the (want - obase)-th set bit of the owner's summary word
                uint32_t bits = osw;
                for (int q = want - obase; q > 0; --q) bits &= bits - 1;
                widx = (s0 + owner) * 32 + (__ffs(bits) - 1);
                word = mw[widx];
            }
            // survivors of these (up to 64) words behind the pending ones, ascending
            const int cnt = __popc(word);
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(incl, off); if (lane >= off) incl += up; }
            const int total = __shfl(incl, 63);
            // (64 words can hold 2,048 survivors: the list takes them in two halves of 32 words)
            for (int half = 0; half < 2; ++half) {
                const int lo_l = 32 * half;
                const int before = half ? __shfl(incl, 31) : 0;
                const int htotal = half ? total - before : __shfl(incl, 31);
                if (htotal == 0) continue;
                if (lane >= lo_l && lane < lo_l + 32) {
                    int pos = fill + incl - cnt - before;
                    const int32_t item_base = (int32_t)(a.item0 + widx * kScanTile);
                    uint32_t bits = word;
                    while (bits) { const int b = __ffs(bits) - 1; bits &= bits - 1; list[pos++] = item_base + b; }
                }
                fill += htotal;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                int b0 = 0;
                for (; fill - b0 >= 64; b0 += 64) batch(b0, 64);
                if (b0) {                                              // the rest (< 64) moves to the front
                    const int rest = fill - b0;
                    const int32_t keep = lane < rest ? list[b0 + lane] : 0;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    if (lane < rest) list[lane] = keep;
                    fill = rest;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    if (fill) batch(0, fill);
    if (lane < N) { a.scores[upos * N + lane] = sa; a.ids[upos * N + lane] = si; }
    if (lane == 0) { unsigned long long *wk = work_slot(a.work, (unsigned)upos); atomicAdd(wk + 1, events); atomicAdd(wk + 2, exact); }
}

}  // namespace yue
