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
    int32_t *flags;        // [0] some user had < N candidates, [1] state-machine events, [2] exact re-scores (bf16 path)
    const float *tile_norm_max; // max ||Q[i]||_2 over each tile of 32 items (bf16 pre-filter margin); unused by the f32 kernel
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

__device__ __forceinline__ void scan_finish(const ScanState &S, int N, int32_t *flags) {
    if (S.cnt < N) {
        atomicOr(flags, 1);
        for (int q = 0; q < N; ++q) { S.g_sc[q] = -INFINITY; S.g_id[q] = -1; }
    } else {
        for (int q = 0; q < N; ++q) { S.g_sc[q] = S.st_a[q]; S.g_id[q] = S.st_id[q]; }
    }
    atomicAdd(flags + 1, S.events);
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

    if (h == 0 && uvalid) scan_finish(S, N, a.flags);
}

// ------------------------------------------------------------------------------------------
// bf16 pre-filter in front of the exact path (k = 16*K16 <= 128): k_topn_scan_mx.
//
// |bf16 score - exact score| <= 2^-8 * sum|p_e q_e| <= 2^-8 ||P_u|| ||Q_i||, so a pair can only matter to the
// state machine if  bf16 score + 2^-7 ||P_u|| ||Q_i|| > threshold_u  (margin doubled for slack).  Those few
// survivors are re-scored exactly -- the same k-ascending fp32 fma chain as k_topn_scan / the oracle -- and only
// the exact score enters the state machine: results are identical to the f32 kernel.
//
// Shape (one wave per SIMD, the large register file):
//   * a wave owns U = 2 blocks of 32 users: lane (r, h) is THE lane of user r of block h -- it keeps that user's
//     whole fp32 row in registers (the exact chain needs no memory for P), runs that user's selection state machine
//     and its mask cursor; the bf16 rows of both blocks are the B operands of v_mfma_f32_32x32x16_bf16 and stay in
//     registers for the whole scan.  A workgroup = 4 waves = 256 users: Q is streamed m / 256 times;
//   * item tiles of 32 are staged ONCE per workgroup into LDS in two forms: fp32 (for the exact chains) and bf16
//     (the A operands: one ds_read_b128 per k-step per lane, used by both blocks); tiles travel global ->
//     registers one tile ahead -> LDS, one barrier per tile;
//   * ITEMS are the rows of the MFMA tile and USERS its columns, so all 32 scores of user r of a block sit on lanes
//     r and r+32: the pre-filter is 16 compares against one per-lane value, the survivor mask is assembled in-lane
//     and completed with one exchange between the two halves of the wave;
//   * survivors: in every phase each lane takes ITS user's next surviving item through the full k-step chain (Q row
//     from the fp32 tile), so both halves of the wave work on different users; exact scores feed the reference's
//     state machine (N slots in LDS) in ascending item order.  Thresholds only rise: a threshold that is one tile
//     old only lets more pairs through.
// U = 1 (N > 45: the N slots of 256 users no longer fit beside the tiles) keeps the upper half of the lanes idle
// in the chains.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__host__ __device__ inline size_t scan_mx_lds_bytes(int k, int U, int NS) {
    const size_t tiles = 2u * kScanTile * (size_t)(k + 4) * sizeof(float) + 2u * kScanTile * (size_t)(k + 8) * 2u;
    const size_t state = (size_t)kScanWaves * 32 * U * NS * (sizeof(float) + sizeof(int32_t));
    return tiles + state;
}

template <int K16, int U>
__global__ void __launch_bounds__(256, 1) k_topn_scan_mx(ScanArgs a, int NS) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int K = 16 * K16, LDF = K + 4, LDH = K + 8, UW = 32 * U;
    const int N = a.N;
    float *tileF = reinterpret_cast<float *>(lds_raw);                                   // [2][32][LDF] fp32
    __bf16 *tileH = reinterpret_cast<__bf16 *>(tileF + 2 * kScanTile * LDF);             // [2][32][LDH] bf16
    float *st_a_all = reinterpret_cast<float *>(tileH + 2 * kScanTile * LDH);            // [4][UW][NS]
    int32_t *st_id_all = reinterpret_cast<int32_t *>(st_a_all + kScanWaves * UW * NS);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // B fragments of both blocks: lane (r,h) supplies B[16*s + 8*h + j][user r of block b]
    bf16x8 bfr[U][K16];
#pragma unroll
    for (int b = 0; b < U; ++b) {
        const int64_t up = ((int64_t)blockIdx.x * kScanWaves + w) * UW + b * 32 + r;
        const float *prow = a.P + (int64_t)a.users[up < a.nu ? up : 0] * K;
#pragma unroll
        for (int s2 = 0; s2 < K16; ++s2) {
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(prow + 16 * s2 + 8 * h), v1 = *reinterpret_cast<const f32x4 *>(prow + 16 * s2 + 8 * h + 4);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { bfr[b][s2][jj] = (__bf16)v0[jj]; bfr[b][s2][jj + 4] = (__bf16)v1[jj]; }
        }
    }
    // my user: user r of block h (U = 1: lanes of the upper half have none)
    const bool has_user = h < U;
    const int64_t upos = ((int64_t)blockIdx.x * kScanWaves + w) * UW + (has_user ? h : 0) * 32 + r;
    const bool uvalid = has_user && upos < a.nu;
    const int32_t uid = a.users[uvalid ? upos : 0];
    float pf[K];                                                 // the exact fp32 row of my user
    float mu_mine;
    {
        const float *prow = a.P + (int64_t)uid * K;
        float ss = 0.0f;
#pragma unroll
        for (int e = 0; e < K; e += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(prow + e);
            pf[e] = v[0]; pf[e + 1] = v[1]; pf[e + 2] = v[2]; pf[e + 3] = v[3];
            ss = __builtin_fmaf(v[0], v[0], ss); ss = __builtin_fmaf(v[1], v[1], ss); ss = __builtin_fmaf(v[2], v[2], ss); ss = __builtin_fmaf(v[3], v[3], ss);
        }
        mu_mine = __builtin_sqrtf(ss) * (1.01f / 128.0f);       // 2^-7 ||P_u||, 1 % slack for the norm roundings
    }
    const float mu_other = __shfl_xor(mu_mine, 32);

    ScanState S;
    S.st_a = st_a_all + ((w * UW) + (has_user ? h : 0) * 32 + r) * NS; S.st_id = st_id_all + ((w * UW) + (has_user ? h : 0) * 32 + r) * NS;
    S.cnt = 0; S.thr = -INFINITY; S.events = 0;
    S.g_sc = a.out_scores + (uvalid ? upos : 0) * N;
    S.g_id = a.out_ids + (uvalid ? upos : 0) * N;
    int rescored = 0;
    float thr_other = -INFINITY;                                 // threshold of the user on the other half's lane (same r)
    int64_t mcur = 0, mend = 0;
    int32_t mnext = 0x7fffffff;
    if (uvalid) {
        const int64_t mrow = a.mask_by_user ? (int64_t)uid : upos;
        mcur = a.mask_ptr[mrow]; mend = a.mask_ptr[mrow + 1];
        if (mcur < mend) mnext = a.mask_idx[mcur];
    }

    // item tiles: global -> registers (one tile ahead) -> LDS (fp32 and bf16), float4 granularity
    constexpr int PF4 = (kScanTile * K / 4 + 255) / 256;
    int offF[PF4], offH[PF4];
#pragma unroll
    for (int q = 0; q < PF4; ++q) { const int el = (tid + 256 * q) * 4; const int row = el / K; offF[q] = row * LDF + (el - row * K); offH[q] = row * LDH + (el - row * K); }
    f32x4 pre[PF4];
    float nu_next = 0.0f;
    auto fetch = [&](int64_t it0) {
        const int64_t limit = (a.n - it0) * K;
        const float *src = a.Q + it0 * K;
#pragma unroll
        for (int q = 0; q < PF4; ++q) {
            const int el = (tid + 256 * q) * 4;
            pre[q] = (el < kScanTile * K && el < limit) ? *reinterpret_cast<const f32x4 *>(src + el) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        nu_next = a.tile_norm_max[it0 / kScanTile];             // largest ||Q_i|| of the tile (wave-uniform)
    };
    auto commit = [&](int buf) {
        float *dF = tileF + buf * kScanTile * LDF;
        __bf16 *dH = tileH + buf * kScanTile * LDH;
#pragma unroll
        for (int q = 0; q < PF4; ++q)
            if ((tid + 256 * q) * 4 < kScanTile * K) {
                *reinterpret_cast<f32x4 *>(dF + offF[q]) = pre[q];
                bf16x4 hv;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) hv[jj] = (__bf16)pre[q][jj];
                *reinterpret_cast<bf16x4 *>(dH + offH[q]) = hv;
            }
    };

    const int64_t ntiles = (a.n + kScanTile - 1) / kScanTile;
    fetch(0);
    commit(0);
    float nu = nu_next;
    __syncthreads();

    for (int64_t t = 0; t < ntiles; ++t) {
        const int cur = (int)(t & 1);
        const int64_t it0 = t * kScanTile;
        const float *tF = tileF + cur * kScanTile * LDF;
        const __bf16 *tH = tileH + cur * kScanTile * LDH;
        if (t + 1 < ntiles) fetch(it0 + kScanTile);

        // bf16 scores: lane (r,h) supplies A[item r][16*s + 8*h + j] and receives D[item (q&3)+8*(q>>2)+4*h][user r of block b]
        f32x16 acc[U];
#pragma unroll
        for (int b = 0; b < U; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[b][q] = 0.0f;
        const __bf16 *irow = tH + r * LDH + 8 * h;
#pragma unroll
        for (int s2 = 0; s2 < K16; ++s2) {
            const bf16x8 itf = *reinterpret_cast<const bf16x8 *>(irow + 16 * s2);
#pragma unroll
            for (int b = 0; b < U; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(itf, bfr[b][s2], acc[b], 0, 0, 0);
        }

        // pre-filter in registers: block b's user r is filtered with the threshold its own lane (r, b) keeps.
        // First only the largest of a lane's 16 scores per block is looked at: in most tiles nothing passes anywhere.
        float bar[U];
        bool some = false;
#pragma unroll
        for (int b = 0; b < U; ++b) {
            bar[b] = (b == h ? S.thr : thr_other) - (b == h ? mu_mine : mu_other) * nu;
            float mx = __builtin_fmaxf(acc[b][0], acc[b][1]);
#pragma unroll
            for (int q = 2; q < 16; q += 2) mx = __builtin_fmaxf(mx, __builtin_fmaxf(acc[b][q], acc[b][q + 1]));
            some = some || bar[b] < mx;
        }
        uint32_t mymask = 0u;
        if (__ballot(some) != 0ull) {
#pragma unroll
            for (int b = 0; b < U; ++b) {
                uint32_t pmask = 0u;
#pragma unroll
                for (int q = 0; q < 16; ++q) pmask |= (bar[b] < acc[b][q] ? 1u : 0u) << ((q & 3) + 8 * (q >> 2));
                pmask <<= 4 * h;                                   // rows of lane (r,1) are shifted by 4
                pmask |= (uint32_t)__shfl_xor((int)pmask, 32);     // both lanes now hold all 32 items of user (b, r)
                if (b == h) mymask = pmask;
            }
        }
        const int64_t left = a.n - it0;
        if (left < kScanTile) mymask &= (1u << (uint32_t)left) - 1u;
        uint32_t cand = 0u;
        if (uvalid) {
            uint32_t mb = 0u;
            while (mnext < it0 + kScanTile) {
                mb |= 1u << (uint32_t)(mnext - it0);
                ++mcur;
                mnext = mcur < mend ? a.mask_idx[mcur] : 0x7fffffff;
            }
            cand = mymask & ~mb;
        }
        // survivors: one phase = every lane takes its user's next TWO surviving items through the exact chain
        // (k-ascending fp32 fma chain == v_mfma_f32_32x32x2_f32 == oracle) -- two independent chains keep the fma
        // pipe busy where one would wait for itself -- then feeds them, in ascending item order, through the state
        // machine.  A slot without a survivor runs on row 0 and is dropped.
        while (__ballot(cand != 0u) != 0ull) {
            int c[2];
            bool v[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                v[s2] = cand != 0u;
                c[s2] = v[s2] ? __ffs(cand) - 1 : 0;
                cand &= cand - 1u;                               // 0 stays 0
            }
            const float *q0 = tF + c[0] * LDF, *q1 = tF + c[1] * LDF;
            float sc[2] = {0.0f, 0.0f};
            // 8 elements per step; the reads of step i+1 are issued before the 16 fmas of step i
            f32x4 x0[2][2], x1[2][2];
#pragma unroll
            for (int g = 0; g < 2; ++g) { x0[0][g] = *reinterpret_cast<const f32x4 *>(q0 + 4 * g); x1[0][g] = *reinterpret_cast<const f32x4 *>(q1 + 4 * g); }
#pragma unroll
            for (int st = 0; st < K / 8; ++st) {
                const int cb = st & 1, nb = cb ^ 1;
                if (st + 1 < K / 8) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) { x0[nb][g] = *reinterpret_cast<const f32x4 *>(q0 + 8 * (st + 1) + 4 * g); x1[nb][g] = *reinterpret_cast<const f32x4 *>(q1 + 8 * (st + 1) + 4 * g); }
                }
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        sc[0] = __builtin_fmaf(pf[8 * st + 4 * g + jj], x0[cb][g][jj], sc[0]);
                        sc[1] = __builtin_fmaf(pf[8 * st + 4 * g + jj], x1[cb][g][jj], sc[1]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                if (v[s2]) {
                    ++rescored;
                    if (!(S.cnt == N && !(S.thr < sc[s2]))) scan_push(S, N, sc[s2], (int32_t)(it0 + c[s2]), a.true_topn);
                }
        }
        thr_other = __shfl_xor(S.thr, 32);
        if (t + 1 < ntiles) commit(cur ^ 1);
        nu = nu_next;
        __syncthreads();
    }

    if (uvalid) { scan_finish(S, N, a.flags); atomicAdd(a.flags + 2, rescored); }
}

template <int K2>
inline void launch_scan_f32(const ScanArgs &a, hipStream_t stream, dim3 grid, size_t lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_topn_scan<K2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_topn_scan<K2>, grid, dim3(256), lds, stream, a);
}
template <int K16, int U>
inline void launch_scan_mx(const ScanArgs &a, hipStream_t stream, int NS) {
    const size_t lds = scan_mx_lds_bytes(16 * K16, U, NS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_topn_scan_mx<K16, U>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const dim3 grid((unsigned)((a.nu + kScanWaves * 32 * U - 1) / (kScanWaves * 32 * U)));
    hipLaunchKernelGGL((k_topn_scan_mx<K16, U>), grid, dim3(256), lds, stream, a, NS);
}
template <int K16>
inline void launch_scan_mx_u(const ScanArgs &a, hipStream_t stream) {
    // 256 users per workgroup while their N slots fit beside the tiles in the 160 KB of LDS (N <= 45), else 128
    if (a.N <= 45) launch_scan_mx<K16, 2>(a, stream, scan_ns(a.N));
    else launch_scan_mx<K16, 1>(a, stream, scan_ns(a.N));
}

// force_f32 != 0: always the exact-f32-MFMA kernel.  Otherwise k in {16,32,64,128} takes the bf16
// pre-filter kernel (identical results), anything else the f32 kernel.  Returns 1 if bf16 was used.
inline int launch_scan(const ScanArgs &a, hipStream_t stream, int force_f32) {
    if (a.k > 256 || a.N > 100) return -1;
    if (!force_f32 && (a.k == 16 || a.k == 32 || a.k == 64 || a.k == 128)) {
        switch (a.k) {
            case 16: launch_scan_mx_u<1>(a, stream); break;
            case 32: launch_scan_mx_u<2>(a, stream); break;
            case 64: launch_scan_mx_u<4>(a, stream); break;
            default: launch_scan_mx_u<8>(a, stream); break;
        }
        return 1;
    }
    const dim3 grid((unsigned)((a.nu + kScanWaves * 32 - 1) / (kScanWaves * 32)));
    const int kp = a.k + (a.k & 1);
    const int K2 = kp / 2 <= 8 ? 8 : kp / 2 <= 16 ? 16 : kp / 2 <= 32 ? 32 : kp / 2 <= 64 ? 64 : 128;
    const size_t lds = scan_lds_bytes(K2, a.N);
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
