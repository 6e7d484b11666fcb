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
    int32_t *flags;        // [0] some user had < N candidates, [1] state-machine events
};

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
    int cnt = 0;
    float thr = -INFINITY;
    int events = 0;
    int64_t mcur = 0, mend = 0;
    int32_t mnext = 0x7fffffff;
    if (h == 0 && uvalid) {
        const int64_t mrow = a.mask_by_user ? (int64_t)uid : upos;
        mcur = a.mask_ptr[mrow]; mend = a.mask_ptr[mrow + 1];
        if (mcur < mend) mnext = a.mask_idx[mcur];
    }
    if (h == 0) thr_w[r] = -INFINITY;
    float *g_sc = a.out_scores + (uvalid ? upos : 0) * N;     // also the unsorted-seed scratch
    int32_t *g_id = a.out_ids + (uvalid ? upos : 0) * N;

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
                if (cnt == N && !(thr < s)) continue;
                const int32_t item = (int32_t)(it0 + c);
                ++events;
                if (cnt < N) {
                    // seed phase (IterativeRecommender.py:107-112): keep id order in global scratch
                    g_sc[cnt] = s; g_id[cnt] = item;
                    // stable descending insertion (:114): later ids go after equal scores
                    int p = cnt;
                    while (p > 0 && st_a[p - 1] < s) { st_a[p] = st_a[p - 1]; st_id[p] = st_id[p - 1]; --p; }
                    st_a[p] = s; st_id[p] = item;
                    ++cnt;
                    if (cnt == N) {
                        // the scan (:121-144) starts over from the first candidate
                        for (int q = 0; q < N; ++q) {
                            const float sq = g_sc[q];
                            if (st_a[N - 1] < sq) {
                                int p2 = 0;
                                while (st_a[p2] >= sq) ++p2;
                                st_a[p2] = sq; st_id[p2] = g_id[q];
                            }
                        }
                        thr = st_a[N - 1];
                    }
                } else {
                    int p = 0;
                    while (st_a[p] >= s) ++p;          // first slot strictly below s
                    st_a[p] = s; st_id[p] = item;      // overwrite, no shift (:142-144)
                    thr = st_a[N - 1];
                }
            }
            thr_w[r] = thr;
        }
        __syncthreads();
    }

    if (h == 0 && uvalid) {
        if (cnt < N) {
            atomicOr(a.flags, 1);
            for (int q = 0; q < N; ++q) { g_sc[q] = -INFINITY; g_id[q] = -1; }
        } else {
            for (int q = 0; q < N; ++q) { g_sc[q] = st_a[q]; g_id[q] = st_id[q]; }
        }
        atomicAdd(a.flags + 1, events);
    }
}

inline int launch_scan(const ScanArgs &a, hipStream_t stream) {
    const int kp = a.k + (a.k & 1);
    const int K2 = kp / 2 <= 8 ? 8 : kp / 2 <= 16 ? 16 : kp / 2 <= 32 ? 32 : 64;
    if (kp / 2 > 64 || a.N > 100) return -1;
    const size_t lds = scan_lds_bytes(K2, a.N);
    const dim3 grid((unsigned)((a.nu + kScanWaves * 32 - 1) / (kScanWaves * 32))), block(256);
    switch (K2) {
        case 8:
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_topn_scan<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_topn_scan<8>, grid, block, lds, stream, a); break;
        case 16:
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_topn_scan<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_topn_scan<16>, grid, block, lds, stream, a); break;
        case 32:
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_topn_scan<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_topn_scan<32>, grid, block, lds, stream, a); break;
        default:
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_topn_scan<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_topn_scan<64>, grid, block, lds, stream, a); break;
    }
    return 0;
}

}  // namespace yue
