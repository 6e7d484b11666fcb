// BPR training kernels for gfx950 (CDNA4).  Wave = 64 lanes.
//
// Layout of one triplet (u, i, j) inside a wave: lanes 0-31 own the positive row Q[i],
// lanes 32-63 the negative row Q[j]; element 32*r + (lane & 31) sits in register r of its
// half, so every load / atomic wave-instruction touches two contiguous 128-byte segments
// (one per row) -- the shape that runs at the full float-atomic rate on MI355X.  P[u] is held
// by both halves.  The k-length dots are reduced with a 32-lane butterfly per half, which is
// the summation order oracle/bpr_oracle.c:dot32 restates.
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
    uint32_t *dirtyP, *dirtyQ;
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

// One wave walks `tpw` consecutive events [base, base+cnt).
//   SAMPLE: draw j in-kernel (fused sampler) and record it in ev_j; else read ev_j.
//   DIRECT: the launch is conflict-free (a dependency level): write the updated rows in place.
//   else  : S-round -- add (new - old) into dP/dQ, factors stay at the round-start snapshot.
template <int KR, bool SAMPLE, bool DIRECT>
__global__ void __launch_bounds__(256) k_bpr_update(TrainArgs a, int64_t e_begin, int64_t e_end, int tpw) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t base = e_begin + wave * tpw;
    if (base >= e_end) return;
    const int cnt = (int)((e_end - base) < (int64_t)tpw ? (e_end - base) : (int64_t)tpw);

    int u = -1, i = 0, j = -1;
    if (lane < cnt) {
        const int64_t e = base + lane;
        u = a.ev_u[e];
        i = a.ev_i[e];
        if (SAMPLE) { j = sample_negative(a, u, e); a.ev_j[e] = j; }
        else j = a.ev_j[e];
    }

    const int hl = lane & 31;
    const bool hi = lane >= 32;
    const int k = a.k;
    float p[KR], dp[KR];
    int cur_u = -1;
    double nll = 0.0;

    for (int t = 0; t < cnt; ++t) {
        const int tu = __builtin_amdgcn_readlane(u, t);
        const int ti = __builtin_amdgcn_readlane(i, t);
        const int tj = __builtin_amdgcn_readlane(j, t);
        if (tj < 0) continue;                       // sampler gave up: triplet skipped (oracle does the same)
        if (DIRECT || tu != cur_u) {
            if (!DIRECT && cur_u >= 0 && !hi) {
#pragma unroll
                for (int r = 0; r < KR; ++r) { const int e = 32 * r + hl; if (e < k) atomicAdd(a.dP + (int64_t)cur_u * k + e, dp[r]); }
                if (lane == 0) a.dirtyP[cur_u] = 1u;
            }
#pragma unroll
            for (int r = 0; r < KR; ++r) { const int e = 32 * r + hl; p[r] = e < k ? a.P[(int64_t)tu * k + e] : 0.0f; dp[r] = 0.0f; }
            cur_u = tu;
        }
        const int64_t row = hi ? tj : ti;
        float q[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) { const int e = 32 * r + hl; q[r] = e < k ? a.Q[row * k + e] : 0.0f; }

        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < KR; ++r) { const float pr = p[r] * q[r]; acc = acc + pr; }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) acc = acc + __shfl_xor(acc, off);
        const float di = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 0));
        const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 32));
        const float x = di - dj;                                   // BPR.py:50, fp32 margin
        const double s = 1.0 / (1.0 + exp(-(double)x));            // qmath.py:115-116
        const float c = (float)(a.lr * (1.0 - s));
        nll += -log(s);                                            // BPR.py:58

#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = 32 * r + hl;
            const float other = __shfl_xor(q[r], 32);
            const float qi = hi ? other : q[r];
            const float qj = hi ? q[r] : other;
            const float d = qi - qj;
            const float td = c * d;
            const float p1 = p[r] + td;                            // :51
            const float tq = c * p1;
            const float q1 = hi ? (q[r] - tq) : (q[r] + tq);       // :53 / :52 (updated P[u])
            const float rq = a.ri * q1;
            const float q2 = q1 - rq;                              // :56 / :57
            const float rp = a.ru * p1;
            const float p2 = p1 - rp;                              // :55
            if (e < k) {
                if (DIRECT) {
                    a.Q[row * k + e] = q2;
                    if (!hi) a.P[(int64_t)tu * k + e] = p2;
                } else {
                    atomicAdd(a.dQ + row * k + e, q2 - q[r]);
                    dp[r] += p2 - p[r];
                }
            }
        }
        if (!DIRECT && hl == 0) a.dirtyQ[row] = 1u;
    }
    if (!DIRECT && cur_u >= 0 && !hi) {
#pragma unroll
        for (int r = 0; r < KR; ++r) { const int e = 32 * r + hl; if (e < k) atomicAdd(a.dP + (int64_t)cur_u * k + e, dp[r]); }
        if (lane == 0) a.dirtyP[cur_u] = 1u;
    }
    if (lane == 0 && nll != 0.0) atomicAdd(a.nll_slots + (wave & (kNllSlots - 1)), nll);
}

// Round end: every row touched in [e_begin, e_end) gets its summed difference added once.
// A row is claimed by whichever lane swaps its dirty flag back to 0 first.
__device__ __forceinline__ void apply_row(float *X, float *dX, int64_t row, int k, int lane) {
    for (int e = lane; e < k; e += 64) {
        const int64_t o = row * k + e;
        X[o] += dX[o];
        dX[o] = 0.0f;
    }
}

__global__ void __launch_bounds__(256) k_apply_round(TrainArgs a, int64_t e_begin, int64_t e_end, int apply_p) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t base = e_begin + wave * 32;
    if (base >= e_end) return;
    const int64_t e = base + (lane & 31);
    const bool valid = e < e_end;
    int32_t row = -1;
    if (valid) row = lane < 32 ? a.ev_i[e] : a.ev_j[e];
    uint32_t won = 0;
    if (row >= 0) won = atomicExch(a.dirtyQ + row, 0u);
    unsigned long long mask = __ballot(won != 0u);
    while (mask) {
        const int b = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        apply_row(a.Q, a.dQ, __builtin_amdgcn_readlane(row, b), a.k, lane);
    }
    if (apply_p) {
        // consecutive events mostly share the user: only the first lane of a run tries the claim
        int32_t ur = (valid && lane < 32) ? a.ev_u[e] : -1;
        const int32_t prev = __shfl_up(ur, 1);
        if (lane != 0 && prev == ur) ur = -1;
        won = 0;
        if (ur >= 0) won = atomicExch(a.dirtyP + ur, 0u);
        mask = __ballot(won != 0u);
        while (mask) {
            const int b = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            apply_row(a.P, a.dP, __builtin_amdgcn_readlane(ur, b), a.k, lane);
        }
    }
}

// Multi-GPU: after the all-reduce of dP[first .. first+count) the same range is applied everywhere.
__global__ void __launch_bounds__(256) k_apply_range(float *X, float *dX, uint32_t *dirty, int64_t first_elem, int64_t count, int k) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
        const int64_t o = first_elem + t;
        X[o] += dX[o];
        dX[o] = 0.0f;
        if (t % k == 0) dirty[o / k] = 0u;
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

__global__ void __launch_bounds__(256) k_sum_slots(const double *slots, int nslots, double *out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double s = 0.0;
        for (int t = 0; t < nslots; ++t) s += slots[t];
        *out = s;
    }
}

}  // namespace yue
