// FISM (recommender/cf/FISM.py of the reference) on gfx950 -- the parity path.
//
// The reference's loop is strictly sequential: every draw of every event of every user rewrites
// Q[i], Q[j], Bi[i], Bi[j] in place, the next draw reads them, and the user's item-history rows P[.]
// are rewritten when the user is done (FISM.py:38-69).  Users share items, so the dependency chain runs
// through the whole epoch: k_fism_epoch is ONE wave that walks the epoch in the reference's order
// (lane l holds elements 64r + l of a row).  It exists to reproduce the reference, not to be fast; the
// throughput version is k_fism_round below (rounds of users, one wave per user).
//
// Types as the reference leaves them: P float64 [n,k] (item-history factors), Q float32 [n,k],
// Bi float64 [n].  Arithmetic follows the NumPy expressions term by term (python-float coefficients,
// float32 products where NumPy forms them in float32, float64 sums rounded into the float32 rows);
// the file is compiled with -ffp-contract=off.  The k-length dots are 64 strided partials + a
// butterfly in double (the reference's BLAS ddot order is not pinned: results agree to ~1e-15 rel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace yue {

struct FismArgs {
    double *P;                   // [n,k]
    float *Q;                    // [n,k]
    double *Bi;                  // [n]
    int64_t n;
    int k;
    const int64_t *user_ptr;     // [m+1] events of user u are ev_i[user_ptr[u] : user_ptr[u+1]]
    int64_t m;
    const int32_t *ev_i;
    const int32_t *negs;         // accepted negatives in processing order, rho per event of users with > 1 event
    int rho;
    const double *coef;          // [m] pow(nu - 1, -alpha), formed on the host as Python forms it (FISM.py:42)
    double lr, regI, regB;
    double *x_rows;              // scratch [max events of one user][k]
    double *out;                 // out[0] = sum of 0.5 * error^2
};

// Sum over the 64 lanes, the butterfly with partners lane^1, ^2, ^4, ^8, ^16, ^32 (the order round 2's __shfl_xor loop had:
// same operands per step, so the same double on every lane), without its twelve trips through the LDS crossbar: DPP moves of
// the two halves for the first four steps (quad_perm, quad_perm, row_half_mirror, row_mirror: the value is already constant over
// the smaller groups), v_permlane16_swap of the value with itself for ^16, lanes 0 and 32 for ^32.  The two sums of a draw sit
// in the dependency chain of every draw of every user.
template <int CTRL>
__device__ __forceinline__ double fism_dpp_mov(double v) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v = v + fism_dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v = v + fism_dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v = v + fism_dpp_mov<0x141>(v);      // row_half_mirror
    v = v + fism_dpp_mov<0x140>(v);      // row_mirror
    {   // rows 0, 0, 2, 2 in one result and 1, 1, 3, 3 in the other (see wave_sum of bpr_device.hpp for the s_nops)
        const uint64_t b = __builtin_bit_cast(uint64_t, v);
        uint32_t lo0 = (uint32_t)b, lo1 = lo0, hi0 = (uint32_t)(b >> 32), hi1 = hi0;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1" : "+v"(lo0), "+v"(lo1), "+v"(hi0), "+v"(hi1));
        v = __builtin_bit_cast(double, ((uint64_t)hi0 << 32) | lo0) + __builtin_bit_cast(double, ((uint64_t)hi1 << 32) | lo1);
    }
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint64_t t0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 0) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 0);
    const uint64_t t1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 32) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 32);
    return __builtin_bit_cast(double, t0) + __builtin_bit_cast(double, t1);
}

// Loads / stores that go to L2 (sc1): the wave re-reads rows it has just rewritten, the L1 must not
// serve them.
__device__ __forceinline__ double ldg_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ldg_f32(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stg_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stg_f32(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int KR>
__global__ void __launch_bounds__(64) k_fism_epoch(FismArgs a) {
    const int lane = threadIdx.x;
    const int k = a.k;
    const float regI32 = (float)a.regI;                  // `regI * Q[i]`: python float times a float32 row stays float32
    double half_sq = 0.0;
    int64_t cursor = 0;
    for (int64_t u = 0; u < a.m; ++u) {
        const int64_t e0 = a.user_ptr[u], e1 = a.user_ptr[u + 1];
        const int64_t nu = e1 - e0;
        if (nu <= 1) continue;                           // FISM.py:40-41 (a user without events has nothing to do either)
        const double coef = a.coef[u];
        double hist[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) hist[r] = 0.0;
        for (int64_t e = e0; e < e1; ++e) {              // :44-46
            const int64_t it = a.ev_i[e];
#pragma unroll
            for (int r = 0; r < KR; ++r) { const int el = 64 * r + lane; if (el < k) hist[r] = hist[r] + ldg_f64(a.P + it * k + el); }
        }
        for (int64_t e = e0; e < e1; ++e) {
            const int64_t i = a.ev_i[e];
            double x[KR];
#pragma unroll
            for (int r = 0; r < KR; ++r) x[r] = 0.0;
            for (int c = 0; c < a.rho; ++c) {
                const int64_t j = a.negs[cursor++];
                double di[KR], dj[KR];
                float qi[KR], qj[KR];
                double ai = 0.0, aj = 0.0;
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const int el = 64 * r + lane;
                    di[r] = dj[r] = 0.0; qi[r] = qj[r] = 0.0f;
                    if (el < k) {
                        di[r] = hist[r] - ldg_f64(a.P + i * k + el);
                        dj[r] = hist[r] - ldg_f64(a.P + j * k + el);
                        qi[r] = ldg_f32(a.Q + i * k + el);
                        qj[r] = ldg_f32(a.Q + j * k + el);
                    }
                    const double m1 = di[r] * (double)qi[r]; ai = ai + m1;
                    const double m2 = dj[r] * (double)qj[r]; aj = aj + m2;
                }
                const double bi = ldg_f64(a.Bi + i), bj = ldg_f64(a.Bi + j);
                const double r_pos = coef * wave_sum_f64(ai) + bi;          // :54
                const double r_neg = coef * wave_sum_f64(aj) + bj;          // :55
                const double err = 1.0 - (r_pos - r_neg);
                half_sq = half_sq + 0.5 * (err * err);                      // :58
                if (lane == 0) {
                    stg_f64(a.Bi + i, bi + a.lr * (err - a.regB * bi));     // :59
                    stg_f64(a.Bi + j, bj - a.lr * (err + a.regB * bj));     // :60
                }
                const double ec = err * coef;
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const int el = 64 * r + lane;
                    const float ri = regI32 * qi[r], rj = regI32 * qj[r];
                    const float ni = (float)((double)qi[r] + a.lr * (ec * di[r] - (double)ri));     // :61
                    const float nj = (float)((double)qj[r] - a.lr * (ec * dj[r] + (double)rj));     // :62
                    const float dq = ni - nj;
                    x[r] = x[r] + err * (double)dq;                                                  // :63
                    if (el < k) { stg_f32(a.Q + i * k + el, ni); stg_f32(a.Q + j * k + el, nj); }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int r = 0; r < KR; ++r) { const int el = 64 * r + lane; if (el < k) stg_f64(a.x_rows + (e - e0) * k + el, x[r]); }
        }
        const double pc = 1.0 / (double)a.rho * coef;                       // :68, left to right
        for (int64_t e = e0; e < e1; ++e) {
            const int64_t it = a.ev_i[e];
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const int el = 64 * r + lane;
                if (el < k) {
                    const double p = ldg_f64(a.P + it * k + el);
                    stg_f64(a.P + it * k + el, p + a.lr * (pc * ldg_f64(a.x_rows + (e - e0) * k + el) - a.regI * p));
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (lane == 0) a.out[0] = half_sq;
}

// ------------------------------------------------------------------------------------------
// Throughput path: rounds of users (oracle/numpy_fism.py: fism_rounds).
// Every user of a round runs the reference's whole per-user loop -- draws and item-history updates in the reference's
// order, on the model as it was when the round started plus the user's OWN changes so far -- in a private working
// copy of the rows it touches (its events' items and its negatives; the host hands over, per user, the list of those
// items and, per event / draw, the position of its item in that list).  One wave per user, all users of the round at
// once.  At the end the wave adds (working row - round-start row) into the difference buffers with atomics;
// k_fism_apply adds the sums to the model once per round.  One user per round reproduces k_fism_epoch.
// ------------------------------------------------------------------------------------------
struct FismRoundArgs {
    int64_t u_begin, u_end;          // users of this round
    const int64_t *uq_ptr;           // [m+1] touched items of user u: uq_items[uq_ptr[u] : uq_ptr[u+1]] (ascending, unique)
    const int32_t *uq_items;
    const int32_t *loc_i;            // [E] position of the event's item in its user's list
    const int32_t *loc_j;            // [n_negs] position of the draw's negative in its user's list
    const int64_t *neg_ptr;          // [m+1] first draw of user u in negs / loc_j
    int64_t w_base;                  // first working row of the round (= uq_ptr[u_begin]); x_base likewise for events
    float *wq;                       // working copies [rows of the round][k]
    double *wp, *wb;                 // [rows][k], [rows]
    float *dQ;                       // difference buffers [n,k], [n,k], [n]
    double *dP, *dB;
};

template <int KR>
__global__ void __launch_bounds__(256) k_fism_round(FismArgs a, FismRoundArgs ra) {
    const int lane = threadIdx.x & 63;
    const int64_t u = ra.u_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= ra.u_end) return;
    const int k = a.k;
    const int64_t e0 = a.user_ptr[u], e1 = a.user_ptr[u + 1];
    if (e1 - e0 <= 1) return;                            // FISM.py:40-41
    const float regI32 = (float)a.regI;
    const int64_t q0 = ra.uq_ptr[u], q1 = ra.uq_ptr[u + 1];
    const int64_t wrow0 = q0 - ra.w_base;                // my first working row
    float *wq = ra.wq + wrow0 * k;
    double *wp = ra.wp + wrow0 * k, *wb = ra.wb + wrow0;
    double *xr = a.x_rows + (e0 - a.user_ptr[ra.u_begin]) * k;
    // copy in the rows this user touches
    for (int64_t s = 0; s < q1 - q0; ++s) {
        const int64_t it = ra.uq_items[q0 + s];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int el = 64 * r + lane;
            if (el < k) { stg_f32(wq + s * k + el, a.Q[it * k + el]); stg_f64(wp + s * k + el, a.P[it * k + el]); }
        }
        if (lane == 0) stg_f64(wb + s, a.Bi[it]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const double coef = a.coef[u];
    double hist[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) hist[r] = 0.0;
    for (int64_t e = e0; e < e1; ++e) {                  // :44-46
        const int64_t s = ra.loc_i[e];
#pragma unroll
        for (int r = 0; r < KR; ++r) { const int el = 64 * r + lane; if (el < k) hist[r] = hist[r] + ldg_f64(wp + s * k + el); }
    }
    double half_sq = 0.0;
    int64_t cursor = ra.neg_ptr[u];
    for (int64_t e = e0; e < e1; ++e) {
        const int64_t i = ra.loc_i[e];
        double x[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) x[r] = 0.0;
        for (int c = 0; c < a.rho; ++c) {
            const int64_t j = ra.loc_j[cursor++];
            double di[KR], dj[KR];
            float qi[KR], qj[KR];
            double ai = 0.0, aj = 0.0;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const int el = 64 * r + lane;
                di[r] = dj[r] = 0.0; qi[r] = qj[r] = 0.0f;
                if (el < k) {
                    di[r] = hist[r] - ldg_f64(wp + i * k + el);
                    dj[r] = hist[r] - ldg_f64(wp + j * k + el);
                    qi[r] = ldg_f32(wq + i * k + el);
                    qj[r] = ldg_f32(wq + j * k + el);
                }
                const double m1 = di[r] * (double)qi[r]; ai = ai + m1;
                const double m2 = dj[r] * (double)qj[r]; aj = aj + m2;
            }
            const double bi = ldg_f64(wb + i), bj = ldg_f64(wb + j);
            const double r_pos = coef * wave_sum_f64(ai) + bi;          // :54
            const double r_neg = coef * wave_sum_f64(aj) + bj;          // :55
            const double err = 1.0 - (r_pos - r_neg);
            half_sq = half_sq + 0.5 * (err * err);                      // :58
            if (lane == 0) {
                stg_f64(wb + i, bi + a.lr * (err - a.regB * bi));       // :59
                stg_f64(wb + j, bj - a.lr * (err + a.regB * bj));       // :60
            }
            const double ec = err * coef;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const int el = 64 * r + lane;
                const float ri = regI32 * qi[r], rj = regI32 * qj[r];
                const float ni = (float)((double)qi[r] + a.lr * (ec * di[r] - (double)ri));     // :61
                const float nj = (float)((double)qj[r] - a.lr * (ec * dj[r] + (double)rj));     // :62
                const float dq = ni - nj;
                x[r] = x[r] + err * (double)dq;                                                  // :63
                if (el < k) { stg_f32(wq + i * k + el, ni); stg_f32(wq + j * k + el, nj); }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int r = 0; r < KR; ++r) { const int el = 64 * r + lane; if (el < k) stg_f64(xr + (e - e0) * k + el, x[r]); }
    }
    const double pc = 1.0 / (double)a.rho * coef;                       // :68, left to right
    for (int64_t e = e0; e < e1; ++e) {
        const int64_t s = ra.loc_i[e];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int el = 64 * r + lane;
            if (el < k) {
                const double p = ldg_f64(wp + s * k + el);
                stg_f64(wp + s * k + el, p + a.lr * (pc * ldg_f64(xr + (e - e0) * k + el) - a.regI * p));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // what this user changed: working row - round-start row, added to the round's difference buffers
    for (int64_t s = 0; s < q1 - q0; ++s) {
        const int64_t it = ra.uq_items[q0 + s];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int el = 64 * r + lane;
            if (el < k) {
                const float dq = ldg_f32(wq + s * k + el) - a.Q[it * k + el];
                if (dq != 0.0f) atomicAdd(ra.dQ + it * k + el, dq);
                const double dp = ldg_f64(wp + s * k + el) - a.P[it * k + el];
                if (dp != 0.0) atomicAdd(ra.dP + it * k + el, dp);
            }
        }
        if (lane == 0) { const double db = ldg_f64(wb + s) - a.Bi[it]; if (db != 0.0) atomicAdd(ra.dB + it, db); }
    }
    if (lane == 0) atomicAdd(a.out, half_sq);
}

// The same round with the user's working rows in LDS (round 3): a wave finds the items its user touches itself (its events'
// items and its negatives, at most 64 in all: one per lane; equal items share a row: first occurrence = the row's owner), copies
// their rows into LDS, walks the reference's per-user loop there -- an LDS round trip per draw instead of an L2 one, and no host
// work per user at all (k_fism_round takes sorted item lists and positions the host prepares) -- and adds (working row - round-start
// row) into the difference buffers.  Same results as k_fism_round up to the order of the float atomics.
struct FismLdsArgs {
    int64_t u_begin, u_end;
    const int64_t *neg_ptr;          // [m+1] first draw of user u in negs
    float *dQ;
    double *dP, *dB;
    int rows_cap;                    // working rows a user may need (<= 64)
    // Rows only ONE user of the round touches (round 4): that user's wave stores its working row in place -- no round-start
    // row read again, no atomic adds, nothing for k_fism_apply to fold.  cnt_cur[item] = users of THIS round touching the item
    // (null: every row goes through the difference buffers, round 3's form); the waves of a round count the NEXT round's
    // users into cnt_next (block b: user next_begin + b) and clear their own entries of cnt_cur behind them -- the two
    // arrays swap per round; a wave that reads an entry another wave has already cleared sees 0 instead of >= 2: shared either way.
    const unsigned *cnt_cur_r;
    unsigned *cnt_cur, *cnt_next;
    int64_t next_begin, next_end;
#ifdef YUE_FISM_STAMPS
    unsigned long long *stamps;      // diagnostic build: [8] summed 100 MHz ticks per phase over all waves, [7] = waves
#endif
};
#ifdef YUE_FISM_STAMPS
#define FISM_STAMP(slot) do { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); if (lane == 0) atomicAdd(ra.stamps + (slot), now_ - t_prev_); t_prev_ = now_; } while (0)
#else
#define FISM_STAMP(slot) do { } while (0)
#endif

// the distinct items user u touches (its events' items and its negatives), each counted once into cnt
__device__ __forceinline__ void fism_count_rows(const FismArgs &a, const int64_t *neg_ptr, int64_t u, unsigned *cnt, int lane) {
    const int64_t e0 = a.user_ptr[u], e1 = a.user_ptr[u + 1];
    const int ne = (int)(e1 - e0);
    if (ne <= 1) return;
    const int c = ne + ne * a.rho;
    const int64_t n0 = neg_ptr[u];
    const int32_t x = lane < ne ? a.ev_i[e0 + lane] : lane < c ? a.negs[n0 + lane - ne] : -1;
    bool first = lane < c;
    for (int t = 0; t < c; ++t) { const int32_t xt = __builtin_amdgcn_readlane(x, t); if (x == xt && t < lane) first = false; }
    if (first) atomicAdd(cnt + x, 1u);
}
// before the first round of a call: its users' rows counted (later rounds are counted by the round in front of them)
__global__ void __launch_bounds__(64) k_fism_count_round(FismArgs a, const int64_t *neg_ptr, int64_t u_begin, int64_t u_end, unsigned *cnt) {
    const int64_t u = u_begin + blockIdx.x;
    if (u < u_end) fism_count_rows(a, neg_ptr, u, cnt, threadIdx.x);
}

// START: the round-start rows stay in LDS beside the working rows (twice the LDS: the host asks for it when it fits), so that the
// differences need no second read of the model.
template <int KR, bool START>
__global__ void __launch_bounds__(64) k_fism_round_lds(FismArgs a, FismLdsArgs ra) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fism_lds[];
    constexpr int kFismBatch = 64 / KR;                 // rows whose loads are in flight together (a batch costs one memory latency)
    const int lane = threadIdx.x;
#ifdef YUE_FISM_STAMPS
    unsigned long long t_prev_ = __builtin_amdgcn_s_memrealtime();
#endif
    // this wave's user, and the user of the NEXT round whose rows it counts: both users' pointers, then both users' items, are
    // asked for together (four dependent memory latencies in a row used to open every wave)
    const int64_t u = ra.u_begin + blockIdx.x, un = ra.next_begin + blockIdx.x;
    const bool has_me = u < ra.u_end, has_next = ra.cnt_next && un < ra.next_end;
    const int k = a.k;
    int64_t e0 = 0, e1 = 0, n0 = 0, f0 = 0, f1 = 0, g0 = 0;
    if (has_me) { e0 = a.user_ptr[u]; e1 = a.user_ptr[u + 1]; n0 = ra.neg_ptr[u]; }
    if (has_next) { f0 = a.user_ptr[un]; f1 = a.user_ptr[un + 1]; g0 = ra.neg_ptr[un]; }
    const int ne = (int)(e1 - e0), nf = (int)(f1 - f0);
    const int cnt = ne > 1 ? ne + ne * a.rho : 0;        // touches of this user (the host guarantees <= 64)
    const int cntn = nf > 1 ? nf + nf * a.rho : 0;
    const int32_t x = lane >= cnt ? -1 : lane < ne ? a.ev_i[e0 + lane] : a.negs[n0 + lane - ne];
    const int32_t xn = lane >= cntn ? -1 : lane < nf ? a.ev_i[f0 + lane] : a.negs[g0 + lane - nf];
    int first = lane;                                    // lowest lane holding the same item
    bool firstn = lane < cntn;
    for (int t = 0; t < (cnt > cntn ? cnt : cntn); ++t) {
        const int32_t xt = __builtin_amdgcn_readlane(x, t), xnt = __builtin_amdgcn_readlane(xn, t);
        if (lane < cnt && x == xt && t < first) first = t;
        if (xn == xnt && t < lane) firstn = false;
    }
    // (here, not beside the differences at the end: measured 57 against 63 us per round -- the end of the wave is where the
    // atomic units are busy)
    if (firstn) atomicAdd(ra.cnt_next + xn, 1u);
    if (!has_me || ne <= 1) return;                      // FISM.py:40-41
    double *wp = reinterpret_cast<double *>(fism_lds);                               // [rows_cap][k]
    double *wb = wp + (size_t)ra.rows_cap * k;                                       // [rows_cap]
    float *wq = reinterpret_cast<float *>(wb + ra.rows_cap);                         // [rows_cap][k]
    double *sp = reinterpret_cast<double *>(wq + (size_t)ra.rows_cap * k + (((size_t)ra.rows_cap * k) & 1));   // START: the same three, as the round started
    double *sb = sp + (size_t)ra.rows_cap * k;
    float *sq = reinterpret_cast<float *>(sb + ra.rows_cap);
    const float regI32 = (float)a.regI;
    const unsigned long long fmask = __ballot(lane < cnt && first == lane);
    const int urow = __popcll(fmask & ((1ull << first) - 1ull));                    // the item's working row
    const int nuniq = __popcll(fmask);
    // rows of mine nobody else in the round touches
    const unsigned long long emask = __ballot(ra.cnt_cur_r && lane < cnt && first == lane && ra.cnt_cur_r[x] == 1u);
    FISM_STAMP(0);                                        // counting the next round's rows, finding my own
    {   // copy in the rows this user touches, a batch of rows' loads in flight at a time
        unsigned long long mleft = fmask;
        for (int s0 = 0; s0 < nuniq; s0 += kFismBatch) {
            int64_t it[kFismBatch];
            float q[kFismBatch][KR];
            double p[kFismBatch][KR], b[kFismBatch];
#pragma unroll
            for (int t = 0; t < kFismBatch; ++t) {
                const int src = mleft ? __ffsll((long long)mleft) - 1 : 0;
                mleft &= mleft - 1;
                it[t] = __builtin_amdgcn_readlane(x, src);
                const bool ex = s0 + t < nuniq;
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const int el = 64 * r + lane;
                    q[t][r] = (ex && el < k) ? a.Q[it[t] * k + el] : 0.0f;
                    p[t][r] = (ex && el < k) ? a.P[it[t] * k + el] : 0.0;
                }
                b[t] = ex ? a.Bi[it[t]] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < kFismBatch; ++t) {
                if (s0 + t < nuniq) {
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const int el = 64 * r + lane;
                        if (el < k) {
                            wq[(s0 + t) * k + el] = q[t][r]; wp[(s0 + t) * k + el] = p[t][r];
                            if (START) { sq[(s0 + t) * k + el] = q[t][r]; sp[(s0 + t) * k + el] = p[t][r]; }
                        }
                    }
                    if (lane == 0) { wb[s0 + t] = b[t]; if (START) sb[s0 + t] = b[t]; }
                }
            }
        }
    }
    FISM_STAMP(1);                                        // copy-in
    double *xr = a.x_rows + (e0 - a.user_ptr[ra.u_begin]) * k;
    const double coef = a.coef[u];
    double hist[KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) hist[r] = 0.0;
    for (int e = 0; e < ne; ++e) {                       // :44-46
        const int s = __builtin_amdgcn_readlane(urow, e);
#pragma unroll
        for (int r = 0; r < KR; ++r) { const int el = 64 * r + lane; if (el < k) hist[r] = hist[r] + wp[s * k + el]; }
    }
    FISM_STAMP(2);                                        // history sum
    double half_sq = 0.0;
    for (int e = 0; e < ne; ++e) {
        // the event's own row stays in registers over its rho draws (round 4): between two draws of an event lies the sum and the
        // update of the row, not also its trip to LDS and back; the bias likewise (every lane forms the same value)
        const int i = __builtin_amdgcn_readlane(urow, e);
        double xs[KR], di[KR];
        float qi[KR];
        double bi = wb[i];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int el = 64 * r + lane;
            xs[r] = 0.0; di[r] = 0.0; qi[r] = 0.0f;
            if (el < k) { di[r] = hist[r] - wp[i * k + el]; qi[r] = wq[i * k + el]; }
        }
        for (int c = 0; c < a.rho; ++c) {
            const int j = __builtin_amdgcn_readlane(urow, ne + e * a.rho + c);
            double dj[KR];
            float qj[KR], nj[KR];
            double ai = 0.0, aj = 0.0;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const int el = 64 * r + lane;
                dj[r] = 0.0; qj[r] = 0.0f;
                if (el < k) { dj[r] = hist[r] - wp[j * k + el]; qj[r] = wq[j * k + el]; }
                const double m1 = di[r] * (double)qi[r]; ai = ai + m1;
                const double m2 = dj[r] * (double)qj[r]; aj = aj + m2;
            }
            const double bj = wb[j];
            const double r_pos = coef * wave_sum_f64(ai) + bi;          // :54
            const double r_neg = coef * wave_sum_f64(aj) + bj;          // :55
            const double err = 1.0 - (r_pos - r_neg);
            half_sq = half_sq + 0.5 * (err * err);                      // :58
            const double bin = bi + a.lr * (err - a.regB * bi);         // :59
            const double bjn = bj - a.lr * (err + a.regB * bj);         // :60
            __builtin_amdgcn_wave_barrier();                            // (every lane has read the negative's bias)
            if (lane == 0) wb[j] = bjn;
            const double ec = err * coef;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const int el = 64 * r + lane;
                const float ri = regI32 * qi[r], rj = regI32 * qj[r];
                const float ni = (float)((double)qi[r] + a.lr * (ec * di[r] - (double)ri));     // :61
                nj[r] = (float)((double)qj[r] - a.lr * (ec * dj[r] + (double)rj));              // :62
                const float dq = ni - nj[r];
                xs[r] = xs[r] + err * (double)dq;                                                // :63
                qi[r] = ni;
                if (el < k) wq[j * k + el] = nj[r];
            }
            bi = bin;
            if (j == i) {                                               // (the reference never draws one of the user's own items; as the
                bi = bjn;                                               // earlier forms: the negative's values are the ones that stay)
#pragma unroll
                for (int r = 0; r < KR; ++r) qi[r] = nj[r];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int r = 0; r < KR; ++r) { const int el = 64 * r + lane; if (el < k) { wq[i * k + el] = qi[r]; xr[(int64_t)e * k + el] = xs[r]; } }
        if (lane == 0) wb[i] = bi;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    FISM_STAMP(3);                                        // draws
    const double pc = 1.0 / (double)a.rho * coef;                       // :68, left to right
    for (int eb = 0; eb < ne; eb += 8) {                                // (the x rows of eight events in flight at a time)
        double xv[8][KR];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < KR; ++r) { const int el = 64 * r + lane; xv[t][r] = (eb + t < ne && el < k) ? xr[(int64_t)(eb + t) * k + el] : 0.0; }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (eb + t < ne) {
                const int s = __builtin_amdgcn_readlane(urow, eb + t);
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const int el = 64 * r + lane;
                    if (el < k) { const double p = wp[s * k + el]; wp[s * k + el] = p + a.lr * (pc * xv[t][r] - a.regI * p); }
                }
            }
        }
    }
    FISM_STAMP(4);                                        // item-history rows
    {   // what this user changed.  Shared rows first: working row - round-start row, formed IN the working rows (the round-start
        // rows read again, a batch at a time; with START they are at hand) ...
        unsigned long long mleft = fmask;
        for (int s0 = 0; !START && s0 < nuniq; s0 += kFismBatch) {
            float q[kFismBatch][KR];
            double p[kFismBatch][KR], b[kFismBatch];
            bool shared[kFismBatch];
#pragma unroll
            for (int t = 0; t < kFismBatch; ++t) {
                const int src = mleft ? __ffsll((long long)mleft) - 1 : 0;
                mleft &= mleft - 1;
                const int64_t it = __builtin_amdgcn_readlane(x, src);
                shared[t] = s0 + t < nuniq && !((emask >> src) & 1ull);
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const int el = 64 * r + lane;
                    q[t][r] = (shared[t] && el < k) ? a.Q[it * k + el] : 0.0f;
                    p[t][r] = (shared[t] && el < k) ? a.P[it * k + el] : 0.0;
                }
                b[t] = shared[t] ? a.Bi[it] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < kFismBatch; ++t) {
                if (shared[t]) {
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const int el = 64 * r + lane;
                        if (el < k) { wq[(s0 + t) * k + el] = wq[(s0 + t) * k + el] - q[t][r]; wp[(s0 + t) * k + el] = wp[(s0 + t) * k + el] - p[t][r]; }
                    }
                    if (lane == 0) wb[s0 + t] = wb[s0 + t] - b[t];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ... then every row on its way with nothing left to wait for: a row of the user's own back in place, a shared row's
        // differences into the round's buffers (no-return atomics; on this target they count on the same counter as loads, and a
        // batch of loads behind them would wait for every one of them to land)
        mleft = fmask;
        for (int s = 0; s < nuniq; ++s) {
            const int src = __ffsll((long long)mleft) - 1;
            mleft &= mleft - 1;
            const int64_t it = __builtin_amdgcn_readlane(x, src);
            const bool own = (emask >> src) & 1ull;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const int el = 64 * r + lane;
                if (el < k) {
                    float vq = wq[s * k + el];
                    double vp = wp[s * k + el];
                    if (own) { a.Q[it * k + el] = vq; a.P[it * k + el] = vp; }
                    else {
                        if (START) { vq = vq - sq[s * k + el]; vp = vp - sp[s * k + el]; }
                        if (vq != 0.0f) atomicAdd(ra.dQ + it * k + el, vq);
                        if (vp != 0.0) atomicAdd(ra.dP + it * k + el, vp);
                    }
                }
            }
            if (lane == 0) {
                double vb = wb[s];
                if (own) a.Bi[it] = vb;
                else { if (START) vb = vb - sb[s]; if (vb != 0.0) atomicAdd(ra.dB + it, vb); }
            }
        }
    }
    if (ra.cnt_cur && lane < cnt && first == lane) ra.cnt_cur[x] = 0u;
    FISM_STAMP(5);                                        // rows back in place / differences
#ifdef YUE_FISM_STAMPS
    if (lane == 0) atomicAdd(ra.stamps + 7, 1ull);
#endif
    if (lane == 0) atomicAdd(a.out, half_sq);
}

// end of a round: model += summed differences, buffers cleared
__global__ void __launch_bounds__(256) k_fism_apply(double *P, float *Q, double *Bi, double *dP, float *dQ, double *dB, int64_t n, int k) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tot = n * k;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += stride) {
        const float q = dQ[t];
        if (q != 0.0f) { Q[t] = Q[t] + q; dQ[t] = 0.0f; }
        const double p = dP[t];
        if (p != 0.0) { P[t] = P[t] + p; dP[t] = 0.0; }
    }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const double b = dB[t];
        if (b != 0.0) { Bi[t] = Bi[t] + b; dB[t] = 0.0; }
    }
}

// FISM.py:70: sum(P*P), sum(Q*Q) (float32 products), Bi.Bi -- accumulated in double.
__global__ void __launch_bounds__(256) k_fism_sumsq(const double *P, const float *Q, const double *Bi, int64_t n, int k, double *out3) {
    __shared__ double part[3][4];
    double sp = 0.0, sq = 0.0, sb = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tot = n * k;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += stride) {
        const double p2 = P[t] * P[t]; sp += p2;
        const float q2 = Q[t] * Q[t]; sq += (double)q2;
    }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) { const double b2 = Bi[t] * Bi[t]; sb += b2; }
    for (int off = 32; off >= 1; off >>= 1) { sp += __shfl_xor(sp, off); sq += __shfl_xor(sq, off); sb += __shfl_xor(sb, off); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = sp; part[1][threadIdx.x >> 6] = sq; part[2][threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(out3 + threadIdx.x, part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3]);
}

// predict (FISM.py:75-83): hist[t] = sum of P rows of the listed user's training events (one block per user) ...
__global__ void __launch_bounds__(256) k_fism_hist(const double *P, int k, const int64_t *row_ptr, const int32_t *row_items, double *hist) {
    const int64_t t = blockIdx.x;
    for (int el = threadIdx.x; el < k; el += 256) {
        double s = 0.0;
        for (int64_t e = row_ptr[t]; e < row_ptr[t + 1]; ++e) s = s + P[(int64_t)row_items[e] * k + el];
        hist[t * k + el] = s;
    }
}

// ... scores[t][it] = Bi[it] + Q[it].hist[t] - sum_e P[it][e] * Q[it][e], element order ascending.
__global__ void __launch_bounds__(256) k_fism_scores(const double *P, const float *Q, const double *Bi, int64_t n, int k,
                                                     const double *hist, int64_t nu, double *scores) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nu * n) return;
    const int64_t t = idx / n, it = idx - t * n;
    const double *h = hist + t * k, *p = P + it * k;
    const float *q = Q + it * k;
    double d = 0.0, s = 0.0;
    for (int e = 0; e < k; ++e) {
        const double qe = (double)q[e];
        const double m1 = qe * h[e]; d = d + m1;
        const double m2 = p[e] * qe; s = s + m2;
    }
    scores[idx] = Bi[it] + d - s;
}

// The reference's selection (base/IterativeRecommender.py:98-145) on explicit float64 score rows: one
// thread per listed user, candidates = item ids ascending minus the user's training items (row_items,
// unsorted, duplicates allowed).  out_ids[t][N], out_scores[t][N]; flags[t] = 1 if fewer than N candidates.
__global__ void __launch_bounds__(64) k_fism_select(const double *scores, int64_t n, int64_t nu, int N,
                                                    const int64_t *row_ptr, const int32_t *row_items,
                                                    int32_t *out_ids, double *out_scores, int32_t *flags) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= nu) return;
    const double *sc = scores + t * n;
    double *a = out_scores + t * N;
    int32_t *id = out_ids + t * N;
    const int64_t m0 = row_ptr[t], m1 = row_ptr[t + 1];
    auto masked = [&](int64_t it) { for (int64_t e = m0; e < m1; ++e) if (row_items[e] == it) return true; return false; };
    int cnt = 0;
    for (int64_t it = 0; it < n && cnt < N; ++it) {
        if (masked(it)) continue;
        int pos = cnt;                                   // stable insertion == list.sort(reverse=True) on the seed
        while (pos > 0 && a[pos - 1] < sc[it]) { a[pos] = a[pos - 1]; id[pos] = id[pos - 1]; --pos; }
        a[pos] = sc[it]; id[pos] = (int32_t)it; ++cnt;
    }
    flags[t] = cnt < N;
    if (cnt < N) { for (int q = cnt; q < N; ++q) { a[q] = -INFINITY; id[q] = -1; } return; }
    for (int64_t it = 0; it < n; ++it) {
        if (masked(it)) continue;
        const double s = sc[it];
        if (a[N - 1] < s) {
            int p = 0;
            while (a[p] >= s) ++p;                       // first slot strictly below s
            a[p] = s; id[p] = (int32_t)it;
        }
    }
}

}  // namespace yue
