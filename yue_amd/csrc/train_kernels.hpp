// BPR training kernels for gfx950 (CDNA4): sampler pass, exact replay levels, the TF-style minibatch step, CUNE's steps,
// the S-round kernel with in-launch retire (k_round) and the small reduction / apply kernels.  Building blocks: bpr_device.hpp.
#pragma once
#include "bpr_device.hpp"

namespace yue {

__global__ void __launch_bounds__(256) k_sample(TrainArgs a, int64_t E) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) a.ev_j[e] = sample_negative(a, a.ev_u[e], e);
}

// ------------------------------------------------------------------------------------------
// Wave layout: lane l holds elements 64*r + l (r < KR) of P[u], Q[i] and Q[j]; every load, store
// and atomic wave-instruction covers 256 contiguous bytes of one row.
//
// Exact replay: one launch = one dependency level (pairwise row-disjoint triplets), rows are
// rewritten in place.  One wave walks `tpw` consecutive triplets of the level.
// ------------------------------------------------------------------------------------------
template <int KR>
__global__ void __launch_bounds__(256) k_bpr_level(TrainArgs a, int64_t e_begin, int64_t e_end, int tpw) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t base = e_begin + wave * tpw;
    if (base >= e_end) return;
    const int cnt = (int)((e_end - base) < (int64_t)tpw ? (e_end - base) : (int64_t)tpw);
    int u = 0, i = 0, j = 0;
    if (lane < cnt) { u = a.ev_u[base + lane]; i = a.ev_i[base + lane]; j = a.ev_j[base + lane]; }
    const int k = a.k;
    double nll = 0.0;
    for (int t = 0; t < cnt; ++t) {
        const int64_t tu = __builtin_amdgcn_readlane(u, t);
        const int64_t ti = __builtin_amdgcn_readlane(i, t);
        const int64_t tj = __builtin_amdgcn_readlane(j, t);
        float p[KR], qi[KR], qj[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = 64 * r + lane;
            p[r] = e < k ? a.P[tu * k + e] : 0.0f;
            qi[r] = e < k ? a.Q[ti * k + e] : 0.0f;
            qj[r] = e < k ? a.Q[tj * k + e] : 0.0f;
        }
        float ai = 0.0f, aj = 0.0f;
#pragma unroll
        for (int r = 0; r < KR; ++r) { const float a1 = p[r] * qi[r]; ai = ai + a1; const float a2 = p[r] * qj[r]; aj = aj + a2; }
        const float x = wave_sum(ai) - wave_sum(aj);               // BPR.py:50, fp32 margin
        const double s = 1.0 / (1.0 + exp(-(double)x));            // qmath.py:115-116
        const float c = (float)(a.lr * (1.0 - s));
        nll += -log(s);                                            // BPR.py:58
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = 64 * r + lane;
            const Elem o = bpr_elem(p[r], qi[r], qj[r], c, a.ru, a.ri);
            if (e < k) { a.P[tu * k + e] = o.p2; a.Q[ti * k + e] = o.qi2; a.Q[tj * k + e] = o.qj2; }
        }
    }
    if (lane == 0) atomicAdd(a.nll_slots + (wave & (kNllSlots - 1)), nll);
}

// ------------------------------------------------------------------------------------------
// The reference's live path (recommender/cf/BPR.py:83-129): one minibatch step of the TensorFlow-1
// graph -- softplus(-(u.vi - u.vj)) summed over the fed triplets plus reg * l2_loss of the three gathered row sets, Adam on
// both embedding matrices -- as two kernels.  Parity unpinned (no TensorFlow here): the checker is oracle/numpy_adam.py.
//   k_mb_grad: a wave takes 32 consecutive triplets (the reference feeds 100 negatives per positive event, so consecutive
//     triplets share u and i: their rows are loaded once per run, their gradient rows are summed in registers and added
//     once per run with float atomics; every negative's gradient row is added as it comes), all in float32 as TensorFlow
//     computes; loss partials in double slots.
//   k_adam: dense Adam over a matrix (TF-1's sparse apply decays m and v and moves EVERY row: a zero gradient on the
//     untouched ones), gradient buffer cleared for the next step.
// ------------------------------------------------------------------------------------------
struct MbArgs {
    const float *U, *V;
    float *gU, *gV;
    int k;
    const int32_t *u, *i, *j;
    int64_t T;
    float reg;
    double *loss_slots;
};

template <int KR>
__global__ void __launch_bounds__(256) k_mb_grad(MbArgs a) {
    constexpr int CH = 32;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t t0 = wave * CH;
    if (t0 >= a.T) return;
    const int64_t t1 = t0 + CH < a.T ? t0 + CH : a.T;
    const int k = a.k;
    float pu[KR], vi[KR], gu[KR], gvi[KR];
    int64_t cu = -1, ci = -1;
    float dui = 0.0f, su = 0.0f, si = 0.0f;          // u.vi and the two squared norms of the current run
    double loss = 0.0;
    auto flush = [&]() {
        if (cu < 0) return;
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = 64 * r + lane;
            if (e < k) { atomicAdd(a.gU + cu * k + e, gu[r]); atomicAdd(a.gV + ci * k + e, gvi[r]); }
        }
    };
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t u = a.u[t], i = a.i[t], j = a.j[t];
        if (u != cu || i != ci) {
            flush();
            cu = u; ci = i;
            float d = 0.0f, q1 = 0.0f, q2 = 0.0f;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const int e = 64 * r + lane;
                pu[r] = e < k ? a.U[u * k + e] : 0.0f;
                vi[r] = e < k ? a.V[i * k + e] : 0.0f;
                gu[r] = 0.0f; gvi[r] = 0.0f;
                d = __builtin_fmaf(pu[r], vi[r], d); q1 = __builtin_fmaf(pu[r], pu[r], q1); q2 = __builtin_fmaf(vi[r], vi[r], q2);
            }
            dui = wave_sum(d); su = wave_sum(q1); si = wave_sum(q2);
        }
        float vn[KR];
        float d = 0.0f, q3 = 0.0f;
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = 64 * r + lane;
            vn[r] = e < k ? a.V[j * k + e] : 0.0f;
            d = __builtin_fmaf(pu[r], vn[r], d); q3 = __builtin_fmaf(vn[r], vn[r], q3);
        }
        const float err = dui - wave_sum(d);                                 // BPR.py:101
        const float sn = wave_sum(q3);
        const float ax = __builtin_fabsf(err);
        loss += (double)(fmaxf(-err, 0.0f) + log1pf(expf(-ax)));             // softplus(-err), :102
        loss += (double)(0.5f * a.reg * (su + si + sn));                     // the three l2 terms of this triplet, :104-107
        const float g = -1.0f / (1.0f + expf(err));                          // d softplus(-e) / d e = -sigmoid(-e)
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = 64 * r + lane;
            gu[r] = gu[r] + (g * (vi[r] - vn[r]) + a.reg * pu[r]);
            gvi[r] = gvi[r] + (g * pu[r] + a.reg * vi[r]);
            if (e < k) atomicAdd(a.gV + j * k + e, -g * pu[r] + a.reg * vn[r]);
        }
    }
    flush();
    if (lane == 0) atomicAdd(a.loss_slots + (wave & (kNllSlots - 1)), loss);
}

__global__ void __launch_bounds__(256) k_adam(float *var, float *m, float *v, float *grad, int64_t count, float lr_t, float beta1, float beta2, float eps) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
        const float g = grad[t];
        const float m1 = beta1 * m[t] + (1.0f - beta1) * g;
        const float v1 = beta2 * v[t] + ((1.0f - beta2) * g) * g;
        m[t] = m1; v[t] = v1;
        var[t] = var[t] - lr_t * m1 / (__builtin_sqrtf(v1) + eps);
        if (g != 0.0f) grad[t] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// CUNE's two-level BPR step (reference recommender/advanced/CUNE.py:126-172), exact sequential
// semantics: ONE wave walks the given (u, i, k, j) steps in order (every step of a user rewrites P[u]: the chain is
// sequential anyway).  k >= 0: (i over k) then (k over j, margin and step scaled by 1/s), then the four decays;
// k < 0: the plain (i over j) step without decay.  As the reference writes it, the sigmoid is evaluated again, on the
// rows already updated, in every one of the statements.  Rounding as NumPy does it: float32 rows, coefficients
// formed in double and rounded to float32 once, the 1/s factor of the margin applied in float32, products and sums
// rounded separately.  loss_out[t] = the step's two (one) -log sigmoid terms on the final rows.
// ------------------------------------------------------------------------------------------
struct CuneArgs {
    float *P, *Q;
    int k;
    const int32_t *u, *i, *kk, *j;
    int64_t T;
    double inv_s, lr;            // 1 / s as Python forms it
    float inv_s32;               // the same, as the float32 NumPy multiplies the float32 margin with
    float ru, ri;                // float32(lr * regU), float32(lr * regI)
    double *loss_out;
};

template <int KR>
__global__ void __launch_bounds__(64) k_cune_steps(CuneArgs a) {
    const int lane = threadIdx.x;
    const int k = a.k;
    auto dot = [&](const float (&x)[KR], const float (&y)[KR]) {
        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < KR; ++r) { const float m = x[r] * y[r]; acc = acc + m; }
        return wave_sum(acc);
    };
    auto sig = [](double x) { return 1.0 / (1.0 + exp(-x)); };
    for (int64_t t = 0; t < a.T; ++t) {
        const int64_t u = a.u[t], i = a.i[t], kq = a.kk[t], j = a.j[t];
        float p[KR], qi[KR], qk[KR], qj[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = 64 * r + lane;
            p[r] = e < k ? __hip_atomic_load(a.P + u * k + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
            qi[r] = e < k ? __hip_atomic_load(a.Q + i * k + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
            qj[r] = e < k ? __hip_atomic_load(a.Q + j * k + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
            qk[r] = (e < k && kq >= 0) ? __hip_atomic_load(a.Q + kq * k + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
        }
        // k == j is possible (both are merely items the user has not listened to): the two names then mean ONE row,
        // every write through one of them is seen through the other (NumPy updates the same memory)
        const bool same = kq == j;
        double loss;
        if (kq >= 0) {
            float c = (float)(a.lr * (1.0 - sig((double)(dot(p, qi) - dot(p, qk)))));                       // :133
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float d = qi[r] - qk[r]; const float td = c * d; p[r] = p[r] + td; }
            c = (float)(a.lr * (1.0 - sig((double)(dot(p, qi) - dot(p, qk)))));                             // :135
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float tq = c * p[r]; qi[r] = qi[r] + tq; }
            c = (float)(a.lr * (1.0 - sig((double)(dot(p, qi) - dot(p, qk)))));                             // :137
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float tq = c * p[r]; qk[r] = qk[r] - tq; if (same) qj[r] = qk[r]; }
            float m2 = a.inv_s32 * (dot(p, qk) - dot(p, qj));
            c = (float)(a.inv_s * a.lr * (1.0 - sig((double)m2)));                                          // :148
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float d = qk[r] - qj[r]; const float td = c * d; p[r] = p[r] + td; }
            m2 = a.inv_s32 * (dot(p, qk) - dot(p, qj));
            c = (float)(a.inv_s * a.lr * (1.0 - sig((double)m2)));                                          // :151
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float tq = c * p[r]; qk[r] = qk[r] + tq; if (same) qj[r] = qk[r]; }
            m2 = a.inv_s32 * (dot(p, qk) - dot(p, qj));
            c = (float)(a.inv_s * a.lr * (1.0 - sig((double)m2)));                                          // :153
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float tq = c * p[r]; qj[r] = qj[r] - tq; if (same) qk[r] = qj[r]; }
#pragma unroll
            for (int r = 0; r < KR; ++r) {                                                                  // :156-159
                const float rp = a.ru * p[r]; p[r] = p[r] - rp;
                const float r1 = a.ri * qi[r]; qi[r] = qi[r] - r1;
                const float r2 = a.ri * qj[r]; qj[r] = qj[r] - r2;
                if (same) qk[r] = qj[r];
                const float r3 = a.ri * qk[r]; qk[r] = qk[r] - r3;
                if (same) qj[r] = qk[r];
            }
            const float m3 = a.inv_s32 * (dot(p, qk) - dot(p, qj));
            loss = -log(sig((double)(dot(p, qi) - dot(p, qk)))) - log(sig((double)m3));                     // :161-162
        } else {
            float c = (float)(a.lr * (1.0 - sig((double)(dot(p, qi) - dot(p, qj)))));                       // :168
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float d = qi[r] - qj[r]; const float td = c * d; p[r] = p[r] + td; }
            c = (float)(a.lr * (1.0 - sig((double)(dot(p, qi) - dot(p, qj)))));                             // :169
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float tq = c * p[r]; qi[r] = qi[r] + tq; }
            c = (float)(a.lr * (1.0 - sig((double)(dot(p, qi) - dot(p, qj)))));                             // :170
#pragma unroll
            for (int r = 0; r < KR; ++r) { const float tq = c * p[r]; qj[r] = qj[r] - tq; }
            loss = -log(sig((double)(dot(p, qi) - dot(p, qj))));                                            // :172
        }
#pragma unroll
        for (int r = 0; r < KR; ++r) {
            const int e = 64 * r + lane;
            if (e < k) {
                __hip_atomic_store(a.P + u * k + e, p[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.Q + i * k + e, qi[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.Q + j * k + e, qj[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (kq >= 0 && !same) __hip_atomic_store(a.Q + kq * k + e, qk[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (lane == 0) a.loss_out[t] = loss;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------------
// S-round launch.  Every wave first takes the tickets of TPW events of the NEXT round (touch count +
// staging-slot table of their two item rows, user-row flush counts; lanes TPW .. 2 TPW - 1), then
// updates TPW events of THIS round -- no separate workgroups for the counting, so every resident
// workgroup carries 4 * TPW events of the round (the negatives come from k_sample, drawn up front): one wave takes
// TPW consecutive events, requests all their rows up front (straight-line code: the compiler's
// counted waits keep every gather in flight), evaluates the TPW sigmoids in one double-precision
// pass (lane t holds triplet t), then writes:
//   * an item row touched exactly once in the round (cnt == 1): the new row, in place, plain
//     stores -- nobody else reads or writes it this round;
//   * a row with 2..kStageMax touches: (new - old) written to the touch's own staging row (behind the
//     item rows in the Q allocation) with
//     write-through (sc1) stores; the last toucher of the row (touch count reaching zero: every other
//     toucher has drained its stores before its decrement) reads the staged rows back with sc1 loads,
//     adds them in event order -- the order of the oracle's sum -- and rewrites the row once.
//     Scattered 512-byte float-atomic rows run at ~2.4e9/s on MI355X, stores and loads at ~1.1e10/s
//     (tools/experiments/rowops.hip), and most contended rows of a round have 2-3 touches;
//   * a hotter row: (new - old) added into dQ with float atomics; the last toucher swaps the sum out
//     of dQ (leaving it zeroed) and rewrites the row.
// P[u] differences are summed in registers over a run of equal users and flushed into dP; user
// rows are finished by the same last-arriver rule (or, on a communicator, by the all-reduce).
// ------------------------------------------------------------------------------------------

// evu / evi / evj: the event arrays again, as restrict-qualified read-only views of THIS round's
// range, so that the wave-uniform reads of a batch's (u, i, j) become scalar loads (s_load): they do
// not queue behind the vector-memory traffic of the other waves.
template <int KR, int TPW>
__global__ void __launch_bounds__(256) k_round(TrainArgs a, RoundArgs ra, const int32_t *__restrict__ evu,
                                               const int32_t *__restrict__ evi, const int32_t *__restrict__ evj) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Tickets of the NEXT round: lanes TPW .. 2 TPW - 1 of wave w take them for events n_begin + w * TPW + (lane - TPW)
    // -- one returning atomic per item row (the low half before the add is the touch's ticket on its row), one
    // flush count per run of equal users.  An updating wave issues them behind its row stores, so that they
    // return inside the drain wait it needs anyway (in front of the gathers they would hold the gathers back:
    // returns are in order and atomics are the slowest); the slot-table entries follow the wait.
    uint32_t tk_i = 0xffffffffu, tk_j = 0xffffffffu, nx_slot = 0u;
    int32_t nx_i = 0, nx_j = -1;
    auto take_tickets = [&]() {
        const int64_t ne = ra.n_begin + wave * TPW + (lane - TPW);
        if (lane >= TPW && lane < 2 * TPW && ne < ra.n_end) {
            nx_i = a.ev_i[ne]; nx_j = a.ev_j[ne];
            nx_slot = 2u * (uint32_t)(ne - ra.n_begin);
            if (nx_j >= 0) { tk_i = (uint32_t)atomicAdd(ra.cnt_next + nx_i, kTouch); tk_j = (uint32_t)atomicAdd(ra.cnt_next + nx_j, kTouch); }
            if (ra.apply_p) {      // one flush per run of equal users inside a TPW-aligned batch
                const int32_t u = a.ev_u[ne];
                if (lane == TPW || a.ev_u[ne - 1] != u) atomicAdd(ra.cntp_next + u, 1u);
            }
        }
    };
    auto publish_slots = [&]() {
        if (ra.staged && nx_j >= 0) {
            if (tk_i < kStageMax) ra.tab_next[(size_t)nx_i * kStageMax + tk_i] = nx_slot;
            if (tk_j < kStageMax) ra.tab_next[(size_t)nx_j * kStageMax + tk_j] = nx_slot + 1u;
        }
    };
    const int64_t base = ra.e_begin + wave * TPW;
    if (base >= ra.e_end) { take_tickets(); publish_slots(); return; }
    YUE_STAMP(0, "");

    // batch header through the scalar unit: (u, i, j) of the TPW events
    int hu[TPW], hi_[TPW], hj[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const bool ex = base + t < ra.e_end;
        const int64_t ix = ex ? base + t : ra.e_end - 1;
        hu[t] = evu[ix]; hi_[t] = evi[ix]; hj[t] = ex ? evj[ix] : -1;
    }
    // lane t keeps event t (the retire phase and the loss work per lane)
    int i = 0, j = -1;
#pragma unroll
    for (int t = 0; t < TPW; ++t) if (lane == t) { i = hi_[t]; j = hj[t]; }
    YUE_STAMP(1, "s_waitcnt lgkmcnt(0)");
    uint32_t ci = 0, cj = 0;             // touches of my rows in this round (the immutable half)
    if (lane < TPW && j >= 0) { ci = (uint32_t)(ra.cnt_cur[i] >> 32); cj = (uint32_t)(ra.cnt_cur[j] >> 32); }
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
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(a.Q, 0, qrec, kRsrcFlags);
    const auto rsdQ = __builtin_amdgcn_make_buffer_rsrc(a.dQ, 0, qrec, kRsrcFlags);
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc(a.P + (uint64_t)u0 * k, 0, prec, kRsrcFlags);
    const auto rsdP = __builtin_amdgcn_make_buffer_rsrc(a.dP + (uint64_t)u0 * k, 0, prec, kRsrcFlags);
    // staging rows of this round: 2 per event (the host keeps 2 * events * row_bytes below 2^31 when staged)
    const auto rsS = __builtin_amdgcn_make_buffer_rsrc(a.stage, 0, ra.staged ? (int)(2u * (unsigned)(ra.e_end - ra.e_begin) * row_bytes) : 0, kRsrcFlags);

    unsigned oi[TPW], oj[TPW], ou[TPW], ru_[TPW], ri_[TPW], rj_[TPW];
    bool ok[TPW];
    float qi[TPW][KR], qj[TPW][KR], p[TPW][KR];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        ru_[t] = (unsigned)hu[t];
        ri_[t] = (unsigned)hi_[t];
        const int tj = hj[t];
        ok[t] = tj >= 0;                                 // no event / sampler gave up: nothing is written
        rj_[t] = ok[t] ? (unsigned)tj : 0u;
        oi[t] = ri_[t] * row_bytes; oj[t] = rj_[t] * row_bytes;
        ou[t] = (base + t < ra.e_end ? ru_[t] - u0 : 0u) * row_bytes;
#pragma unroll
        for (int r = 0; r < KR; ++r) { qi[t][r] = YUE_BLOAD(rsQ, vo[r], oi[t]); qj[t][r] = YUE_BLOAD(rsQ, vo[r], oj[t]); }
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
    int nruns = 0;                                       // runs of equal users in this batch (wave-uniform)
    unsigned run_u = 0;                                  // lane q holds the user of run q
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const float c = rdlane(cs, t);
        const unsigned cti = (unsigned)__builtin_amdgcn_readlane(ci, t), ctj = (unsigned)__builtin_amdgcn_readlane(cj, t);
        const bool uniq_i = cti == 1u, uniq_j = ctj == 1u;
        const bool stg_i = ra.staged && cti <= kStageMax, stg_j = ra.staged && ctj <= kStageMax;
        const unsigned ss = 2u * ((unsigned)wave * TPW + t) * row_bytes;     // my two staging rows
        if (ok[t]) {                                     // wave-uniform
            run_ok = true;
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const Elem o = bpr_elem(p[t][r], qi[t][r], qj[t][r], c, a.ru, a.ri);
                if (uniq_i) YUE_BSTORE(o.qi2, rsQ, vo[r], oi[t]);
                else if (stg_i) YUE_BSTORE_SC1(o.qi2 - qi[t][r], rsS, vo[r], ss);
                else YUE_BATOMIC(o.qi2 - qi[t][r], rsdQ, vo[r], oi[t]);
                if (uniq_j) YUE_BSTORE(o.qj2, rsQ, vo[r], oj[t]);
                else if (stg_j) YUE_BSTORE_SC1(o.qj2 - qj[t][r], rsS, vo[r], ss + row_bytes);
                else YUE_BATOMIC(o.qj2 - qj[t][r], rsdQ, vo[r], oj[t]);
                dp[r] += o.p2 - p[t][r];
            }
            if (lane == 0) {                             // sole toucher: reset the counter here
                if (uniq_i) ra.cnt_cur[ri_[t]] = 0ull;
                if (uniq_j) ra.cnt_cur[rj_[t]] = 0ull;
            }
        }
        // end of a run of equal users (or of the batch): flush the summed P[u] differences
        const bool exists = base + t < ra.e_end;
        const bool last = exists && ((t == TPW - 1) || base + t + 1 >= ra.e_end || ru_[t + 1 < TPW ? t + 1 : t] != ru_[t]);
        if (last) {
            if (run_ok) {
#pragma unroll
                for (int r = 0; r < KR; ++r) YUE_BATOMIC(dp[r], rsdP, vo[r], ou[t]);
            }
            run_u = lane == nruns ? ru_[t] : run_u;
            ++nruns;
#pragma unroll
            for (int r = 0; r < KR; ++r) dp[r] = 0.0f;
            run_ok = false;
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // Retire this batch's touches of contended rows.  Every toucher decrements the row's count once
    // its own adds are acknowledged; whoever takes the count to zero knows every add of the round
    // has been performed and every toucher has read the row: it swaps the sum out of dQ (returning
    // atomic: coherent at the memory side, leaves dQ zeroed) and rewrites the row.
    YUE_STAMP(4, "");
    // staging slots of my contended rows (written by the previous launch): fetched behind the drain of
    // my stores, only where a last toucher could need them (rows with 2..kStageMax touches)
    uint32_t si[kStageMax], sj[kStageMax];
#pragma unroll
    for (unsigned q = 0; q < kStageMax; ++q) si[q] = sj[q] = 0xffffffffu;
    if (ra.staged && lane < TPW && j >= 0) {
        if (ci != 1u && ci <= kStageMax) {
            const uint4 wi = *reinterpret_cast<const uint4 *>(ra.tab_cur + (size_t)i * kStageMax);
            si[0] = wi.x; si[1] = wi.y; si[2] = wi.z; si[3] = wi.w;
        }
        if (cj != 1u && cj <= kStageMax) {
            const uint4 wj = *reinterpret_cast<const uint4 *>(ra.tab_cur + (size_t)j * kStageMax);
            sj[0] = wj.x; sj[1] = wj.y; sj[2] = wj.z; sj[3] = wj.w;
        }
    }
    take_tickets();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    YUE_STAMP(5, "");
    publish_slots();
    // entries beyond the row's touch count are leftovers of earlier rounds: drop them, then put the
    // slots in ascending order = event order, the order in which the oracle sums a row's differences
    static_assert(kStageMax == 4, "the sorting network below is written for four slots");
#pragma unroll
    for (unsigned q = 0; q < kStageMax; ++q) { if (q >= ci) si[q] = 0xffffffffu; if (q >= cj) sj[q] = 0xffffffffu; }
#define YUE_CSWAP(x, y) { const uint32_t lo_ = min(x, y), hi2_ = max(x, y); x = lo_; y = hi2_; }
    YUE_CSWAP(si[0], si[1]) YUE_CSWAP(si[2], si[3]) YUE_CSWAP(si[0], si[2]) YUE_CSWAP(si[1], si[3]) YUE_CSWAP(si[1], si[2])
    YUE_CSWAP(sj[0], sj[1]) YUE_CSWAP(sj[2], sj[3]) YUE_CSWAP(sj[0], sj[2]) YUE_CSWAP(sj[1], sj[3]) YUE_CSWAP(sj[1], sj[2])
#undef YUE_CSWAP

    bool last_i = false, last_j = false, last_p = false;
    if (lane < TPW && j >= 0) {
        if (ci != 1u) last_i = (uint32_t)atomicAdd(ra.cnt_cur + i, ~0ull) == 1u;     // -1 on the low half
        if (cj != 1u) last_j = (uint32_t)atomicAdd(ra.cnt_cur + j, ~0ull) == 1u;
    }
    if (ra.apply_p && lane < nruns) last_p = atomicSub(ra.cntp_cur + run_u, 1u) == 1u;
    // winners: bits [0,TPW) = item row i of that lane, [TPW,2*TPW) = row j, [2*TPW,3*TPW) = user run
    unsigned long long win = (__ballot(last_i) & ((1ull << TPW) - 1)) | ((__ballot(last_j) & ((1ull << TPW) - 1)) << TPW) |
                             ((__ballot(last_p) & ((1ull << TPW) - 1)) << (2 * TPW));
    YUE_STAMP(6, "s_waitcnt vmcnt(0)");
    while (win) {
        // up to four rows per pass: all swaps and row loads are issued before the first store
        float *xp[4], *dx[4];
        bool act[4];
        unsigned nst[4], so[4][kStageMax];               // staged rows to add (0: the row went through dQ / dP)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            act[sl] = win != 0;
            const int b = act[sl] ? __ffsll((long long)win) - 1 : 0;
            if (act[sl]) win &= win - 1;
            const int src = b % TPW;
            const unsigned row = b < TPW ? (unsigned)__builtin_amdgcn_readlane(i, src)
                                 : b < 2 * TPW ? (unsigned)__builtin_amdgcn_readlane(j, src)
                                               : (unsigned)__builtin_amdgcn_readlane((int)run_u, src);
            const unsigned touches = b < TPW ? (unsigned)__builtin_amdgcn_readlane(ci, src)
                                     : b < 2 * TPW ? (unsigned)__builtin_amdgcn_readlane(cj, src) : 0u;
            nst[sl] = act[sl] && ra.staged && b < 2 * TPW && touches <= kStageMax ? touches : 0u;
#pragma unroll
            for (unsigned q = 0; q < kStageMax; ++q)
                so[sl][q] = (b < TPW ? (unsigned)__builtin_amdgcn_readlane(si[q], src) : (unsigned)__builtin_amdgcn_readlane(sj[q], src)) * row_bytes;
            const uint64_t o = (uint64_t)row * k;
            xp[sl] = (b < 2 * TPW ? a.Q : a.P) + o;
            dx[sl] = (b < 2 * TPW ? a.dQ : a.dP) + o;
            if (act[sl] && b < 2 * TPW && lane == 0) ra.cnt_cur[row] = 0ull;      // every touch retired: clear the word
        }
        float d[4][KR], x[4][KR];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
            if (act[sl]) {
                if (nst[sl]) {
                    // every load of the staged bytes is an sc1 load issued after this wave's own
                    // decrement returned; slots beyond the count get an out-of-range offset (reads 0)
                    float st[kStageMax][KR];
#pragma unroll
                    for (unsigned q = 0; q < kStageMax; ++q)
#pragma unroll
                        for (int r = 0; r < KR; ++r)
                            st[q][r] = YUE_BLOAD_SC1(rsS, q < nst[sl] ? vo[r] : kOobOffset, q < nst[sl] ? so[sl][q] : 0u);
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const unsigned e = 64u * r + lane;
                        if (e < k) x[sl][r] = xp[sl][e];
                        float acc = st[0][r];
#pragma unroll
                        for (unsigned q = 1; q < kStageMax; ++q) acc = acc + st[q][r];
                        d[sl][r] = acc;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const unsigned e = 64u * r + lane;
                        if (e < k) { d[sl][r] = atomicExch(dx[sl] + e, 0.0f); x[sl][r] = xp[sl][e]; }
                    }
                }
            }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
            if (act[sl]) {
#pragma unroll
                for (int r = 0; r < KR; ++r) {
                    const unsigned e = 64u * r + lane;
                    if (e < k) xp[sl][e] = x[sl][r] + d[sl][r];
                }
            }
    }
    YUE_STAMP(7, "s_waitcnt vmcnt(0)");
#pragma unroll
    for (int off = 1; off < TPW; off <<= 1) nll += __shfl_xor(nll, off);
    if (lane == 0 && nll != 0.0) atomicAdd(a.nll_slots + (wave & (kNllSlots - 1)), nll);
}

// Multi-GPU: after the all-reduce of dP[first .. first+count) the same range is applied everywhere.
__global__ void __launch_bounds__(256) k_apply_range(float *X, float *dX, int64_t first_elem, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += stride) {
        const int64_t o = first_elem + t;
        X[o] += dX[o];
        dX[o] = 0.0f;
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

__global__ void __launch_bounds__(64) k_sum_slots(const double *slots, int nslots, double *out) {
    double s = 0.0;
    for (int t = threadIdx.x; t < nslots; t += 64) s += slots[t];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) *out = s;
}

}  // namespace yue
