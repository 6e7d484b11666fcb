// libyue_hip.so -- FISM (recommender/cf/FISM.py): parity path and throughput form (include/yue_hip.h).
#include "host_common.hpp"

#include "fism_kernels.hpp"

using yue_host::fail;
using yue_host::kr_of;

extern "C" {


// ---------------------------------------------------------------------------------------------
// FISM (recommender/cf/FISM.py), parity path
// ---------------------------------------------------------------------------------------------
int yue_fism_set_model(yue_ctx *c, const double *P, const float *Q, const double *Bi, int64_t n, int k) {
    if (!c || !P || !Q || !Bi) return fail(YUE_ERR_ARG, "yue_fism_set_model: null argument");
    if (n <= 0 || k <= 0 || k > 256 || n >= (1ll << 31)) return fail(YUE_ERR_ARG, "yue_fism_set_model: need 0 < n < 2^31 and 1 <= k <= 256");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(c->fP.resize((size_t)n * k)); HIPCHK(c->fQ.resize((size_t)n * k)); HIPCHK(c->fBi.resize((size_t)n));
    HIPCHK(hipMemcpyAsync(c->fP.p, P, (size_t)n * k * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->fQ.p, Q, (size_t)n * k * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->fBi.p, Bi, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->fn = n; c->fk = k;
    return YUE_OK;
}

int yue_fism_get_model(yue_ctx *c, double *P, float *Q, double *Bi) {
    if (!c || c->fn == 0) return fail(YUE_ERR_ARG, "yue_fism_get_model: no FISM model uploaded");
    HIPCHK(hipSetDevice(c->device));
    if (P) HIPCHK(hipMemcpyAsync(P, c->fP.p, (size_t)c->fn * c->fk * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (Q) HIPCHK(hipMemcpyAsync(Q, c->fQ.p, (size_t)c->fn * c->fk * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (Bi) HIPCHK(hipMemcpyAsync(Bi, c->fBi.p, (size_t)c->fn * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return YUE_OK;
}

namespace {
// uploads a CSR of item rows (user_ptr[rows+1], items) after checking it
int fism_upload_rows(yue_ctx *c, const int64_t *ptr, int64_t rows, const int32_t *items, const char *who) {
    if (rows < 0 || !ptr || ptr[0] != 0) return fail(YUE_ERR_ARG, std::string(who) + ": bad row pointer");
    for (int64_t r = 0; r < rows; ++r) if (ptr[r + 1] < ptr[r]) return fail(YUE_ERR_ARG, std::string(who) + ": row pointer must be non-decreasing");
    const int64_t E = ptr[rows];
    if (E > 0 && !items) return fail(YUE_ERR_ARG, std::string(who) + ": null items");
    for (int64_t e = 0; e < E; ++e) if (items[e] < 0 || items[e] >= c->fn) return fail(YUE_ERR_ARG, std::string(who) + ": item id out of range");
    HIPCHK(c->f_ptr.resize((size_t)rows + 1)); HIPCHK(c->f_items.resize((size_t)std::max<int64_t>(E, 1)));
    HIPCHK(hipMemcpyAsync(c->f_ptr.p, ptr, ((size_t)rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    if (E > 0) HIPCHK(hipMemcpyAsync(c->f_items.p, items, (size_t)E * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    return YUE_OK;
}
}  // namespace

int yue_fism_epoch(yue_ctx *c, const int64_t *user_ptr, int64_t m, const int32_t *ev_i, const int32_t *negs, int64_t n_negs, int rho,
                   const double *coef, double lr, double regI, double regB, double *half_sq_out, double *sumsq3_out) {
    if (!c || c->fn == 0) return fail(YUE_ERR_ARG, "yue_fism_epoch: no FISM model uploaded");
    if (m <= 0 || rho < 1 || !coef || (n_negs > 0 && !negs)) return fail(YUE_ERR_ARG, "yue_fism_epoch: bad argument");
    HIPCHK(hipSetDevice(c->device));
    int rc = fism_upload_rows(c, user_ptr, m, ev_i, "yue_fism_epoch");
    if (rc) return rc;
    int64_t need = 0, widest = 1;
    for (int64_t u = 0; u < m; ++u) { const int64_t nu = user_ptr[u + 1] - user_ptr[u]; if (nu > 1) need += nu * rho; widest = std::max(widest, nu); }
    if (need != n_negs) return fail(YUE_ERR_ARG, "yue_fism_epoch: need rho negatives per event of every user with more than one event (" + std::to_string(need) + "), got " + std::to_string(n_negs));
    for (int64_t t = 0; t < n_negs; ++t) if (negs[t] < 0 || negs[t] >= c->fn) return fail(YUE_ERR_ARG, "yue_fism_epoch: negative item id out of range");
    HIPCHK(c->f_negs.resize((size_t)std::max<int64_t>(n_negs, 1))); HIPCHK(c->f_coef.resize((size_t)m)); HIPCHK(c->f_x.resize((size_t)widest * c->fk));
    if (n_negs > 0) HIPCHK(hipMemcpyAsync(c->f_negs.p, negs, (size_t)n_negs * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->f_coef.p, coef, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    double *sc = c->scal.p + yue_host::kNllSlotsHost;              // [0] half_sq, [1..3] sums of squares
    HIPCHK(hipMemsetAsync(sc, 0, 4 * sizeof(double), c->stream));
    yue::FismArgs a{};
    a.P = c->fP.p; a.Q = c->fQ.p; a.Bi = c->fBi.p; a.n = c->fn; a.k = c->fk;
    a.user_ptr = c->f_ptr.p; a.m = m; a.ev_i = c->f_items.p; a.negs = c->f_negs.p; a.rho = rho; a.coef = c->f_coef.p;
    a.lr = lr; a.regI = regI; a.regB = regB; a.x_rows = c->f_x.p; a.out = sc;
    switch (kr_of(c->fk)) {
        case 1: hipLaunchKernelGGL(yue::k_fism_epoch<1>, dim3(1), dim3(64), 0, c->stream, a); break;
        case 2: hipLaunchKernelGGL(yue::k_fism_epoch<2>, dim3(1), dim3(64), 0, c->stream, a); break;
        default: hipLaunchKernelGGL(yue::k_fism_epoch<4>, dim3(1), dim3(64), 0, c->stream, a); break;
    }
    hipLaunchKernelGGL(yue::k_fism_sumsq, dim3(256), dim3(256), 0, c->stream, c->fP.p, c->fQ.p, c->fBi.p, c->fn, c->fk, sc + 1);
    HIPCHK(hipGetLastError());
    double h[4];
    HIPCHK(hipMemcpyAsync(h, sc, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (half_sq_out) *half_sq_out = h[0];
    if (sumsq3_out) { sumsq3_out[0] = h[1]; sumsq3_out[1] = h[2]; sumsq3_out[2] = h[3]; }
    return YUE_OK;
}

int yue_fism_rounds(yue_ctx *c, const int64_t *user_ptr, int64_t m, const int32_t *ev_i, const int32_t *negs, int64_t n_negs, int rho,
                    const double *coef, int64_t round_users, double lr, double regI, double regB, double *half_sq_out, double *sumsq3_out) {
    if (!c || c->fn == 0) return fail(YUE_ERR_ARG, "yue_fism_rounds: no FISM model uploaded");
    if (m <= 0 || rho < 1 || round_users < 1 || !coef || (n_negs > 0 && !negs)) return fail(YUE_ERR_ARG, "yue_fism_rounds: bad argument");
    HIPCHK(hipSetDevice(c->device));
    int rc = fism_upload_rows(c, user_ptr, m, ev_i, "yue_fism_rounds");
    if (rc) return rc;
    const int64_t E = user_ptr[m];
    // per user: first draw, the sorted unique list of the items it touches, and every event's / draw's position in it
    std::vector<int64_t> neg_ptr((size_t)m + 1, 0), uq_ptr((size_t)m + 1, 0);
    for (int64_t u = 0; u < m; ++u) { const int64_t nu = user_ptr[u + 1] - user_ptr[u]; neg_ptr[(size_t)u + 1] = neg_ptr[(size_t)u] + (nu > 1 ? nu * rho : 0); }
    if (neg_ptr[(size_t)m] != n_negs) return fail(YUE_ERR_ARG, "yue_fism_rounds: need rho negatives per event of every user with more than one event (" + std::to_string(neg_ptr[(size_t)m]) + "), got " + std::to_string(n_negs));
    for (int64_t t = 0; t < n_negs; ++t) if (negs[t] < 0 || negs[t] >= c->fn) return fail(YUE_ERR_ARG, "yue_fism_rounds: negative item id out of range");
    {
        // Fast form (round 3): every user touches at most 64 rows and they fit in LDS -> k_fism_round_lds finds the rows itself;
        // the host prepares nothing per user (the sorted item lists below took most of the call's time).
        int64_t cnt_max = 0, ev_max = 1;
        for (int64_t u = 0; u < m; ++u) { const int64_t nu = user_ptr[u + 1] - user_ptr[u]; if (nu > 1) cnt_max = std::max(cnt_max, nu + nu * rho); }
        for (int64_t u0 = 0; u0 < m; u0 += round_users) ev_max = std::max(ev_max, user_ptr[std::min(m, u0 + round_users)] - user_ptr[u0]);
        const size_t lds = (size_t)cnt_max * ((size_t)c->fk * 12 + 8);
        if (c->opt_fism_lds && cnt_max >= 1 && cnt_max <= 64 && lds <= 160u * 1024u) {      // (gfx950: 160 KB of LDS per CU, all of it one workgroup's if asked for)
            const size_t nk = (size_t)c->fn * c->fk;
            HIPCHK(c->f_negs.resize((size_t)std::max<int64_t>(n_negs, 1))); HIPCHK(c->f_neg_ptr.resize((size_t)m + 1)); HIPCHK(c->f_coef.resize((size_t)m));
            HIPCHK(c->f_x.resize((size_t)ev_max * c->fk));
            HIPCHK(c->f_dQ.resize(nk)); HIPCHK(c->f_dP.resize(nk)); HIPCHK(c->f_dB.resize((size_t)c->fn));
            if (n_negs > 0) HIPCHK(hipMemcpyAsync(c->f_negs.p, negs, (size_t)n_negs * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(c->f_neg_ptr.p, neg_ptr.data(), ((size_t)m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(c->f_coef.p, coef, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemsetAsync(c->f_dQ.p, 0, nk * sizeof(float), c->stream));
            HIPCHK(hipMemsetAsync(c->f_dP.p, 0, nk * sizeof(double), c->stream));
            HIPCHK(hipMemsetAsync(c->f_dB.p, 0, (size_t)c->fn * sizeof(double), c->stream));
            double *sc = c->scal.p + yue_host::kNllSlotsHost;      // [0] half_sq, [1..3] sums of squares
            HIPCHK(hipMemsetAsync(sc, 0, 4 * sizeof(double), c->stream));
            yue::FismArgs a{};
            a.P = c->fP.p; a.Q = c->fQ.p; a.Bi = c->fBi.p; a.n = c->fn; a.k = c->fk;
            a.user_ptr = c->f_ptr.p; a.m = m; a.ev_i = c->f_items.p; a.negs = c->f_negs.p; a.rho = rho; a.coef = c->f_coef.p;
            a.lr = lr; a.regI = regI; a.regB = regB; a.x_rows = c->f_x.p; a.out = sc;
            yue::FismLdsArgs la{};
            la.neg_ptr = c->f_neg_ptr.p; la.dQ = c->f_dQ.p; la.dP = c->f_dP.p; la.dB = c->f_dB.p; la.rows_cap = (int)cnt_max;
            // rows one user of a round has to itself go back in place (option fism_inplace, default 1): two count arrays, swapped per round
            unsigned *cnt[2] = {nullptr, nullptr};
            if (c->opt_fism_inplace) {
                HIPCHK(c->f_cnt.resize(2 * (size_t)c->fn));
                HIPCHK(hipMemsetAsync(c->f_cnt.p, 0, 2 * (size_t)c->fn * sizeof(unsigned), c->stream));
                cnt[0] = c->f_cnt.p; cnt[1] = c->f_cnt.p + c->fn;
                hipLaunchKernelGGL(yue::k_fism_count_round, dim3((unsigned)std::min(m, round_users)), dim3(64), 0, c->stream, a, la.neg_ptr, (int64_t)0, std::min(m, round_users), cnt[0]);
            }
#ifdef YUE_FISM_STAMPS
            unsigned long long *d_stamps = nullptr;
            HIPCHK(hipMalloc(&d_stamps, 8 * sizeof(unsigned long long)));
            HIPCHK(hipMemsetAsync(d_stamps, 0, 8 * sizeof(unsigned long long), c->stream));
            la.stamps = d_stamps;
#endif
            // the round-start rows beside the working rows when both fit in the CU's LDS (a wave has the CU to itself anyway)
            // -- as long as that leaves the round's waves resident at once: a round larger than the CUs can hold with the doubled
            // LDS runs in generations (rounds of 512 users at 93 KB: one workgroup per CU, 130 against 68 us per round of 256)
            const size_t lds_rows = (lds + 7) & ~(size_t)7;
            int cus = 0;
            HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            const bool start = 2 * lds_rows <= 160u * 1024u && std::min(m, round_users) <= (int64_t)cus * (int64_t)((160u * 1024u) / (2 * lds_rows));
            const size_t lds_launch = start ? 2 * lds_rows : lds;
            const void *kfn = nullptr;
            switch (kr_of(c->fk)) {
                case 1: kfn = start ? (const void *)yue::k_fism_round_lds<1, true> : (const void *)yue::k_fism_round_lds<1, false>; break;
                case 2: kfn = start ? (const void *)yue::k_fism_round_lds<2, true> : (const void *)yue::k_fism_round_lds<2, false>; break;
                default: kfn = start ? (const void *)yue::k_fism_round_lds<4, true> : (const void *)yue::k_fism_round_lds<4, false>; break;
            }
            HIPCHK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch));
            const dim3 apply_grid((unsigned)std::min<int64_t>(1024, (int64_t)(nk + 255) / 256));
            for (int64_t u0 = 0; u0 < m; u0 += round_users) {
                const int64_t u1 = std::min(m, u0 + round_users);
                la.u_begin = u0; la.u_end = u1;
                const int par = (int)((u0 / round_users) & 1);
                la.cnt_cur_r = la.cnt_cur = cnt[par]; la.cnt_next = cnt[par ^ 1];
                la.next_begin = u1; la.next_end = std::min(m, u1 + round_users);
                const dim3 grid((unsigned)(u1 - u0));
                void *kargs[2] = {&a, &la};
                HIPCHK(hipLaunchKernel(kfn, grid, dim3(64), kargs, lds_launch, c->stream));
                hipLaunchKernelGGL(yue::k_fism_apply, apply_grid, dim3(256), 0, c->stream, c->fP.p, c->fQ.p, c->fBi.p, c->f_dP.p, c->f_dQ.p, c->f_dB.p, c->fn, c->fk);
            }
            hipLaunchKernelGGL(yue::k_fism_sumsq, dim3(256), dim3(256), 0, c->stream, c->fP.p, c->fQ.p, c->fBi.p, c->fn, c->fk, sc + 1);
            HIPCHK(hipGetLastError());
            double h[4];
            HIPCHK(hipMemcpyAsync(h, sc, sizeof h, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
#ifdef YUE_FISM_STAMPS
            {
                unsigned long long st[8];
                HIPCHK(hipMemcpy(st, d_stamps, sizeof st, hipMemcpyDeviceToHost));
                HIPCHK(hipFree(d_stamps));
                const char *names[6] = {"count + find rows", "copy-in", "history sum", "draws", "item-history rows", "rows back / differences"};
                for (int q = 0; q < 6; ++q) std::fprintf(stderr, "fism stamps: %-24s %8.2f us per wave\n", names[q], st[7] ? 0.01 * (double)st[q] / (double)st[7] : 0.0);
            }
#endif
            if (half_sq_out) *half_sq_out = h[0];
            if (sumsq3_out) { sumsq3_out[0] = h[1]; sumsq3_out[1] = h[2]; sumsq3_out[2] = h[3]; }
            return YUE_OK;
        }
    }
    std::vector<int32_t> uq_items, loc_i((size_t)std::max<int64_t>(E, 1)), loc_j((size_t)std::max<int64_t>(n_negs, 1)), tmp;
    for (int64_t u = 0; u < m; ++u) {
        tmp.assign(ev_i + user_ptr[u], ev_i + user_ptr[u + 1]);
        tmp.insert(tmp.end(), negs + neg_ptr[(size_t)u], negs + neg_ptr[(size_t)u + 1]);
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        for (int64_t e = user_ptr[u]; e < user_ptr[u + 1]; ++e) loc_i[(size_t)e] = (int32_t)(std::lower_bound(tmp.begin(), tmp.end(), ev_i[e]) - tmp.begin());
        for (int64_t t = neg_ptr[(size_t)u]; t < neg_ptr[(size_t)u + 1]; ++t) loc_j[(size_t)t] = (int32_t)(std::lower_bound(tmp.begin(), tmp.end(), negs[t]) - tmp.begin());
        uq_items.insert(uq_items.end(), tmp.begin(), tmp.end());
        uq_ptr[(size_t)u + 1] = (int64_t)uq_items.size();
    }
    int64_t rows_max = 1, ev_max = 1;                          // working rows / events of the largest round
    for (int64_t u0 = 0; u0 < m; u0 += round_users) {
        const int64_t u1 = std::min(m, u0 + round_users);
        rows_max = std::max(rows_max, uq_ptr[(size_t)u1] - uq_ptr[(size_t)u0]);
        ev_max = std::max(ev_max, user_ptr[u1] - user_ptr[u0]);
    }
    const size_t nk = (size_t)c->fn * c->fk;
    HIPCHK(c->f_uq_ptr.resize((size_t)m + 1)); HIPCHK(c->f_neg_ptr.resize((size_t)m + 1)); HIPCHK(c->f_uq_items.resize(std::max<size_t>(uq_items.size(), 1)));
    HIPCHK(c->f_loc_i.resize(loc_i.size())); HIPCHK(c->f_loc_j.resize(loc_j.size())); HIPCHK(c->f_coef.resize((size_t)m));
    HIPCHK(c->f_wq.resize((size_t)rows_max * c->fk)); HIPCHK(c->f_wp.resize((size_t)rows_max * c->fk)); HIPCHK(c->f_wb.resize((size_t)rows_max));
    HIPCHK(c->f_x.resize((size_t)ev_max * c->fk));
    HIPCHK(c->f_dQ.resize(nk)); HIPCHK(c->f_dP.resize(nk)); HIPCHK(c->f_dB.resize((size_t)c->fn));
    HIPCHK(hipMemcpyAsync(c->f_uq_ptr.p, uq_ptr.data(), ((size_t)m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->f_neg_ptr.p, neg_ptr.data(), ((size_t)m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    if (!uq_items.empty()) HIPCHK(hipMemcpyAsync(c->f_uq_items.p, uq_items.data(), uq_items.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    if (E > 0) HIPCHK(hipMemcpyAsync(c->f_loc_i.p, loc_i.data(), (size_t)E * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    if (n_negs > 0) HIPCHK(hipMemcpyAsync(c->f_loc_j.p, loc_j.data(), (size_t)n_negs * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->f_coef.p, coef, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->f_dQ.p, 0, nk * sizeof(float), c->stream));
    HIPCHK(hipMemsetAsync(c->f_dP.p, 0, nk * sizeof(double), c->stream));
    HIPCHK(hipMemsetAsync(c->f_dB.p, 0, (size_t)c->fn * sizeof(double), c->stream));
    double *sc = c->scal.p + yue_host::kNllSlotsHost;              // [0] half_sq, [1..3] sums of squares
    HIPCHK(hipMemsetAsync(sc, 0, 4 * sizeof(double), c->stream));
    yue::FismArgs a{};
    a.P = c->fP.p; a.Q = c->fQ.p; a.Bi = c->fBi.p; a.n = c->fn; a.k = c->fk;
    a.user_ptr = c->f_ptr.p; a.m = m; a.ev_i = c->f_items.p; a.negs = nullptr; a.rho = rho; a.coef = c->f_coef.p;
    a.lr = lr; a.regI = regI; a.regB = regB; a.x_rows = c->f_x.p; a.out = sc;
    yue::FismRoundArgs ra{};
    ra.uq_ptr = c->f_uq_ptr.p; ra.uq_items = c->f_uq_items.p; ra.loc_i = c->f_loc_i.p; ra.loc_j = c->f_loc_j.p; ra.neg_ptr = c->f_neg_ptr.p;
    ra.wq = c->f_wq.p; ra.wp = c->f_wp.p; ra.wb = c->f_wb.p; ra.dQ = c->f_dQ.p; ra.dP = c->f_dP.p; ra.dB = c->f_dB.p;
    const dim3 apply_grid((unsigned)std::min<int64_t>(1024, (int64_t)(nk + 255) / 256));
    for (int64_t u0 = 0; u0 < m; u0 += round_users) {
        const int64_t u1 = std::min(m, u0 + round_users);
        ra.u_begin = u0; ra.u_end = u1; ra.w_base = uq_ptr[(size_t)u0];
        const dim3 grid((unsigned)((u1 - u0 + 3) / 4));
        switch (kr_of(c->fk)) {
            case 1: hipLaunchKernelGGL(yue::k_fism_round<1>, grid, dim3(256), 0, c->stream, a, ra); break;
            case 2: hipLaunchKernelGGL(yue::k_fism_round<2>, grid, dim3(256), 0, c->stream, a, ra); break;
            default: hipLaunchKernelGGL(yue::k_fism_round<4>, grid, dim3(256), 0, c->stream, a, ra); break;
        }
        hipLaunchKernelGGL(yue::k_fism_apply, apply_grid, dim3(256), 0, c->stream, c->fP.p, c->fQ.p, c->fBi.p, c->f_dP.p, c->f_dQ.p, c->f_dB.p, c->fn, c->fk);
    }
    hipLaunchKernelGGL(yue::k_fism_sumsq, dim3(256), dim3(256), 0, c->stream, c->fP.p, c->fQ.p, c->fBi.p, c->fn, c->fk, sc + 1);
    HIPCHK(hipGetLastError());
    double h[4];
    HIPCHK(hipMemcpyAsync(h, sc, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (half_sq_out) *half_sq_out = h[0];
    if (sumsq3_out) { sumsq3_out[0] = h[1]; sumsq3_out[1] = h[2]; sumsq3_out[2] = h[3]; }
    return YUE_OK;
}

namespace {
// scores[nu][n] of the users whose training rows were uploaded with fism_upload_rows
int fism_score_rows(yue_ctx *c, int64_t nu) {
    HIPCHK(c->f_hist.resize((size_t)nu * c->fk)); HIPCHK(c->f_scores.resize((size_t)nu * c->fn));
    hipLaunchKernelGGL(yue::k_fism_hist, dim3((unsigned)nu), dim3(256), 0, c->stream, c->fP.p, c->fk, c->f_ptr.p, c->f_items.p, c->f_hist.p);
    const int64_t tot = nu * c->fn;
    hipLaunchKernelGGL(yue::k_fism_scores, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                       c->fP.p, c->fQ.p, c->fBi.p, c->fn, c->fk, c->f_hist.p, nu, c->f_scores.p);
    HIPCHK(hipGetLastError());
    return YUE_OK;
}
}  // namespace

int yue_fism_scores(yue_ctx *c, const int32_t *items, int64_t n_items, double *out_n) {
    if (!c || c->fn == 0 || !out_n) return fail(YUE_ERR_ARG, "yue_fism_scores: no FISM model uploaded");
    HIPCHK(hipSetDevice(c->device));
    const int64_t ptr[2] = {0, n_items};
    int rc = fism_upload_rows(c, ptr, 1, items, "yue_fism_scores");
    if (rc) return rc;
    if ((rc = fism_score_rows(c, 1))) return rc;
    HIPCHK(hipMemcpyAsync(out_n, c->f_scores.p, (size_t)c->fn * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return YUE_OK;
}

int yue_fism_topn_scan(yue_ctx *c, const int64_t *row_ptr, const int32_t *row_items, int64_t nu, int N, int32_t *out_ids, double *out_scores) {
    if (!c || c->fn == 0) return fail(YUE_ERR_ARG, "yue_fism_topn_scan: no FISM model uploaded");
    if (nu < 0 || (nu > 0 && (!out_ids || !out_scores))) return fail(YUE_ERR_ARG, "yue_fism_topn_scan: null argument");
    if (N < 1 || N > 100) return fail(YUE_ERR_ARG, "yue_fism_topn_scan: N must be in 1..100");
    if (nu == 0) return YUE_OK;
    if (nu * c->fn >= (1ll << 31)) return fail(YUE_ERR_ARG, "yue_fism_topn_scan: users x items of one call must stay below 2^31 (call in chunks)");
    HIPCHK(hipSetDevice(c->device));
    int rc = fism_upload_rows(c, row_ptr, nu, row_items, "yue_fism_topn_scan");
    if (rc) return rc;
    if ((rc = fism_score_rows(c, nu))) return rc;
    HIPCHK(c->f_ids.resize((size_t)nu * N)); HIPCHK(c->f_out_sc.resize((size_t)nu * N)); HIPCHK(c->f_flags.resize((size_t)nu));
    hipLaunchKernelGGL(yue::k_fism_select, dim3((unsigned)((nu + 63) / 64)), dim3(64), 0, c->stream, c->f_scores.p, c->fn, nu, N,
                       c->f_ptr.p, c->f_items.p, c->f_ids.p, c->f_out_sc.p, c->f_flags.p);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> flags((size_t)nu);
    HIPCHK(hipMemcpyAsync(out_ids, c->f_ids.p, (size_t)nu * N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(out_scores, c->f_out_sc.p, (size_t)nu * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(flags.data(), c->f_flags.p, (size_t)nu * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int64_t t = 0; t < nu; ++t) if (flags[(size_t)t]) return fail(YUE_ERR_FEW_ITEMS, "yue_fism_topn_scan: user at position " + std::to_string(t) + " has fewer than N candidates");
    return YUE_OK;
}

}  // extern "C"
