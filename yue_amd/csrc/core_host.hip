// libyue_hip.so -- context, factor matrices and interactions (include/yue_hip.h).
#include "host_common.hpp"

namespace yue_host {
thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }
}  // namespace yue_host
using yue_host::fail;
void yue_comm_release(yue_ctx *c);       // comm.hip

extern "C" {

const char *yue_last_error(void) { return yue_host::g_err.c_str(); }
int yue_version(void) { return 1; }

int yue_ctx_create(int device, yue_ctx **out) {
    if (!out) return fail(YUE_ERR_ARG, "yue_ctx_create: out is NULL");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(YUE_ERR_HIP, "no HIP device visible: libyue_hip needs an MI355X (no CPU fallback exists)");
    if (device < 0 || device >= ndev) return fail(YUE_ERR_ARG, "device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    yue_ctx *c = new yue_ctx();
    c->device = device;
    const auto init = [c]() -> int {
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_rounds, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming));
        HIPCHK(hipEventCreate(&c->ev_t_rounds));
        HIPCHK(hipEventCreate(&c->ev_t_comm));
        HIPCHK(hipEventCreate(&c->ev_scan0));
        HIPCHK(hipEventCreate(&c->ev_scan1));
        HIPCHK(c->scal.resize(yue_host::kNllSlotsHost + 8));
        return YUE_OK;
    };
    const int rc = init();
    if (rc) { const std::string msg = yue_host::g_err; (void)yue_ctx_destroy(c); yue_host::g_err = msg; return rc; }     // nothing of a half-built context survives
    *out = c;
    return YUE_OK;
}

int yue_ctx_destroy(yue_ctx *c) {
    if (!c) return YUE_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    yue_comm_release(c);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    if (c->ev_rounds) (void)hipEventDestroy(c->ev_rounds);
    if (c->ev_comm) (void)hipEventDestroy(c->ev_comm);
    if (c->ev_t_rounds) (void)hipEventDestroy(c->ev_t_rounds);
    if (c->ev_t_comm) (void)hipEventDestroy(c->ev_t_comm);
    if (c->ev_scan0) (void)hipEventDestroy(c->ev_scan0);
    if (c->ev_chain0) (void)hipEventDestroy(c->ev_chain0);
    if (c->ev_chain1) (void)hipEventDestroy(c->ev_chain1);
    if (c->ev_scan1) (void)hipEventDestroy(c->ev_scan1);
    for (auto &pr : c->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    c->P.release(); c->Q.release(); c->dP.release(); c->dQ.release(); c->margins.release();
    c->cnt0.release(); c->cnt1.release(); c->cntp0.release(); c->cntp1.release();
    c->tab0.release(); c->tab1.release();
    c->meta_i.release(); c->meta_j.release(); c->round_rows.release(); c->d_bounds.release(); c->fold.release(); c->bk_touch.release(); c->bk_ptr.release();
    c->ev_u.release(); c->ev_i.release(); c->ev_j.release(); c->indices.release(); c->indptr.release();
    c->xu.release(); c->xi.release(); c->xj.release(); c->xk.release(); c->x_loss.release(); c->scal.release();
    c->aU_m.release(); c->aU_v.release(); c->aV_m.release(); c->aV_v.release();
    c->s_users.release(); c->s_ids.release(); c->s_mask_idx.release(); c->s_flags.release();
    c->s_mask_ptr.release(); c->s_scores.release(); c->s_row.release(); c->s_norms.release(); c->s_work.release(); c->s_qb.release(); c->s_masks.release(); c->s_few.release(); c->s_few_ids.release(); c->s_few_scores.release();
    c->fP.release(); c->fBi.release(); c->f_coef.release(); c->f_x.release(); c->f_hist.release(); c->f_scores.release();
    c->f_out_sc.release(); c->fQ.release(); c->f_ptr.release(); c->f_items.release(); c->f_negs.release(); c->f_ids.release(); c->f_flags.release();
    c->f_uq_ptr.release(); c->f_neg_ptr.release(); c->f_uq_items.release(); c->f_loc_i.release(); c->f_loc_j.release();
    c->f_wq.release(); c->f_dQ.release(); c->f_wp.release(); c->f_wb.release(); c->f_dP.release(); c->f_dB.release();
    c->ch_key.release(); c->ch_val.release(); c->ch_key2.release(); c->ch_val2.release(); c->ch_seg.release(); c->ch_ord_i.release(); c->ch_ord_j.release();
    c->ch_head.release(); c->ch_incl.release(); c->ch_ord_u.release(); c->ch_rkey.release(); c->ch_rval.release(); c->ch_run_ptr.release(); c->d_ev_ptr.release();
    c->ch_run_u.release(); c->ch_tmp.release(); c->ch_Qv.release(); c->ch_Pv.release(); c->ch_ctl.release(); c->ch_stats.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return YUE_OK;
}

int yue_sync(yue_ctx *c) {
    if (!c) return fail(YUE_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return YUE_OK;
}

int yue_set_factors(yue_ctx *c, const float *P, int64_t m, const float *Q, int64_t n, int k) {
    if (!c || !P || !Q) return fail(YUE_ERR_ARG, "yue_set_factors: null argument");
    if (m <= 0 || n <= 0 || k <= 0 || k > 256) return fail(YUE_ERR_ARG, "yue_set_factors: need m,n > 0 and 1 <= k <= 256");
    if (n >= (1ll << 31) || m >= (1ll << 31)) return fail(YUE_ERR_ARG, "yue_set_factors: ids must fit int32");
    // (item matrices of 2 GiB and more: yue_bpr_epoch's default path and the scoring calls take them -- 64-bit row addressing,
    // round_kernels.hpp: BIGQ --; explicit rounds, the exact path and round_meta = 0 say so and refuse)
    HIPCHK(hipSetDevice(c->device));
    if (c->have_inter && (m != c->m || n != c->n || k != c->k)) c->have_inter = false;   // new shape (k enters the offset bounds checked by yue_set_interactions): upload the interactions again
    if (k != c->k) c->opt_round_tpw = 0;                   // the events-per-wave option was validated against the old k
    c->m = m; c->n = n; c->k = k;
    HIPCHK(c->P.resize(m * k)); HIPCHK(c->Q.resize(n * k));
    HIPCHK(c->dP.resize(m * k)); HIPCHK(c->dQ.resize(n * k));
    HIPCHK(c->cnt0.resize(n)); HIPCHK(c->cnt1.resize(n)); HIPCHK(c->cntp0.resize(m)); HIPCHK(c->cntp1.resize(m));
    HIPCHK(hipMemcpyAsync(c->P.p, P, m * k * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->Q.p, Q, n * k * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->dP.p, 0, m * k * sizeof(float), c->stream));
    HIPCHK(hipMemsetAsync(c->dQ.p, 0, n * k * sizeof(float), c->stream));
    HIPCHK(hipMemsetAsync(c->cntp0.p, 0, m * sizeof(uint32_t), c->stream));
    HIPCHK(hipMemsetAsync(c->cntp1.p, 0, m * sizeof(uint32_t), c->stream));
    HIPCHK(hipMemsetAsync(c->cnt0.p, 0, n * sizeof(unsigned long long), c->stream));
    HIPCHK(hipMemsetAsync(c->cnt1.p, 0, n * sizeof(unsigned long long), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_factors = true;
    return YUE_OK;
}

int yue_get_factors(yue_ctx *c, float *P, float *Q) {
    if (!c || !c->have_factors) return fail(YUE_ERR_ARG, "yue_get_factors: no factors uploaded");
    HIPCHK(hipSetDevice(c->device));
    if (P) HIPCHK(hipMemcpyAsync(P, c->P.p, c->m * c->k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (Q) HIPCHK(hipMemcpyAsync(Q, c->Q.p, c->n * c->k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return YUE_OK;
}

int yue_set_interactions(yue_ctx *c, const int64_t *indptr, const int32_t *indices, const int64_t *ev_ptr, const int32_t *ev_i) {
    if (!c || !indptr || !indices || !ev_ptr || !ev_i) return fail(YUE_ERR_ARG, "yue_set_interactions: null argument");
    if (!c->have_factors) return fail(YUE_ERR_ARG, "yue_set_interactions: call yue_set_factors first (m, n)");
    HIPCHK(hipSetDevice(c->device));
    const int64_t m = c->m, n = c->n;
    if (indptr[0] != 0 || ev_ptr[0] != 0) return fail(YUE_ERR_ARG, "yue_set_interactions: indptr[0] and ev_ptr[0] must be 0");
    for (int64_t u = 0; u < m; ++u) {
        if (indptr[u + 1] < indptr[u] || ev_ptr[u + 1] < ev_ptr[u]) return fail(YUE_ERR_ARG, "yue_set_interactions: offsets must be non-decreasing");
        for (int64_t t = indptr[u]; t < indptr[u + 1]; ++t) {
            if (indices[t] < 0 || indices[t] >= n) return fail(YUE_ERR_ARG, "yue_set_interactions: item id out of range");
            if (t > indptr[u] && indices[t] <= indices[t - 1]) return fail(YUE_ERR_ARG, "yue_set_interactions: rows must be sorted and unique");
        }
        if (indptr[u + 1] - indptr[u] >= n) return fail(YUE_ERR_ARG, "yue_set_interactions: a user listened to every item (the reference's sampler would never return, BPR.py:47)");
    }
    const int64_t nnz = indptr[m], E = ev_ptr[m];
    {   // the update kernel addresses P relative to a batch's first user with 31-bit byte offsets
        int64_t prev = -1, max_gap = 0;
        for (int64_t u = 0; u < m; ++u) if (ev_ptr[u + 1] > ev_ptr[u]) { if (prev >= 0) max_gap = std::max(max_gap, u - prev); prev = u; }
        if (max_gap * c->k * 4 >= (1ll << 31)) return fail(YUE_ERR_ARG, "yue_set_interactions: more than 2 GiB of user-factor rows between two consecutive users with events");
    }
    std::vector<int32_t> evu((size_t)E);
    for (int64_t u = 0; u < m; ++u)
        for (int64_t e = ev_ptr[u]; e < ev_ptr[u + 1]; ++e) {
            if (ev_i[e] < 0 || ev_i[e] >= n) return fail(YUE_ERR_ARG, "yue_set_interactions: event item out of range");
            evu[(size_t)e] = (int32_t)u;
        }
    c->E = E; c->nnz = nnz;
    c->h_ev_ptr.assign(ev_ptr, ev_ptr + m + 1);
    c->d_ev_ptr_valid = false;
    HIPCHK(c->indptr.resize(m + 1)); HIPCHK(c->indices.resize(std::max<int64_t>(nnz, 1)));
    HIPCHK(c->ev_u.resize((size_t)E + yue_host::kHeaderSlackHost)); HIPCHK(c->ev_i.resize((size_t)E + yue_host::kHeaderSlackHost)); HIPCHK(c->ev_j.resize((size_t)E + yue_host::kHeaderSlackHost));   // (+ slack: k_round_m reads whole header blocks)
    HIPCHK(hipMemcpyAsync(c->indptr.p, indptr, (m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->indices.p, indices, nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->ev_u.p, evu.data(), E * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->ev_i.p, ev_i, E * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_inter = true;
    return YUE_OK;
}


}  // extern "C"
