// libyue_hip.so -- predict and evalRanking's selection (include/yue_hip.h).
#include "host_common.hpp"

#include "score_kernels.hpp"

using yue_host::fail;

extern "C" {


int yue_scores(yue_ctx *c, int32_t user, float *out_n) {
    if (!c || !c->have_factors || !out_n) return fail(YUE_ERR_ARG, "yue_scores: no factors uploaded");
    if (user < 0 || user >= c->m) return fail(YUE_ERR_ARG, "yue_scores: user id out of range");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(c->s_row.resize(c->n));
    hipLaunchKernelGGL(yue::k_scores_one, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream,
                       c->P.p + (int64_t)user * c->k, c->Q.p, c->n, c->k, c->s_row.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_n, c->s_row.p, c->n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return YUE_OK;
}

int yue_topn_scan(yue_ctx *c, const int32_t *users, int64_t nu, int N, const int64_t *mask_indptr, const int32_t *mask_indices,
                  int32_t *out_ids, float *out_scores) {
    if (!c || !c->have_factors) return fail(YUE_ERR_ARG, "yue_topn_scan: no factors uploaded");
    if (nu < 0 || (nu > 0 && (!users || !out_ids || !out_scores))) return fail(YUE_ERR_ARG, "yue_topn_scan: null argument");
    if (N < 1 || N > 100) return fail(YUE_ERR_ARG, "yue_topn_scan: N must be in 1..100");
    if ((mask_indptr == nullptr) != (mask_indices == nullptr)) return fail(YUE_ERR_ARG, "yue_topn_scan: pass both mask arrays or neither");
    if (!mask_indptr && !c->have_inter) return fail(YUE_ERR_ARG, "yue_topn_scan: no mask given and no interactions uploaded");
    if (nu == 0) return YUE_OK;
    HIPCHK(hipSetDevice(c->device));
    for (int64_t t = 0; t < nu; ++t) if (users[t] < 0 || users[t] >= c->m) return fail(YUE_ERR_ARG, "yue_topn_scan: user id out of range");
    HIPCHK(c->s_users.resize(nu)); HIPCHK(c->s_ids.resize(nu * N)); HIPCHK(c->s_scores.resize(nu * N)); HIPCHK(c->s_flags.resize(4)); HIPCHK(c->s_work.resize(4));
    HIPCHK(hipMemcpyAsync(c->s_users.p, users, nu * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    yue::ScanArgs sa{};
    sa.P = c->P.p; sa.Q = c->Q.p; sa.n = c->n; sa.k = c->k; sa.users = c->s_users.p; sa.nu = nu; sa.N = N;
    sa.out_ids = c->s_ids.p; sa.out_scores = c->s_scores.p; sa.flags = c->s_flags.p;
    sa.true_topn = c->opt_topn_true;
    if (mask_indptr) {
        const int64_t mnnz = mask_indptr[nu];
        for (int64_t t = 0; t < nu; ++t) {
            if (mask_indptr[t + 1] < mask_indptr[t]) return fail(YUE_ERR_ARG, "yue_topn_scan: mask_indptr must be non-decreasing");
            for (int64_t q = mask_indptr[t]; q < mask_indptr[t + 1]; ++q)
                if (mask_indices[q] < 0 || mask_indices[q] >= c->n || (q > mask_indptr[t] && mask_indices[q] <= mask_indices[q - 1]))
                    return fail(YUE_ERR_ARG, "yue_topn_scan: mask rows must be sorted, unique and in range");
        }
        HIPCHK(c->s_mask_ptr.resize(nu + 1)); HIPCHK(c->s_mask_idx.resize(std::max<int64_t>(mnnz, 1)));
        HIPCHK(hipMemcpyAsync(c->s_mask_ptr.p, mask_indptr, (nu + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->s_mask_idx.p, mask_indices, mnnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        sa.mask_ptr = c->s_mask_ptr.p; sa.mask_idx = c->s_mask_idx.p; sa.mask_by_user = 0;
    } else {
        sa.mask_ptr = c->indptr.p; sa.mask_idx = c->indices.p; sa.mask_by_user = 1;
    }
    HIPCHK(hipMemsetAsync(c->s_flags.p, 0, 4 * sizeof(int32_t), c->stream));
    HIPCHK(hipMemsetAsync(c->s_work.p, 0, 4 * sizeof(unsigned long long), c->stream));
    sa.work = c->s_work.p;
    const int64_t ntile = (c->n + 31) / 32;
    HIPCHK(c->s_norms.resize(2 * ntile));
    hipLaunchKernelGGL(yue::k_tile_norm_max, dim3((unsigned)((ntile + 3) / 4)), dim3(256), 0, c->stream, c->Q.p, c->n, c->k, c->s_norms.p);
    hipLaunchKernelGGL(yue::k_tile_norm_sufmax, dim3(1), dim3(64), 0, c->stream, c->s_norms.p, ntile, c->s_norms.p + ntile);
    sa.tile_norm_max = c->s_norms.p;
    sa.tile_norm_sufmax = c->s_norms.p + ntile;
    const hipEvent_t t0 = c->ev_scan0, t1 = c->ev_scan1;
    HIPCHK(hipEventRecord(t0, c->stream));
    int rc = yue::launch_scan(sa, c->stream, c->opt_scan_f32, c->opt_scan_batch);
    HIPCHK(hipEventRecord(t1, c->stream));
    if (rc < 0) return fail(YUE_ERR_ARG, "yue_topn_scan: unsupported (k, N) combination (k <= 256; for k > 128 the list length N is limited to 66)");
    c->scan_used_bf16 = rc;
    HIPCHK(hipGetLastError());
    int32_t flags[4];
    HIPCHK(hipMemcpyAsync(out_ids, c->s_ids.p, nu * N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(out_scores, c->s_scores.p, nu * N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(flags, c->s_flags.p, sizeof flags, hipMemcpyDeviceToHost, c->stream));
    unsigned long long work[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(work, c->s_work.p, sizeof work, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, t0, t1));
    c->scan_ms = ms;
    c->scan_events = (int64_t)work[1];
    c->scan_rescored = (int64_t)work[2];
    c->scan_tiles_done = (int64_t)work[0];
    c->scan_tiles_total = ((nu + 31) / 32) * ntile;
    if (flags[0]) return fail(YUE_ERR_FEW_ITEMS, "a user has fewer than N candidate items (the reference raises IndexError, base/IterativeRecommender.py:126)");
    return YUE_OK;
}

}  // extern "C"
