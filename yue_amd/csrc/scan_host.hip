// libyue_hip.so -- predict and evalRanking's selection (include/yue_hip.h).
#include "host_common.hpp"

#include "score2_kernels.hpp"

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
    HIPCHK(c->s_users.resize(nu)); HIPCHK(c->s_ids.resize(nu * N)); HIPCHK(c->s_scores.resize(nu * N)); HIPCHK(c->s_flags.resize(4)); HIPCHK(c->s_work.resize(4 * yue::kWorkSlots));
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
    HIPCHK(hipMemsetAsync(c->s_work.p, 0, 4 * yue::kWorkSlots * sizeof(unsigned long long), c->stream));
    sa.work = c->s_work.p;
    const int64_t ntile = (c->n + 31) / 32;
    HIPCHK(c->s_norms.resize(2 * ntile));
    hipLaunchKernelGGL(yue::k_tile_norm_max, dim3((unsigned)((ntile + 3) / 4)), dim3(256), 0, c->stream, c->Q.p, c->n, c->k, c->s_norms.p);
    hipLaunchKernelGGL(yue::k_tile_norm_sufmax, dim3(1), dim3(64), 0, c->stream, c->s_norms.p, ntile, c->s_norms.p + ntile);
    sa.tile_norm_max = c->s_norms.p;
    sa.tile_norm_sufmax = c->s_norms.p + ntile;
    const hipEvent_t t0 = c->ev_scan0, t1 = c->ev_scan1;
    HIPCHK(hipEventRecord(t0, c->stream));
    int rc = 0;
    c->scan_chunks = 0;
    c->scan_few_users = 0;
    // Long catalogues: the first chunk through the fused kernel (it seeds the lists), the rest in chunks of doubling size
    // through k_scan_filter / k_scan_select (score2_kernels.hpp).  Same lists and scores by construction.
    constexpr int64_t kChunk0 = 512;       // (the fused kernel is slow where the thresholds still move fast)
    const bool two_phase = c->opt_scan_two_phase && !c->opt_scan_f32 && (c->k == 16 || c->k == 32 || c->k == 64 || c->k == 128) && N <= 64 &&
                           c->n >= 16384 && yue::scan_bf16p_lds_bytes(c->k, N, 2, 8) <= 160u * 1024u;
    bool done = false;
    if (two_phase) {
        yue::ScanArgs s0 = sa;
        s0.n = kChunk0;
        rc = yue::launch_scan(s0, c->stream, 0, 0);
        int32_t few = 0;
        std::vector<unsigned long long> w0(4 * yue::kWorkSlots, 0ull);
        // (flags[2..3] are free during the scan: sampled / settled users for the choice of the filter variant below)
        unsigned *settle_counts = reinterpret_cast<unsigned *>(c->s_flags.p + 2);
        constexpr int kSettleStep = 16;
        const int64_t settle_item = std::min<int64_t>(c->n - 1, 8 * kChunk0);
        hipLaunchKernelGGL(yue::k_scan_settled_sample, dim3((unsigned)(((nu + kSettleStep - 1) / kSettleStep + 63) / 64)), dim3(256), 0, c->stream,
                           c->P.p, c->s_users.p, nu, c->k, N, c->s_scores.p, c->s_norms.p + ntile, settle_item, kSettleStep, settle_counts);
        unsigned settle_host[2] = {0u, 0u};
        HIPCHK(hipMemcpyAsync(settle_host, settle_counts, sizeof settle_host, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&few, c->s_flags.p, sizeof few, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(w0.data(), c->s_work.p, w0.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        unsigned long long ev0 = 0;
        for (int q = 0; q < yue::kWorkSlots; ++q) ev0 += w0[(size_t)4 * q + 1];
        // thresholds that still move fast (many list updates among the first items: e.g. untrained factors) want short chunks
        // Most users settled against the catalogue's tail already (norm bounds: factors of a few epochs)?  Then the filter's
        // workgroups check for that and stop (k_scan_filter<.., SETTLE>: 41 -> 26 ms per 1M users after 6 epochs); where nothing
        // settles the check would only cost (2 % after 25 epochs).
        const bool settle = 2u * settle_host[1] > settle_host[0];
        c->scan_settle = settle ? 1 : 0;
        const int growth = c->opt_scan_growth > 0 ? c->opt_scan_growth : ((double)ev0 / (double)nu > 4.0 * N ? 2 : 8);
        // Some users have fewer than N unmasked candidates among the first items (a heavy listener of the catalogue's head): with the
        // training CSR as the mask they alone go through the fused kernel over all items afterwards, everybody else through the
        // filter / select pair as usual.  (An explicit mask is indexed by position in users[]: there the whole call falls back.)
        int64_t nfew = 0;
        if (few && sa.mask_by_user) {
            HIPCHK(c->s_few.resize((size_t)(2 * nu + 2)));
            unsigned *cnt = reinterpret_cast<unsigned *>(c->s_few.p + 2 * nu);
            HIPCHK(hipMemsetAsync(cnt, 0, sizeof(unsigned), c->stream));
            hipLaunchKernelGGL(yue::k_scan_collect_few, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, c->stream, c->s_ids.p, c->s_scores.p, c->s_users.p, nu, N,
                               c->s_few.p, c->s_few.p + nu, cnt);
            unsigned h = 0;
            HIPCHK(hipMemcpyAsync(&h, cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemsetAsync(c->s_flags.p, 0, sizeof(int32_t), c->stream));          // (the fused pass over these users raises it again if they really have too few)
            HIPCHK(hipStreamSynchronize(c->stream));
            nfew = h;
        }
        c->scan_few_users = nfew;
        if (few && (!sa.mask_by_user || nfew > std::max<int64_t>(64, nu / 8))) {      // explicit masks, or too many such users: the fused kernel walks everything
            nfew = 0;
            c->scan_few_users = 0;
            HIPCHK(hipMemsetAsync(c->s_flags.p, 0, 4 * sizeof(int32_t), c->stream));
            HIPCHK(hipMemsetAsync(c->s_work.p, 0, 4 * yue::kWorkSlots * sizeof(unsigned long long), c->stream));
        } else {
            const int k = c->k;
            const int64_t npad = (c->n + 63) / 64 * 64;
            HIPCHK(c->s_qb.resize((size_t)npad * k));
            hipLaunchKernelGGL(yue::k_q_to_bf16, dim3(2048), dim3(256), 0, c->stream, c->Q.p, reinterpret_cast<__bf16 *>(c->s_qb.p), c->n * (int64_t)k, npad * (int64_t)k);
            int cus = 0;
            HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            // chunk boundaries and the widest chunk
            std::vector<int64_t> cb{kChunk0};
            while (cb.back() < c->n) cb.push_back(std::min<int64_t>(c->n, cb.back() * growth));
            int64_t stride = 2;
            for (size_t q = 0; q + 1 < cb.size(); ++q) stride = std::max<int64_t>(stride, ((cb[q + 1] - cb[q] + 127) / 128) * 4);
            // users in slabs, so that the survivor words of a chunk stay below 12 GB -- and, for many users, at least four slabs that
            // alternate between two streams (each with its own survivor words): the selection of one slab (row gathers, HBM-bound)
            // runs beside the filter of the next (MFMA-bound); a slab's own chunks stay in order on its stream
            const int lanes = (c->opt_scan_streams >= 2 && nu >= c->opt_scan_streams_min_users) ? 2 : 1;
            int64_t slab = std::max<int64_t>(256, std::min<int64_t>(nu, ((12ll << 30) / (stride * 4)) / 256 * 256));
            if (lanes == 2) slab = std::min<int64_t>(slab, ((nu + c->opt_scan_slabs - 1) / c->opt_scan_slabs + 255) / 256 * 256);
            const int64_t sstride = (stride + 31) / 32 + 1;
            const int64_t lane_words = std::min(slab, nu) * (stride + sstride);
            HIPCHK(c->s_masks.resize((size_t)(lanes * lane_words)));
            yue::FilterArgs fa{};
            fa.P = c->P.p; fa.Qb = reinterpret_cast<const __bf16 *>(c->s_qb.p); fa.N = N; fa.tile_norm_max = c->s_norms.p; fa.tile_norm_sufmax = c->s_norms.p + ntile; fa.masks = c->s_masks.p;
            fa.mask_stride = stride; fa.work = c->s_work.p; fa.n = c->n; fa.sum_stride = sstride;
            yue::SelectArgs xa{};
            xa.P = c->P.p; xa.Q = c->Q.p; xa.n = c->n; xa.k = k; xa.N = N; xa.mask_ptr = sa.mask_ptr; xa.mask_idx = sa.mask_idx; xa.mask_by_user = sa.mask_by_user;
            xa.mask_stride = stride; xa.sum_stride = sstride; xa.work = c->s_work.p; xa.true_topn = c->opt_topn_true;
            constexpr int kSelWaves = 2;
            const size_t sel_lds = yue::select_lds_bytes(kSelWaves);
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&yue::k_scan_select<kSelWaves>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds));
            if (lanes == 2) { HIPCHK(hipEventRecord(c->ev_rounds, c->stream)); HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_rounds, 0)); }
            int64_t slab_ix = 0;
            for (int64_t u0 = 0; u0 < nu; u0 += slab, ++slab_ix) {
                const int64_t un = std::min(slab, nu - u0);
                const int ln = lanes == 2 ? (int)(slab_ix & 1) : 0;
                hipStream_t st = ln ? c->comm_stream : c->stream;
                xa.masks = fa.masks = c->s_masks.p + ln * lane_words;
                xa.summary = fa.summary = c->s_masks.p + ln * lane_words + std::min(slab, nu) * stride;
                fa.users = c->s_users.p + u0; fa.nu = un; fa.thr_rows = c->s_scores.p + u0 * N;
                xa.users = c->s_users.p + u0; xa.nu = un; xa.ids = c->s_ids.p + u0 * N; xa.scores = c->s_scores.p + u0 * N;
                // (an explicit mask is indexed by position in users[]: shift its row pointer with the slab)
                if (!sa.mask_by_user) xa.mask_ptr = sa.mask_ptr + u0;
                for (size_t q = 0; q + 1 < cb.size(); ++q) {
                    fa.item0 = xa.item0 = cb[q]; fa.item1 = xa.item1 = cb[q + 1];
                    // scan_filter_ub: 3 = two blocks of 32 users per wave, 4 waves per workgroup, item rows by DMA into LDS (k = 64 / 128; other k: as 2);
                    // 2 = the same blocking with rows staged through registers; 1 = one block per wave, 8 waves
                    const int fmode = c->opt_scan_filter_ub == 3 && k != 64 && k != 128 ? 2 : c->opt_scan_filter_ub;
                    const int kUB = fmode == 1 ? 1 : 2, kFW = kUB == 2 ? 4 : 8;
                    const int64_t ublocks = (un + 32 * kFW * kUB - 1) / (32 * kFW * kUB), iters = (cb[q + 1] - cb[q] + 63) / 64;
                    const int64_t splits = std::max<int64_t>(1, std::min<int64_t>(iters, (2 * cus + ublocks - 1) / ublocks));
                    fa.iters_per_block = ((iters + splits - 1) / splits + 15) / 16 * 16;      // a summary word (16 stages) belongs to one workgroup
                    const dim3 fgrid((unsigned)ublocks, (unsigned)((iters + fa.iters_per_block - 1) / fa.iters_per_block));
                    const size_t flds = yue::filter_lds_bytes(k);
#define YUE_FILTER_V(K16_, FW_, UB_, DMA_) do { if (settle) hipLaunchKernelGGL((yue::k_scan_filter<K16_, FW_, UB_, true, DMA_>), fgrid, dim3(64 * FW_), flds, st, fa); \
                                            else hipLaunchKernelGGL((yue::k_scan_filter<K16_, FW_, UB_, false, DMA_>), fgrid, dim3(64 * FW_), flds, st, fa); } while (0)
#define YUE_FILTER(K16_) do { if (kUB == 2) YUE_FILTER_V(K16_, 4, 2, false); else YUE_FILTER_V(K16_, 8, 1, false); } while (0)
#define YUE_FILTER_DMA(K16_) do { if (fmode == 3) YUE_FILTER_V(K16_, 4, 2, true); else YUE_FILTER(K16_); } while (0)
                    if (k == 16) YUE_FILTER(1); else if (k == 32) YUE_FILTER(2); else if (k == 64) YUE_FILTER_DMA(4); else YUE_FILTER_DMA(8);
#undef YUE_FILTER_DMA
#undef YUE_FILTER
#undef YUE_FILTER_V
                    hipLaunchKernelGGL((yue::k_scan_select<kSelWaves>), dim3((unsigned)((un + kSelWaves - 1) / kSelWaves)), dim3(64 * kSelWaves), sel_lds, st, xa);
                    if (u0 == 0) c->scan_chunks++;
                }
            }
            if (lanes == 2) { HIPCHK(hipEventRecord(c->ev_comm, c->comm_stream)); HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0)); }
            if (nfew > 0) {      // the few-candidates users: the fused kernel over all items, rows back to their places
                HIPCHK(c->s_few_ids.resize((size_t)(nfew * N))); HIPCHK(c->s_few_scores.resize((size_t)(nfew * N)));
                yue::ScanArgs sf = sa;
                sf.users = c->s_few.p + nu; sf.nu = nfew; sf.out_ids = c->s_few_ids.p; sf.out_scores = c->s_few_scores.p;
                (void)yue::launch_scan(sf, c->stream, 0, 0);
                hipLaunchKernelGGL(yue::k_scan_scatter_rows, dim3((unsigned)((nfew * N + 255) / 256)), dim3(256), 0, c->stream, c->s_few_ids.p, c->s_few_scores.p,
                                   c->s_few.p, nfew, N, c->s_ids.p, c->s_scores.p);
            }
            HIPCHK(hipGetLastError());
            done = true;
            rc = 1;
        }
    }
    if (!done) rc = yue::launch_scan(sa, c->stream, c->opt_scan_f32, c->opt_scan_batch);
    HIPCHK(hipEventRecord(t1, c->stream));
    if (rc < 0) return fail(YUE_ERR_ARG, "yue_topn_scan: unsupported (k, N) combination (k <= 256; for k > 128 the list length N is limited to 66)");
    c->scan_used_bf16 = rc;
    HIPCHK(hipGetLastError());
    int32_t flags[4];
    HIPCHK(hipMemcpyAsync(out_ids, c->s_ids.p, nu * N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(out_scores, c->s_scores.p, nu * N * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(flags, c->s_flags.p, sizeof flags, hipMemcpyDeviceToHost, c->stream));
    std::vector<unsigned long long> wslots(4 * yue::kWorkSlots, 0ull);
    HIPCHK(hipMemcpyAsync(wslots.data(), c->s_work.p, wslots.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, t0, t1));
    c->scan_ms = ms;
    unsigned long long work[4] = {0, 0, 0, 0};
    for (int q = 0; q < yue::kWorkSlots; ++q) for (int z = 0; z < 4; ++z) work[z] += wslots[(size_t)4 * q + z];
    c->scan_events = (int64_t)work[1];
    c->scan_rescored = (int64_t)work[2];
    c->scan_tiles_done = (int64_t)work[0];
    c->scan_tiles_total = ((nu + 31) / 32) * ntile;
    if (flags[0]) return fail(YUE_ERR_FEW_ITEMS, "a user has fewer than N candidate items (the reference raises IndexError, base/IterativeRecommender.py:126)");
    return YUE_OK;
}

}  // extern "C"
