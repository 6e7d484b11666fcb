// libyue_hip.so -- exact sequential semantics of recommender/cf/BPR.py:40-62 as a dataflow launch (chain_kernels.hpp):
// yue_bpr_replay on any triplet stream, and yue_bpr_epoch's exact mode on the uploaded events (option "epoch_exact").
// Everything that scales with the number of triplets runs on the device: the row ordinals come from a stable radix sort
// of the touches by row (rocPRIM, the ROCm primitives library -- a pre-pass, not the hot loop), run boundaries from a
// prefix sum.  The host issues launches and copies the stream up once.
#include "host_common.hpp"

#include <rocprim/rocprim.hpp>

#include "chain_kernels.hpp"

using yue_host::fail;
using yue_host::kr_of;

namespace {

int bits_for(uint64_t values) { int b = 1; while ((1ull << b) < values) ++b; return b; }

// ord[code] = number of earlier entries with the same key (stable sort by key, position minus segment start).
// key / val hold `count` entries (clobbered); keys are < nkeys, or == nkeys for entries to ignore.
int rank_in_stream(yue_ctx *c, DevBuf<uint32_t> &key, DevBuf<uint32_t> &val, int64_t count, uint32_t nkeys, uint32_t *ord_a, uint32_t *ord_b) {
    if (count == 0) return YUE_OK;
    HIPCHK(c->ch_key2.resize((size_t)count)); HIPCHK(c->ch_val2.resize((size_t)count)); HIPCHK(c->ch_seg.resize((size_t)nkeys + 1));
    rocprim::double_buffer<uint32_t> dk(key.p, c->ch_key2.p), dv(val.p, c->ch_val2.p);
    size_t tmp = 0;
    const unsigned end_bit = (unsigned)bits_for((uint64_t)nkeys + 1);
    HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp, dk, dv, (size_t)count, 0u, end_bit, c->stream));
    HIPCHK(c->ch_tmp.resize(tmp + 16));
    HIPCHK(rocprim::radix_sort_pairs(c->ch_tmp.p, tmp, dk, dv, (size_t)count, 0u, end_bit, c->stream));
    const dim3 grid((unsigned)((count + 255) / 256));
    hipLaunchKernelGGL(yue::k_chain_seg, grid, dim3(256), 0, c->stream, dk.current(), count, nkeys, c->ch_seg.p);
    hipLaunchKernelGGL(yue::k_chain_ord, grid, dim3(256), 0, c->stream, dk.current(), dv.current(), count, nkeys, c->ch_seg.p, ord_a, ord_b);
    HIPCHK(hipGetLastError());
    return YUE_OK;
}

// Runs the dataflow launch over the triplets (ev_i, ev_j)[T] grouped into R runs.  ev_u / run_u null: run r is user r.
int chain_launch(yue_ctx *c, const int32_t *ev_u, const int32_t *ev_i, const int32_t *ev_j, int64_t T, const int64_t *run_ptr, const int32_t *run_u,
                 const uint32_t *ord_u, int64_t R, double lr, double regU, double regI) {
    const int64_t n = c->n, m = c->m;
    const int k = c->k;
    if (2 * T >= (1ll << 32)) return fail(YUE_ERR_ARG, "exact path: at most 2^31 - 1 triplets per call");
    HIPCHK(c->ch_key.resize((size_t)(2 * T))); HIPCHK(c->ch_val.resize((size_t)(2 * T)));
    HIPCHK(c->ch_ord_i.resize((size_t)T + yue_host::kHeaderSlackHost)); HIPCHK(c->ch_ord_j.resize((size_t)T + yue_host::kHeaderSlackHost));   // (+ slack: the chain kernel reads whole header blocks)
    HIPCHK(c->ch_ctl.resize(4));
    HIPCHK(hipMemsetAsync(c->ch_ctl.p, 0, 4 * sizeof(unsigned long long), c->stream));
    uint32_t *flags = reinterpret_cast<uint32_t *>(c->ch_ctl.p + 1), *status = reinterpret_cast<uint32_t *>(c->ch_ctl.p + 2);
    const dim3 tgrid((unsigned)((T + 255) / 256));
    hipLaunchKernelGGL(yue::k_chain_keys, tgrid, dim3(256), 0, c->stream, ev_u, ev_i, ev_j, T, m, (uint32_t)n, c->ch_key.p, c->ch_val.p, flags);
    int rc = rank_in_stream(c, c->ch_key, c->ch_val, 2 * T, (uint32_t)n, c->ch_ord_i.p, c->ch_ord_j.p);
    if (rc) return rc;
    {   // ids are checked on the device; nothing has been written to the factors yet
        unsigned long long fl = 0;
        HIPCHK(hipMemcpyAsync(&fl, c->ch_ctl.p + 1, sizeof fl, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (fl & 1u) return fail(YUE_ERR_ARG, "a triplet's user or item id is out of range");
        if (fl & 2u) return fail(YUE_ERR_ARG, "a triplet has i == j");
    }
    // item rows -> granules (and the user rows, when a user may have several runs)
    const size_t gwords = (size_t)yue::chain_granules_per_row(k) * 2;      // 32-bit words of a granule row
    if (n * (int64_t)gwords * 4 >= (1ll << 31)) return fail(YUE_ERR_ARG, "exact path: the granule copy of the item matrix must stay below 2 GiB");
    HIPCHK(c->ch_Qv.resize((size_t)n * gwords));
    hipLaunchKernelGGL(yue::k_chain_pack, dim3(2048), dim3(256), 0, c->stream, c->Q.p, c->ch_Qv.p, n, k);
    if (ord_u) {
        HIPCHK(c->ch_Pv.resize((size_t)m * gwords));
        hipLaunchKernelGGL(yue::k_chain_pack, dim3(2048), dim3(256), 0, c->stream, c->P.p, c->ch_Pv.p, m, k);
    }
    yue::ChainArgs a{};
    a.P = c->P.p; a.Pv = ord_u ? c->ch_Pv.p : nullptr; a.Qv = c->ch_Qv.p;
    a.run_ptr = run_ptr; a.run_u = run_u; a.ord_u = ord_u; a.ev_i = ev_i; a.ev_j = ev_j; a.ord_i = c->ch_ord_i.p; a.ord_j = c->ch_ord_j.p;
    a.R = R; a.claim = c->ch_ctl.p; a.status = status; a.nll_slots = c->scal.p; a.m = m; a.n = n; a.k = k;
    a.ru = (float)(lr * regU); a.ri = (float)(lr * regI); a.lr = lr;          // BPR.py:55: python-float product, cast to fp32 by NumPy
    a.spin_limit = c->opt_chain_spin > 0 ? (uint32_t)c->opt_chain_spin : (1u << 22);
#ifdef YUE_CHAIN_STATS
    HIPCHK(c->ch_stats.resize(8));
    HIPCHK(hipMemsetAsync(c->ch_stats.p, 0, 8 * sizeof(unsigned long long), c->stream));
    a.stats = c->ch_stats.p;
#endif
    // One persistent launch of resident waves only (a wave that is not resident cannot be waited for).  The epoch's time is its
    // longest chain of dependent triplets times the latency of a step, and a wave alone on its SIMD has the shortest step
    // (measured on BASELINE config 3 with k_bpr_chain: 913 ms per epoch with 1,024 waves, 1,131 ms with 3,072), so the default
    // is ONE workgroup per CU (option chain_waves = workgroups per CU):
    //   chain_split = 1: k_bpr_chain3, a workgroup = five waves (2 x loads / chain / 2 x stores) walking one run;
    //   chain_split = 0: k_bpr_chain, a workgroup = four waves, each walking a run of its own;
    //   chain_split = -1 (default): the first for streams of at least 16 triplets per run.
    // chain_fast = 1: single-precision coefficient and one 64-lane sum per triplet (within 1e-5 of the reference, not bit-equal).
    int per_cu = 0, cus = 0;
    const int kr = kr_of(k);
    dim3 block(256), grid(1);
    int waves_per_block = 4;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    bool one_xcd = false, allow_one_xcd = true;
    // which kernel: the wave group shortens the step of a run (config 3: 940 -> 607 ms per epoch, bit-equal) but makes the
    // start of a run and every cross-wave hand-off dearer (one rank's share of config 4, 6 triplets per run: 181 -> 291 ms)
    const bool split = c->opt_chain_split < 0 ? T >= 16 * R : c->opt_chain_split != 0;
    auto launch = [&](auto kernel, int threads, int64_t runs_per_block) -> int {
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0));
        per_cu = std::max(1, std::min(per_cu, c->opt_chain_waves > 0 ? c->opt_chain_waves : (one_xcd ? 2 : 1)));
        block = dim3((unsigned)threads);
        // (one XCD: a workgroup in eight stays, and the XCD has an eighth of the CUs -- the same grid size for per_cu per CU of it)
        grid = dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)per_cu * cus, one_xcd ? (int64_t)per_cu * cus : (R + runs_per_block - 1) / runs_per_block)));
        HIPCHK(hipEventRecord(c->ev_chain0, c->stream));
        hipLaunchKernelGGL(kernel, grid, block, 0, c->stream, a, ev_i, ev_j, c->ch_ord_i.p, c->ch_ord_j.p);
        HIPCHK(hipEventRecord(c->ev_chain1, c->stream));
        return YUE_OK;
    };
    if (!c->ev_chain0) { HIPCHK(hipEventCreate(&c->ev_chain0)); HIPCHK(hipEventCreate(&c->ev_chain1)); }
    a.xcd = 0u; a.groups = c->ch_ctl.p + 3;
    // ring of 8 prefetched triplets per wave (k_bpr_chain at k > 128: 4, register budget)
#define YUE_CHAIN3(KR_, PV_, G_, X_) (c->opt_chain_fast ? launch(yue::k_bpr_chain3<KR_, PV_, G_, true, X_>, 320, 1) : launch(yue::k_bpr_chain3<KR_, PV_, G_, false, X_>, 320, 1))
#define YUE_CHAIN_RUN(KR_, PV_, G1_)                                                                                    \
    do {                                                                                                                  \
        if (split) {                                                                                                      \
            waves_per_block = 5;                                                                                          \
            one_xcd = c->opt_chain_xcd != 0 && allow_one_xcd;                                                                           \
            if (c->opt_chain_ring == 16 && KR_ <= 2) rc = one_xcd ? YUE_CHAIN3(KR_, PV_, (KR_ <= 2 ? 16 : 8), true) : YUE_CHAIN3(KR_, PV_, (KR_ <= 2 ? 16 : 8), false); \
            else rc = one_xcd ? YUE_CHAIN3(KR_, PV_, 8, true) : YUE_CHAIN3(KR_, PV_, 8, false);                           \
        } else {                                                                                                          \
            rc = c->opt_chain_fast ? launch(yue::k_bpr_chain<KR_, PV_, G1_, true>, 256, 4)                                \
                                   : launch(yue::k_bpr_chain<KR_, PV_, G1_, false>, 256, 4);                              \
        }                                                                                                                 \
    } while (0)
    for (;;) {
        if (ord_u) { if (kr == 1) YUE_CHAIN_RUN(1, true, 8); else if (kr == 2) YUE_CHAIN_RUN(2, true, 8); else YUE_CHAIN_RUN(4, true, 4); }
        else { if (kr == 1) YUE_CHAIN_RUN(1, false, 8); else if (kr == 2) YUE_CHAIN_RUN(2, false, 8); else YUE_CHAIN_RUN(4, false, 4); }
        if (rc || !one_xcd) break;
        // one XCD: did any workgroup land on it?  (Placement is the dispatcher's business; with none, nothing has been touched
        // and the launch is repeated on all XCDs with write-through hand-offs.)
        unsigned long long stayed = 0;
        HIPCHK(hipMemcpyAsync(&stayed, c->ch_ctl.p + 3, sizeof stayed, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        c->chain_groups = (int64_t)stayed;
        if (stayed > 0) break;
        allow_one_xcd = false;
    }
#undef YUE_CHAIN_RUN
#undef YUE_CHAIN3
    if (rc) return rc;
    const int64_t blocks = grid.x;
    hipLaunchKernelGGL(yue::k_chain_unpack, dim3(2048), dim3(256), 0, c->stream, c->ch_Qv.p, c->Q.p, n, k);
    if (ord_u) hipLaunchKernelGGL(yue::k_chain_unpack, dim3(2048), dim3(256), 0, c->stream, c->ch_Pv.p, c->P.p, m, k);
    HIPCHK(hipGetLastError());
    unsigned long long ctl[4];
    HIPCHK(hipMemcpyAsync(ctl, c->ch_ctl.p, sizeof ctl, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint32_t st = (uint32_t)ctl[2];
#ifdef YUE_CHAIN_STATS
    {
        unsigned long long h[8];
        HIPCHK(hipMemcpy(h, c->ch_stats.p, sizeof h, hipMemcpyDeviceToHost));
        if (split)
            fprintf(stderr, "[chain3 stats] per triplet: wave C %.0f cycles of work + %.0f waiting for a packet; wave L %.0f cycles per step without a wait (%llu steps) and %.0f waiting for room in the ring; wave S %.0f cycles waiting for packets or answers (%llu triplets)\n",
                    h[1] ? (double)h[0] / h[1] : 0.0, h[1] ? (double)h[2] / h[1] : 0.0, h[4] ? (double)h[3] / h[4] : 0.0, h[4], h[1] ? (double)h[7] / h[1] : 0.0, h[6] ? (double)h[5] / h[6] : 0.0, h[6]);
        else
        fprintf(stderr, "[chain stats] steps without a wait: %llu, %.0f cycles each; steps that waited: %llu (%.2f %%), %.0f cycles each; per run outside the steps: %.0f cycles (%llu runs)\n",
                h[1], h[1] ? (double)h[0] / h[1] : 0.0, h[3], 100.0 * h[3] / (double)(h[1] + h[3] + 1e-9), h[3] ? (double)h[2] / h[3] : 0.0, h[5] ? (double)h[4] / h[5] : 0.0, h[5]);
    }
#endif
    c->chain_runs = R; c->chain_waves = (one_xcd ? c->chain_groups : blocks) * waves_per_block;
    { float ms = 0.0f; HIPCHK(hipEventElapsedTime(&ms, c->ev_chain0, c->ev_chain1)); c->chain_kernel_us = (int64_t)(1e3 * ms); }
    if (st) return fail(YUE_ERR_HIP, std::string("exact path: a wave gave up waiting for a row (") + ((st & 2u) ? "a row's version ran past a waiting touch" : "spin limit") +
                                    "): internal error, the factors on the device are not usable");
    return YUE_OK;
}

}  // namespace

namespace yue_host {

// The uploaded events (user-major, the sampler's negatives in ev_j) with exact sequential semantics: a run = a user.
int chain_epoch(yue_ctx *c, double lr, double regU, double regI) {
    if (c->E == 0) return YUE_OK;
    HIPCHK(c->d_ev_ptr.resize((size_t)c->m + 1));
    if (!c->d_ev_ptr_valid) {
        HIPCHK(hipMemcpyAsync(c->d_ev_ptr.p, c->h_ev_ptr.data(), ((size_t)c->m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        c->d_ev_ptr_valid = true;
    }
    return chain_launch(c, nullptr, c->ev_i.p, c->ev_j.p, c->E, c->d_ev_ptr.p, nullptr, nullptr, c->m, lr, regU, regI);
}

// Any triplet stream on the device (xu, xi, xj): runs of equal consecutive users, user rows versioned per run.
int chain_stream(yue_ctx *c, int64_t T, double lr, double regU, double regI) {
    if (T == 0) return YUE_OK;
    if (T >= (1ll << 31)) return fail(YUE_ERR_ARG, "exact path: at most 2^31 - 1 triplets per call");
    HIPCHK(c->ch_head.resize((size_t)T)); HIPCHK(c->ch_incl.resize((size_t)T));
    const dim3 tgrid((unsigned)((T + 255) / 256));
    hipLaunchKernelGGL(yue::k_chain_heads, tgrid, dim3(256), 0, c->stream, c->xu.p, T, c->ch_head.p);
    size_t tmp = 0;
    HIPCHK(rocprim::inclusive_scan(nullptr, tmp, c->ch_head.p, c->ch_incl.p, (size_t)T, rocprim::plus<uint32_t>(), c->stream));
    HIPCHK(c->ch_tmp.resize(tmp + 16));
    HIPCHK(rocprim::inclusive_scan(c->ch_tmp.p, tmp, c->ch_head.p, c->ch_incl.p, (size_t)T, rocprim::plus<uint32_t>(), c->stream));
    uint32_t R32 = 0;
    HIPCHK(hipMemcpyAsync(&R32, c->ch_incl.p + (T - 1), sizeof R32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const int64_t R = R32;
    HIPCHK(c->ch_run_ptr.resize((size_t)R + 1)); HIPCHK(c->ch_run_u.resize((size_t)R)); HIPCHK(c->ch_ord_u.resize((size_t)R));
    HIPCHK(c->ch_rkey.resize((size_t)R)); HIPCHK(c->ch_rval.resize((size_t)R));
    hipLaunchKernelGGL(yue::k_chain_runs, tgrid, dim3(256), 0, c->stream, c->xu.p, c->ch_head.p, c->ch_incl.p, T, c->ch_run_ptr.p, c->ch_run_u.p, c->ch_rkey.p, c->ch_rval.p, (uint32_t)c->m);
    // (user ids out of range sort behind all users here and are reported by k_chain_keys before anything is written)
    int rc = rank_in_stream(c, c->ch_rkey, c->ch_rval, R, (uint32_t)c->m, c->ch_ord_u.p, nullptr);
    if (rc) return rc;
    return chain_launch(c, c->xu.p, c->xi.p, c->xj.p, T, c->ch_run_ptr.p, c->ch_run_u.p, c->ch_ord_u.p, R, lr, regU, regI);
}

}  // namespace yue_host
