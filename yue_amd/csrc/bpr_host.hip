// libyue_hip.so -- BPR training entry points: exact replay by dependency levels, S-rounds, fused epochs, CUNE's steps, the
// TF-style Adam step, options and kernel timing (include/yue_hip.h).
#include "host_common.hpp"

#include "train_kernels.hpp"
#include "round_kernels.hpp"

using yue_host::fail;
using yue_host::kr_of;
static_assert(yue::kNllSlots == yue_host::kNllSlotsHost && yue::kHeaderSlack == yue_host::kHeaderSlackHost, "host_common.hpp mirrors these");

namespace {

yue::TrainArgs make_args(yue_ctx *c, double lr, double regU, double regI) {
    yue::TrainArgs a{};
    a.P = c->P.p; a.Q = c->Q.p; a.dP = c->dP.p; a.dQ = c->dQ.p;
    a.stage = c->Q.p ? c->Q.p + (size_t)c->n * c->k : nullptr;
    a.ev_u = c->ev_u.p; a.ev_i = c->ev_i.p; a.ev_j = c->ev_j.p;
    a.indptr = c->indptr.p; a.indices = c->indices.p;
    a.nll_slots = c->scal.p;
    a.m = c->m; a.n = c->n; a.k = c->k;
    a.lr = lr;
    a.ru = (float)(lr * regU);      // BPR.py:55: python-float product, cast to fp32 by NumPy
    a.ri = (float)(lr * regI);
    a.neg_lo = 0; a.neg_range = (int32_t)c->n;
    return a;
}

void launch_level(yue_ctx *c, const yue::TrainArgs &a, int64_t e0, int64_t e1) {
    // enough waves to cover the chip before a wave takes more triplets of the level
    const int tpw = (int)std::min<int64_t>(64, std::max<int64_t>(1, (e1 - e0) / 4096));
    const int64_t waves = (e1 - e0 + tpw - 1) / tpw;
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    switch (kr_of(c->k)) {
        case 1: hipLaunchKernelGGL(yue::k_bpr_level<1>, grid, block, 0, c->stream, a, e0, e1, tpw); break;
        case 2: hipLaunchKernelGGL(yue::k_bpr_level<2>, grid, block, 0, c->stream, a, e0, e1, tpw); break;
        default: hipLaunchKernelGGL(yue::k_bpr_level<4>, grid, block, 0, c->stream, a, e0, e1, tpw); break;
    }
}

// `meta`: for k_round_m (update launch on pre-pass metadata) instead of k_round
int tpw_of(const yue_ctx *c, bool meta) {
    if (c->opt_round_tpw) return c->opt_round_tpw;
    // measured: k = 128 (C3): 8 events per wave 53.9 ms/epoch, 4 -> 58.7, 16 -> 60.7; k = 64 (C2): k_round 16 events per wave
    // 5.2 ms/epoch, 8 -> 5.7; k_round_m + fold (round 3, profiles/r03_round_w_stage_sweep.txt) 8 -> 3.0 ms, 16 -> 3.3
    const int kr = kr_of(c->k);
    return kr == 4 ? 4 : kr == 2 ? 8 : meta ? 8 : 16;
}

// Epoch path: the pre-pass over all rounds (k_round_meta) -- one LDS word per item row of a range, ranges of at most
// kMetaRangeMax rows.  Past kMetaRangesMax ranges every work item would re-read its round too often: the caller then
// stays on k_round (touches counted inside the round launches).
constexpr int64_t kMetaRangeMax = 37 * 1024;       // 148 KB of the CU's 160 KB of LDS
constexpr int64_t kMetaRangesMax = 12;
constexpr int64_t kRoundGenerationsMax = 6;
constexpr double kRoundEventsPerItemRow = 4.0;

// Larger catalogues (up to kBucketRangesMax ranges of 2^kBucketShift rows = 8.4M item rows per GPU) sort a round's touches into
// their ranges first (k_round_bucket), so that a work item reads its own touches only.
bool meta_bucketed(const yue_ctx *c) { return c->n > kMetaRangeMax * kMetaRangesMax || c->opt_round_bucket; }
bool meta_path_fits(const yue_ctx *c) { return c->opt_round_meta && c->n <= ((int64_t)yue::kBucketRangesMax << yue::kBucketShift); }
bool fold_path(const yue_ctx *c) { return meta_path_fits(c); }

// Default round size.
//   * k_round / k_round_m with the retire phase inside the launch: the events ONE resident set of waves takes (workgroups
//     of 4 waves x TPW events) -- a launch is a single wave generation, the measured optimum of a kernel whose waves wait
//     for each other (49,152 at k = 128);
//   * the fold path (k_round_m without retire + k_round_fold: no wave waits for another): up to kRoundGenerationsMax
//     resident sets, as long as a round stays at or below kRoundEventsPerItemRow events per item row (the staleness of
//     the S-round semantics is bounded relative to the item count) -- 6 x 57,344 = 344,064 on C3, 3 x 57,344 on C2.
//     Round 3 measured (profiles/r03_deviation_*.jsonl): the distance from the sequential loop is
//     set by a user's events sharing one round, not by W -- C3: loss +0.320 % at W = 172,032, +0.332 % at 344,064;
//     rounds of 6 sets with staging blocks sized for them are 6 % faster than rounds of 3.
// `n_rows` = item rows per rank (job-wide average on a communicator, so that all ranks agree).
int default_round_events(yue_ctx *c, double n_rows, int64_t *out) {
    const bool fold = fold_path(c);
    const int tpw = tpw_of(c, fold);
    int per_cu = 0, cus = 0;
    hipError_t e = hipSuccess;
#define YUE_OCC(KR_, TPW_) \
    e = fold ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, yue::k_round_m<KR_, TPW_>, 256, 0) \
             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, yue::k_round<KR_, TPW_>, 256, 0); \
    break;
    switch (kr_of(c->k) * 16 + tpw) {
        case 1 * 16 + 8: YUE_OCC(1, 8)
        case 1 * 16 + 16: YUE_OCC(1, 16)
        case 2 * 16 + 8: YUE_OCC(2, 8)
        case 4 * 16 + 4: YUE_OCC(4, 4)
        case 1 * 16 + 4: YUE_OCC(1, 4)
        case 2 * 16 + 4: YUE_OCC(2, 4)
        case 1 * 16 + 2: YUE_OCC(1, 2)
        case 2 * 16 + 2: YUE_OCC(2, 2)
        default: return fail(YUE_ERR_ARG, "unsupported (k, TPW) combination");
    }
#undef YUE_OCC
    HIPCHK(e);
    // A CU admits at most floor(800 / (ceil(SGPRs / 16) * 16 + 16)) workgroups of 4 waves whatever the occupancy query
    // says (MI355X_MICROARCH.md, residency rule; seen on C2: the query answers 7, a round sized for 7 runs as two
    // generations): k_round uses all 106 SGPRs -> 6; k_round_m is compiled with at most 96 -> 7
    // (yue_amd/csrc/resource_usage.txt).
    per_cu = std::min(per_cu, fold ? 7 : 6);
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    const double slots = (double)per_cu * (double)cus;                    // resident workgroups of 256 threads
    int64_t w = (int64_t)(slots * 4.0 * tpw);
    w -= w % 1024;
    w = std::max<int64_t>(w, 1024);
    if (fold) {
        int64_t sets = std::min<int64_t>(kRoundGenerationsMax, std::max<int64_t>(1, (int64_t)(kRoundEventsPerItemRow * n_rows / (double)w)));
        // wide catalogues (fewer than two touches per item row and round): nearly every row is written in place, a larger
        // round amortises nothing and the rows it writes outgrow the memory-side cache (measured on c3wide, 1M item rows:
        // 29.4 ms/epoch at 3 and 4 sets, 31.6 at 6)
        if (2.0 * (double)(w * sets) < 2.0 * n_rows) sets = std::min<int64_t>(sets, 4);
        w *= sets;
    }
    *out = w;
    return YUE_OK;
}

// One S-round launch: update [e0,e1) with the counts in cnt_cur, prepare [n0,n1) into cnt_next.
// Every timing_stride-th launch is bracketed with HIP events on the library's stream.
int launch_round(yue_ctx *c, const yue::TrainArgs &a_in, int64_t e0, int64_t e1, int64_t n0, int64_t n1,
                 int parity, int apply_p) {
    unsigned long long *cnt[2] = {c->cnt0.p, c->cnt1.p};
    uint32_t *cntp[2] = {c->cntp0.p, c->cntp1.p};
    yue::RoundArgs ra{};
    ra.e_begin = e0; ra.e_end = e1; ra.n_begin = n0; ra.n_end = n1;
    ra.cnt_cur = cnt[parity]; ra.cnt_next = cnt[parity ^ 1]; ra.cntp_cur = cntp[parity]; ra.cntp_next = cntp[parity ^ 1];
    uint32_t *tab[2] = {c->tab0.p, c->tab1.p};
    ra.tab_cur = tab[parity]; ra.tab_next = tab[parity ^ 1]; ra.staged = c->staged ? 1 : 0;
    ra.apply_p = apply_p;
    const int tpw = tpw_of(c, false);
    const int64_t waves = (std::max(e1 - e0, n1 - n0) + tpw - 1) / tpw;      // every wave: tickets of the next round + its update batch
    const int64_t blocks = (waves + 3) / 4;
    if (blocks == 0) return YUE_OK;
    const dim3 grid((unsigned)blocks), block(256);
#ifdef YUE_STAMPS
    yue::TrainArgs a = a_in;
    a.stamps = nullptr;
    if (e1 > e0 && c->update_launches++ == c->stamp_launch) {
        HIPCHK(c->stamps.resize((size_t)waves * 8));
        HIPCHK(hipMemsetAsync(c->stamps.p, 0, (size_t)waves * 64, c->stream));
        a.stamps = c->stamps.p;
        c->stamp_waves = (e1 - e0 + tpw - 1) / tpw;
    }
#else
    const yue::TrainArgs &a = a_in;
#endif
    switch (kr_of(c->k) * 16 + tpw) {
        case 1 * 16 + 8: hipLaunchKernelGGL((yue::k_round<1, 8>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j); break;
        case 1 * 16 + 16: hipLaunchKernelGGL((yue::k_round<1, 16>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j); break;
        case 2 * 16 + 8: hipLaunchKernelGGL((yue::k_round<2, 8>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j); break;
        case 4 * 16 + 4: hipLaunchKernelGGL((yue::k_round<4, 4>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j); break;
        case 1 * 16 + 4: hipLaunchKernelGGL((yue::k_round<1, 4>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j); break;
        case 2 * 16 + 4: hipLaunchKernelGGL((yue::k_round<2, 4>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j); break;
        case 1 * 16 + 2: hipLaunchKernelGGL((yue::k_round<1, 2>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j); break;
        case 2 * 16 + 2: hipLaunchKernelGGL((yue::k_round<2, 2>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j); break;
        default: return fail(YUE_ERR_ARG, "unsupported (k, TPW) combination");
    }
    return YUE_OK;
}

int launch_round_meta(yue_ctx *c, const yue::TrainArgs &a, const std::vector<int64_t> &bounds) {
    const int64_t R = (int64_t)bounds.size() - 1;
    const int64_t E = bounds.back();
    HIPCHK(c->meta_i.resize((size_t)E + yue::kHeaderSlack)); HIPCHK(c->meta_j.resize((size_t)E + yue::kHeaderSlack));   // (+ slack: k_round_m reads whole header blocks)
    HIPCHK(c->round_rows.resize((size_t)R)); HIPCHK(c->d_bounds.resize((size_t)R + 1));
    if (c->h_bounds != bounds) {                       // the same blocks epoch after epoch: uploaded once
        HIPCHK(hipStreamSynchronize(c->stream));       // an earlier upload may still read h_bounds
        c->h_bounds = bounds;
        HIPCHK(hipMemcpyAsync(c->d_bounds.p, c->h_bounds.data(), (size_t)(R + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipMemsetAsync(c->round_rows.p, 0, (size_t)R * sizeof(unsigned long long), c->stream));
    HIPCHK(c->fold.resize((size_t)(E + 4 * R + 8)));
    yue::MetaArgs ma{};
    ma.ev_i = a.ev_i; ma.ev_j = a.ev_j; ma.bounds = c->d_bounds.p; ma.R = R; ma.n = (int32_t)c->n;
    const bool bucketed = meta_bucketed(c);
    if (bucketed) {
        ma.range = 1 << yue::kBucketShift;
        ma.G = (int32_t)((c->n + ma.range - 1) >> yue::kBucketShift);
        HIPCHK(c->bk_touch.resize((size_t)(2 * E + 2))); HIPCHK(c->bk_ptr.resize((size_t)R * (ma.G + 1)));
        yue::BucketArgs ba{};
        ba.ev_i = a.ev_i; ba.ev_j = a.ev_j; ba.bounds = c->d_bounds.p; ba.R = R; ba.G = ma.G;
        ba.bk_touch = c->bk_touch.p; ba.bk_ptr = c->bk_ptr.p; ba.meta_i = c->meta_i.p; ba.meta_j = c->meta_j.p;
        int cus0 = 0;
        HIPCHK(hipDeviceGetAttribute(&cus0, hipDeviceAttributeMultiprocessorCount, c->device));
        hipLaunchKernelGGL(yue::k_round_bucket, dim3((unsigned)std::min<int64_t>(R, 2 * cus0)), dim3(1024), 0, c->stream, ba);
        ma.bk_touch = c->bk_touch.p; ma.bk_ptr = c->bk_ptr.p;
    } else {
        ma.G = (int32_t)((c->n + kMetaRangeMax - 1) / kMetaRangeMax);
        ma.range = (int32_t)((c->n + ma.G - 1) / ma.G);
    }
    ma.chunk = (ma.range + 1023) / 1024; ma.chunk |= 1;
    // Largest staged block: rows with more touches take float atomics (executed memory-side on this chip, ~1 TB/s) -- sized
    // from the round's mean touches per item row so that the blocks cover the bulk of the distribution; a block is summed
    // by one wave of the fold launch, very long ones would be its tail.  Measured per k (profiles/r03_round_w_stage_sweep.txt):
    // k <= 64 (rows of 256 B: an atomic row costs as many requests as bytes/64) 4 x the mean, else 2 x.
    int64_t widest = 0;
    for (int64_t r = 0; r < R; ++r) widest = std::max(widest, bounds[(size_t)r + 1] - bounds[(size_t)r]);
    const double mean_touches = 2.0 * (double)widest / (double)std::max<int64_t>(1, c->n);
    const uint32_t stage_auto = (uint32_t)std::min<double>((double)yue::kMetaStageMax, std::max<double>((double)yue::kMetaStageDefault,
                                                         std::ceil(mean_touches * (kr_of(c->k) == 1 ? 4.0 : 2.0))));
    ma.stage_max = !c->staged ? 1u : c->opt_round_stage >= 2 ? (uint32_t)c->opt_round_stage : stage_auto;
    c->last_stage_max = (int)ma.stage_max;
    ma.meta_i = c->meta_i.p; ma.meta_j = c->meta_j.p; ma.round_rows = c->round_rows.p; ma.fold = c->fold.p;
    const size_t lds = (size_t)ma.range * sizeof(uint32_t);
    const void *kfn = bucketed ? (const void *)yue::k_round_meta<true> : (const void *)yue::k_round_meta<false>;
    HIPCHK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0, cus = 0;
    if (bucketed) { HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, yue::k_round_meta<true>, 1024, lds)); }
    else { HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, yue::k_round_meta<false>, 1024, lds)); }
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    // a multiple of 8 workgroups (k_round_meta deals the work out per XCD), at most one resident set
    const int64_t grid = std::max<int64_t>(8, std::min<int64_t>((R * ma.G + 7) / 8 * 8, (int64_t)std::max(per_cu, 1) * cus / 8 * 8));
    if (bucketed) hipLaunchKernelGGL(yue::k_round_meta<true>, dim3((unsigned)grid), dim3(1024), lds, c->stream, ma);
    else
        hipLaunchKernelGGL(yue::k_round_meta<false>, dim3((unsigned)grid), dim3(1024), lds, c->stream, ma);
    HIPCHK(hipGetLastError());
    return YUE_OK;
}

void launch_round_fold(yue_ctx *c, const yue::TrainArgs &a, int64_t e0, int64_t e1, int64_t round_index) {
    yue::FoldArgs f{};
    f.fold = c->fold.p + yue::fold_base(e0, round_index); f.round_rows = c->round_rows.p + round_index; f.Q = a.Q; f.dQ = a.dQ; f.stage = a.stage; f.k = c->k; f.capacity = (uint32_t)((e1 - e0) & ~(int64_t)3);
    const dim3 fgrid((unsigned)std::min<int64_t>(c->opt_fold_blocks, std::max<int64_t>(1, (e1 - e0 + 15) / 16))), block(256);
    switch (kr_of(c->k)) {
        case 1: hipLaunchKernelGGL(yue::k_round_fold<1>, fgrid, block, 0, c->stream, f); break;
        case 2: hipLaunchKernelGGL(yue::k_round_fold<2>, fgrid, block, 0, c->stream, f); break;
        default: hipLaunchKernelGGL(yue::k_round_fold<4>, fgrid, block, 0, c->stream, f); break;
    }
}

int launch_round_m(yue_ctx *c, const yue::TrainArgs &a_in, int64_t e0, int64_t e1, int64_t round_index) {
    yue::RoundMArgs ra{};
    ra.e_begin = e0; ra.e_end = e1; ra.staged = c->staged ? 1 : 0;
    const int tpw = tpw_of(c, true);
    const int64_t waves = (e1 - e0 + tpw - 1) / tpw;
    const int64_t blocks = (waves + 3) / 4;
    if (blocks == 0) return YUE_OK;
    const dim3 grid((unsigned)blocks), block(256);
#ifdef YUE_STAMPS
    yue::TrainArgs a = a_in;
    a.stamps = nullptr;
    if (c->update_launches++ == c->stamp_launch) {
        HIPCHK(c->stamps.resize((size_t)waves * 8));
        HIPCHK(hipMemsetAsync(c->stamps.p, 0, (size_t)waves * 64, c->stream));
        a.stamps = c->stamps.p;
        c->stamp_waves = waves;
    }
#else
    const yue::TrainArgs &a = a_in;
#endif
    const uint32_t *mi = c->meta_i.p, *mj = c->meta_j.p;
    if (c->bigq) {              // (the default events-per-wave instances only)
        switch (kr_of(c->k) * 16 + tpw) {
            case 1 * 16 + 8: hipLaunchKernelGGL((yue::k_round_m<1, 8, true>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j, mi, mj); break;
            case 2 * 16 + 8: hipLaunchKernelGGL((yue::k_round_m<2, 8, true>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j, mi, mj); break;
            case 4 * 16 + 4: hipLaunchKernelGGL((yue::k_round_m<4, 4, true>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j, mi, mj); break;
            default: return fail(YUE_ERR_ARG, "item matrices of 2 GiB and more: only the default events-per-wave setting (round_tpw = 0)");
        }
    } else
#define YUE_RM(KR_, TPW_) hipLaunchKernelGGL((yue::k_round_m<KR_, TPW_>), grid, block, 0, c->stream, a, ra, a.ev_u, a.ev_i, a.ev_j, mi, mj); break;
    switch (kr_of(c->k) * 16 + tpw) {
        case 1 * 16 + 8: YUE_RM(1, 8)
        case 1 * 16 + 16: YUE_RM(1, 16)
        case 2 * 16 + 8: YUE_RM(2, 8)
        case 4 * 16 + 4: YUE_RM(4, 4)
        case 1 * 16 + 4: YUE_RM(1, 4)
        case 2 * 16 + 4: YUE_RM(2, 4)
        case 1 * 16 + 2: YUE_RM(1, 2)
        case 2 * 16 + 2: YUE_RM(2, 2)
        default: return fail(YUE_ERR_ARG, "unsupported (k, TPW) combination");
    }
#undef YUE_RM
    launch_round_fold(c, a, e0, e1, round_index);
    return YUE_OK;
}

// The S-round update launch with sequential user rows (k_round_u: one wave per user of the round) + the fold launch.
int launch_round_u(yue_ctx *c, const yue::TrainArgs &a, int64_t u0, int64_t u1, int64_t e0, int64_t e1, int64_t round_index) {
    yue::RoundUArgs ra{};
    ra.u_begin = u0; ra.u_end = u1; ra.ev_ptr = c->d_ev_ptr.p; ra.e_begin = e0; ra.e_end = e1; ra.staged = c->staged ? 1 : 0; ra.margins = c->margins.p;
    const int64_t blocks = (u1 - u0 + 3) / 4;
    if (blocks == 0) return YUE_OK;
    const dim3 grid((unsigned)blocks), block(256);
    const uint32_t *mi = c->meta_i.p, *mj = c->meta_j.p;
#define YUE_RU(KR_) do { if (c->opt_round_fast) hipLaunchKernelGGL((yue::k_round_u<KR_, true>), grid, block, 0, c->stream, a, ra, a.ev_i, a.ev_j, mi, mj); \
                         else hipLaunchKernelGGL((yue::k_round_u<KR_, false>), grid, block, 0, c->stream, a, ra, a.ev_i, a.ev_j, mi, mj); } while (0)
    switch (kr_of(c->k)) {
        case 1: YUE_RU(1); break;
        case 2: YUE_RU(2); break;
        default: YUE_RU(4); break;
    }
#undef YUE_RU
    launch_round_fold(c, a, e0, e1, round_index);
    return YUE_OK;
}

// Runs the non-empty rounds bounds[r]..bounds[r+1] in order.  after_round(r) is called once the
// launches of round r are queued (the communicator path hooks its all-reduce there).
template <typename F>
int run_rounds(yue_ctx *c, yue::TrainArgs a, const std::vector<int64_t> &bounds, int apply_p, F after_round, bool meta = false, const std::vector<int64_t> *user_blocks = nullptr) {
    const int64_t R = (int64_t)bounds.size() - 1;
    // staging rows: two per event of the largest round, addressed with 31-bit byte offsets
    int64_t widest = 0;
    for (int64_t r = 0; r < R; ++r) widest = std::max(widest, bounds[(size_t)r + 1] - bounds[(size_t)r]);
    // the metadata word keeps a staging row index or a row's touch count in 25 bits: rounds of 2^24 events and more stay on k_round
    if (widest >= (1ll << 24)) meta = false;
    const bool fits31 = (c->n + 2 * widest) * (int64_t)c->k * 4 < (1ll << 31);     // item rows + staging rows within 31-bit byte offsets
    // beyond that the update launch of the epoch path addresses rows through 64-bit pointers (k_round_m<.., BIGQ>); k_round cannot
    c->bigq = meta && !fits31;
    if (!meta && (int64_t)c->n * c->k * 4 >= (1ll << 31))
        return fail(YUE_ERR_ARG, "item matrices of 2 GiB and more per GPU are supported by yue_bpr_epoch's default path only (not by explicit rounds / round_meta = 0 / rounds of 2^24 events and more)");
    c->staged = c->opt_round_stage && widest > 0 && (fits31 || c->bigq);
    if (c->staged) {
        const size_t need = (size_t)(c->n + 2 * widest) * (size_t)c->k;
        if (c->Q.n < need) {                                // grow the item allocation, keep the rows
            DevBuf<float> bigger;
            HIPCHK(bigger.resize(need));
            HIPCHK(hipMemcpyAsync(bigger.p, c->Q.p, (size_t)c->n * c->k * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            c->Q.release();
            c->Q = bigger;
            a.Q = c->Q.p;
        }
        if (!meta) {
            HIPCHK(c->tab0.resize((size_t)c->n * yue::kStageMax));
            HIPCHK(c->tab1.resize((size_t)c->n * yue::kStageMax));
        }
        a.stage = c->Q.p + (size_t)c->n * c->k;
    }
    std::vector<int64_t> ne;                                // indices of non-empty rounds
    for (int64_t r = 0; r < R; ++r) if (bounds[(size_t)r + 1] > bounds[(size_t)r]) ne.push_back(r);
    int rc;
    if (meta) {
        if (!ne.empty() && (rc = launch_round_meta(c, a, bounds))) return rc;
    } else if (!ne.empty()) {      // prologue: touch counts of the first round
        const int64_t r0 = ne[0];
        if ((rc = launch_round(c, a, 0, 0, bounds[(size_t)r0], bounds[(size_t)r0 + 1], 1, apply_p))) return rc;
    }
    // kernel timing (yue_set_kernel_timing): one HIP-event bracket around all round launches of this call
    const bool timed = c->timing_stride > 0 && !ne.empty();
    if (timed) {
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t s, t;
            HIPCHK(hipEventCreate(&s));
            HIPCHK(hipEventCreate(&t));
            c->ev_pool.emplace_back(s, t);
            c->ev_triplets.push_back(0);
            c->ev_launches.push_back(0);
        }
        HIPCHK(hipEventRecord(c->ev_pool[c->ev_used].first, c->stream));
    }
    size_t pos = 0;
    for (int64_t r = 0; r < R; ++r) {
        if (pos < ne.size() && ne[pos] == r) {
            const int64_t e0 = bounds[(size_t)r], e1 = bounds[(size_t)r + 1];
            int64_t n0 = 0, n1 = 0;
            if (pos + 1 < ne.size()) { n0 = bounds[(size_t)ne[pos + 1]]; n1 = bounds[(size_t)ne[pos + 1] + 1]; }
            // (user_blocks: the rounds are blocks of whole users and this GPU holds all their events -- sequential user rows)
            const bool seq_user = meta && user_blocks && !c->bigq;
            if ((rc = seq_user ? launch_round_u(c, a, (*user_blocks)[(size_t)r], (*user_blocks)[(size_t)r + 1], e0, e1, r)
                      : meta ? launch_round_m(c, a, e0, e1, r) : launch_round(c, a, e0, e1, n0, n1, (int)(pos & 1), apply_p))) return rc;
            ++pos;
        }
        if ((rc = after_round(r))) return rc;
    }
    if (timed) {
        HIPCHK(hipEventRecord(c->ev_pool[c->ev_used].second, c->stream));
        c->ev_triplets[c->ev_used] = bounds.back() - bounds.front();
        c->ev_launches[c->ev_used] = (int64_t)ne.size();
        c->ev_used++;
    }
    return YUE_OK;
}

// After a failed round launch the touch counters and difference buffers may hold a half-finished round: clear them,
// so that the next call starts from a clean state instead of silently producing wrong factors.
void reset_round_state(yue_ctx *c) {
    (void)hipStreamSynchronize(c->stream);
    if (c->cnt0.p) (void)hipMemsetAsync(c->cnt0.p, 0, c->cnt0.n * sizeof(unsigned long long), c->stream);
    if (c->cnt1.p) (void)hipMemsetAsync(c->cnt1.p, 0, c->cnt1.n * sizeof(unsigned long long), c->stream);
    if (c->cntp0.p) (void)hipMemsetAsync(c->cntp0.p, 0, c->cntp0.n * sizeof(uint32_t), c->stream);
    if (c->cntp1.p) (void)hipMemsetAsync(c->cntp1.p, 0, c->cntp1.n * sizeof(uint32_t), c->stream);
    if (c->dP.p) (void)hipMemsetAsync(c->dP.p, 0, c->dP.n * sizeof(float), c->stream);
    if (c->dQ.p) (void)hipMemsetAsync(c->dQ.p, 0, c->dQ.n * sizeof(float), c->stream);
    (void)hipStreamSynchronize(c->stream);
}


}  // namespace

namespace yue_host {
int zero_scalars(yue_ctx *c) {
    HIPCHK(hipMemsetAsync(c->scal.p, 0, (yue::kNllSlots + 8) * sizeof(double), c->stream));
    return YUE_OK;
}

int read_scalars(yue_ctx *c, double *nll, double *sp, double *sq) {
    double *sc = c->scal.p + yue::kNllSlots;
    hipLaunchKernelGGL(yue::k_sum_slots, dim3(1), dim3(64), 0, c->stream, c->scal.p, yue::kNllSlots, sc);
    double h[3];
    HIPCHK(hipMemcpyAsync(h, sc, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (nll) *nll = h[0];
    if (sp) *sp = h[1];
    if (sq) *sq = h[2];
    return YUE_OK;
}

int sumsq_async(yue_ctx *c) {
    double *sc = c->scal.p + yue::kNllSlots;
    HIPCHK(hipMemsetAsync(sc + 1, 0, 2 * sizeof(double), c->stream));
    hipLaunchKernelGGL(yue::k_sumsq, dim3(2048), dim3(256), 0, c->stream, c->P.p, c->m * c->k, sc + 1);
    hipLaunchKernelGGL(yue::k_sumsq, dim3(2048), dim3(256), 0, c->stream, c->Q.p, c->n * c->k, sc + 2);
    return YUE_OK;
}

int upload_triplets(yue_ctx *c, const int32_t *u, const int32_t *i, const int32_t *j, int64_t T, bool validate) {
    for (int64_t t = 0; validate && t < T; ++t) {
        if (u[t] < 0 || u[t] >= c->m || i[t] < 0 || i[t] >= c->n || j[t] >= c->n)
            return fail(YUE_ERR_ARG, "triplet " + std::to_string(t) + " out of range");
    }
    HIPCHK(c->xu.resize((size_t)T + yue::kHeaderSlack)); HIPCHK(c->xi.resize((size_t)T + yue::kHeaderSlack)); HIPCHK(c->xj.resize((size_t)T + yue::kHeaderSlack));   // (+ slack: header blocks)
    HIPCHK(hipMemcpyAsync(c->xu.p, u, T * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->xi.p, i, T * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->xj.p, j, T * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    return YUE_OK;
}
}  // namespace yue_host
using yue_host::zero_scalars; using yue_host::read_scalars; using yue_host::sumsq_async; using yue_host::upload_triplets;

extern "C" {

int yue_bpr_replay(yue_ctx *c, const int32_t *u, const int32_t *i, const int32_t *j, int64_t T,
                   double lr, double regU, double regI, double *nll_out) {
    if (!c || !c->have_factors) return fail(YUE_ERR_ARG, "yue_bpr_replay: no factors uploaded");
    if (T < 0 || (T > 0 && (!u || !i || !j))) return fail(YUE_ERR_ARG, "yue_bpr_replay: bad triplet arrays");
    HIPCHK(hipSetDevice(c->device));
    if (!c->opt_replay_levels) {
        // default: one dataflow launch (chain_kernels.hpp); ids are checked on the device, the host only copies the stream up
        int rc = upload_triplets(c, u, i, j, T, false);
        if (rc) return rc;
        if ((rc = zero_scalars(c))) return rc;
        if ((rc = yue_host::chain_stream(c, T, lr, regU, regI))) return rc;
        return read_scalars(c, nll_out, nullptr, nullptr);
    }
    // option replay_levels (the round-1 path, kept for comparison):
    // Dependency levels: a triplet runs one level after the latest earlier triplet that touches
    // P[u], Q[i] or Q[j].  Triplets of a level are pairwise row-disjoint, so running a level
    // concurrently gives exactly the state of the sequential loop (BPR.py:42-58).
    std::vector<int32_t> lastP((size_t)c->m, 0), lastQ((size_t)c->n, 0), lvl((size_t)T);
    int32_t nlev = 0;
    for (int64_t t = 0; t < T; ++t) {
        if (u[t] < 0 || u[t] >= c->m || i[t] < 0 || i[t] >= c->n || j[t] >= c->n) return fail(YUE_ERR_ARG, "triplet " + std::to_string(t) + " out of range");
        if (j[t] < 0) { lvl[(size_t)t] = 0; continue; }
        if (i[t] == j[t]) return fail(YUE_ERR_ARG, "triplet " + std::to_string(t) + ": i == j");
        int32_t l = std::max(lastP[(size_t)u[t]], std::max(lastQ[(size_t)i[t]], lastQ[(size_t)j[t]])) + 1;
        lvl[(size_t)t] = l;
        lastP[(size_t)u[t]] = lastQ[(size_t)i[t]] = lastQ[(size_t)j[t]] = l;
        nlev = std::max(nlev, l);
    }
    std::vector<int64_t> lptr((size_t)nlev + 2, 0);
    for (int64_t t = 0; t < T; ++t) if (lvl[(size_t)t] > 0) lptr[(size_t)lvl[(size_t)t] + 1]++;
    for (int32_t l = 1; l <= nlev + 0; ++l) lptr[(size_t)l + 1] += lptr[(size_t)l];
    const int64_t Tv = lptr[(size_t)nlev + 1];
    std::vector<int32_t> pu((size_t)std::max<int64_t>(Tv, 1)), pi((size_t)std::max<int64_t>(Tv, 1)), pj((size_t)std::max<int64_t>(Tv, 1));
    {
        std::vector<int64_t> cur(lptr.begin(), lptr.end());
        for (int64_t t = 0; t < T; ++t) {
            const int32_t l = lvl[(size_t)t];
            if (l == 0) continue;
            const int64_t o = cur[(size_t)l]++;
            pu[(size_t)o] = u[t]; pi[(size_t)o] = i[t]; pj[(size_t)o] = j[t];
        }
    }
    int rc = upload_triplets(c, pu.data(), pi.data(), pj.data(), Tv, true);
    if (rc) return rc;
    yue::TrainArgs a = make_args(c, lr, regU, regI);
    a.ev_u = c->xu.p; a.ev_i = c->xi.p; a.ev_j = c->xj.p;
    if ((rc = zero_scalars(c))) return rc;
    c->replay_levels = nlev;
    for (int32_t l = 1; l <= nlev; ++l) {
        const int64_t e0 = lptr[(size_t)l], e1 = lptr[(size_t)l + 1];
        if (e1 > e0) launch_level(c, a, e0, e1);
    }
    HIPCHK(hipGetLastError());
    return read_scalars(c, nll_out, nullptr, nullptr);
}

int yue_bpr_rounds(yue_ctx *c, const int32_t *u, const int32_t *i, const int32_t *j, const int64_t *round_ptr, int64_t n_rounds,
                   double lr, double regU, double regI, double *nll_out) {
    if (!c || !c->have_factors) return fail(YUE_ERR_ARG, "yue_bpr_rounds: no factors uploaded");
    if (!round_ptr || n_rounds < 0) return fail(YUE_ERR_ARG, "yue_bpr_rounds: bad round_ptr");
    HIPCHK(hipSetDevice(c->device));
    const int64_t T = round_ptr[n_rounds];
    if (round_ptr[0] != 0) return fail(YUE_ERR_ARG, "yue_bpr_rounds: round_ptr[0] must be 0");
    for (int64_t r = 0; r < n_rounds; ++r) if (round_ptr[r + 1] < round_ptr[r]) return fail(YUE_ERR_ARG, "yue_bpr_rounds: round_ptr must be non-decreasing");
    if (c->m * (int64_t)c->k * 4 >= (1ll << 31)) {
        // the round kernel addresses P relative to the smallest user of a wave's batch with 31-bit offsets
        const int tpw = tpw_of(c, false);
        for (int64_t r = 0; r < n_rounds; ++r)
            for (int64_t b = round_ptr[r]; b < round_ptr[r + 1]; b += tpw) {
                int32_t lo = u[b], hi = u[b];
                for (int64_t t = b; t < std::min(round_ptr[r + 1], b + tpw); ++t) { lo = std::min(lo, u[t]); hi = std::max(hi, u[t]); }
                if ((int64_t)(hi - lo) * c->k * 4 >= (1ll << 31)) return fail(YUE_ERR_ARG, "yue_bpr_rounds: users of neighbouring triplets are more than 2 GiB of factor rows apart; group the triplets by user");
            }
    }
    int rc = upload_triplets(c, u, i, j, T, true);
    if (rc) return rc;
    yue::TrainArgs a = make_args(c, lr, regU, regI);
    a.ev_u = c->xu.p; a.ev_i = c->xi.p; a.ev_j = c->xj.p;
    if ((rc = zero_scalars(c))) return rc;
    std::vector<int64_t> bounds(round_ptr, round_ptr + n_rounds + 1);
    if ((rc = run_rounds(c, a, bounds, 1, [](int64_t) { return YUE_OK; }))) { reset_round_state(c); return rc; }
    HIPCHK(hipGetLastError());
    return read_scalars(c, nll_out, nullptr, nullptr);
}

int yue_cune_steps(yue_ctx *c, const int32_t *u, const int32_t *i, const int32_t *k, const int32_t *j, int64_t T,
                   double s, double lr, double regU, double regI, double *loss_out) {
    if (!c || !c->have_factors) return fail(YUE_ERR_ARG, "yue_cune_steps: no factors uploaded");
    if (T < 0 || (T > 0 && (!u || !i || !k || !j || !loss_out)) || !(s > 0.0)) return fail(YUE_ERR_ARG, "yue_cune_steps: bad argument");
    if (T == 0) return YUE_OK;
    HIPCHK(hipSetDevice(c->device));
    for (int64_t t = 0; t < T; ++t) {
        if (u[t] < 0 || u[t] >= c->m || i[t] < 0 || i[t] >= c->n || j[t] < 0 || j[t] >= c->n || k[t] >= c->n) return fail(YUE_ERR_ARG, "yue_cune_steps: step " + std::to_string(t) + " out of range");
        if (i[t] == j[t] || k[t] == i[t]) return fail(YUE_ERR_ARG, "yue_cune_steps: step " + std::to_string(t) + ": i must differ from k and from j (k == j is allowed, as in the reference)");
    }
    int rc = upload_triplets(c, u, i, j, T, true);
    if (rc) return rc;
    HIPCHK(c->xk.resize(T)); HIPCHK(c->x_loss.resize(T));
    HIPCHK(hipMemcpyAsync(c->xk.p, k, T * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    yue::CuneArgs a{};
    a.P = c->P.p; a.Q = c->Q.p; a.k = c->k; a.u = c->xu.p; a.i = c->xi.p; a.kk = c->xk.p; a.j = c->xj.p; a.T = T;
    a.inv_s = 1.0 / s; a.inv_s32 = (float)(1.0 / s); a.lr = lr;
    a.ru = (float)(lr * regU); a.ri = (float)(lr * regI);          // CUNE.py:156: python-float product, cast to float32 by NumPy
    a.loss_out = c->x_loss.p;
    switch (kr_of(c->k)) {
        case 1: hipLaunchKernelGGL(yue::k_cune_steps<1>, dim3(1), dim3(64), 0, c->stream, a); break;
        case 2: hipLaunchKernelGGL(yue::k_cune_steps<2>, dim3(1), dim3(64), 0, c->stream, a); break;
        default: hipLaunchKernelGGL(yue::k_cune_steps<4>, dim3(1), dim3(64), 0, c->stream, a); break;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(loss_out, c->x_loss.p, T * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return YUE_OK;
}

int yue_adam_reset(yue_ctx *c) {
    if (!c || !c->have_factors) return fail(YUE_ERR_ARG, "yue_adam_reset: no factors uploaded");
    HIPCHK(hipSetDevice(c->device));
    const size_t mk = (size_t)c->m * c->k, nk = (size_t)c->n * c->k;
    HIPCHK(c->aU_m.resize(mk)); HIPCHK(c->aU_v.resize(mk)); HIPCHK(c->aV_m.resize(nk)); HIPCHK(c->aV_v.resize(nk));
    HIPCHK(hipMemsetAsync(c->aU_m.p, 0, mk * sizeof(float), c->stream)); HIPCHK(hipMemsetAsync(c->aU_v.p, 0, mk * sizeof(float), c->stream));
    HIPCHK(hipMemsetAsync(c->aV_m.p, 0, nk * sizeof(float), c->stream)); HIPCHK(hipMemsetAsync(c->aV_v.p, 0, nk * sizeof(float), c->stream));
    HIPCHK(hipMemsetAsync(c->dP.p, 0, mk * sizeof(float), c->stream)); HIPCHK(hipMemsetAsync(c->dQ.p, 0, nk * sizeof(float), c->stream));
    c->adam_m = c->m; c->adam_n = c->n; c->adam_k = c->k;
    return YUE_OK;
}

int yue_adam_step(yue_ctx *c, const int32_t *u, const int32_t *i, const int32_t *j, int64_t T, double lr, double reg, int64_t step, double *loss_out) {
    if (!c || !c->have_factors) return fail(YUE_ERR_ARG, "yue_adam_step: no factors uploaded");
    if (T <= 0 || !u || !i || !j || step < 1) return fail(YUE_ERR_ARG, "yue_adam_step: bad argument (T > 0, step >= 1)");
    HIPCHK(hipSetDevice(c->device));
    if (c->adam_m != c->m || c->adam_n != c->n || c->adam_k != c->k) { const int rc0 = yue_adam_reset(c); if (rc0) return rc0; }
    for (int64_t t = 0; t < T; ++t)
        if (u[t] < 0 || u[t] >= c->m || i[t] < 0 || i[t] >= c->n || j[t] < 0 || j[t] >= c->n) return fail(YUE_ERR_ARG, "yue_adam_step: triplet " + std::to_string(t) + " out of range");
    int rc = upload_triplets(c, u, i, j, T, true);
    if (rc) return rc;
    if ((rc = zero_scalars(c))) return rc;
    yue::MbArgs a{};
    a.U = c->P.p; a.V = c->Q.p; a.gU = c->dP.p; a.gV = c->dQ.p; a.k = c->k; a.u = c->xu.p; a.i = c->xi.p; a.j = c->xj.p; a.T = T;
    a.reg = (float)reg; a.loss_slots = c->scal.p;
    const dim3 grid((unsigned)(((T + 31) / 32 + 3) / 4));
    switch (kr_of(c->k)) {
        case 1: hipLaunchKernelGGL(yue::k_mb_grad<1>, grid, dim3(256), 0, c->stream, a); break;
        case 2: hipLaunchKernelGGL(yue::k_mb_grad<2>, grid, dim3(256), 0, c->stream, a); break;
        default: hipLaunchKernelGGL(yue::k_mb_grad<4>, grid, dim3(256), 0, c->stream, a); break;
    }
    // tf.train.AdamOptimizer defaults; lr_t as _prepare / _apply_sparse_shared form it
    const double b1 = 0.9, b2 = 0.999;
    const float lr_t = (float)(lr * std::sqrt(1.0 - std::pow(b2, (double)step)) / (1.0 - std::pow(b1, (double)step)));
    const int64_t mk = c->m * (int64_t)c->k, nk = c->n * (int64_t)c->k;
    hipLaunchKernelGGL(yue::k_adam, dim3((unsigned)std::min<int64_t>(8192, (mk + 255) / 256)), dim3(256), 0, c->stream, c->P.p, c->aU_m.p, c->aU_v.p, c->dP.p, mk, lr_t, 0.9f, 0.999f, 1e-8f);
    hipLaunchKernelGGL(yue::k_adam, dim3((unsigned)std::min<int64_t>(8192, (nk + 255) / 256)), dim3(256), 0, c->stream, c->Q.p, c->aV_m.p, c->aV_v.p, c->dQ.p, nk, lr_t, 0.9f, 0.999f, 1e-8f);
    HIPCHK(hipGetLastError());
    return read_scalars(c, loss_out, nullptr, nullptr);
}

int yue_sample_negatives(yue_ctx *c, uint64_t seed, uint32_t epoch, int32_t *j_out) {
    if (!c || !c->have_inter || !j_out) return fail(YUE_ERR_ARG, "yue_sample_negatives: no interactions uploaded");
    HIPCHK(hipSetDevice(c->device));
    yue::TrainArgs a = make_args(c, 0, 0, 0);
    a.seed = seed; a.epoch = epoch;
    if (c->E > 0) {
        hipLaunchKernelGGL(yue::k_sample, dim3((unsigned)((c->E + 255) / 256)), dim3(256), 0, c->stream, a, c->E);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(j_out, c->ev_j.p, c->E * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return YUE_OK;
}

int yue_sumsq(yue_ctx *c, double *sp, double *sq) {
    if (!c || !c->have_factors) return fail(YUE_ERR_ARG, "yue_sumsq: no factors uploaded");
    HIPCHK(hipSetDevice(c->device));
    int rc = sumsq_async(c);
    if (rc) return rc;
    double h[2];
    HIPCHK(hipMemcpyAsync(h, c->scal.p + yue::kNllSlots + 1, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (sp) *sp = h[0];
    if (sq) *sq = h[1];
    return YUE_OK;
}

int yue_bpr_epoch(yue_ctx *c, uint64_t seed, uint32_t epoch, int64_t round_events, double lr, double regU, double regI,
                  double *nll_out, double *sumsqP_out, double *sumsqQ_out) {
    if (!c || !c->have_factors || !c->have_inter) return fail(YUE_ERR_ARG, "yue_bpr_epoch: upload factors and interactions first");
    if (round_events < 0) return fail(YUE_ERR_ARG, "yue_bpr_epoch: round_events must be positive (or 0 for the device's default)");
    HIPCHK(hipSetDevice(c->device));
    yue::TrainArgs a = make_args(c, lr, regU, regI);
    a.seed = seed + 0x632BE59BD9B4E019ull * (uint64_t)c->rank;   // independent stream per item shard
    a.epoch = epoch;
    int rc = zero_scalars(c);
    if (rc) return rc;
    const int64_t E = c->E;
    // negatives of the whole epoch in one pass
    if (E > 0) hipLaunchKernelGGL(yue::k_sample, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, c->stream, a, E);
    if (c->opt_epoch_exact) {
        // exact sequential semantics over the epoch's triplets (the reference's loop, BPR.py:42-58, on the device sampler's
        // negatives): one dataflow launch, no rounds.  One GPU only: the order of the whole stream is the semantics.
        if (yue_host::on_communicator(c) && c->nranks > 1) return fail(YUE_ERR_ARG, "yue_bpr_epoch: option epoch_exact runs on one GPU (the sequential order spans all item shards)");
        if ((rc = yue_host::chain_epoch(c, lr, regU, regI))) return rc;
        if ((rc = sumsq_async(c))) return rc;
        return read_scalars(c, nll_out, sumsqP_out, sumsqQ_out);
    }
    // Rounds are blocks of whole users of about round_events events: a user never straddles rounds, so a
    // block's user-row differences are only needed again in the next epoch -- they stay in dP and a group of
    // blocks is applied by k_apply_range.  On a communicator the group is first summed over the ranks, and
    // both run on the second stream while the compute stream goes on with the next group's rounds (other
    // users' rows of P and dP only; item rows are rank-local).  The block width comes from job-wide counts:
    // the same blocks on every rank.
    std::vector<int64_t> bounds;
    double tot[2] = {(double)E, (double)c->n};            // job-wide events and item rows
    if (yue_host::on_communicator(c)) {
        if ((rc = yue_allreduce_f64(c, tot, 2))) return rc;
        if ((rc = zero_scalars(c))) return rc;            // the all-reduce used the scalar scratch
    }
    const double etot = tot[0];
    if (round_events == 0 && (rc = default_round_events(c, tot[1] / c->nranks, &round_events))) return rc;
    int64_t ub = 1, group = 1;
    if ((rc = yue_epoch_plan(c->m, c->k, round_events, etot, c->nranks, &ub, &group, nullptr))) return rc;
    // one GPU: nobody waits for the user rows of a group (a user's events all lie in one round, P[u] is not read again this epoch),
    // so the differences are applied once behind the last round instead of group by group between the rounds
    if (!yue_host::on_communicator(c)) group = std::max<int64_t>(1, (c->m + ub - 1) / ub);
    // communicator: the size of one all-reduce is a knob (option comm_group_mb; a user's dP is not needed before the next epoch,
    // so groups may be large -- RCCL reaches its bus bandwidth at tens of MB)
    else group = std::max<int64_t>(1, ((int64_t)c->opt_comm_group_mb << 20) / std::max<int64_t>(1, ub * c->k * 4));
    std::vector<int64_t> ublock;
    for (int64_t u0 = 0; u0 < c->m; u0 += ub) { ublock.push_back(u0); bounds.push_back(c->h_ev_ptr[(size_t)u0]); }
    ublock.push_back(c->m); bounds.push_back(E);
    const int64_t R = (int64_t)ublock.size() - 1;
    // One GPU: the rank holds all events of a user, one wave can walk them in order -- sequential user rows (k_round_u), nothing
    // left in dP.  (A communicator spreads a user's events over the ranks by item shard: there the user rows keep round semantics
    // and dP carries their differences to the all-reduce.)
    const bool seq_user = !yue_host::on_communicator(c) && c->opt_round_user_seq && meta_path_fits(c);
    if (seq_user) {
        HIPCHK(c->d_ev_ptr.resize((size_t)c->m + 1));
        if (!c->d_ev_ptr_valid) {
            HIPCHK(hipMemcpyAsync(c->d_ev_ptr.p, c->h_ev_ptr.data(), ((size_t)c->m + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
            c->d_ev_ptr_valid = true;
        }
        HIPCHK(c->margins.resize((size_t)std::max<int64_t>(E, 1)));
    }
    c->last_round_user_seq = 0;
    auto after = [&](int64_t r) -> int {
        if (seq_user && !c->bigq) { c->last_round_user_seq = 1; return YUE_OK; }          // (bigq is set by run_rounds before the first round)
        if ((r + 1) % group != 0 && r + 1 != R) return YUE_OK;
        const int64_t g_first = ublock[(size_t)(r - (r % group))], g_last = ublock[(size_t)r + 1];
        const int64_t first = g_first * c->k, count = (g_last - g_first) * c->k;
        const dim3 grid((unsigned)std::min<int64_t>(4096, (count + 255) / 256));
        if (!yue_host::on_communicator(c)) {
            // one GPU: nothing to wait for -- apply between two rounds on the compute stream (a few microseconds per
            // group; on a second stream its workgroups would take residency slots from the round kernel, which
            // fills the chip exactly, and push single launches into a second wave generation)
            hipLaunchKernelGGL(yue::k_apply_range, grid, dim3(256), 0, c->stream, c->P.p, c->dP.p, first, count);
            return YUE_OK;
        }
        HIPCHK(hipEventRecord(c->ev_rounds, c->stream));
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_rounds, 0));
        c->comm_collectives += 1;
        c->comm_bytes += (double)count * sizeof(float);
        if (const int rcr = yue_host::reduce_user_block(c, first, count, c->comm_stream)) return rcr;
        hipLaunchKernelGGL(yue::k_apply_range, grid, dim3(256), 0, c->comm_stream, c->P.p, c->dP.p, first, count);
        return YUE_OK;
    };
    c->comm_collectives = 0; c->comm_bytes = 0.0; c->comm_wait_ms = 0.0;
    if ((rc = run_rounds(c, a, bounds, 0, after, meta_path_fits(c), seq_user ? &ublock : nullptr))) { reset_round_state(c); return rc; }
    if (c->last_round_user_seq && E > 0)      // the loss of the epoch from the margins k_round_u left
        hipLaunchKernelGGL(yue::k_loss_margins, dim3(2048), dim3(256), 0, c->stream, c->margins.p, c->ev_j.p, E, c->scal.p);
    // the epoch's user rows must be complete before the loss sums and before the next epoch reads P
    if (yue_host::on_communicator(c)) {       // how long the compute stream has to wait for the last group's all-reduce + apply
        HIPCHK(hipEventRecord(c->ev_t_rounds, c->stream));
        HIPCHK(hipEventRecord(c->ev_t_comm, c->comm_stream));
    }
    HIPCHK(hipEventRecord(c->ev_comm, c->comm_stream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));      // the ONLY wait of the compute stream for the collective stream in an epoch
    c->comm_compute_waits = 1;
    HIPCHK(hipGetLastError());
    if ((rc = sumsq_async(c))) return rc;
    if ((rc = read_scalars(c, nll_out, sumsqP_out, sumsqQ_out))) return rc;
    if (yue_host::on_communicator(c)) {
        float ms = 0.f;
        HIPCHK(hipEventSynchronize(c->ev_t_comm));
        HIPCHK(hipEventElapsedTime(&ms, c->ev_t_rounds, c->ev_t_comm));
        c->comm_wait_ms = ms > 0.f ? ms : 0.0;
    }
    return YUE_OK;
}

int yue_epoch_plan(int64_t m, int k, int64_t round_events, double events_total, int nranks,
                   int64_t *user_block, int64_t *blocks_per_group, int64_t *n_blocks) {
    if (m <= 0 || k <= 0 || round_events <= 0 || nranks < 1 || !(events_total >= 0.0)) return fail(YUE_ERR_ARG, "yue_epoch_plan: bad argument");
    const double per_user = events_total / (double)nranks / (double)m;
    const int64_t ub = std::max<int64_t>(1, (int64_t)std::floor((double)round_events / std::max(per_user, 1e-9) + 0.5));
    if (user_block) *user_block = ub;
    // apply / all-reduce granularity: at least ~8 MB of user-factor differences per group
    if (blocks_per_group) *blocks_per_group = std::max<int64_t>(1, (8ll << 20) / std::max<int64_t>(1, ub * k * 4));
    if (n_blocks) *n_blocks = (m + ub - 1) / ub;
    return YUE_OK;
}

int yue_default_round_events(yue_ctx *c, int64_t *out) {
    if (!c || !c->have_factors || !out) return fail(YUE_ERR_ARG, "yue_default_round_events: upload factors first (the value depends on k)");
    HIPCHK(hipSetDevice(c->device));
    double n_rows = (double)c->n;
    if (yue_host::on_communicator(c)) {                    // collective on a communicator: every rank gets the same value
        const int rc = yue_allreduce_f64(c, &n_rows, 1);
        if (rc) return rc;
        n_rows /= c->nranks;
    }
    return default_round_events(c, n_rows, out);
}

int yue_set_kernel_timing(yue_ctx *c, int stride) {
    if (!c || stride < 0) return fail(YUE_ERR_ARG, "yue_set_kernel_timing: bad argument");
    c->timing_stride = stride;
    c->ev_used = 0;
    return YUE_OK;
}

int yue_get_kernel_timing(yue_ctx *c, double *total_ms, int64_t *launches, int64_t *triplets) {
    if (!c) return fail(YUE_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    double tot = 0.0;
    int64_t trip = 0, nl = 0;
    for (size_t t = 0; t < c->ev_used; ++t) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->ev_pool[t].first, c->ev_pool[t].second));
        tot += ms;
        trip += c->ev_triplets[t];
        nl += c->ev_launches[t];
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = nl;
    if (triplets) *triplets = trip;
    c->ev_used = 0;
    return YUE_OK;
}

int yue_get_scan_stats(yue_ctx *c, double *kernel_ms, int64_t *events, int64_t *rescored, int *used_bf16) {
    if (!c) return fail(YUE_ERR_ARG, "null context");
    if (kernel_ms) *kernel_ms = c->scan_ms;
    if (events) *events = c->scan_events;
    if (rescored) *rescored = c->scan_rescored;
    if (used_bf16) *used_bf16 = c->scan_used_bf16;
    return YUE_OK;
}

int yue_get_scan_work(yue_ctx *c, int64_t *tiles_scored, int64_t *tiles_total) {
    if (!c) return fail(YUE_ERR_ARG, "null context");
    if (tiles_scored) *tiles_scored = c->scan_tiles_done;
    if (tiles_total) *tiles_total = c->scan_tiles_total;
    return YUE_OK;
}

int yue_get_option(yue_ctx *c, const char *name, int64_t *value) {
    if (!c || !name || !value) return fail(YUE_ERR_ARG, "yue_get_option: null argument");
    const std::string key(name);
    if (key == "scan_f32") *value = c->opt_scan_f32;
    else if (key == "scan_batch") *value = c->opt_scan_batch;
    else if (key == "scan_two_phase") *value = c->opt_scan_two_phase;
    else if (key == "fism_lds") *value = c->opt_fism_lds;
    else if (key == "fism_inplace") *value = c->opt_fism_inplace;
    else if (key == "scan_growth") *value = c->opt_scan_growth;
    else if (key == "scan_filter_ub") *value = c->opt_scan_filter_ub;
    else if (key == "scan_streams") *value = c->opt_scan_streams;
    else if (key == "scan_slabs") *value = c->opt_scan_slabs;
    else if (key == "scan_streams_min_users") *value = c->opt_scan_streams_min_users;
    else if (key == "scan_last_chunks") *value = c->scan_chunks;
    else if (key == "scan_last_few_users") *value = c->scan_few_users;
    else if (key == "scan_last_settle") *value = c->scan_settle;
    else if (key == "topn_true") *value = c->opt_topn_true;
    else if (key == "round_stage") *value = c->opt_round_stage;
    else if (key == "round_last_stage_max") *value = c->last_stage_max;
    else if (key == "round_meta") *value = c->opt_round_meta;
    else if (key == "comm_group_mb") *value = c->opt_comm_group_mb;
    else if (key == "round_cus_reserved") *value = c->opt_round_cus_reserved;
    else if (key == "comm_last_compute_waits") *value = c->comm_compute_waits;
    else if (key == "round_user_seq") *value = c->opt_round_user_seq;
    else if (key == "round_fast") *value = c->opt_round_fast;
    else if (key == "round_last_user_seq") *value = c->last_round_user_seq;
    else if (key == "round_bucket") *value = c->opt_round_bucket;
    else if (key == "fold_blocks") *value = c->opt_fold_blocks;
    else if (key == "round_tpw") *value = c->opt_round_tpw;
    else if (key == "epoch_exact") *value = c->opt_epoch_exact;
    else if (key == "replay_levels") *value = c->opt_replay_levels;
    else if (key == "chain_waves") *value = c->opt_chain_waves;
    else if (key == "chain_split") *value = c->opt_chain_split;
    else if (key == "chain_fast") *value = c->opt_chain_fast;
    else if (key == "chain_xcd") *value = c->opt_chain_xcd;
    else if (key == "chain_last_us") *value = c->chain_kernel_us;
    else if (key == "chain_ring") *value = c->opt_chain_ring;
    else if (key == "chain_spin") *value = c->opt_chain_spin;
    else if (key == "chain_last_runs") *value = c->chain_runs;          // last exact launch: runs walked, waves launched
    else if (key == "chain_last_waves") *value = c->chain_waves;
    else if (key == "replay_last_levels") *value = c->replay_levels;    // last levelled replay: dependency levels = launches
    // which kernels yue_bpr_epoch runs for the uploaded factors: 0 k_round, 1 k_round_meta + k_round_m + k_round_fold
    else if (key == "round_path") *value = fold_path(c) ? 1 : 0;
    else return fail(YUE_ERR_ARG, "yue_get_option: unknown option " + key);
    return YUE_OK;
}

int yue_set_option(yue_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return fail(YUE_ERR_ARG, "yue_set_option: null argument");
    const std::string key(name);
    if (key == "scan_f32") { c->opt_scan_f32 = value != 0; return YUE_OK; }
    if (key == "scan_batch") { if (value != 0 && value != 1) return fail(YUE_ERR_ARG, "yue_set_option: scan_batch must be 0 or 1"); c->opt_scan_batch = (int)value; return YUE_OK; }
    if (key == "topn_true") { c->opt_topn_true = value != 0; return YUE_OK; }
    if (key == "scan_two_phase") { c->opt_scan_two_phase = value != 0; return YUE_OK; }
    if (key == "fism_lds") { c->opt_fism_lds = value != 0; return YUE_OK; }
    if (key == "fism_inplace") { c->opt_fism_inplace = value != 0; return YUE_OK; }
    if (key == "scan_streams_min_users") { if (value < 1024) return fail(YUE_ERR_ARG, "yue_set_option: scan_streams_min_users must be at least 1024"); c->opt_scan_streams_min_users = value; return YUE_OK; }
    if (key == "scan_slabs") { if (value < 2 || value > 64) return fail(YUE_ERR_ARG, "yue_set_option: scan_slabs must be 2..64"); c->opt_scan_slabs = (int)value; return YUE_OK; }
    if (key == "scan_streams") { if (value != 1 && value != 2) return fail(YUE_ERR_ARG, "yue_set_option: scan_streams must be 1 or 2"); c->opt_scan_streams = (int)value; return YUE_OK; }
    if (key == "scan_filter_ub") { if (value < 1 || value > 3) return fail(YUE_ERR_ARG, "yue_set_option: scan_filter_ub must be 1, 2 or 3"); c->opt_scan_filter_ub = (int)value; return YUE_OK; }
    if (key == "scan_growth") { if (value != 0 && (value < 2 || value > 64)) return fail(YUE_ERR_ARG, "yue_set_option: scan_growth must be 0 (automatic) or 2..64"); c->opt_scan_growth = (int)value; return YUE_OK; }
    if (key == "round_stage") {
        if (value < 0 || value > (int64_t)yue::kMetaStageMax) return fail(YUE_ERR_ARG, "yue_set_option: round_stage must be 0, 1 or 2..64");
        c->opt_round_stage = (int)value; return YUE_OK;
    }
    if (key == "round_meta") { c->opt_round_meta = value != 0; return YUE_OK; }
    if (key == "comm_group_mb") { if (value < 1 || value > 4096) return fail(YUE_ERR_ARG, "yue_set_option: comm_group_mb must be 1..4096"); c->opt_comm_group_mb = (int)value; return YUE_OK; }
    if (key == "round_cus_reserved") {
        // the compute stream is re-created with a CU mask that leaves the LAST `value` CUs of the device free: RCCL's kernels
        // (collective stream, no mask) find room beside round launches that would otherwise fill the chip exactly
        int cus = 0;
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
        if (value < 0 || value >= cus) return fail(YUE_ERR_ARG, "yue_set_option: round_cus_reserved must be 0 .. CUs - 1");
        HIPCHK(hipStreamSynchronize(c->stream));
        hipStream_t fresh = nullptr;
        if (value == 0) HIPCHK(hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
        else {
            std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
            for (int cu = 0; cu < cus - (int)value; ++cu) mask[(size_t)cu / 32] |= 1u << (cu % 32);
            HIPCHK(hipExtStreamCreateWithCUMask(&fresh, (uint32_t)mask.size(), mask.data()));
        }
        (void)hipStreamDestroy(c->stream);
        c->stream = fresh;
        c->opt_round_cus_reserved = (int)value;
        return YUE_OK;
    }
    if (key == "round_user_seq") { c->opt_round_user_seq = value != 0; return YUE_OK; }
    if (key == "round_fast") { c->opt_round_fast = value != 0; return YUE_OK; }
    if (key == "epoch_exact") { c->opt_epoch_exact = value != 0; return YUE_OK; }
    if (key == "replay_levels") { c->opt_replay_levels = value != 0; return YUE_OK; }
    if (key == "chain_split") { if (value < -1 || value > 1) return fail(YUE_ERR_ARG, "yue_set_option: chain_split must be -1, 0 or 1"); c->opt_chain_split = (int)value; return YUE_OK; }
    if (key == "chain_fast") { c->opt_chain_fast = value != 0; return YUE_OK; }
    if (key == "chain_xcd") { c->opt_chain_xcd = value != 0; return YUE_OK; }
    if (key == "chain_ring") { if (value != 0 && value != 8 && value != 16) return fail(YUE_ERR_ARG, "yue_set_option: chain_ring must be 0, 8 or 16"); c->opt_chain_ring = (int)value; return YUE_OK; }
    if (key == "chain_waves") { if (value < 0 || value > 8) return fail(YUE_ERR_ARG, "yue_set_option: chain_waves must be 0..8"); c->opt_chain_waves = (int)value; return YUE_OK; }
    if (key == "chain_spin") { if (value < 0 || value > 0x7fffffff) return fail(YUE_ERR_ARG, "yue_set_option: chain_spin out of range"); c->opt_chain_spin = value; return YUE_OK; }
    if (key == "round_bucket") { c->opt_round_bucket = value != 0; return YUE_OK; }
    if (key == "fold_blocks") { if (value < 1 || value > 65536) return fail(YUE_ERR_ARG, "yue_set_option: fold_blocks out of range"); c->opt_fold_blocks = (int)value; return YUE_OK; }
#ifdef YUE_STAMPS
    if (key == "debug_stamp_launch") { c->stamp_launch = value; c->update_launches = 0; return YUE_OK; }
#endif
    if (key == "round_tpw") {
        if (value != 0 && value != 2 && value != 4 && value != 8 && value != 16) return fail(YUE_ERR_ARG, "yue_set_option: round_tpw must be 0, 2, 4, 8 or 16");
        if (value == 8 && kr_of(c->k) == 4) return fail(YUE_ERR_ARG, "yue_set_option: round_tpw 8 needs k <= 128");
        if (value == 16 && kr_of(c->k) != 1) return fail(YUE_ERR_ARG, "yue_set_option: round_tpw 16 needs k <= 64");
        c->opt_round_tpw = (int)value;
        return YUE_OK;
    }
    return fail(YUE_ERR_ARG, "yue_set_option: unknown option " + key);
}

#ifdef YUE_STAMPS
// Diagnostic build only (make stamps; not part of include/yue_hip.h): the 8 wall_clock64 stamps
// (100 MHz) of every update wave of the launch chosen with yue_set_option("debug_stamp_launch", i).
int yue_debug_get_stamps(yue_ctx *c, unsigned long long *out, int64_t max_waves, int64_t *n_waves) {
    HIPCHK(hipStreamSynchronize(c->stream));
    const int64_t w = std::min(max_waves, c->stamp_waves);
    if (w > 0) HIPCHK(hipMemcpy(out, c->stamps.p, (size_t)w * 64, hipMemcpyDeviceToHost));
    *n_waves = w;
    return YUE_OK;
}
#endif

}  // extern "C"
