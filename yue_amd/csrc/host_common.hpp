// Host side of libyue_hip.so, shared by its translation units: the opaque context (device buffers, streams, options),
// error plumbing, and the few helpers more than one subsystem uses.  The C ABI is include/yue_hip.h.
//   core_host.hip   context, factors, interactions            bpr_host.hip   replay levels, S-rounds, epochs, CUNE, Adam, options
//   chain_host.hip  exact sequential semantics as dataflow    scan_host.hip  predict + evalRanking's selection
//   fism_host.hip   FISM                                      comm.hip       RCCL
#pragma once
#include "../../include/yue_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct ncclComm;

namespace yue_host {

int fail(int code, const std::string &msg);          // sets the thread-local message of yue_last_error(), returns code

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return yue_host::fail(YUE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t resize(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; n = 0; if (e != hipSuccess) return e; }
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

constexpr int kNllSlotsHost = 1024;   // == yue::kNllSlots (bpr_device.hpp; checked where both are visible)
constexpr int kHeaderSlackHost = 16;  // == yue::kHeaderSlack (round_kernels.hpp)

inline int kr_of(int k) { return k <= 64 ? 1 : k <= 128 ? 2 : 4; }   // registers per lane per row (64 lanes)

}  // namespace yue_host

using yue_host::DevBuf;

struct yue_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t m = 0, n = 0, E = 0, nnz = 0;
    int k = 0;
    bool have_factors = false, have_inter = false;
    DevBuf<float> P, Q, dP, dQ;
    DevBuf<float> margins;                       // [E] the events' margins of the running epoch (k_round_u -> k_loss_margins)
    DevBuf<unsigned long long> cnt0, cnt1;       // item-row touch counters of the even / odd round (total | remaining)
    DevBuf<uint32_t> cntp0, cntp1;               // user-row flushes of the even / odd round
    DevBuf<uint32_t> tab0, tab1;                 // staging-slot tables of the even / odd round (kStageMax words per item row)
    // staged item rows (2 per event of the widest round) live behind the n item rows in the Q allocation,
    // so that "new row in place" and "new row to my staging row" are the same store with another offset
    bool staged = false;                         // the running call uses the staging rows
    bool bigq = false;                           // ... and addresses item / staging rows through 64-bit pointers (2 GiB and more)
    // epoch path: touch metadata of all rounds from one pre-pass (round_kernels.hpp)
    DevBuf<uint32_t> meta_i, meta_j;
    DevBuf<unsigned long long> round_rows;
    DevBuf<uint2> fold, bk_touch;
    DevBuf<uint32_t> bk_ptr;
    DevBuf<int64_t> d_bounds;
    std::vector<int64_t> h_bounds;               // outlives the asynchronous upload
#ifdef YUE_STAMPS
    DevBuf<unsigned long long> stamps;           // diagnostic build: phase stamps of one chosen round launch
    int64_t stamp_launch = -1, update_launches = 0, stamp_waves = 0;
#endif
    DevBuf<int32_t> ev_u, ev_i, ev_j, indices;
    DevBuf<int64_t> indptr;
    DevBuf<int32_t> xu, xi, xj, xk;      // explicit triplets (replay / rounds), CUNE's fourth row
    DevBuf<double> x_loss;               // per-step losses (CUNE)
    DevBuf<float> aU_m, aU_v, aV_m, aV_v;  // Adam moments of the live TF-style path (yue_adam_step); gradients use dP / dQ
    int64_t adam_m = 0, adam_n = 0; int adam_k = 0;
    DevBuf<double> scal;                 // [kNllSlots] nll slots + [8] scalars
    std::vector<int64_t> h_ev_ptr;       // host copy: user -> first event
    // scoring scratch
    DevBuf<int32_t> s_users, s_ids, s_mask_idx, s_flags;
    DevBuf<int32_t> s_few, s_few_ids;   // two-phase scoring: positions / ids of the users with fewer than N candidates in the first chunk, their lists
    DevBuf<float> s_few_scores;
    int64_t scan_few_users = 0;         // such users of the last scan (read-only option scan_last_few_users)
    DevBuf<int64_t> s_mask_ptr;
    DevBuf<float> s_scores, s_row, s_norms;
    double scan_ms = 0.0;
    int64_t scan_events = 0, scan_rescored = 0, scan_tiles_done = 0, scan_tiles_total = 0;
    DevBuf<unsigned long long> s_work;
    DevBuf<unsigned short> s_qb;         // two-phase scoring: bf16 copy of the item factors
    DevBuf<uint32_t> s_masks;            // ... and the survivor words of a chunk
    int opt_scan_two_phase = 1;          // 0: always the fused kernels (k_topn_scan*)
    int64_t opt_scan_streams_min_users = 262144;   // ... from this many users of a call on (tests lower it)
    int opt_scan_slabs = 4;              // ... slabs of users at least, with two streams
    int opt_scan_streams = 2;            // two-phase path, 262,144 users and more: slabs of users alternate between two streams (1: one stream)
    int opt_scan_filter_ub = 3;          // two-phase path: form of k_scan_filter (scan_host.hip): 3 = two blocks of 32 users per wave, rows by DMA into LDS; 2 = rows through registers; 1 = one block, 8 waves
    int opt_scan_growth = 0;             // two-phase path: a chunk ends at this many times the items scanned so far (0: 2 or 8 by the first items' update rate)
    int scan_chunks = 0;                 // chunks the last scan ran through the filter / select pair
    int scan_used_bf16 = 0;
    // options (yue_set_option)
    int opt_scan_f32 = 0;                // 1: always the exact-f32-MFMA scoring kernel
    int opt_scan_batch = 0;              // bf16 scoring kernel: 0 = two tiles per iteration, 256 users per workgroup (default); 1 = one tile, 128 users
    int opt_round_tpw = 0;               // 0: default events per wave in the round kernel
    int opt_topn_true = 0;               // 1: yue_topn_scan returns a real top-N instead of the reference's overwrite-scan
    int scan_settle = 0;                 // the last two-phase scan used the filter variant that stops settled workgroups (read-only option scan_last_settle)
    int last_stage_max = 0;              // largest staged block of the last pre-pass (read-only option round_last_stage_max)
    int opt_round_stage = 1;             // 0: every contended item row goes through float atomics (no staging rows); 1: sized from the round's mean touches per row; 2..64: rows with up to that many touches are staged (epoch path)
    int opt_round_bucket = 0;            // 1: the bucketed pre-pass also for small catalogues (tests)
    int opt_fold_blocks = 1536;           // workgroups of k_round_fold
    int opt_round_user_seq = 1;          // epoch path on one GPU: 1 = a wave owns a user and walks the user's events in order (k_round_u); 0 = user rows with round semantics (k_round_m + dP)
    int opt_round_fast = 1;              // k_round_u: the step's coefficient in single precision (0: the reference's double-precision sigmoid)
    int last_round_user_seq = 0;         // the last epoch ran k_round_u (read-only option round_last_user_seq)
    int opt_comm_group_mb = 8;           // communicator: user-factor differences per all-reduce (groups of user blocks), MB (8 = what yue_epoch_plan announces; RCCL reaches its bus bandwidth at tens of MB: try 32..64 on a real node)
    int opt_round_cus_reserved = 0;      // CUs the compute stream leaves free (CU mask of the stream) for RCCL's kernels beside the round launches
    int64_t comm_compute_waits = 0;      // times the compute stream waited for the collective stream in the last epoch (read-only option comm_last_compute_waits)
    int opt_round_meta = 1;              // 0: the epoch path counts touches inside the round launches (k_round) as the explicit-rounds path does
    // kernel timing
    int timing_stride = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    std::vector<int64_t> ev_triplets, ev_launches;
    size_t ev_used = 0;
    // FISM (parity path): item-history factors (f64), item factors (f32), item bias (f64)
    DevBuf<double> fP, fBi, f_coef, f_x, f_hist, f_scores, f_out_sc;
    DevBuf<float> fQ;
    DevBuf<int64_t> f_ptr;
    DevBuf<int32_t> f_items, f_negs, f_ids, f_flags;
    // FISM rounds: touched-item lists, positions, working copies, difference buffers
    DevBuf<int64_t> f_uq_ptr, f_neg_ptr;
    DevBuf<int32_t> f_uq_items, f_loc_i, f_loc_j;
    DevBuf<float> f_wq, f_dQ;
    DevBuf<double> f_wp, f_wb, f_dP, f_dB;
    DevBuf<unsigned> f_cnt;              // [2][n] users of a round touching an item (k_fism_round_lds)
    int64_t fn = 0;
    int fk = 0;
    int opt_fism_inplace = 1;            // 0: every row a round's users touch goes through the difference buffers (round 3's form)
    int opt_fism_lds = 1;                // 0: yue_fism_rounds always through k_fism_round (working rows in global memory, item lists from the host)
    // exact path (chain_host.hip): touch keys / ordinals, runs, granule copies of the factor rows, control words
    DevBuf<uint32_t> ch_key, ch_val, ch_key2, ch_val2, ch_seg, ch_ord_i, ch_ord_j, ch_head, ch_incl, ch_ord_u, ch_rkey, ch_rval;
    DevBuf<int64_t> ch_run_ptr, d_ev_ptr;
    DevBuf<int32_t> ch_run_u;
    DevBuf<unsigned char> ch_tmp;
    DevBuf<unsigned> ch_Qv, ch_Pv;
    DevBuf<unsigned long long> ch_ctl;   // [0] run claim counter, [1] validation flags, [2] wait status
    DevBuf<unsigned long long> ch_stats; // diagnostic build (make chainstats) only
    bool d_ev_ptr_valid = false;
    int opt_epoch_exact = 0;             // 1: yue_bpr_epoch applies the epoch's triplets with exact sequential semantics (k_bpr_chain)
    int opt_replay_levels = 0;           // 1: yue_bpr_replay by host-computed dependency levels, one launch per level (the round-1 path)
    int opt_chain_split = -1;            // 1: a run is walked by a group of five waves (k_bpr_chain3: 2 x loads / dependency chain / 2 x stores); 0: by one wave (k_bpr_chain); -1: by the stream's mean run length
    int opt_chain_ring = 0;              // wave-group kernel: triplets whose rows a loading wave keeps in flight (0 = 8; 16 for k <= 128)
    int opt_chain_xcd = 0;               // wave-group kernel: 1 = all working waves on ONE XCD, rows handed over through its L2
    int opt_chain_fast = 0;              // 1: single-precision coefficient, one 64-lane sum per triplet (within 1e-5, not bit-equal)
    int opt_chain_waves = 0;             // workgroups per CU of the persistent launch (0: what fits, at most 8)
    int64_t opt_chain_spin = 0;          // polls per wait before a wave gives up (0: 2^22)
    int64_t chain_runs = 0, chain_waves = 0, chain_groups = 0, chain_kernel_us = 0, replay_levels = 0;
    hipEvent_t ev_chain0 = nullptr, ev_chain1 = nullptr;   // brackets of the dataflow launch (read-only option chain_last_us)
    // RCCL
    ncclComm *comm = nullptr;              // ncclComm_t (comm.hip)
    int rank = 0, nranks = 1;
    hipStream_t comm_stream = nullptr;   // (all-reduce +) user-row apply of yue_bpr_epoch run here, beside the next rounds
    hipEvent_t ev_rounds = nullptr, ev_comm = nullptr;
    hipEvent_t ev_t_rounds = nullptr, ev_t_comm = nullptr;   // timing pair of the last epoch on a communicator: end of its round launches / end of its last apply
    // TEST SEAM (libyue_hip_seam.so only, make test-seam: the product library has no way to set these): the collective of
    // reduce_user_block / yue_allreduce_f64 as a host-staged sum through a callback, so that the 2-rank block / group /
    // stream logic of yue_bpr_epoch runs on a one-GPU box (tests/test_gpu_multi.py).  dtype 0 = float, 1 = double.
    int (*seam_reduce)(void *host, int64_t count, int dtype, void *user) = nullptr;
    void *seam_user = nullptr;
    std::vector<float> seam_buf;
    // communicator statistics of the last yue_bpr_epoch (yue_get_comm_stats)
    int64_t comm_collectives = 0;
    double comm_bytes = 0.0, comm_wait_ms = 0.0;
    int comm_version = 0, comm_nranks_reported = 0;
    hipEvent_t ev_scan0 = nullptr, ev_scan1 = nullptr;      // brackets of the scoring kernel (yue_get_scan_stats)
};

namespace yue_host {
inline bool on_communicator(const yue_ctx *c) { return c->comm != nullptr || c->seam_reduce != nullptr; }
// bpr_host.hip: loss / scalar scratch shared by the training entry points
int zero_scalars(yue_ctx *c);
int read_scalars(yue_ctx *c, double *nll, double *sp, double *sq);
int sumsq_async(yue_ctx *c);
int upload_triplets(yue_ctx *c, const int32_t *u, const int32_t *i, const int32_t *j, int64_t T, bool validate);
// comm.hip: sum dP[first .. first + count) over the ranks on `stream` (in place); identity without a communicator
int reduce_user_block(yue_ctx *c, int64_t first, int64_t count, hipStream_t stream);
// chain_host.hip: exact sequential semantics over the uploaded events (negatives in ev_j) / over the stream in xu, xi, xj
int chain_epoch(yue_ctx *c, double lr, double regU, double regI);
int chain_stream(yue_ctx *c, int64_t T, double lr, double regU, double regI);
}  // namespace yue_host
