/*
 * yue_hip.h -- C ABI of libyue_hip.so: the MI355X (gfx950) BPR training + top-N scoring path
 * that replaces the NumPy loop of 0411tony/Yue behind its Recommender plugin surface.
 *
 * Plain pointers and sizes only; all pointers are HOST pointers owned by the caller and are
 * borrowed for the duration of the call (NumPy owns P/Q).  Device memory lives
 * behind the opaque context.  Every function returns 0 on success; otherwise a negative
 * status and yue_last_error() holds a thread-local message (the reference's convention is
 * print + exit(-1): tool/config.py:9-11, base/IterativeRecommender.py:64-66 -- the Python
 * shim does exactly that with the message).
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   yue_set_factors / yue_get_factors   state contract of IterativeRecommender.initModel
 *                                       (base/IterativeRecommender.py:36-39): P[m,k], Q[n,k] fp32 C-order
 *   yue_set_interactions                userListen + userRecord iteration of BPR.buildModel
 *                                       (recommender/cf/BPR.py:32-35,42-45; data/record.py:138-163)
 *   yue_bpr_replay                      the epoch body recommender/cf/BPR.py:42-58 on an explicit
 *                                       (u,i,j) stream, exact sequential semantics
 *   yue_cune_steps                      the two-level BPR loop of recommender/advanced/CUNE.py:126-172
 *   yue_bpr_rounds / yue_bpr_epoch      the same triplet update in rounds (DESIGN.md "S-round"),
 *                                       yue_bpr_epoch fuses the negative sampler of BPR.py:46-48
 *   yue_sumsq                           the regulariser sums of BPR.py:59
 *   yue_scores                          BPR.predict / IterativeRecommender.predict
 *                                       (recommender/cf/BPR.py:131-134, base/IterativeRecommender.py:58-60)
 *   yue_topn_scan                       the per-user mask + seed + overwrite-scan of
 *                                       IterativeRecommender.evalRanking (base/IterativeRecommender.py:96-145)
 *                                       and ranking_performance (:186-228)
 */
#ifndef YUE_HIP_H
#define YUE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct yue_ctx yue_ctx;

#define YUE_OK              0
#define YUE_ERR_ARG        -1   /* bad argument / call order */
#define YUE_ERR_HIP        -2   /* HIP runtime error */
#define YUE_ERR_FEW_ITEMS  -3   /* a user has fewer than N candidates (reference: IndexError) */
#define YUE_ERR_COMM       -4   /* RCCL error */

#define YUE_MAX_ATTEMPTS   64   /* negative-sampler attempts before a triplet is skipped */
#define YUE_UNIQUE_ID_BYTES 128

const char *yue_last_error(void);
int yue_version(void);

/* device = HIP ordinal.  HIP is initialised here, never earlier (fork-safety: yue.py:94-105). */
int yue_ctx_create(int device, yue_ctx **out);
int yue_ctx_destroy(yue_ctx *ctx);
int yue_sync(yue_ctx *ctx);

/* Factor matrices, fp32 row-major.  k <= 256, m and n below 2^31 rows, any size in bytes: item matrices of 2 GiB and more per GPU
 * are taken by yue_bpr_epoch's default path and by the scoring calls; yue_bpr_rounds, yue_bpr_replay, the option epoch_exact and
 * round_meta = 0 refuse them (their kernels address item rows with 31-bit byte offsets). */
int yue_set_factors(yue_ctx *ctx, const float *P, int64_t m, const float *Q, int64_t n, int k);
int yue_get_factors(yue_ctx *ctx, float *P, float *Q);

/*
 * Training interactions of this context's item range (all items on one GPU).
 *   indptr[m+1], indices[nnz]  sorted-unique listened items per user (negative rejection)
 *   ev_ptr[m+1], ev_i[E]       events in userRecord order: user-major, users by ascending id,
 *                              duplicates kept (one triplet per event, BPR.py:44)
 */
int yue_set_interactions(yue_ctx *ctx, const int64_t *indptr, const int32_t *indices,
                         const int64_t *ev_ptr, const int32_t *ev_i);

/* Exact sequential semantics of BPR.py:42-58 on explicit triplets, any order of users: one dataflow launch in which every
 * triplet waits for exactly the earlier triplets that share a row with it (chain_kernels.hpp; row ordinals and runs of
 * equal users are computed on the device, the host only copies the stream up).  j[t] < 0 skips the triplet.
 * nll_out = sum of -log(s).  Ids are checked on the device before anything is written. */
int yue_bpr_replay(yue_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *j, int64_t T,
                   double lr, double regU, double regI, double *nll_out);

/* The reference's LIVE path (recommender/cf/BPR.py:83-129, a TensorFlow-1 graph): one minibatch step on the fed triplets
 * (the reference feeds 512 events x 100 negatives, :65-81) -- loss = sum softplus(-(U[u].V[i] - U[u].V[j])) + reg * (l2_loss of
 * the three gathered row sets), then Adam (beta1 0.9, beta2 0.999, epsilon 1e-8, TF-1 sparse apply = dense Adam with zero
 * gradients on untouched rows) on both factor matrices, all in float32.  step = 1, 2, ... (Adam's bias correction).
 * yue_adam_reset clears the moments (also done implicitly when the factor shapes change).  PARITY UNPINNED: no TensorFlow
 * here; the checker is oracle/numpy_adam.py, a restatement of the graph as written. */
int yue_adam_reset(yue_ctx *ctx);
int yue_adam_step(yue_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *j, int64_t T,
                  double lr, double reg, int64_t step, double *loss_out);

/* CUNE's two-level BPR steps (reference recommender/advanced/CUNE.py:126-172) on explicit (u, i, k, j) steps, exact
 * sequential semantics in the given order: k[t] >= 0 is the step with a friends' item k ((i over k), then (k over j) with
 * margin and step scaled by 1/s, then the decays :156-159), k[t] < 0 the plain (i over j) step of :166-172.
 * loss_out[T] = every step's -log sigmoid term(s) on its final rows (the caller adds them up the way :161 / :175 do).
 * The user network / Word2Vec stage that produces the friends' items (CUNE.py:34-118) is not part of this library. */
int yue_cune_steps(yue_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *k, const int32_t *j, int64_t T,
                   double s, double lr, double regU, double regI, double *loss_out);

/* Explicit triplets, S-round semantics: rounds are round_ptr[r]..round_ptr[r+1] (n_rounds+1 entries). */
int yue_bpr_rounds(yue_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *j,
                   const int64_t *round_ptr, int64_t n_rounds,
                   double lr, double regU, double regI, double *nll_out);

/*
 * One epoch over the uploaded events; negatives from the device's counter-based sampler.
 * Rounds are blocks of whole users holding about round_events events (0 = the device's default,
 * yue_default_round_events): users per block = floor(round_events / (events per user) + 1/2) from the
 * job-wide event count, the same blocks on every rank of a communicator, where the blocks' user-factor
 * differences are all-reduced.  (yue_amd/dist.py: epoch_round_ptr restates the rule.)
 * Outputs: nll (sum of -log s over this rank's triplets), sums of squares of P and of this
 * rank's Q after the epoch (for BPR.py:59).  Any output pointer may be NULL.
 */
int yue_bpr_epoch(yue_ctx *ctx, uint64_t seed, uint32_t epoch, int64_t round_events,
                  double lr, double regU, double regI,
                  double *nll_out, double *sumsqP_out, double *sumsqQ_out);

/* The host-side schedule of yue_bpr_epoch, as a pure function (no device, no context): users per round
 * (user_block = floor(round_events / (events_total / nranks / m) + 1/2), at least 1), rounds per apply / all-reduce
 * group (at least ~8 MB of user-factor differences per collective) and the number of rounds.  Every rank computes
 * the same values from job-wide counts; tests check it against yue_amd/dist.py: epoch_block_plan. */
int yue_epoch_plan(int64_t m, int k, int64_t round_events, double events_total, int nranks,
                   int64_t *user_block, int64_t *blocks_per_group, int64_t *n_blocks);

/* The default round size of yue_bpr_epoch for the uploaded factors on this device.  One resident set of waves of the
 * round kernel takes 57,344 events on MI355X at k = 128 (49,152 for the kernels that finish contended rows inside
 * the launch: option round_fold / round_meta = 0, and yue_bpr_rounds); the default is up to 6 such sets, as long as
 * a round holds at most four events per item row of this rank (job-wide average on a communicator: the call is then
 * collective), at most 4 sets for rounds with fewer than two touches per item row -- 344,064 on BASELINE config 3,
 * 172,032 on config 2.  Results depend on the round size (DESIGN.md section 3 tabulates the distance from the
 * sequential loop against it): pass an explicit value where runs must be comparable across devices. */
int yue_default_round_events(yue_ctx *ctx, int64_t *out);

/* Negatives the device sampler draws for (seed, epoch): j_out[E], -1 where all attempts were rejected. */
int yue_sample_negatives(yue_ctx *ctx, uint64_t seed, uint32_t epoch, int32_t *j_out);

int yue_sumsq(yue_ctx *ctx, double *sumsqP_out, double *sumsqQ_out);

/* predict(user): scores of all n items in id order, fp32, k-ascending fused-multiply-add chain. */
int yue_scores(yue_ctx *ctx, int32_t user, float *out_n);

/*
 * evalRanking's selection for nu users.  Masked items per user come from mask_indptr[nu+1] /
 * mask_indices (rows sorted ascending, indexed by position in users[]); pass NULL for both to
 * mask the uploaded training items (evalRanking).  out_ids[nu*N], out_scores[nu*N], 1 <= N <= 100 (the
 * reference caps N at 100, :84-86).  Scores are the exact fp32 fma chain of yue_scores; for
 * k in {16,32,64,128} a bf16 MFMA tile pre-filters the pairs that can matter (same results).
 * Returns YUE_ERR_FEW_ITEMS if some user has fewer than N candidates (their rows are -1 / -inf).
 */
int yue_topn_scan(yue_ctx *ctx, const int32_t *users, int64_t nu, int N,
                  const int64_t *mask_indptr, const int32_t *mask_indices,
                  int32_t *out_ids, float *out_scores);

/* Timing of the dominant training kernel with HIP events on the library's stream: stride = 0 disables; any other
 * value brackets ALL round launches of each yue_bpr_epoch / yue_bpr_rounds call with one event pair (launch
 * boundaries and the interleaved user-row apply launches included).  yue_get_kernel_timing returns the summed
 * time, the round launches and the triplets inside the brackets since the last call. */
int yue_set_kernel_timing(yue_ctx *ctx, int stride);
int yue_get_kernel_timing(yue_ctx *ctx, double *total_ms, int64_t *launches_timed, int64_t *triplets_timed);
/* Last yue_topn_scan: time of its scoring kernel (HIP events), state-machine events, exact re-scores
 * done behind the bf16 pre-filter, and whether the bf16 pre-filter kernel ran (k in 16/32/64/128). */
int yue_get_scan_stats(yue_ctx *ctx, double *kernel_ms, int64_t *events, int64_t *rescored, int *used_bf16);
/* Last yue_topn_scan: 32-user x 32-item tiles the kernel actually scored, and the tiles of the full user x item product.  The
 * bf16 kernel skips a tile -- and stops a workgroup -- when a norm bound shows that none of its exact scores can reach any of the
 * users' thresholds (||P_u|| * max ||Q_i|| <= threshold; exact: the lists do not change). */
int yue_get_scan_work(yue_ctx *ctx, int64_t *tiles_scored, int64_t *tiles_total);

/* Tuning / diagnostic knobs (results do not depend on them, except that round_stage changes the order of some fp32 sums):
 *   "scan_f32"  1 = always score with the exact f32-MFMA kernel instead of bf16 pre-filter + exact re-score
 *   "scan_batch" bf16 scoring kernel: 0 (default) two item tiles per loop iteration and 256 users per workgroup, 1 one tile and 128 users
 *   "round_tpw" events per wave in the training round kernel: 0 = default (8 for k <= 128 -- 16 for k <= 64 in yue_bpr_rounds --, else 4), 2, 4, 8, 16 (k <= 64 only)
 *   "round_stage" 1 (default): item rows touched a few times in a round collect their differences in staging rows,
 *               summed in ticket / event order when the row is rewritten -- up to 4 touches in yue_bpr_rounds; in the
 *               epoch path up to a bound sized from the round's mean touches per item row (2x the mean at k > 64,
 *               4x at k <= 64, between 4 and 64; read it back as "round_last_stage_max"); 2..64: that bound chosen
 *               explicitly (epoch path); 0: every contended row goes through float atomics
 *   "round_meta" 1 (default): yue_bpr_epoch takes the touch metadata of all rounds from one pre-pass per epoch
 *               (k_round_meta), round launches without a retire phase (k_round_m) and a fold launch behind each
 *               (k_round_fold) -- item shards of up to 8.4M rows (above 454,656: touches bucketed by item range first); 0: touches counted and contended rows finished
 *               inside the round launches (k_round, the kernel of yue_bpr_rounds).  Also moves yue_default_round_events.
 *   "round_bucket" 1: the bucketed pre-pass also for small catalogues (tests)
 *   "fold_blocks" workgroups of the fold launch (default 1536)
 *   "scan_two_phase" 1 (default): yue_topn_scan on catalogues of >= 16,384 items (k in {16,32,64,128}, N <= 64, bf16 path) scores
 *               the first 512 items with the fused kernel and the rest in chunks through k_scan_filter + k_scan_select; 0: the
 *               fused kernel for everything (same lists and scores)
 *   "scan_growth" 0 (default): each chunk of the two-phase scan ends at 2x or 8x the items scanned so far, chosen from the
 *               list-update rate of the first 512 items; 2..64: that factor
 *   "scan_filter_ub" form of k_scan_filter: 3 (default) = two blocks of 32 users per wave in workgroups of four waves, item rows by
 *               LDS-DMA (k = 64 / 128; other k: as 2); 2 = the same blocking, rows staged through registers; 1 = one block per
 *               wave, eight waves (round 3).  Same lists and scores
 *   "scan_streams" 2 (default): calls of at least "scan_streams_min_users" users (262,144) split them into four or more slabs that
 *               alternate between two streams -- one slab's selection beside the next slab's filter; 1: one stream
 *   "fism_lds"   1 (default): yue_fism_rounds keeps a user's working rows in LDS when they fit (k_fism_round_lds); 0: always the
 *               form with working rows in global memory and host-built item lists
 *   "fism_inplace" 1 (default): in k_fism_round_lds a row only ONE user of the round touches goes back to the model in place
 *               (no second read of the round-start row, no atomic adds; the round in front counts the users per row);
 *               0: every row through the difference buffers
 *   "chain_waves" exact path: workgroups per CU of the dataflow launch, 1..8 (0 = default: 1; 2 with chain_xcd)
 *   "chain_split" exact path: 1 = a run is walked by a GROUP of five waves (k_bpr_chain3: two keep the memory side -- headers,
 *               prefetch rings, version checks, polling; even / odd triplets --, one the dependency chain margin -> sigmoid ->
 *               user row, two store the updated item rows; hand-over through LDS; same results bit for bit); 0 = by one wave
 *               (k_bpr_chain); -1 (default) = the group for streams of at least 16 triplets per run (BASELINE config 3: 940 ->
 *               607 ms per epoch), one wave otherwise (short runs: the group's hand-offs and run starts cost more than they save)
 *   "chain_fast"  exact path: 1 = the step's coefficient fp32(lr (1 - sigmoid(x))) in single precision and the margin as one
 *               64-lane sum -- the reference's ORDER of updates, not its last bit: factors within BASELINE.json's 1e-5 of the
 *               sequential loop (measured 3e-7..6e-7 on full configs 2, 3 and config 4's shard), 1.26e8 instead of 8.2e7 triplets/s
 *               on config 3; 0 (default) = every operation as the reference rounds it (bit-equal to the oracle)
 *   "chain_xcd"   exact path, wave group only: 1 = every working wave on ONE XCD (the other XCDs' workgroups leave at once), item rows
 *               handed over through that XCD's L2 instead of the memory side; same results; default 0
 *   "chain_ring"  exact path, wave group: triplets whose rows a loading wave keeps in flight, 8 (default) or 16 (k <= 128)
 *   "chain_spin"  exact path: polls a wave spends on one wait before it gives up with an error (0 = default: 2^22)
 *   "round_user_seq" epoch path on one GPU: 1 (default) = a wave owns a user and applies the user's triplets in the reference's
 *               order with P[u] in registers (k_round_u), only the item rows keep round semantics; 0 = user rows under round
 *               semantics too (k_round_m, differences through dP: the form a communicator always runs)
 *   "round_fast"  k_round_u: 1 (default) = the step's coefficient in single precision (as chain_fast); 0 = double precision
 *   "comm_group_mb" communicator: MB of user-factor differences per ncclAllReduce (groups of user blocks), 1..4096, default 8
 *               (what yue_epoch_plan announces); RCCL reaches its bus bandwidth at tens of MB
 *   "round_cus_reserved" CUs the compute stream leaves free (the stream is re-created with a CU mask) for RCCL's kernels beside
 *               round launches that otherwise fill the chip exactly; default 0
 * Behaviour switches:
 *   "epoch_exact" 1 = yue_bpr_epoch applies the epoch's triplets (device sampler's negatives) with the reference's exact
 *               sequential semantics (recommender/cf/BPR.py:42-58) instead of S-rounds: round_events is ignored, one GPU only
 *   "replay_levels" 1 = yue_bpr_replay by host-computed dependency levels, one launch per level (the round-1 path, kept for
 *               comparison; same results)
 * Read-only (yue_get_option): "chain_last_runs" / "chain_last_waves" (runs walked / waves launched by the last exact launch),
 *   "scan_last_few_users" (users of the last two-phase scan whose first 512 items held fewer than N candidates: they alone went
 *   through the fused kernel over all items), "chain_last_us" (HIP-event time of that launch alone, microseconds), "round_last_user_seq" (1: the last epoch ran k_round_u),
 *   "comm_last_compute_waits" (times the compute stream waited for the collective stream in the last epoch: 1),
 *   "replay_last_levels" (dependency levels of the last levelled replay), "scan_last_chunks" (filter + select launches of the
 *   last two-phase scan; 0: the fused kernel ran), "scan_last_settle" (1: most sampled users were settled against the catalogue's
 *   tail after the first chunk, the filter ran in the variant whose workgroups stop then), "round_last_stage_max" (largest staged block of the last epoch's pre-pass)
 * Behaviour switch (off by default = the reference's behaviour):
 *   "topn_true" 1 = yue_topn_scan returns a real top-N (descending, ties: lower item id first) instead of
 *               the reference's order-dependent overwrite-scan */
int yue_set_option(yue_ctx *ctx, const char *name, int64_t value);
/* Reads an option back; "round_path" = the kernels yue_bpr_epoch runs for the uploaded factors: 0 k_round,
 * 1 k_round_meta + k_round_m + k_round_fold. */
int yue_get_option(yue_ctx *ctx, const char *name, int64_t *value);

/*
 * FISM (reference recommender/cf/FISM.py) -- parity path: the reference's strictly
 * sequential epoch on the device, in the reference's types (item-history factors P float64 [n,k],
 * item factors Q float32 [n,k], item bias Bi float64 [n]).
 *   yue_fism_set_model / yue_fism_get_model   replace FISM.initModel's arrays (FISM.py:15-18) on / from the device
 *   yue_fism_epoch   replaces one pass of FISM.buildModel's loop (FISM.py:38-69): users in user_ptr order, users
 *                    with one event skipped; negs = the accepted negatives in processing order (rho per event,
 *                    drawn by the caller as FISM.py:50-53 does); coef[u] = pow(nu - 1, -alpha) (FISM.py:42).
 *                    Outputs: sum of 0.5*error^2 (:58) and {sum(P*P), sum(Q*Q), Bi.Bi} after the pass (:70).
 *   yue_fism_rounds  the throughput form of the same pass (ours; DESIGN.md section 10, oracle/numpy_fism.py: fism_rounds):
 *                    rounds of round_users consecutive users; every user of a round runs the reference's whole per-user
 *                    loop on the model as it was when the round started plus its own changes, the per-row differences of
 *                    the round's users are summed and added once.  round_users = 1 is yue_fism_epoch.  Same arguments
 *                    and outputs otherwise.
 *   yue_fism_scores  replaces FISM.predict (FISM.py:75-83) for a user whose training events are `items`.
 *   yue_fism_topn_scan  predict + the selection of base/IterativeRecommender.py:98-145 for nu users given as a
 *                    CSR of their training events (mask = those items).  YUE_ERR_FEW_ITEMS as yue_topn_scan.
 */
int yue_fism_set_model(yue_ctx *ctx, const double *P, const float *Q, const double *Bi, int64_t n, int k);
int yue_fism_get_model(yue_ctx *ctx, double *P, float *Q, double *Bi);
int yue_fism_epoch(yue_ctx *ctx, const int64_t *user_ptr, int64_t m, const int32_t *ev_i, const int32_t *negs, int64_t n_negs, int rho,
                   const double *coef, double lr, double regI, double regB, double *half_sq_out, double *sumsq3_out);
int yue_fism_rounds(yue_ctx *ctx, const int64_t *user_ptr, int64_t m, const int32_t *ev_i, const int32_t *negs, int64_t n_negs, int rho,
                    const double *coef, int64_t round_users, double lr, double regI, double regB, double *half_sq_out, double *sumsq3_out);
int yue_fism_scores(yue_ctx *ctx, const int32_t *items, int64_t n_items, double *out_n);
int yue_fism_topn_scan(yue_ctx *ctx, const int64_t *row_ptr, const int32_t *row_items, int64_t nu, int N, int32_t *out_ids, double *out_scores);

/* Multi-GPU (one process per GPU, RCCL over xGMI).  Rank 0 creates the id, the caller ships
 * the 128 bytes to the other ranks (any side channel), every rank calls yue_comm_init. */
int yue_comm_unique_id(void *id128_out);
int yue_comm_init(yue_ctx *ctx, const void *id128, int rank, int nranks);
/* Sum a double across ranks (loss terms); identity without a communicator. */
int yue_allreduce_f64(yue_ctx *ctx, double *vals, int count);
/* The last yue_bpr_epoch on a communicator: bytes this rank handed to ncclAllReduce (user-factor differences, fp32), the
 * number of collectives, the time the compute stream had to wait for the second stream (all-reduce + apply of the last
 * group) after its own last round launch had finished (HIP events), the communicator's rank count as RCCL reports it and
 * the RCCL version (ncclGetVersion).  Zeros / 1 without a communicator. */
int yue_get_comm_stats(yue_ctx *ctx, double *allreduce_bytes, int64_t *collectives, double *wait_ms, int *nranks, int *rccl_version);

#ifdef __cplusplus
}
#endif
#endif
