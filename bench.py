#!/usr/bin/env python3
"""Headline benchmark: BPR triplet-updates/s at k=128 (BASELINE.json metric) on N MI355X.

A "step" is one epoch: one pass of the counter-based sampler + BPR update path over the whole synthetic
event list (BASELINE config 3: 1M users x 200K items, 50 events/user = 50M triplets), factors and
interactions resident in HBM before the timed region starts.  For N > 1 the driver launches one
process per GPU (torch.distributed.run); every rank owns an item shard of the same shape (weak
scaling), users are replicated, user-factor differences are all-reduced over RCCL inside the
library; yue_amd/dist.py (standard-library TCP, no torch in the ranks) only ships the RCCL id, runs the
barriers and takes the max-over-ranks time.

For N > 1 the default workload is one GPU's share of BASELINE config 4 per rank (`c4shard`: 10M replicated users, a
125K-item shard and 60M events per rank -- at N = 8 exactly config 4; `--workload c3` keeps config 3 per rank), and the
line carries `comm` (all-reduce bytes and collectives per epoch and rank, the time the compute stream waited for the
collective stream, RCCL version and rank count).

One JSON line on rank 0, with `roofline` (dominant kernel, HIP events on the library's stream),
`cpu_baseline` (oracle/ timed on one host core, N=1 only) and, at N=1:
  `config.deviation_vs_sequential`  how far ONE epoch of the timed S-round semantics lands from the reference's sequential
                  loop run on the same factors and negatives (the exact device path, itself checked against the oracle);
  `secondary.exact`  the EXACT path (recommender/cf/BPR.py:40-62 semantics, chain_kernels.hpp) on config 3, config 2 and one rank's share of config 4:
                  triplets/s of a whole epoch, dependency depth of the stream, and the same stream through yue_bpr_replay
                  from host arrays;
  `secondary.c5`  the top-N scoring path (BASELINE config 5: all 1M users x 200K items, the factors the timed epochs left)
                  with its own roofline (MFMA) and CPU baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from yue_amd import synth            # noqa: E402
from yue_amd import _shim          # noqa: E402
from yue_amd._shim import Device   # noqa: E402
if os.environ.get('YUE_LIB'):         # tuning experiments: another build of the library
    _shim.LIB_PATH = os.environ['YUE_LIB']
# Rehearsal of the N > 1 flow on ONE GPU (tests/test_gpu_multi.py): every rank on device 0, the collective behind the test build's
# host-staged seam (libyue_hip_seam.so) instead of RCCL.  Exercises this file's multi-rank logic, not the interconnect: never a result.
SEAM_REHEARSAL = bool(os.environ.get('YUE_BENCH_SEAM'))
if SEAM_REHEARSAL:
    _shim.LIB_PATH = os.path.join(ROOT, 'yue_amd', 'csrc', 'libyue_hip_seam.so')
from yue_amd.dist import ControlPlane, attach_device   # noqa: E402

WORKLOADS = {
    # name: (users, items per GPU, events per user, k)
    'c3': (1000000, 200000, 50, 128),
    'c2': (100000, 50000, 50, 64),
    'tiny': (20000, 5000, 20, 128),
    # a catalogue beyond the plain pre-pass (454,656 item rows): C3's users and events on 1M items (bucketed pre-pass)
    'c3wide': (1000000, 1000000, 50, 128),
    # an item matrix beyond 2 GiB on one GPU (2.36 GB: 64-bit row addressing in the update launch, 141 item ranges in the pre-pass)
    'c3big': (1000000, 4600000, 50, 128),
    # one GPU's share of BASELINE config 4 (10M users x 1M items over 8 GPUs): replicated 5.1 GB of user factors,
    # a 125K-item shard, 60M of the 500M events; run with --force-comm to take the all-reduce path on one rank
    'c4shard': (10000000, 125000, 6, 128),
    # scoring (BASELINE config 5): all users x all items, top-20 selection, training items masked
    'c5': (1000000, 200000, 50, 128),
    'c5small': (65536, 200000, 50, 128),
    # FISM: the round form on a 100K-user problem, FISM.conf's rho / alpha / learning rate
    'fism': (100000, 20000, 20, 64),
}
MFMA_F32_PEAK = 157.3e12   # dense f32-input MFMA, MI355X (MI355X_MICROARCH.md)
MFMA_BF16_PEAK = 2.5e15    # dense bf16 MFMA
LR, REG_U, REG_I = 0.02, 0.01, 0.01
HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)


def algorithmic_bytes(k):
    # 3 rows read + 3 rows written (fp32) + one (u,i) int32 pair per triplet
    return 24 * k + 8


def cpu_baseline(data, P0, Q0, j_first, k, budget_s=8.0):
    """The reference's sequential loop on the host, bounded samples of the same workload:
    `value` = oracle/bpr_oracle.c on one core (exact semantics); beside it the loop as NumPy runs it
    (oracle/numpy_loop.py, what the reference executes per triplet minus its dict lookups) and the C
    loop raced Hogwild-style by the box's CPU share."""
    import oracle
    from oracle.numpy_loop import bpr_loop
    orc = oracle.Oracle()
    ev_u = np.repeat(np.arange(data['m'], dtype=np.int32), np.diff(data['ev_ptr']))
    probe = min(len(j_first), 200000)
    P, Q = P0.copy(), Q0.copy()
    t0 = time.perf_counter()
    orc.bpr_sequential(P, Q, ev_u[:probe], data['ev_i'][:probe], j_first[:probe], LR, REG_U, REG_I)
    rate = probe / (time.perf_counter() - t0)
    S = int(min(len(j_first), max(probe, rate * budget_s)))
    P, Q = P0.copy(), Q0.copy()
    t0 = time.perf_counter()
    orc.bpr_sequential(P, Q, ev_u[:S], data['ev_i'][:S], j_first[:S], LR, REG_U, REG_I)
    dt = time.perf_counter() - t0
    cpu_baseline.prefix = (S, P, Q)                       # the exact device path is checked against this state (secondary_exact)
    # NumPy loop: ~1e5 triplets/s
    Sn = min(len(j_first), 300000)
    P, Q = P0.copy(), Q0.copy()
    t0 = time.perf_counter()
    bpr_loop(P, Q, ev_u[:Sn], data['ev_i'][:Sn], j_first[:Sn], LR, REG_U, REG_I)
    dtn = time.perf_counter() - t0
    # Hogwild with the threads this process may use (16 per GPU on the bench boxes)
    try:
        threads = max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        threads = max(1, min(16, os.cpu_count() or 1))
    Sh = int(min(len(j_first), max(probe, rate * threads * 4.0)))
    P, Q = P0.copy(), Q0.copy()
    t0 = time.perf_counter()
    orc.bpr_hogwild(P, Q, ev_u[:Sh], data['ev_i'][:Sh], j_first[:Sh], LR, REG_U, REG_I, threads)
    dth = time.perf_counter() - t0
    return {'value': S / dt, 'unit': 'triplets/s', 'cores': 1, 'kind': 'port',
            'sample': 'first %d triplets of epoch 0 of the same workload, sequential loop of oracle/bpr_oracle.c '
                      '(restates recommender/cf/BPR.py:42-58), %.1f s on %s' % (S, dt, _cpu_name()),
            'numpy_loop': {'value': Sn / dtn, 'cores': 1, 'sample': 'first %d triplets, oracle/numpy_loop.py (the per-triplet NumPy statements of BPR.py:50-58; bit-equal to the reference on its goldens), %.1f s' % (Sn, dtn)},
            'hogwild': {'value': Sh / dth, 'cores': threads, 'sample': 'first %d triplets, the C loop raced by %d threads over slices of the stream (result depends on the interleaving), %.1f s' % (Sh, threads, dth)}}


def scan_kernel_label(dev, k):
    chunks = dev.get_option('scan_last_chunks')
    if chunks > 0:      # two-phase scan of the last call: the first 512 items fused, then `chunks` filter + select launches
        return ('k_topn_scan_bf16p (first 512 items) + %d x [k_scan_filter<K16=%d> (bf16 MFMA + margin filter -> survivor bits) + '
                'k_scan_select (exact re-score of the survivors, list per wave)]' % (chunks, k // 16))
    if dev.get_option('scan_batch') == 1:
        return 'k_topn_scan_bf16<K16=%d, 1 tile per iteration, 4 waves>' % (k // 16)
    return 'k_topn_scan_bf16p<K16=%d, 2 tiles per iteration, 8 waves>' % (k // 16)


_SEAM_KEEP = []


def seam_attach(dev, cp):
    """SEAM_REHEARSAL: the library's collective = a float sum over the control plane, staged through the host."""
    import ctypes as C
    fn_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)

    def reduce(host, count, dtype, user):
        try:
            arr = np.ctypeslib.as_array(((C.c_double if dtype else C.c_float) * count).from_address(host))
            cp.allreduce_sum(arr)
            return 0
        except Exception as exc:                                   # never let an exception cross the C boundary
            print('seam reduce failed:', exc, file=sys.stderr)
            return 1
    cb = fn_t(reduce)
    _SEAM_KEEP.append(cb)
    dev._lib.yue_seam_init.restype = C.c_int
    if dev._lib.yue_seam_init(dev._ctx, C.c_int(cp.rank), C.c_int(cp.world), cb, None) != 0:
        sys.exit('yue_seam_init failed')


def round_kernel_label(dev, k):
    kr = 1 if k <= 64 else 2 if k <= 128 else 4
    if dev.get_option('round_path') == 1 and dev.get_option('round_last_user_seq') == 1:
        return ('k_round_u<KR=%d> + k_round_fold<KR=%d> (one round: a wave per user walks the user\'s events in order on the pre-pass metadata, '
                'then the rewrite of the round\'s contended item rows)' % (kr, kr))
    if dev.get_option('round_path') == 1:
        return 'k_round_m<KR=%d> + k_round_fold<KR=%d> (one round: update launch on the pre-pass metadata + rewrite of its contended rows)' % (kr, kr)
    return 'k_round<KR=%d> (update + touch tickets of the next round)' % kr


def measured_traffic(workload, round_events):
    """(HBM bytes per launch of the dominant kernel, where the number comes from) from the newest committed
    rocprofv3 --pmc summary (profiles/r*_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes);
    None if that profile was taken on another workload / round size.  Not a live measurement of this run."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')), reverse=True):
        try:
            t = json.load(open(path))
            if t['workload'] == workload and t['round_events'] == round_events:
                return t['traffic_bytes_per_launch'], 'committed profile %s (%s)' % (os.path.relpath(path, ROOT), t.get('taken', 'rocprofv3 --pmc passes'))
        except (OSError, ValueError, KeyError):
            pass
    return None


def _cpu_name():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip() + ' (%d logical cpus)' % os.cpu_count()
    except OSError:
        pass
    return 'unknown cpu'


def scoring_cpu_baseline(dev, data, users, ids, N, n, budget_s):
    """oracle/ (restates base/IterativeRecommender.py:96-145) on one core, a bounded sample of the same users;
    its lists must equal the GPU's."""
    import oracle
    orc = oracle.Oracle()
    P, Q = dev.get_factors()

    def rows_of(sel):
        rws = [data['indices'][data['indptr'][x]:data['indptr'][x + 1]] for x in sel]
        mp = np.zeros(len(sel) + 1, np.int64)
        mp[1:] = np.cumsum([len(x) for x in rws])
        return mp, np.concatenate(rws)
    t1 = time.perf_counter()
    orc.topn_scan(P, Q, users[:4], N, *rows_of(users[:4]))
    per_user = (time.perf_counter() - t1) / 4
    S = int(max(4, min(len(users), budget_s / per_user)))
    t1 = time.perf_counter()
    oid, _, _ = orc.topn_scan(P, Q, users[:S], N, *rows_of(users[:S]))
    cdt = time.perf_counter() - t1
    if not np.array_equal(oid, ids[:S]):
        sys.exit('scoring bench: lists differ from the oracle')
    return {'value': S / cdt, 'unit': 'users/s', 'cores': 1, 'kind': 'port',
            'sample': 'first %d users of the same workload (all %d items each), oracle/bpr_oracle.c:orc_topn_scan, %.1f s on %s; lists equal the GPU lists'
                      % (S, n, cdt, _cpu_name())}


def secondary_scoring(dev, data, m, n, k, no_cpu, state):
    """BASELINE config 5 at its stated size: all users x all items, N = 20, training items masked, with the factors on the device."""
    N, nu, steps = 20, m, 3
    users = np.arange(nu, dtype=np.int32)
    dev.topn_scan(users, N)
    t0 = time.perf_counter()
    kms = 0.0
    for _ in range(steps):
        ids, sc = dev.topn_scan(users, N)
        ms, events, rescored, used_bf16 = dev.scan_stats()
        kms += ms
    dt = time.perf_counter() - t0
    done, total = dev.scan_work()
    ach = 2.0 * 1024 * k * done * steps / (kms * 1e-3)              # flops of the tiles the kernel scored (the rest is skipped by a norm bound)
    peak = MFMA_BF16_PEAK if used_bf16 else MFMA_F32_PEAK
    return {'metric': 'top-%d scoring users/sec (P.Q^T + overwrite-scan selection), k=%d' % (N, k), 'value': nu * steps / dt, 'unit': 'users/s',
            'steps': steps, 'ms_per_step': 1e3 * dt / steps, 'dtype': 'bf16 pre-filter + f32 exact re-score' if used_bf16 else 'f32',
            'config': {'workload': 'C5: all %d users x %d items, k=%d, N=%d, training items masked; factors: %s; '
                                   'host copies of ids/scores included in value' % (m, n, k, N, state),
                       'state_machine_events_per_user': events / nu, 'exact_rescores_per_user': rescored / nu,
                       'tiles_scored_fraction': done / max(1, total)},
            'roofline': {'bound': 'mfma', 'kernel': scan_kernel_label(dev, k) if used_bf16 else 'k_topn_scan (f32 MFMA)', 'achieved': ach / 1e12,
                         'peak': peak / 1e12, 'unit': 'TFLOP/s', 'frac': ach / peak, 'kernel_ms_per_scan': kms / steps, 'traffic': None},
            'cpu_baseline': None if no_cpu else scoring_cpu_baseline(dev, data, users, ids, N, n, 4.0)}



def _rms(a):
    a = a.astype(np.float64, copy=False).ravel()
    return float(np.sqrt(np.dot(a, a) / a.size))


def deviation_vs_sequential(dev, P0, Q0, seed, W, E):
    """One epoch from (P0, Q0) on the device sampler's epoch-0 negatives, twice: with the reference's exact sequential semantics
    (option epoch_exact: chain_kernels.hpp, bit-checked against oracle/ by tests/test_gpu_exact.py and by secondary.exact below)
    and with the S-round semantics at round size W.  Returns the distance between the two end points."""
    dev.set_factors(P0, Q0)
    dev.set_option('epoch_exact', 1)
    t0 = time.perf_counter()
    nll_e = dev.bpr_epoch(seed, 0, 0, LR, REG_U, REG_I)[0]
    t_exact = time.perf_counter() - t0
    dev.set_option('epoch_exact', 0)
    Pe, Qe = dev.get_factors()
    dev.set_factors(P0, Q0)
    nll_r = dev.bpr_epoch(seed, 0, W, LR, REG_U, REG_I)[0]
    Pr, Qr = dev.get_factors()
    out = {'round_events': W,
           'what': 'one epoch from the initial factors on the same negatives: S-round end point against the exact sequential loop on the device',
           'loss_rel': (nll_r - nll_e) / nll_e,
           'nll_per_triplet': {'sequential': nll_e / E, 's_round': nll_r / E},
           'rms_distance_over_rms_movement': {'P': _rms(Pr - Pe) / _rms(Pe - P0), 'Q': _rms(Qr - Qe) / _rms(Qe - Q0)},
           'normwise_rel': {'P': float(np.abs(Pr - Pe).max() / np.abs(Pe).max()), 'Q': float(np.abs(Qr - Qe).max() / np.abs(Qe).max())}}
    return out, (nll_e, t_exact)


EXACT_MODES = (
    # (key, options, what)
    ('bit_equal', {'chain_fast': 0, 'chain_xcd': 0},
     'every operation of recommender/cf/BPR.py:50-57 as the reference rounds it (sigmoid in double precision): bit-equal to oracle/bpr_oracle.c'),
    ('within_1e-5', {'chain_fast': 1, 'chain_xcd': 0},
     'the same sequential order, the step\'s coefficient lr (1 - sigmoid(x)) in single precision and the margin as one 64-lane sum: not bit-equal, '
     'factors within BASELINE.json\'s 1e-5 of the oracle (measured 3e-7 .. 6e-7: tests/test_gpu_baseline_configs.py)'),
    ('within_1e-5_one_xcd', {'chain_fast': 1, 'chain_xcd': 1},
     'as within_1e-5 with every working wave on one XCD, rows handed over through its L2 instead of the memory side (same results)'),
)


def exact_line(dev, name, data, P0, Q0, seed, k, steps, no_cpu, prefix=None):
    """The exact path on one workload: per mode `steps` whole epochs through yue_bpr_epoch(option epoch_exact) -- sampler pass, row
    ordinals (device sort), granule copies and the dataflow launch all inside the timed call -- and the first epoch's stream
    once more through yue_bpr_replay from host arrays (upload of 12 bytes per triplet inside the timed call).
    `value` is the rate of the mode that meets north_star's tolerance (within_1e-5); the bit-equal mode stands beside it."""
    m, n = P0.shape[0], Q0.shape[0]
    E = int(data['ev_ptr'][-1])
    ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(data['ev_ptr']))
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    j0 = dev.sample_negatives(seed, 0)
    out = {'workload': name, 'triplets_per_epoch': E}
    modes = {}
    for key, opts, what in EXACT_MODES:
        for o, v in opts.items():
            dev.set_option(o, v)
        check = None
        if prefix is not None and key != 'within_1e-5_one_xcd':
            # exactness on the bench workload itself: the first S triplets of epoch 0 against the CPU oracle's state after them
            S, Ps, Qs = prefix
            dev.set_factors(P0, Q0)
            dev.bpr_replay(ev_u[:S], data['ev_i'][:S], j0[:S], LR, REG_U, REG_I)
            P, Q = dev.get_factors()
            check = {'triplets': S, 'rel_err_P': float(np.abs(P - Ps).max() / np.abs(Ps).max()), 'rel_err_Q': float(np.abs(Q - Qs).max() / np.abs(Qs).max()),
                     'bit_equal_fraction_P': float(np.mean(P[:int(ev_u[S - 1]) + 1] == Ps[:int(ev_u[S - 1]) + 1])), 'bit_equal_fraction_Q': float(np.mean(Q == Qs)),
                     'against': 'oracle/bpr_oracle.c: orc_bpr_sequential on the same prefix'}
            if max(check['rel_err_P'], check['rel_err_Q']) > (1e-6 if key == 'bit_equal' else 1e-5):
                sys.exit('exact path (%s) differs from the oracle: %r' % (key, check))
            del P, Q
        dev.set_factors(P0, Q0)
        dev.set_option('epoch_exact', 1)
        dev.bpr_epoch(seed, 0, 0, LR, REG_U, REG_I)                      # warm-up (allocations of the pre-pass)
        dev.sync()
        t0 = time.perf_counter()
        kus = 0
        for ep in range(1, steps + 1):
            nll = dev.bpr_epoch(seed, ep, 0, LR, REG_U, REG_I)[0]
            kus += dev.get_option('chain_last_us')
        dt = (time.perf_counter() - t0) / steps
        dev.set_option('epoch_exact', 0)
        modes[key] = {'value': E / dt, 'unit': 'triplets/s', 'ms_per_epoch': 1e3 * dt, 'dataflow_launch_ms': 1e-3 * kus / steps, 'steps': steps,
                      'final_nll_per_triplet': nll / E, 'runs': dev.get_option('chain_last_runs'), 'waves': dev.get_option('chain_last_waves'),
                      'what': what, 'checked_against_oracle': check}
    dev.set_option('chain_fast', 0)
    dev.set_option('chain_xcd', 0)
    dev.set_factors(P0, Q0)
    t0 = time.perf_counter()
    dev.bpr_replay(ev_u, data['ev_i'], j0, LR, REG_U, REG_I)
    dt_replay = time.perf_counter() - t0
    head = modes['within_1e-5']
    out.update({'value': head['value'], 'unit': 'triplets/s', 'ms_per_epoch': head['ms_per_epoch'], 'value_mode': 'within_1e-5', 'modes': modes,
                'checked_against_oracle': head['checked_against_oracle'],
                'replay_from_host_arrays': {'value': E / dt_replay, 'unit': 'triplets/s', 'ms': 1e3 * dt_replay,
                                            'what': 'yue_bpr_replay(u, i, j) of epoch 0 (bit-equal mode): upload of the stream, runs and ordinals on the device, user rows versioned per run'}})
    if not no_cpu:
        import oracle
        t0 = time.perf_counter()
        depth, row_max = oracle.Oracle().dependency_depth(ev_u, data['ev_i'], j0, m, n)
        out['dependency'] = {'depth': depth, 'hottest_row_touches': row_max, 'us_per_dependent_step': 1e6 * head['ms_per_epoch'] * 1e-3 / depth,
                             'what': 'longest chain of dependent triplets of the epoch-0 stream (a triplet depends on the latest earlier one sharing P[u], Q[i] or Q[j]): '
                                     'no schedule of the exact loop takes fewer steps; the levelled replay of round 1 (option replay_levels) needs this many LAUNCHES; '
                                     'computed on the host by oracle/ outside the timed region (%.1f s)' % (time.perf_counter() - t0)}
    return out


def _oracle_prefix(dev, data, P0, Q0, seed, S):
    """state of the CPU oracle's sequential loop after the first S triplets of epoch 0 (the exact device path is checked against it)"""
    import oracle
    m = P0.shape[0]
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    j = dev.sample_negatives(seed, 0)
    ev_u = np.repeat(np.arange(m, dtype=np.int32), np.diff(data['ev_ptr']))
    users = int(ev_u[S - 1]) + 1                          # the prefix only touches the first users' rows
    Ps, Qs = P0[:users].copy(), Q0.copy()
    oracle.Oracle().bpr_sequential(Ps, Qs, ev_u[:S], data['ev_i'][:S], j[:S], LR, REG_U, REG_I)
    Pfull = P0.copy()
    Pfull[:users] = Ps
    return (S, Pfull, Qs)


def secondary_exact(dev, data, P0, Q0, seed, k, no_cpu):
    out = {'metric': 'BPR triplet-updates/sec with the reference\'s exact sequential semantics (recommender/cf/BPR.py:40-62)',
           'kernel': 'k_bpr_chain3 (one dataflow launch per epoch: a group of five waves per user run -- 2 x loads / dependency chain / 2 x stores, hand-over through LDS -- '
                     'item rows as versioned 8-byte granules, row ordinals from a device sort); k_bpr_chain (a wave per run) for streams of fewer than 16 triplets per run',
           'c3': exact_line(dev, 'C3: 1000000 users x 200000 items, k=128, 50 events/user', data, P0, Q0, seed, k, 2, no_cpu, getattr(cpu_baseline, 'prefix', None))}
    m2, n2, d2, k2 = WORKLOADS['c2']
    data2 = synth.make_arrays(m2, n2, d2, seed=20260001)
    P2, Q2 = synth.init_factors(m2, n2, k2, 20260002)
    out['c2'] = exact_line(dev, 'C2: 100000 users x 50000 items, k=64, 50 events/user', data2, P2, Q2, seed, k2, 2, no_cpu,
                           None if no_cpu else _oracle_prefix(dev, data2, P2, Q2, seed, 2000000))
    del data2, P2, Q2
    # one rank's share of config 4: 6 events per user -> a shallow dependency graph
    m4, n4, d4, k4 = WORKLOADS['c4shard']
    data4 = synth.make_arrays(m4, n4, d4, seed=20260001)
    P4, Q4 = synth.init_factors(m4, n4, k4, 20260002)
    out['c4shard'] = exact_line(dev, 'C4 shard: 10000000 users x 125000 items, k=128, 6 events/user', data4, P4, Q4, seed, k4, 2, no_cpu,
                                None if no_cpu else _oracle_prefix(dev, data4, P4, Q4, seed, 2000000))
    return out


def bench_scoring(args, cp):
    """Secondary line: evalRanking's scoring + selection for every user (C5).  One step = one scan."""
    m, n, d, k = WORKLOADS[args.workload]
    N = 20
    data = synth.make_arrays(m, n, d, seed=20260001)
    P0, Q0 = synth.init_factors(m, n, k, 20260002)
    dev = Device(cp.local_rank, raise_errors=True)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    lo, hi = cp.rank * m // cp.world, (cp.rank + 1) * m // cp.world        # users shard over GPUs, no collective
    users = np.arange(lo, hi, dtype=np.int32)
    if args.scan_f32:
        dev.set_option('scan_f32', 1)
    for kv in args.opt:
        name, value = kv.split('=')
        dev.set_option(name, int(value))
    for _ in range(args.warmup):
        dev.topn_scan(users, N)
    cp.barrier()
    t0 = time.perf_counter()
    kms = 0.0
    for _ in range(args.steps):
        ids, sc = dev.topn_scan(users, N)
        ms, events, rescored, used_bf16 = dev.scan_stats()
        kms += ms
    cp.barrier()
    dt = cp.reduce_max(time.perf_counter() - t0)
    cpu = None
    if cp.rank == 0 and cp.world == 1 and not args.no_cpu_baseline:
        cpu = scoring_cpu_baseline(dev, data, users, ids, N, n, 12.0)
    if cp.rank == 0:
        done, total = dev.scan_work()
        ach = 2.0 * 1024 * k * done * args.steps / (kms * 1e-3)     # flops of the tiles the kernel scored
        peak = MFMA_BF16_PEAK if used_bf16 else MFMA_F32_PEAK
        print(json.dumps({
            'metric': 'top-%d scoring users/sec (P.Q^T + overwrite-scan selection), k=%d' % (N, k), 'value': m * args.steps / dt,
            'unit': 'users/s', 'n_gpus': cp.world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'bf16 pre-filter + f32 exact re-score' if used_bf16 else 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: %d users x %d items, k=%d, N=%d, training items masked, host copies of ids/scores included in value'
                                   % (args.workload.upper(), m, n, k, N), 'state_machine_events_per_user': events / max(1, len(users)),
                       'exact_rescores_per_user': rescored / max(1, len(users)), 'bf16_prefilter': used_bf16, 'tiles_scored_fraction': done / max(1, total)},
            'roofline': {'bound': 'mfma', 'kernel': scan_kernel_label(dev, k) if used_bf16 else ('k_topn_scan<K2=%d>' % (k // 2)), 'achieved': ach / 1e12, 'peak': peak / 1e12,
                         'unit': 'TFLOP/s', 'frac': ach / peak, 'kernel_ms_per_scan': kms / args.steps, 'traffic': None},
            'cpu_baseline': cpu}))
    dev.close()
    cp.close()


def bench_fism(args, cp):
    """Extra line for FISM: draws/s of the round form (yue_fism_rounds, one wave per user, the users of a
    round at once) on a 100K-user problem, beside the sequential device pass (one wave by construction, the reference's exact
    order) and the reference's loop as NumPy runs it (oracle/numpy_fism.py) on samples of the same workload.  Not the headline."""
    m, n, d, k = WORKLOADS['fism']
    rho, alpha, lr, reg = 2, 0.5, 0.015, 0.01                      # the reference's FISM.conf
    round_users = args.round_events if args.round_events > 0 else 256     # larger rounds diverge on this problem (sums of stale differences on the popular items)
    data = synth.make_arrays(m, n, d, seed=20260001)
    rs = np.random.RandomState(20260005)
    Q0 = (rs.rand(n, k).astype(np.float32) / 10)
    B0, P0 = rs.rand(n) / 100, rs.rand(n, k) / 100
    ptr, ev_i = data['ev_ptr'], data['ev_i']
    draws = int(np.diff(ptr)[np.diff(ptr) > 1].sum()) * rho

    def negatives(epoch):
        # rejection against the user's items (FISM.py:50-53), vectorised: redraw the hits
        g = np.random.RandomState(77 + epoch)
        ev_u = np.repeat(np.arange(m, dtype=np.int64), np.diff(ptr))
        uu = np.repeat(ev_u, rho)
        out = g.randint(0, n, size=len(uu)).astype(np.int64)
        keys = np.unique(ev_u * n + ev_i)
        while True:
            bad = np.isin(uu * n + out, keys)
            if not bad.any():
                break
            out[bad] = g.randint(0, n, size=int(bad.sum()))
        return out.astype(np.int32)
    negs = [negatives(e) for e in range(args.warmup + args.steps)]
    coef = np.array([pow(int(x) - 1, -alpha) if x > 1 else 0.0 for x in np.diff(ptr)], np.float64)
    dev = Device(cp.local_rank, raise_errors=True)
    for kv in args.opt:
        name, value = kv.split('=')
        dev.set_option(name, int(value))
    dev.fism_set_model(P0, Q0, B0)
    for e in range(args.warmup):
        dev.fism_rounds(ptr, ev_i, negs[e], rho, coef, round_users, lr, reg, reg)
    t0 = time.perf_counter()
    for e in range(args.warmup, args.warmup + args.steps):
        half = dev.fism_rounds(ptr, ev_i, negs[e], rho, coef, round_users, lr, reg, reg)[0]
    dt = time.perf_counter() - t0
    # the sequential device pass and the NumPy loop on the first users of the same problem
    su = min(m, 2000)
    sl = slice(0, int(ptr[su]))
    nsl = slice(0, int(np.diff(ptr[:su + 1])[np.diff(ptr[:su + 1]) > 1].sum()) * rho)
    dev.fism_set_model(P0, Q0, B0)
    t1 = time.perf_counter()
    dev.fism_epoch(ptr[:su + 1], ev_i[sl], negs[0][nsl], rho, coef[:su], lr, reg, reg)
    seq_dt = time.perf_counter() - t1
    cpu = None
    if not args.no_cpu_baseline:
        from oracle.numpy_fism import fism_epoch
        P, Q, Bi = P0.copy(), Q0.copy(), B0.copy()
        t1 = time.perf_counter()
        fism_epoch(P, Q, Bi, ptr[:su + 1], ev_i[sl], negs[0][nsl], rho, alpha, lr, reg, reg)
        cdt = time.perf_counter() - t1
        cpu = {'value': (nsl.stop - nsl.start) / cdt, 'unit': 'draws/s', 'cores': 1, 'kind': 'port',
               'sample': 'the first %d users of the same workload, oracle/numpy_fism.py (the NumPy statements of recommender/cf/FISM.py:38-69, bit-equal to the reference on its goldens), %.1f s on %s' % (su, cdt, _cpu_name())}
    bytes_per_draw = 32 * k + 32                                    # P[i], P[j] read (f64), Q[i], Q[j] read + written (f32), four bias accesses
    ach = bytes_per_draw * draws * args.steps / dt
    print(json.dumps({
        'metric': 'FISM draw-updates/sec, rounds of %d users, k=%d' % (round_users, k), 'value': draws * args.steps / dt, 'unit': 'draws/s', 'n_gpus': 1,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64 (P, Bi) / f32 (Q)', 'data': 'synthetic',
        'config': {'workload': 'FISM: %d users x %d items, %d events/user, k=%d, rho=%d, alpha=%g, lr=%g, reg=%g, rounds of %d users; host work of a call '
                               '(upload of events and negatives) included in value' % (m, n, d, k, rho, alpha, lr, reg, round_users),
                   'final_half_sq_error': half,
                   'sequential_device_pass_draws_per_s': (nsl.stop - nsl.start) / seq_dt},
        'roofline': {'bound': 'hbm', 'kernel': 'k_fism_round_lds<KR=%d> (one wave per user, working and round-start rows in LDS, rows of one user stored in place: the user\'s chain of draws, then the contended atomics of the shared rows)' % (1 if k <= 64 else 2 if k <= 128 else 4),
                     'achieved': ach / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': ach / HBM_PEAK, 'algorithmic_bytes_per_draw': bytes_per_draw, 'traffic': None},
        'cpu_baseline': cpu}))
    dev.close()
    cp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default=None, choices=sorted(WORKLOADS), help='default: c3 on one GPU, c4shard per rank on several')
    ap.add_argument('--round-events', type=int, default=0, help='events per S-round; 0 = the library\'s default for this device and problem (yue_default_round_events)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the scoring line (secondary.c5) of the default run')
    ap.add_argument('--scan-f32', action='store_true', help='scoring workloads: force the exact f32-MFMA kernel')
    ap.add_argument('--tpw', type=int, default=0, help='tuning: events per wave in the round kernel (0 = default)')
    ap.add_argument('--opt', action='append', default=[], metavar='NAME=VALUE', help='tuning: yue_set_option(NAME, VALUE), repeatable')
    ap.add_argument('--force-comm', action='store_true', help='N=1 only: run the communicator code path with a 1-rank RCCL communicator')
    args = ap.parse_args()

    cp = ControlPlane()
    rank, world, local_rank = cp.rank, cp.world, cp.local_rank
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run (one process per GPU)' % args.gpus)
        args.gpus = world

    if args.workload is None:
        args.workload = 'c3' if world == 1 else 'c4shard'
    if args.workload.startswith('c5'):
        return bench_scoring(args, cp)
    if args.workload == 'fism':
        return bench_fism(args, cp)
    m, n, d, k = WORKLOADS[args.workload]
    t_setup = time.perf_counter()
    data = synth.make_arrays(m, n, d, seed=20260001 + 7919 * rank)     # this rank's item shard
    P0, Q0 = synth.init_factors(m, n, k, 20260002)
    if rank != 0:
        Q0 = synth.init_factors(1, n, k, 20260002 + rank)[1]
    E = int(data['ev_ptr'][-1])

    dev = Device(0 if SEAM_REHEARSAL else local_rank, raise_errors=True)
    dev.set_factors(P0, Q0)
    dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
    if SEAM_REHEARSAL and world > 1:
        seam_attach(dev, cp)
    else:
        attach_device(dev, cp)
    if args.tpw:
        dev.set_option('round_tpw', args.tpw)
    for kv in args.opt:
        name, value = kv.split('=')
        dev.set_option(name, int(value))
    if args.force_comm and world == 1:
        from yue_amd._shim import comm_unique_id
        from yue_amd.dist import stdout_to_stderr
        with stdout_to_stderr():                       # RCCL's version banner must not land in front of the JSON line
            dev.comm_init(comm_unique_id(), 0, 1)
    if args.round_events <= 0:
        args.round_events = dev.default_round_events()
    setup_s = time.perf_counter() - t_setup

    seed = 20260003
    epoch = 0
    for _ in range(args.warmup):
        dev.bpr_epoch(seed, epoch, args.round_events, LR, REG_U, REG_I)
        epoch += 1
    dev.set_kernel_timing(1)                            # one HIP-event bracket around each epoch's round launches

    def barrier():
        dev.sync()
        cp.barrier()
        dev.sync()

    barrier()
    t0 = time.perf_counter()
    nll = 0.0
    for _ in range(args.steps):
        nll, sp, sq = dev.bpr_epoch(seed, epoch, args.round_events, LR, REG_U, REG_I)
        epoch += 1
    barrier()
    dt = time.perf_counter() - t0
    dt_ranks = cp.gather_floats(dt) if world > 1 else [dt]       # a straggler shows up here; the reported time is the maximum
    dt = cp.reduce_max(dt)
    k_ms, k_launches, k_triplets = dev.get_kernel_timing()
    dev.set_kernel_timing(0)
    comm = dev.comm_stats()
    if not np.isfinite(nll):
        sys.exit('loss is not finite')

    if rank == 0:
        total = float(E) * world * args.steps
        value = total / dt
        ab = algorithmic_bytes(k)
        achieved = ab * k_triplets / (k_ms * 1e-3) if k_ms > 0 else 0.0
        traffic = measured_traffic(args.workload, args.round_events)
        out = {
            'metric': 'BPR triplet-updates/sec at k=%d (S-round semantics%s; the reference\'s exact sequential semantics: secondary.exact)'
                      % (k, ', user rows sequential' if dev.get_option('round_last_user_seq') else ''), 'value': value, 'unit': 'triplets/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic' if not SEAM_REHEARSAL else 'synthetic -- REHEARSAL on one GPU through the test seam: not a measurement',
            'config': {'workload': '%s: BPR k=%d, %d users x %d items per GPU, %d events/user (%d triplets per epoch per GPU), '
                                   'counter-based sampler (one pass per epoch, inside the timed step), S-round W=%d events (the device default unless --round-events is given), lr=%g regU=regI=%g'
                                   % (args.workload.upper(), k, m, n, d, E, args.round_events, LR, REG_U),
                       'semantics': ('S-round with sequential user rows (DESIGN.md section 3): a wave owns a user and applies the user\'s triplets in the reference\'s order '
                                     'with P[u] in registers; item rows are read as the round of W events started, their differences summed per row and added once per round. '
                                     'NOT the reference\'s strictly sequential loop: config.deviation_vs_sequential says how far one epoch lands from it; '
                                     'secondary.exact is the path with the reference\'s semantics')
                                    if dev.get_option('round_last_user_seq') else
                                    ('S-round (DESIGN.md section 3): every triplet of a round of W events is evaluated on the factors as the round started, per-row differences '
                                     'summed and added once; NOT the reference\'s strictly sequential loop (secondary.exact)'),
                       'round_events': args.round_events, 'parallelism': 'items sharded x%d, users replicated' % world,
                       'setup_s': round(setup_s, 1), 'final_nll_per_triplet': nll / E},
            'roofline': {'bound': 'hbm', 'kernel': round_kernel_label(dev, k),
                         'achieved': achieved / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK,
                         'algorithmic_bytes_per_triplet': ab, 'launches_timed': k_launches,
                         'avg_launch_ms': (k_ms / k_launches) if k_launches else None,
                         'triplets_per_launch': (k_triplets / k_launches) if k_launches else None,
                         'timing': 'HIP events on the library\'s stream around all round launches of each timed epoch (a launch here = one round: its update '
                                   'launch and its fold launch; launch boundaries and the user-row apply launches of the epoch included): avg_launch_ms x launches '
                                   'per epoch <= ms_per_step by construction; the kernel-only averages are in the rocprofv3 summary under profiles/',
                         'traffic': traffic[0] if traffic else None, 'traffic_source': traffic[1] if traffic else None},
        }
        if world > 1 or args.force_comm:
            out['per_rank_ms_per_step'] = {'min': 1e3 * min(dt_ranks) / args.steps, 'max': 1e3 * max(dt_ranks) / args.steps, 'all': [1e3 * x / args.steps for x in dt_ranks]}
            out['comm'] = {'allreduce_bytes_per_epoch_per_rank': comm['allreduce_bytes'], 'collectives_per_epoch': comm['collectives'],
                           'allreduce_group_mb': dev.get_option('comm_group_mb'), 'round_cus_reserved': dev.get_option('round_cus_reserved'),
                           'compute_stream_waits_for_the_collective_stream_per_epoch': dev.get_option('comm_last_compute_waits'),
                           'knobs': '--opt comm_group_mb=N (MB of user-factor differences per ncclAllReduce; 8 by default, RCCL reaches its bus bandwidth at tens of MB: try 32..64) and '
                                    '--opt round_cus_reserved=N (CUs the compute stream leaves to RCCL\'s kernels; 0 by default, try 8..16 when compute_stream_wait_ms is not small)',
                           'compute_stream_wait_ms_last_epoch': comm['wait_ms'], 'nranks_rccl': comm['nranks'], 'rccl_version': comm['rccl_version'],
                           'what': 'ncclAllReduce (fp32 sum, in place) of the user-factor differences of a group of user blocks, on a second HIP stream beside the next '
                                   'group\'s rounds; wait = end of the last group\'s all-reduce + apply minus end of the last round launch (HIP events)'}
        if world == 1 and not args.no_cpu_baseline:
            j0 = dev.sample_negatives(seed, 0)
            out['cpu_baseline'] = cpu_baseline(data, P0, Q0, j0, k)
        if world == 1 and not args.no_secondary and not args.force_comm and args.workload == 'c3':
            out['secondary'] = {'c5': secondary_scoring(dev, data, m, n, k, args.no_cpu_baseline,
                                                        'as the %d S-round epochs of this run (warm-up + timed) left them' % (args.warmup + args.steps))}
            dev_out, _ = deviation_vs_sequential(dev, P0, Q0, seed, args.round_events, E)
            out['config']['deviation_vs_sequential'] = dev_out
            out['secondary']['exact'] = secondary_exact(dev, data, P0, Q0, seed, k, args.no_cpu_baseline)
        print(json.dumps(out))
    dev.close()
    cp.close()


if __name__ == '__main__':
    main()
