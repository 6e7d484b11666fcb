#!/usr/bin/env python3
"""How different is the MODEL the throughput semantics trains?  python tools/quality_probe.py [c3|c2] [epochs] [users scored]
The same problem, initial factors, seed and epochs through (a) the exact sequential path (option epoch_exact) and (b) the S-round
path at the device's default round size: loss per epoch of both, and agreement of the top-20 lists the two models produce
(training items masked, true top-N so that list order does not depend on the scan order): one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yue_amd import synth                      # noqa: E402
from yue_amd._shim import Device               # noqa: E402

W = {'c3': (1000000, 200000, 50, 128), 'c2': (100000, 50000, 50, 64)}
LR, REG = 0.02, 0.01


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    m, n, d, k = W[name]
    nu = int(sys.argv[3]) if len(sys.argv) > 3 else min(m, 200000)
    data = synth.make_arrays(m, n, d, seed=20260001)
    P0, Q0 = synth.init_factors(m, n, k, 20260002)
    E = int(data['ev_ptr'][-1])
    users = np.arange(nu, dtype=np.int32)
    dev = Device(0, raise_errors=True)
    dev.set_option('topn_true', 1)
    out = {'workload': name, 'epochs': epochs, 'users_scored': nu}
    lists = {}
    for label, exact in (('sequential', 1), ('s_round', 0)):
        dev.set_factors(P0, Q0)
        dev.set_interactions(data['indptr'], data['indices'], data['ev_ptr'], data['ev_i'])
        dev.set_option('epoch_exact', exact)
        w = 0 if exact else dev.default_round_events()
        nll = [dev.bpr_epoch(20260003, ep, w, LR, REG, REG)[0] / E for ep in range(epochs)]
        dev.set_option('epoch_exact', 0)
        ids, sc = dev.topn_scan(users, 20)
        lists[label] = ids
        out[label] = {'round_events': w, 'nll_per_triplet': [round(x, 6) for x in nll]}
    a, b = lists['sequential'], lists['s_round']
    inter = np.array([len(np.intersect1d(a[u], b[u])) for u in range(nu)])
    out['top20'] = {'mean_overlap': float(inter.mean()) / 20.0, 'lists_with_the_same_items': float(np.mean(inter == 20)),
                    'lists_equal_in_order': float(np.mean(np.all(a == b, axis=1))), 'first_item_equal': float(np.mean(a[:, 0] == b[:, 0]))}
    out['loss_rel_last_epoch'] = out['s_round']['nll_per_triplet'][-1] / out['sequential']['nll_per_triplet'][-1] - 1.0
    print(json.dumps(out))
    dev.close()


if __name__ == '__main__':
    main()
