#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the two PMC summaries of tools/profile_round.sh (what bench.py reports as roofline.traffic).

usage: python tools/traffic_json.py FETCH.txt WRITE.txt WORKLOAD ROUND_EVENTS "taken ..." > profiles/rNN_traffic.json

FETCH_SIZE is reported in KiB and reads exactly 1/2 of the bytes on gfx950 for these 4-byte-per-lane loads: the correction
factor is calibrated in the same pass on k_sumsq, whose bytes are known (it reads P then Q once: the larger of its two
launches reads m * k * 4 bytes -- passed as CALIBRATION_BYTES, default the C3 user factors).  WRITE_SIZE (KiB) is exact
for stores and float atomics."""
import json
import os
import re
import sys


def parse(path):
    out, name = {}, None
    for ln in open(path):
        if not ln.startswith(' '):
            name = ln.strip()
            out[name] = {}
            continue
        m = re.match(r'\s+(\S+)\s+avg\s+([0-9.eE+-]+)\s+over\s+(\d+)', ln)
        if m and name:
            out[name][m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out


def pick(tab, part):
    hits = [k for k in tab if part in k]
    if not hits:
        raise SystemExit('no kernel matching %r' % part)
    return hits[0]


def main():
    fetch_path, write_path, workload, round_events = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    taken = sys.argv[5] if len(sys.argv) > 5 else 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum, separate passes'
    f, w = parse(fetch_path), parse(write_path)
    # the round's update launch: k_round_u (round 4: sequential user rows, one GPU) where the pass ran it, else k_round_m
    upd = 'k_round_u' if any('k_round_u<' in name for name in f) else 'k_round_m'
    km, kf = pick(f, upd + '<'), pick(f, 'k_round_fold<')
    fetch = {upd: f[km]['FETCH_SIZE'][0], 'k_round_fold': f[kf]['FETCH_SIZE'][0]}
    write = {upd: w[pick(w, upd + '<')]['WRITE_SIZE'][0], 'k_round_fold': w[pick(w, 'k_round_fold<')]['WRITE_SIZE'][0]}
    atomics = w[pick(w, upd + '<')].get('TCC_EA0_ATOMIC_sum', (0.0, 0))[0]
    # calibration: k_sumsq reads the user factors and the item factors once each (two launches per epoch, averaged here)
    cal_bytes = float(os.environ.get('CALIBRATION_BYTES', (1000000 + 200000) * 128 * 4 / 2))
    cal_kib = f[pick(f, 'k_sumsq')]['FETCH_SIZE'][0]
    corr = cal_bytes / 1024.0 / cal_kib
    traffic = (fetch[upd] + fetch['k_round_fold']) * 1024.0 * round(corr) + (write[upd] + write['k_round_fold']) * 1024.0
    print(json.dumps({
        '_comment': 'HBM-side traffic of one S-round of the epoch path = its update launch (' + upd + ') + its fold launch (k_round_fold), '
                    'from rocprofv3 --pmc passes of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary` '
                    '(tools/profile_round.sh -> tools/traffic_json.py). FETCH_SIZE is in KiB and reads 1/2 of the bytes on gfx950 for '
                    'these 4-byte-per-lane loads (calibrated in the same pass on k_sumsq: %.1f KiB reported for %.0f KiB read -> x%.3f, '
                    'applied as x%d); WRITE_SIZE (KiB) is exact for stores and float atomics.' % (cal_kib, cal_bytes / 1024.0, corr, round(corr)),
        'taken': taken, 'workload': workload, 'round_events': round_events,
        'kernel': '%s + %s' % (km.replace('void yue::', ''), kf.replace('void yue::', '')),
        'fetch_size_kib': fetch, 'fetch_correction': float(round(corr)), 'fetch_correction_measured': corr,
        'write_size_kib': write, 'atomic_requests_64B': atomics,
        'dispatches': f[km]['FETCH_SIZE'][1],
        'traffic_bytes_per_launch': int(traffic),
        'files': [os.path.basename(fetch_path), os.path.basename(write_path)]}, indent=1))


if __name__ == '__main__':
    main()
