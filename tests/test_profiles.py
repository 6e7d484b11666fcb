"""The traffic figure bench.py reports (roofline.traffic) is derived, not typed in: tools/traffic_json.py on the committed PMC
summaries must reproduce the committed profiles/rNN_traffic.json."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_traffic_json():
    import glob
    return sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))[-1]


def test_traffic_json_follows_from_the_committed_pmc_summaries():
    prof = os.path.join(ROOT, 'profiles')
    committed = json.load(open(_newest_traffic_json()))
    fetch, write = (os.path.join(prof, f) for f in committed['files'])
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'traffic_json.py'), fetch, write, committed['workload'], str(committed['round_events'])],
                         stdout=subprocess.PIPE, check=True).stdout
    again = json.loads(out)                                    # (its `files` are the names the passes had under gpurun_out/pm/)
    assert again['traffic_bytes_per_launch'] == committed['traffic_bytes_per_launch']
    assert again['fetch_size_kib'] == committed['fetch_size_kib'] and again['write_size_kib'] == committed['write_size_kib']
    assert abs(again['fetch_correction_measured'] - 2.0) < 0.01          # the gfx950 FETCH_SIZE factor, calibrated on k_sumsq in the same pass
    # and bench.py picks exactly this file for the default workload at the default round size
    sys.path.insert(0, ROOT)
    import bench
    got = bench.measured_traffic('c3', committed['round_events'])
    assert got is not None and got[0] == committed['traffic_bytes_per_launch'] and os.path.basename(_newest_traffic_json()) in got[1]
