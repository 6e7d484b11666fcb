"""CPU: the C-ABI library is built for gfx950, loads, and exports every symbol the header declares.
No compute is called (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__
    __graft_entry__.build()
    from yue_amd import _shim
    return _shim


def test_library_exports_header_symbols(built):
    hdr = open(os.path.join(ROOT, 'include', 'yue_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(yue_[a-z0-9_]+)\s*\(', hdr)))
    assert declared, 'no declarations parsed'
    lib = ctypes.CDLL(built.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), 'libyue_hip.so lacks ' + name
    assert sorted(built.SYMBOLS) == declared
    assert lib.yue_version() >= 1


def test_product_library_has_no_test_seam(built):
    # the host-staged collective of tests/test_gpu_multi.py exists in libyue_hip_seam.so only
    blob = open(built.LIB_PATH, 'rb').read()
    assert b'yue_seam_init' not in blob
    assert not hasattr(ctypes.CDLL(built.LIB_PATH), 'yue_seam_init')
    seam = os.path.join(os.path.dirname(built.LIB_PATH), 'libyue_hip_seam.so')
    assert os.path.exists(seam) and hasattr(ctypes.CDLL(seam), 'yue_seam_init')


def test_code_object_targets_gfx950(built):
    blob = open(built.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob


def test_product_has_no_cpu_fallback(built, monkeypatch):
    # the product path must fail loudly when the HIP extension is missing
    monkeypatch.setattr(built, 'LIB_PATH', '/nonexistent/libyue_hip.so')
    monkeypatch.setattr(built, '_lib', None)
    with pytest.raises(built.YueHipError):
        built.load_library()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'yue_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(import|from)\s+oracle\b', src, re.M), f
                assert 'liboracle' not in src, f


def test_round_kernels_fit_the_residency_the_host_assumes(tmp_path):
    """default_round_events (csrc/bpr_host.hip) sizes a round for 7 resident workgroups per CU of k_round_m and 6 of k_round:
    that holds while the kernels stay within 96 / 112 allocated SGPRs and 72 / 80 VGPRs without scratch (MI355X residency
    rule: floor(800 / (SGPRs rounded up to 16 + 16)) workgroups of 4 waves, 512 VGPRs per SIMD).  Compiles the device code
    once more with the compiler's resource remarks (about a minute)."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    src = os.path.join(ROOT, 'yue_amd', 'csrc', 'bpr_host.hip')
    out = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S', '--cuda-device-only',
                          '-Rpass-analysis=kernel-resource-usage', '-o', str(tmp_path / 'bpr_host.s'), src],
                         capture_output=True, text=True, cwd=os.path.dirname(src))
    assert out.returncode == 0, out.stderr[-2000:]
    usage = {}
    name = None
    for line in out.stderr.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r'remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]): (\d+)', line)
        if m and name:
            usage[name][m.group(1)] = int(m.group(2))
    # the instantiations the default events-per-wave choice launches: k_round_m k <= 64 -> <1,8>, k <= 128 -> <2,8>, else <4,4>;
    # k_round <1,16>, <2,8>, <4,4>
    seen = 0
    for fn, u in usage.items():
        for kr, tpw in ((1, 8), (2, 8), (4, 4)):
            for bigq in (0, 1):                                  # (Lb1: the instances for item matrices of 2 GiB and more)
                if 'k_round_mILi%dELi%dELb%dEE' % (kr, tpw, bigq) in fn:
                    assert u['TotalSGPRs'] <= 96 and u['VGPRs'] <= 72 and u['ScratchSize [bytes/lane]'] == 0, (fn, u)
                    seen += 1
        for kr, tpw in ((1, 16), (2, 8), (4, 4)):
            if '7k_roundILi%dELi%dEE' % (kr, tpw) in fn:
                assert u['TotalSGPRs'] <= 112 and u['VGPRs'] <= 80 + 8 * (kr == 4) and u['ScratchSize [bytes/lane]'] == 0, (fn, u)
                seen += 1
    assert seen == 9, sorted(usage)
