"""CPU: static check of the exact path's hand-written vector-memory instructions against the one hazard the compiler cannot
cover for them.  On gfx9-family parts a vector-memory instruction may read an SGPR that the vector ALU wrote (v_readlane,
v_readfirstlane, a compare into an SGPR pair) only 5 wait states later.  The compiler counts those for its own instructions,
but chain_kernels.hpp issues its granule loads / stores as inline assembly (counted s_waitcnt vmcnt(N)), which it does not
look into: round 3 had two memory-access faults from exactly this (a descriptor word out of v_readfirstlane read by the next
buffer_load).  Every such operand now passes through vmem_sgpr_guard() (an s_nop 4); this test reads the device listing of
chain_host.hip and fails if any buffer_/global_/flat_ instruction reads an SGPR written by a VALU instruction fewer than 5
wait states earlier.  (Straight-line scan: a writer on another path into a label is not seen -- the guard sits directly in
front of the memory instructions, so the fall-through path is the one that matters.)"""
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VMEM = re.compile(r'^(buffer_|global_|flat_|scratch_)')
SREG = re.compile(r'\bs\[(\d+):(\d+)\]|\bs(\d+)\b')


def _sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan(listing, need=5):
    """-> list of (line number, instruction, sgpr, wait states seen) for every violation."""
    bad = []
    age = {}                                            # sgpr -> wait states since a VALU instruction wrote it
    for ln, raw in enumerate(listing.splitlines(), 1):
        line = raw.split(';')[0].strip()
        if not line or line.endswith(':') or line.startswith('.'):
            continue
        parts = line.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else '')
        if VMEM.match(op):
            for r in _sregs(args):
                if r in age and age[r] < need:
                    bad.append((ln, line, r, age[r]))
        step = 1
        if op == 's_nop':
            step = int(args.strip(), 0) + 1
        for r in list(age):
            age[r] += step
            if age[r] >= need:
                del age[r]
        if op.startswith('v_'):
            dst = args.split(',')[0]
            for r in _sregs(dst):                       # a VALU instruction whose destination is scalar
                age[r] = 0
        elif op.startswith('s_') and op not in ('s_nop', 's_waitcnt', 's_sleep', 's_branch') and not op.startswith('s_cbranch'):
            dst = args.split(',')[0]
            for r in _sregs(dst):                       # rewritten by the scalar ALU: no VALU hazard on it any more
                age.pop(r, None)
    return bad


VREG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')


def _vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan_wide_stores(listing, need=2):
    """A vector-memory store of more than 64 bits reads its data registers late: a VALU instruction that writes one of them needs
    `need` wait states behind the store (the compiler counts them for its own stores; the hand-written granule-pair stores carry
    an s_nop inside their statement).  -> list of (line number, store, overwriting instruction)."""
    bad = []
    pending = []                                        # [data registers, wait states since the store, line, text]
    for ln, raw in enumerate(listing.splitlines(), 1):
        line = raw.split(';')[0].strip()
        if not line or line.endswith(':') or line.startswith('.'):
            continue
        parts = line.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else '')
        if op.startswith('v_') and pending:
            dst = _vregs(args.split(',')[0])
            for regs, age, sl, st in pending:
                if age < need and dst & regs:
                    bad.append((sl, st, line))
        step = int(args.strip(), 0) + 1 if op == 's_nop' else 1
        pending = [[r, a + step, sl, st] for r, a, sl, st in pending if a + step < need]
        if re.match(r'(buffer|global|flat)_store_dwordx[34]', op):
            pending.append([_vregs(args.split(',')[0]), 0, ln, line])
    return bad


def test_checker_sees_the_hazard_and_the_guard():
    wide = "\tbuffer_store_dwordx4 v[4:7], v9, s[4:7], s9 offen sc1\n\tv_mov_b32_e32 v5, v1\n"
    assert len(scan_wide_stores(wide)) == 1
    assert scan_wide_stores(wide.replace('\tv_mov', '\ts_nop 1\n\tv_mov')) == []
    assert scan_wide_stores(wide.replace('v5, v1', 'v8, v1')) == [] and scan_wide_stores(wide.replace('dwordx4 v[4:7]', 'dwordx2 v[4:5]')) == []
    racy = "\tv_readfirstlane_b32 s4, v1\n\tv_readfirstlane_b32 s5, v2\n\tbuffer_load_dwordx2 v[0:1], v3, s[4:7], 0 offen sc1\n"
    assert [b[2] for b in scan(racy)] == [4, 5]
    assert scan(racy.replace('\tbuffer', '\ts_nop 4\n\tbuffer')) == []
    assert scan(racy.replace('\tbuffer', '\ts_nop 2\n\tbuffer')) != []
    soff = "\tv_readlane_b32 s9, v7, 3\n\ts_nop 1\n\tbuffer_store_dwordx2 v[0:1], v3, s[4:7], s9 offen sc1\n"
    assert [b[2] for b in scan(soff)] == [9]
    assert scan("\tv_readlane_b32 s9, v7, 3\n\ts_lshl_b32 s9, s9, 10\n\tbuffer_store_dwordx2 v[0:1], v3, s[4:7], s9 offen\n") == []


def test_exact_path_listing_has_no_valu_sgpr_to_vmem_hazard(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    src = os.path.join(ROOT, 'yue_amd', 'csrc', 'chain_host.hip')
    lst = tmp_path / 'chain_host.s'
    out = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S', '--cuda-device-only', '-o', str(lst), src],
                         capture_output=True, text=True, cwd=os.path.dirname(src))
    assert out.returncode == 0, out.stderr[-2000:]
    text = lst.read_text()
    assert text.count('buffer_load_dwordx2') + text.count('buffer_load_dwordx4') > 100 and text.count('s_nop 4') > 50       # the listing is the one with the hand-written accesses
    bad = scan(text)
    assert not bad, bad[:10]
    wide = scan_wide_stores(text)
    assert text.count('buffer_store_dwordx4') > 50 and not wide, wide[:10]
