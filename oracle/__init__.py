"""CPU oracle for the BPR hot path -- TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package (as the checker or the timed CPU baseline).  yue_amd/ never does.
See oracle/bpr_oracle.c for the restated reference lines; parity is pinned by
tests/golden/ (generated from the reference by tools/make_goldens.py).
"""
from .loader import Oracle, build, lib_path  # noqa: F401
