"""ORACLE — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Two restatements of the same pipeline, same operation order:
  * numpy:   mjmath.py (MuJoCo arithmetic), lie.py, ik.py (mink's Python), qp_gi.py (Goldfarb–Idnani)
  * plain C: c/mink_oracle.{h,c} (+ cport.py, its ctypes binding) — fast enough to check every problem of a
    65 536 batch and to serve as bench.py's CPU baseline
Pinning: the mink layer of both is pinned against fixtures recorded from the REAL mink Python imported from
/root/reference (tests/golden/make_golden.py, tests/test_oracle_{lie,ik,c}.py); the MuJoCo and quadprog layers
are third-party code absent from the reference checkout and from this image — restated from the published
algorithms, "parity unpinned" against the wheels themselves.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
package (mink_amd/) never does (tests/test_abi.py::test_product_never_imports_oracle).
"""
