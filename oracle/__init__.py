"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU checker for the HIP path: `liboracle.so` (our restatement, som_lvq_oracle.c) and,
when built, `_ref/libref_harness.so` (the unmodified reference behind a flat-array
driver).  Only tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this package; the product never does.
"""
from .binding import Oracle, RefHarness, build, ref_available, ref_tool  # noqa: F401
