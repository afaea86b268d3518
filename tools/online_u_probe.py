#!/usr/bin/env python3
"""online_u_probe.py -- the online (batch 1, reference-exact) engine at configs[3] for several register-buffer depths of
k_som_online_step (SOMHIP_ONLINE_U): microseconds per iteration at the head of the schedule (radius 128: every row
read, most written) and near its end (radius ~ 2: every row read, few written), final codebook bits compared."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from som_lvq_pak_amd import engine as E
from som_lvq_pak_amd._lib import SomParams

L, xdim, ydim, d = 10_000_000, 256, 256, 512
eng = E.Engine(0)
ds = E.Dataset(eng, generate=(3456, 256, d, 0, 1 << 17))
lo, hi, cnt = E.column_minmax(ds)
init = E.randinit_from_bbox(lo, hi, cnt, xdim, ydim, 7)
cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xdim, ydim)
N = 16384
ref = {}
for u in [int(v) for v in os.environ.get("ONLINE_U_LIST", "2,4,8,16").split(",")]:
    os.environ["SOMHIP_ONLINE_U"] = str(u)
    for tag, it0 in (("head (radius 128)", 0), ("tail (radius 2)", 9_900_000)):
        cb.upload(init)
        p = SomParams(L, 0.05, 128.0, E.ALPHA_LINEAR, 0, 0, 1, it0, 2048, 0)
        E.check(eng.lib.somhip_som_train(cb.h, ds.h, C.byref(p), None, None))      # warm (graph capture)
        cb.upload(init)
        eng.sync()
        t0 = time.perf_counter()
        p = SomParams(L, 0.05, 128.0, E.ALPHA_LINEAR, 0, 0, 1, it0, N, 0)
        E.check(eng.lib.somhip_som_train(cb.h, ds.h, C.byref(p), None, None))
        eng.sync()
        dt = time.perf_counter() - t0
        got = cb.download()
        same = ref.setdefault(tag, got) is got or np.array_equal(ref[tag].view(np.uint32), got.view(np.uint32))
        print("U %2d  %-18s %7.2f us/iteration  %8.0f vectors/s  bits %s" % (u, tag, 1e6 * dt / N, N / dt, "equal" if same else "DIFFER"), flush=True)
