#!/usr/bin/env python3
"""step_probe.py -- per-kernel HIP-event times of ONE mini-batch step of the configs[3] workload at several points of the
10 M-iteration schedule (the radius and alpha of that point; the map is first trained for a few batches of the schedule's
head so that winners are spread).  Shows which kernels cost what at which radius.

    python tools/step_probe.py [batch]      default 32768
"""
import os
import sys

import numpy as np
import torch

torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from som_lvq_pak_amd import engine as E
from som_lvq_pak_amd import sharded
from som_lvq_pak_amd._lib import SomParams

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L, xdim, ydim, d = 10_000_000, 256, 256, 512
eng = E.Engine(0)
eng.set_update_mode("gemm")
ds = E.Dataset(eng, generate=(3456, 256, d, 0, 8 * B))
lo, hi, cnt = E.column_minmax(ds)
init = E.randinit_from_bbox(lo, hi, cnt, xdim, ydim, 7)
cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xdim, ydim)
gs = sharded.GpuShard(eng, cb, ds, lambda: SomParams(L, 0.05, 128.0, E.ALPHA_LINEAR, 0, 0, B, 0, 0, 0), B)
ss = sharded.ShardedSom(gs, B, L)
for k in range(4):
    ss.step(k * B, k * B, B)
eng.sync()
names = None
for frac in (0.0, 0.25, 0.5, 0.75, 0.85, 0.95, 0.99):
    it0 = int(frac * L) // B * B
    ss.step(it0, 4 * B, B)                                     # warm at this radius
    eng.sync()
    eng.timing(True); eng.timing_reset()
    reps = 3
    for r in range(reps):
        ss.step(it0, (5 + r) * B % (7 * B), B)
    eng.sync()
    eng.timing(False)
    tab = {k: v[1] / reps * 1e3 for k, v in eng.timing_table().items() if v[0]}
    if names is None:
        names = sorted(tab, key=lambda k: -tab[k])
        print("%-10s %8s  " % ("at", "radius") + " ".join("%14s" % n[:14] for n in names) + "   total us")
    print("%-10.2f %8.1f  " % (frac, 1 + 127 * (1 - it0 / L)) + " ".join("%14.1f" % tab.get(n, 0.0) for n in names) + "   %8.1f" % sum(tab.values()), flush=True)
print("scan stats", eng.scan_stats())
