"""gemm_ablate.py -- time of k_som_update_gemm for one batch at 256x256x512 (library chosen by SOMHIP_LIB)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from som_lvq_pak_amd import engine as E
eng = E.Engine(0)
xd, yd, d, n, B = 256, 256, 512, 8192, 4096
ds = E.Dataset(eng, generate=(11, 16, d, 0, n))
lo, hi, cnt = E.column_minmax(ds)
init = E.randinit_from_bbox(lo, hi, cnt, xd, yd, 5)
eng.set_update_mode("gemm")
for radius in (128.0, 40.0, 10.0, 3.0):
    cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xd, yd)
    E.som_train(cb, ds, B, 0.05, radius, batch=B, trace=False)
    ts = []
    for rep in range(3):
        cb.upload(init)
        s0 = eng.scan_stats()
        eng.timing(True); eng.timing_reset()
        E.som_train(cb, ds, B, 0.05, radius, batch=B, trace=False)
        eng.sync(); eng.timing(False)
        tab = eng.timing_table()
        s1 = eng.scan_stats()
        ts.append(tab["k_som_update_gemm"][1] * 1e-3)
    walked = s1["gemm_entries"] - s0["gemm_entries"]
    print("%s radius %g: update %.3f ms (min of 3), walked %d entries, %.1f TFLOP/s executed"
          % (os.environ.get("SOMHIP_LIB", "default"), radius, 1e3 * min(ts), walked, 2.0 * d * 64 * walked / min(ts) / 1e12))
    cb.close()
