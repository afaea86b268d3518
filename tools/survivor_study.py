"""survivor_study.py -- how selective would a ONE-product bf16 pre-filter be over a real configs[3] run?
For several positions of the 10 M-iteration schedule: the group minima of s~ = ||c||^2 - 2<c,x> for 512 samples
(somhip_debug_prefilter), and how many of the 1024 row groups lie within tau1 = 2 * 2^-7 ||c||max ||x|| of the
sample's minimum (the error of dropping both lo terms), against the three-product tau."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from som_lvq_pak_amd import engine as E
import ctypes as C
from som_lvq_pak_amd import _lib
L = 10000000
eng = E.Engine(0)
eng.set_update_mode("gemm")
ds = E.Dataset(eng, generate=(3456, 256, 512, 0, L))
lo, hi, cnt = E.column_minmax(ds)
init = E.randinit_from_bbox(lo, hi, cnt, 256, 256, 7)
cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, 256, 256)
pos = 0
ng, nb = 1024, 512
for frac in (0.0, 0.002, 0.01, 0.03, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99):
    target = int(frac * L) // 4096 * 4096
    if target > pos:
        E.som_train(cb, ds, L, 0.05, 128.0, batch=4096, start_iter=pos, count=target - pos, data_first=pos, trace=False)
        pos = target
    wmin = np.empty((ng, nb), dtype=np.float32)
    tau = np.empty(nb, dtype=np.float32)
    bpad = C.c_int64(0)
    E.check(eng.lib.somhip_debug_prefilter(cb.h, ds.h, pos, nb, wmin.ctypes.data_as(_lib.c_float_p), tau.ctypes.data_as(_lib.c_float_p), C.byref(bpad)))
    codes = cb.download()
    cmax = float(np.sqrt((codes.astype(np.float64) ** 2).sum(1).max()))
    x = ds.rows(pos, nb)
    xn = np.sqrt((x.astype(np.float64) ** 2).sum(1))
    tau1 = 2.0 * 2.0 ** -7 * cmax * xn * 1.05
    gmin = wmin.min(0)
    surv3 = (wmin <= gmin[None, :] + tau[None, :]).sum(0)
    surv1 = (wmin <= gmin[None, :] + tau1[None, :]).sum(0)
    print("at %5.1f %% (radius %5.1f): ||c||max %.1f, tau3 %.3f, tau1 %.1f; groups within tau3 %.2f, within tau1 mean %.1f median %.0f max %d of %d"
          % (100 * frac, 1 + 127 * (1 - frac), cmax, float(tau.mean()), float(tau1.mean()), surv3.mean(), surv1.mean(), np.median(surv1), surv1.max(), ng), flush=True)
