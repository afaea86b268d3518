"""gemm_check.py -- the GEMM-form mini-batch update (kernels/som_update_gemm.hpp) against the exact kernels: one batch from
the same codebook at several shapes / radii, largest difference and the time of the update kernel alone."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from som_lvq_pak_amd import engine as E
eng = E.Engine(0)
rs = np.random.RandomState(3)
for (xd, yd, d, n, B, radius, alpha) in ((64, 64, 128, 8192, 2048, 30.0, 0.05), (64, 64, 128, 8192, 2048, 3.0, 0.02),
                                          (40, 24, 256, 4096, 4096, 20.0, 0.05), (256, 256, 512, 8192, 4096, 128.0, 0.05),
                                          (256, 256, 512, 8192, 4096, 20.0, 0.01)):
    ds = E.Dataset(eng, generate=(11, 16, d, 0, n))
    lo, hi, cnt = E.column_minmax(ds)
    init = E.randinit_from_bbox(lo, hi, cnt, xd, yd, 5)
    res = {}
    for mode in ("exact", "gemm"):
        eng.set_update_mode(mode)
        cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xd, yd)
        E.som_train(cb, ds, B, alpha, radius, batch=B, trace=False)   # warm
        cb.upload(init)
        s0 = eng.scan_stats()
        eng.timing(True); eng.timing_reset()
        eng.sync(); t0 = time.time()
        ti, _ = E.som_train(cb, ds, B, alpha, radius, batch=B)
        eng.sync(); dt = time.time() - t0
        eng.timing(False)
        tab = eng.timing_table()
        dt = sum(tab[k][1] for k in ("k_som_update_gemm", "k_som_update_bubble_s", "k_som_update_run")) * 1e-3
        s1 = eng.scan_stats()
        res[mode] = (cb.download(), ti, dt, s1["gemm_entries"] - s0["gemm_entries"], s1["group_updates"] - s0["group_updates"], s1["row_updates"] - s0["row_updates"])
        cb.close()
    st = eng.scan_stats()
    a, b = res["exact"][0], res["gemm"][0]
    scale = np.abs(a).max()
    print("%dx%dx%d B %d r %g a %g: winners equal %s, max|diff| %.3e (scale %.2f, rel %.2e), rms %.2e; update kernel: exact %.3f ms gemm %.3f ms"
          % (xd, yd, d, B, radius, alpha, np.array_equal(res["exact"][1], res["gemm"][1]), np.abs(a - b).max(), scale,
             np.abs(a - b).max() / scale, np.sqrt(((a - b) ** 2).mean()), 1e3 * res["exact"][2], 1e3 * res["gemm"][2]))
    print("      list entries %d, walked %d (%.2f), row updates %d; gemm executed %.1f TFLOP/s, algorithmic %.1f"
          % (res["gemm"][4], res["gemm"][3], res["gemm"][3] / max(res["gemm"][4], 1), res["gemm"][5],
             2.0 * d * 64 * res["gemm"][3] / res["gemm"][2] / 1e12, 3.0 * d * res["gemm"][5] / res["gemm"][2] / 1e12))
    ds.close()
