#!/usr/bin/env python3
"""gauss_probe.py -- the bit-equal gaussian update (k_som_update_gauss_s) at the configs[3] shape: time per batch of
4096 vectors at several radii: K4g (k_som_update_gauss_s, SOMHIP_GAUSS_K4G=1), K4h (k_som_update_gauss_h) with the library
chain for the rate (SOMHIP_GAUSS_LIBM=1) and with the short form (kernels/gauss_rate.hpp); the three codebooks compared bit
for bit.
    python tools/gauss_probe.py [batch]"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from som_lvq_pak_amd import engine as E  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    eng = E.Engine(0)
    eng.set_update_mode("exact")
    ds = E.Dataset(eng, generate=(3456, 256, 512, 0, 4 * B))
    lo, hi, cnt = E.column_minmax(ds)
    init = E.randinit_from_bbox(lo, hi, cnt, 256, 256, 7)
    bad = 0
    for radius in [float(r) for r in os.environ.get("GAUSS_PROBE_RADII", "128,40,8,2").split(",")]:
        res = {}
        for mode in ("libm", "short", "k4h"):                # K4g; K4h with the library chain; K4h
            os.environ.pop("SOMHIP_GAUSS_LIBM", None); os.environ.pop("SOMHIP_GAUSS_K4G", None)
            if mode == "libm":
                os.environ["SOMHIP_GAUSS_K4G"] = "1"
            if mode == "short":
                os.environ["SOMHIP_GAUSS_LIBM"] = "1"
            cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_GAUSSIAN, 256, 256)
            E.som_train(cb, ds, 10_000_000, 0.05, radius, batch=B, start_iter=0, count=B, trace=False)   # warm
            eng.timing(True); eng.timing_reset()
            E.som_train(cb, ds, 10_000_000, 0.05, radius, batch=B, start_iter=B, count=2 * B, data_first=B, trace=False)
            eng.sync()
            tab = eng.timing_table(); eng.timing(False)
            upd = tab.get("k_som_update_run", (0, 0.0))
            res[mode] = (cb.download(), upd[1] / max(upd[0], 1), sum(v[1] for v in tab.values()) / 2)
            cb.close()
        same = all(np.array_equal(res["libm"][0].view(np.uint32), res[m][0].view(np.uint32)) for m in ("short", "k4h"))
        bad += 0 if same else 1
        print("radius %6.1f batch %d: update kernel %7.3f ms (K4g) -> %7.3f ms (K4h, library chain for the rate) -> %7.3f ms (K4h); whole step %7.3f -> %7.3f -> %7.3f ms; codebooks %s" % (
            radius, B, res["libm"][1], res["short"][1], res["k4h"][1], res["libm"][2], res["short"][2], res["k4h"][2], "bit-equal" if same else "DIFFERENT"), flush=True)
    os.environ.pop("SOMHIP_GAUSS_LIBM", None); os.environ.pop("SOMHIP_GAUSS_K4G", None)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
