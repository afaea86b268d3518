#!/usr/bin/env python3
"""Quick kernel timing probe on the GPU box (not the bench): times the scan / update /
online-step kernels on BASELINE-shaped maps with the engine's HIP-event table.

  python tools/perf_probe.py [xdim ydim dim batch]      default 256 256 512 4096
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from som_lvq_pak_amd import engine as E  # noqa: E402


def main():
    xdim, ydim, dim, B = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (256, 256, 512, 4096)
    n = xdim * ydim
    rs = np.random.RandomState(1)
    x = rs.standard_normal((max(B * 2, 8192), dim)).astype(np.float32)
    codes = rs.standard_normal((n, dim)).astype(np.float32)
    eng = E.Engine(0)
    cb = E.Codebook(eng, codes, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xdim, ydim)
    ds = E.Dataset(eng, x)
    eng.timing(True)
    import os
    eng.set_scan_mode(os.environ.get("SCAN", "mfma"))
    for label, kw, iters in (("minibatch r=big", dict(radius=xdim / 2.0, batch=B), 2 * B),
                             ("minibatch r=3", dict(radius=3.0, batch=B), 2 * B),
                             ("online r=big", dict(radius=xdim / 2.0, batch=1), 256),
                             ("online r=3", dict(radius=3.0, batch=1), 256)):
        E.som_train(cb, ds, iters, 0.05, trace=False, **kw)      # warm
        eng.timing_reset()
        t0 = time.time()
        E.som_train(cb, ds, iters, 0.05, trace=False, **kw)
        eng.sync()
        dt = time.time() - t0
        tab = eng.timing_table()
        print("%-18s %8.1f vec/s  wall %.3fs" % (label, iters / dt, dt))
        for k, (cnt, ms) in tab.items():
            if cnt:
                print("     %-20s launches %6d  avg %9.3f us  total %9.3f ms" % (k, cnt, 1e3 * ms / cnt, ms))
    eng.timing(False)
    for label, kw, iters in (("online r=big (no events)", dict(radius=xdim / 2.0, batch=1), 1024),
                             ("online r=3 (no events)", dict(radius=3.0, batch=1), 1024)):
        t0 = time.time()
        E.som_train(cb, ds, iters, 0.05, trace=False, **kw)
        eng.sync()
        dt = time.time() - t0
        print("%-26s %8.1f vec/s  %.2f us/iter" % (label, iters / dt, 1e6 * dt / iters))
    print("scan stats", eng.scan_stats())
    cbytes = n * dim * 4
    print("codebook bytes %.1f MiB, scan flops/sample (3*N*d) %.1f M" % (cbytes / 2**20, 3 * n * dim / 1e6))


if __name__ == "__main__":
    main()
