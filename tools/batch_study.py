#!/usr/bin/env python3
"""How far does the mini-batch schedule drift from the reference's online algorithm?

Trains the same map on the same stream with batch = 1 (bit-exact with the CPU reference)
and batch = 16 ... 4096, and prints the final qerror (qerror.c semantics: mean over the
data of sqrt(min squared distance)) of each against the online run.

  python tools/batch_study.py [xdim ydim dim nvec radius]     default 32 32 128 100000 10
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from som_lvq_pak_amd import engine as E  # noqa: E402


def mixture(seed, n, d, k):
    rs = np.random.RandomState(seed)
    centres = (4.0 * rs.standard_normal((k, d))).astype(np.float32)
    which = rs.randint(0, k, size=n)
    return (centres[which] + rs.standard_normal((n, d)).astype(np.float32)).astype(np.float32)


def main():
    xdim, ydim, dim, nvec, radius = (32, 32, 128, 100000, 10.0)
    if len(sys.argv) >= 6:
        xdim, ydim, dim, nvec = [int(a) for a in sys.argv[1:5]]
        radius = float(sys.argv[5])
    x = mixture(1234, nvec, dim, 16)
    rs = np.random.RandomState(7)
    lo, hi = x.min(0), x.max(0)
    init = (lo + (hi - lo) * rs.rand(xdim * ydim, dim)).astype(np.float32)
    eng = E.Engine(0)
    ds = E.Dataset(eng, x)
    base = None
    print("map %dx%d dim %d, %d vectors, alpha 0.05 linear, radius %g -> 1" % (xdim, ydim, dim, nvec, radius))
    print("%8s %14s %14s %12s %10s" % ("batch", "qerror", "delta", "rel", "vec/s"))
    for batch in (1, 4, 16, 64, 256, 1024, 4096):
        cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xdim, ydim)
        t0 = time.time()
        E.som_train(cb, ds, nvec, 0.05, radius, batch=batch, trace=False)
        eng.sync()
        dt = time.time() - t0
        _, diff, ret = E.find_winners(cb, ds)
        q = float(E.qerror_sum(diff, ret) / np.float32(nvec))
        if base is None:
            base = q
        print("%8d %14.6f %14.6f %12.3e %10.0f" % (batch, q, q - base, (q - base) / base, nvec / dt))
        cb.close()


if __name__ == "__main__":
    main()
