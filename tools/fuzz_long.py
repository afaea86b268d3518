#!/usr/bin/env python3
"""Randomised sweep of LONG mini-batches (16384 ... 32768 iterations per batch) against the batch oracle: the paths only
long runs take -- winners decoded once (k_decode_winners), k_som_members in two phases per trip with deep trips, the
persistent level-1 ring kernel on maps with few row groups (dims in whole 64s), the row-major re-rank copy -- on small
maps the CPU oracle can replay.  Exact update kernels; codebook bits and winner traces must be equal.
    python tools/fuzz_long.py [seconds] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import synth  # noqa: E402
from oracle import Oracle  # noqa: E402
from som_lvq_pak_amd import engine as E  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    orc, eng = Oracle(), E.Engine(0)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        d = int(rs.choice([8, 12, 16, 64, 64]))
        side = [16, 24, 32, 40, 48, 64] if d < 64 else [16, 24, 32]
        xdim, ydim = [int(v) for v in rs.choice(side, 2)]
        topol = int(rs.choice([3, 4]))
        B = int(rs.choice([16384, 20000, 32768]))
        nvec = int(rs.choice([6000, 9000, 40000]))
        nb = int(rs.choice([1, 2, 3]))
        L = nb * B + int(rs.choice([0, 0, 777]))
        radius = float(rs.uniform(0.8, max(xdim, ydim) / 1.5))
        alpha = float(rs.choice([0.02, 0.05, 0.2]))
        x, _ = synth(int(rs.randint(1, 10000)), nvec, d, k=int(rs.randint(2, 12)), spread=float(rs.uniform(1, 4)))
        fixed = None
        if rs.rand() < 0.4:
            fixed = np.full((nvec, 2), -1, dtype=np.int16)
            for r in rs.choice(nvec, 20, replace=False):
                fixed[r] = (rs.randint(0, xdim + 3), rs.randint(0, ydim + 3))
        ini = orc.randinit(x, xdim, ydim, int(rs.randint(1, 100)))
        kw = dict(fixed_xy=fixed, fixed_on=1) if fixed is not None else {}
        oc, oi, od = orc.som_train(ini, xdim, ydim, topol, 1, x, L, alpha, radius, batch=B, **kw)
        cb = E.Codebook(eng, ini, topol, 1, xdim, ydim)
        ds = E.Dataset(eng, x, fixed_xy=fixed)
        ti, td = E.som_train(cb, ds, L, alpha, radius, use_fixed=1 if fixed is not None else 0, batch=B)
        ok = np.array_equal(ti, oi) and np.array_equal(bits(td), bits(od)) and np.array_equal(bits(cb.download()), bits(oc))
        n += 1
        if not ok:
            bad += 1
            print("MISMATCH", dict(d=d, xdim=xdim, ydim=ydim, topol=topol, B=B, nvec=nvec, L=L, radius=radius, alpha=alpha,
                                   fixed=fixed is not None, winners=int((ti != oi).sum())), flush=True)
        cb.close(); ds.close()
    print("fuzz_long: %d cases in %.0f s, %d mismatches" % (n, time.time() - t0, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
