#!/usr/bin/env python3
"""conformity_study.py -- does a mini-batch schedule end where the reference's online algorithm ends?

BASELINE.json's tolerance is "qerror within 1e-4 of the CPU reference".  The reference is strictly
online (som_rout.c:600-662); the engine's batch = 1 path is bit-exact with it.  This tool trains the
configs[3] map (256x256 hexa bubble, dim 512, alpha 0.05 linear, radius 128 -> 1) on the seeded
generator stream (`-din gen:k=256,dim=512,n=..,seed=3456`, randinit -rand 7) at several run lengths
with several batch sizes, and -- for the lengths listed in --online -- with the online engine, and
prints the final qerror (qerror.c semantics on the first --eval vectors of the stream) of every run.

    python tools/conformity_study.py --lengths 1048576 10000000 --batches 256 1024 4096 --online 10000000

Progress is printed at least every few seconds (the 10 M online run takes minutes).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--xdim", type=int, default=256)
    ap.add_argument("--ydim", type=int, default=256)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--seed", type=int, default=3456)
    ap.add_argument("--init-seed", type=int, default=7)
    ap.add_argument("--alpha", type=float, default=0.05)
    ap.add_argument("--radius", type=float, default=None)
    ap.add_argument("--lengths", type=int, nargs="+", default=[1048576])
    ap.add_argument("--batches", type=int, nargs="+", default=[256, 1024, 4096])
    ap.add_argument("--online", type=int, nargs="*", default=[])
    ap.add_argument("--eval", type=int, default=262144)
    ap.add_argument("--out", default="gpurun_out/conformity.json")
    a = ap.parse_args()
    from som_lvq_pak_amd import engine as E

    radius = a.radius if a.radius is not None else max(a.xdim, a.ydim) / 2.0
    nmax = max(a.lengths + a.online)
    eng = E.Engine(0)
    t0 = time.time()
    ds = E.Dataset(eng, generate=(a.seed, a.k, a.dim, 0, nmax))
    lo, hi, cnt = E.column_minmax(ds)
    init = E.randinit_from_bbox(lo, hi, cnt, a.xdim, a.ydim, a.init_seed)
    print("data %d x %d generated + bbox + init in %.1f s" % (nmax, a.dim, time.time() - t0), flush=True)
    ne = min(a.eval, min(a.lengths + a.online))
    results = []

    def qerr(cb):
        _, diff, ret = E.find_winners(cb, ds, 0, ne)
        return float(E.qerror_sum(diff, ret) / np.float32(ne))

    def record(**kw):
        results.append(kw)
        print(json.dumps(kw), flush=True)
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump({"config": vars(a), "radius": radius, "eval_vectors": ne, "runs": results}, open(a.out, "w"), indent=1)

    for L in a.lengths:
        for B in a.batches:
            cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, a.xdim, a.ydim)
            t0 = time.time()
            E.som_train(cb, ds, L, a.alpha, radius, batch=B, trace=False)
            eng.sync()
            dt = time.time() - t0
            record(length=L, batch=B, qerror=qerr(cb), seconds=dt, vectors_per_s=L / dt)
            cb.close()
    for L in a.online:
        cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, a.xdim, a.ydim)
        t0 = time.time()
        seg = 1 << 18
        for s in range(0, L, seg):
            c = min(seg, L - s)
            E.som_train(cb, ds, L, a.alpha, radius, batch=1, start_iter=s, count=c, data_first=s, trace=False)
            print("  online %d / %d  (%.0f s)" % (s + c, L, time.time() - t0), flush=True)
        eng.sync()
        dt = time.time() - t0
        q = qerr(cb)
        record(length=L, batch=1, qerror=q, seconds=dt, vectors_per_s=L / dt)
        cb.close()
    # summary: every mini-batch run against the online run of the same length (when there is one)
    base = {r["length"]: r["qerror"] for r in results if r["batch"] == 1}
    for r in results:
        if r["batch"] != 1 and r["length"] in base:
            print("L %9d  B %5d  qerror %.6f  online %.6f  rel %.3e" % (
                r["length"], r["batch"], r["qerror"], base[r["length"]], (r["qerror"] - base[r["length"]]) / base[r["length"]]), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
