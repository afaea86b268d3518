#!/usr/bin/env python3
"""conformity_r3.py -- where does a mini-batch schedule's final qerror differ from the online (= reference) one, and why?

For one (stream seed, init seed) pair of the configs[3] workload (256x256 hexa bubble, dim 512, alpha 0.05 linear,
radius 128 -> 1, 10 M vectors of `gen:k=256,dim=512,seed=S`, `randinit -rand I`):
  * the online engine over the whole schedule (bit-exact with som_training, som_rout.c:600-662; ~6 min), unless
    --online-from names a file of an earlier call (per-sample distances of the online result);
  * every schedule given on the command line ("0.9:32768,0.1:4096" = fractions of the run : batch; a leading "x" =
    exact update mode, "auto" = SOMHIP_BATCH_AUTO);
and for each final map, over the first --eval vectors of the stream:
  qerror_f32   find_qerror's own arithmetic (float accumulator of double square roots, som_rout.c:698-715)
  mean_f64     the same mean accumulated in double (what the float accumulator approximates)
  rms_delta    root mean square of the per-sample distance differences to the online map
so that the accumulator's rounding, the sampling noise and the schedule's systematic shift can be told apart.
Results: one JSON line per run on stdout and in --out; the online per-sample distances in --save-online.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

import torch

torch.zeros(1, device="cuda")          # torch's HIP runtime first, then the engine's (see sharded.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from som_lvq_pak_amd import engine as E
from som_lvq_pak_amd import sharded
from som_lvq_pak_amd._lib import SomParams


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("schedules", nargs="*")
    ap.add_argument("--seed", type=int, default=3456)
    ap.add_argument("--init-seed", type=int, default=7)
    ap.add_argument("--xdim", type=int, default=256)
    ap.add_argument("--ydim", type=int, default=256)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--length", type=int, default=10_000_000)
    ap.add_argument("--alpha", type=float, default=0.05)
    ap.add_argument("--radius", type=float, default=None)
    ap.add_argument("--eval", type=int, default=262144)
    ap.add_argument("--eval-wide", type=int, default=2097152, help="a second, larger evaluation set (double accumulation only)")
    ap.add_argument("--online-from", default=None, help=".npz of an earlier call's online result (skips the online run)")
    ap.add_argument("--save-online", default=None)
    ap.add_argument("--no-online", action="store_true")
    ap.add_argument("--perturb-init", action="store_true",
                    help="control: the live online run starts from the initial map with ONE component moved by one ulp "
                         "(use with --online-from: the unperturbed online result is then the reference of the deltas)")
    ap.add_argument("--out", default="gpurun_out/conformity_r3.jsonl")
    a = ap.parse_args()
    L, xdim, ydim, d = a.length, a.xdim, a.ydim, a.dim
    radius = a.radius if a.radius is not None else max(xdim, ydim) / 2.0
    eng = E.Engine(0)
    ds = E.Dataset(eng, generate=(a.seed, a.k, d, 0, L))
    lo, hi, cnt = E.column_minmax(ds)
    init = E.randinit_from_bbox(lo, hi, cnt, xdim, ydim, a.init_seed)
    cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xdim, ydim)
    gs = sharded.GpuShard(eng, cb, ds, lambda: SomParams(L, a.alpha, radius, E.ALPHA_LINEAR, 0, 0, 4096, 0, 0, 0), 8192)
    ne, nw = min(a.eval, L), min(a.eval_wide, L)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    fout = open(a.out, "a")

    def distances(n):
        parts = []
        for f in range(0, n, 8192):
            ek = gs.winner_keys(f, min(8192, n - f))
            eng.sync()
            parts.append(ek.cpu().numpy().copy())
        diffs, _ = sharded.unpack_keys(np.concatenate(parts))
        return np.sqrt(diffs.astype(np.float64))               # sqrt((double)diff), som_rout.c:715 (exact image of the fp32 squared distances)

    def f32_sum(r):
        q = np.float32(0.0)
        for v in r:
            q = np.float32(np.float64(q) + v)
        return q

    def measures(tag, secs, online):
        r = distances(max(ne, nw))
        m = {"seed": a.seed, "init_seed": a.init_seed, "schedule": tag, "seconds": round(secs, 3),
             "vectors_per_s": round(L / secs) if secs else None,
             "qerror_f32": float(f32_sum(r[:ne]) / np.float32(ne)), "mean_f64": float(r[:ne].mean()),
             "mean_f64_wide": float(r[:nw].mean()), "eval": ne, "eval_wide": nw}
        if online is not None:
            m["abs_delta_f32"] = m["qerror_f32"] - online["qerror_f32"]
            m["abs_delta_f64"] = m["mean_f64"] - online["mean_f64"]
            m["abs_delta_f64_wide"] = m["mean_f64_wide"] - online["mean_f64_wide"]
            if "r" in online:
                k = min(len(online["r"]), len(r))
                m["rms_delta"] = float(np.sqrt(np.mean((r[:k] - online["r"][:k]) ** 2)))
        line = json.dumps(m)
        print(line, flush=True)
        fout.write(line + "\n")
        fout.flush()
        return m, r

    online = None
    if a.perturb_init:                                   # the reference's own algorithm, one last bit of its input changed
        base = None
        if a.online_from:
            z = np.load(a.online_from)
            r = np.sqrt(z["diff"].astype(np.float64))
            base = {"r": r, "qerror_f32": float(f32_sum(r[:ne]) / np.float32(ne)), "mean_f64": float(r[:ne].mean()),
                    "mean_f64_wide": float(r[:nw].mean())}
        pin = init.copy()
        pin[0, 0] = np.nextafter(pin[0, 0], np.float32(np.inf))
        cb.upload(pin)
        eng.sync()
        t0 = time.perf_counter()
        seg = 1 << 18
        for s in range(0, L, seg):
            p = SomParams(L, a.alpha, radius, E.ALPHA_LINEAR, 0, 0, 1, s, min(seg, L - s), s)
            E.check(eng.lib.somhip_som_train(cb.h, ds.h, __import__("ctypes").byref(p), None, None))
            print("online (perturbed init) %d / %d (%.0f s)" % (min(s + seg, L), L, time.perf_counter() - t0), file=sys.stderr, flush=True)
        eng.sync()
        measures("online, init[0][0] + 1 ulp", time.perf_counter() - t0, base)
        online = base
    elif a.online_from:
        z = np.load(a.online_from)
        r = np.sqrt(z["diff"].astype(np.float64))
        online = {"r": r, "qerror_f32": float(f32_sum(r[:ne]) / np.float32(ne)), "mean_f64": float(r[:ne].mean()),
                  "mean_f64_wide": float(r[:nw].mean())}
    elif not a.no_online:
        eng.sync()
        t0 = time.perf_counter()
        seg = 1 << 18
        for s in range(0, L, seg):
            p = SomParams(L, a.alpha, radius, E.ALPHA_LINEAR, 0, 0, 1, s, min(seg, L - s), s)
            E.check(eng.lib.somhip_som_train(cb.h, ds.h, __import__("ctypes").byref(p), None, None))
            print("online %d / %d (%.0f s)" % (min(s + seg, L), L, time.perf_counter() - t0), file=sys.stderr, flush=True)
        eng.sync()
        m, r = measures("online", time.perf_counter() - t0, None)
        online = dict(m, r=r)
        if a.save_online:
            np.savez_compressed(a.save_online, diff=(r * r).astype(np.float32), seed=a.seed, init_seed=a.init_seed)

    for spec in a.schedules:
        exact = spec.startswith("x")
        body = spec[1:] if exact else spec
        eng.set_update_mode("exact" if exact else "gemm")
        cb.upload(init)
        eng.sync()
        t0 = time.perf_counter()
        if body == "auto":
            E.som_train(cb, ds, L, a.alpha, radius, batch=E.BATCH_AUTO, trace=False)
        else:
            segs = [(float(s.split(":")[0]), int(s.split(":")[1])) for s in body.split(",")]
            it, acc = 0, 0.0
            for i, (frac, B) in enumerate(segs):
                acc += frac
                end = L if i == len(segs) - 1 else min(L, (int(acc * L) // B) * B)   # (an unaligned start makes the first batch a short one)
                if end > it:
                    E.som_train(cb, ds, L, a.alpha, radius, batch=B, start_iter=it, count=end - it, trace=False)
                it = max(it, end)
        eng.sync()
        measures(spec, time.perf_counter() - t0, online)


if __name__ == "__main__":
    main()
