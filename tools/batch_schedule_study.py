#!/usr/bin/env python3
"""batch_schedule_study.py -- final qerror of configs[3] at its real length for mini-batch sizes that change along the
schedule: segments (fraction of the run, batch) given as "0.75:16384,0.25:4096".  Same stream, initial map and evaluation
as bench.py / profiles/r02_c4_full_length.json (online result 22.569894790649414)."""
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

ONLINE = 22.569894790649414
L, xdim, ydim, d = 10_000_000, 256, 256, 512
eng = E.Engine(0)
eng.set_update_mode("gemm")
ds = E.Dataset(eng, generate=(3456, 256, d, 0, L))
lo, hi, cnt = E.column_minmax(ds)
init = E.randinit_from_bbox(lo, hi, cnt, xdim, ydim, 7)
cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xdim, ydim)
for spec in sys.argv[1:]:
    segs = [(float(s.split(":")[0]), int(s.split(":")[1])) for s in spec.split(",")]
    cb.upload(init)
    eng.sync()
    t0 = time.perf_counter()
    it = 0
    for i, (frac, B) in enumerate(segs):
        end = L if i == len(segs) - 1 else min(L, (int(it + frac * L) // B) * B)
        if end > it:
            E.som_train(cb, ds, L, 0.05, 128.0, batch=B, start_iter=it, count=end - it, trace=False)
        it = end
    eng.sync()
    secs = time.perf_counter() - t0
    ne = 262144
    parts = []
    # qerror exactly as bench.py does it (winner keys over the first 262144 vectors)
    from som_lvq_pak_amd._lib import SomParams
    gs = sharded.GpuShard(eng, cb, ds, lambda: SomParams(L, 0.05, 128.0, E.ALPHA_LINEAR, 0, 0, 4096, 0, 0, 0), 8192)
    for f in range(0, ne, 8192):
        ek = gs.winner_keys(f, 8192)
        eng.sync()
        parts.append(ek.cpu().numpy().copy())
    diffs, _ = sharded.unpack_keys(np.concatenate(parts))
    q = float(E.qerror_sum(diffs) / np.float32(ne))
    print(json.dumps({"schedule": spec, "seconds": round(secs, 3), "vectors_per_s": round(L / secs), "final_qerror": q,
                      "rel_to_online": (q - ONLINE) / ONLINE}), flush=True)
