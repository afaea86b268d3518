#!/usr/bin/env python3
"""prune_study.py -- how much of the winner search a triangle-inequality pre-pass over one representative row per 64-row
group would prune along configs[3]'s schedule: a (group, sample) pair survives iff d(x, rep_g) - r_g <= min_g' d(x, rep_g')
(rep_g = a row of the group, r_g = max_n ||c_n - rep_g||).  Also: the share of (4 groups x 256 samples) GEMM tiles that
still hold a survivor when the batch's samples are sorted by their nearest representative."""
import json
import os
import sys

import numpy as np
import torch

torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from som_lvq_pak_amd import engine as E

L, xdim, ydim, d, B = 10_000_000, 256, 256, 512, 16384
eng = E.Engine(0)
eng.set_update_mode("gemm")
ds = E.Dataset(eng, generate=(3456, 256, d, 0, L))
lo, hi, cnt = E.column_minmax(ds)
init = E.randinit_from_bbox(lo, hi, cnt, xdim, ydim, 7)
cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xdim, ydim)
dev = torch.device("cuda")
it = 0
for frac in (0.0, 0.02, 0.05, 0.1, 0.2, 0.35, 0.5, 0.65, 0.8, 0.9, 0.97, 1.0):
    end = min(L, int(frac * L) // B * B)
    if end > it:
        E.som_train(cb, ds, L, 0.05, 128.0, batch=-1, start_iter=it, count=end - it, trace=False)
        it = end
    codes = torch.from_numpy(cb.download()).to(dev)                       # [65536, 512], unit order
    # 8x8 patches of units = the engine's row groups
    c4 = codes.view(ydim // 8, 8, xdim // 8, 8, d).permute(0, 2, 1, 3, 4).reshape(-1, 64, d)   # [1024 groups, 64, d]
    rep = c4[:, 27, :]                                                    # a row near the patch centre
    r = (c4 - rep[:, None, :]).double().norm(dim=2).max(dim=1).values     # [1024]
    first = min(it, L - B)
    x = torch.from_numpy(ds.rows(first, B)).to(dev)
    dist = torch.cdist(x.double(), rep.double())                           # [B, 1024]
    U = dist.min(dim=1).values
    surv = (dist - r[None, :]) <= U[:, None]                               # [B, 1024]
    g1 = dist.argmin(dim=1)
    order = torch.argsort(g1, stable=True)
    s2 = surv[order].view(B // 256, 256, 1024 // 4, 4)
    tiles = s2.any(dim=3).any(dim=1)                                       # [B/256, 256 group-quads]
    radius = 1.0 + (128.0 - 1.0) * (L - it) / L
    print(json.dumps({"iteration": it, "radius": round(radius, 1), "pairs_surviving": round(float(surv.float().mean()), 4),
                      "tiles_surviving_sorted": round(float(tiles.float().mean()), 4),
                      "tiles_surviving_unsorted": round(float(surv.view(B // 256, 256, 256, 4).any(dim=3).any(dim=1).float().mean()), 4),
                      "r_mean": round(float(r.mean()), 2), "U_mean": round(float(U.mean()), 2)}), flush=True)
