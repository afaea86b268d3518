"""pds_study.py -- would a partial-distance level in front of level 1 pay?  (an exact technique: the squared distance
over the first K' dims, plus the squared difference of the norms of the remaining dims block by block, is a lower
bound of the whole squared distance, so a (code, sample) pair whose bound exceeds the sample's best whole distance
cannot win).  For several positions of the configs[3] schedule: how many of the 1024 row groups hold a code whose
bound is below the sample's best whole distance, for K' = 64 / 128 / 256 and remaining dims in blocks of 32 / all."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from som_lvq_pak_amd import engine as E
L = 10000000
eng = E.Engine(0)
eng.set_update_mode("gemm")
ds = E.Dataset(eng, generate=(3456, 256, 512, 0, L))
lo, hi, cnt = E.column_minmax(ds)
init = E.randinit_from_bbox(lo, hi, cnt, 256, 256, 7)
cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, 256, 256)
pos, nb = 0, 1024
dev = "cuda"
for frac in (0.0, 0.01, 0.05, 0.15, 0.3, 0.5, 0.7, 0.85, 0.95, 0.995):
    target = int(frac * L) // 32768 * 32768
    if target > pos:
        E.som_train(cb, ds, L, 0.05, 128.0, batch=32768, start_iter=pos, count=target - pos, data_first=pos, trace=False)
        pos = target
    c = torch.from_numpy(cb.download()).to(dev).double()
    x = torch.from_numpy(ds.rows(pos, nb)).to(dev).double()
    full = (x * x).sum(1)[:, None] + (c * c).sum(1)[None, :] - 2 * x @ c.T            # [nb, N]
    best = full.min(1).values
    line = "at %5.1f %% (radius %5.1f): best d2 %.0f |" % (100 * frac, 1 + 127 * (1 - frac), float(best.mean()))
    for K in (64, 128, 256):
        for blk in (0, 448 if K == 64 else 512 - K, 32):
            xp, cp = x[:, :K], c[:, :K]
            if blk:
                xr = x[:, K:].reshape(nb, -1, blk).norm(dim=2)
                cr = c[:, K:].reshape(c.shape[0], -1, blk).norm(dim=2)
                xp, cp = torch.cat([xp, xr], 1), torch.cat([cp, cr], 1)
            lb = (xp * xp).sum(1)[:, None] + (cp * cp).sum(1)[None, :] - 2 * xp @ cp.T
            assert bool((lb <= full + 1e-6).all())
            g = lb.reshape(nb, 1024, 64).min(2).values
            surv = (g <= best[:, None]).sum(1).double()
            line += " K'=%d+%s: %.1f" % (K, "0" if not blk else "%dx1" % ((512 - K) // blk), float(surv.mean()))
        line += " |"
    print(line, flush=True)
