#!/usr/bin/env python3
"""Randomised parity sweep: GPU engine vs the CPU oracle on generated shapes and options.
Not part of the pytest suite (minutes); run on a GPU box:  python tools/fuzz_parity.py [seconds] [seed]
Every case must agree bit for bit (codebook, winner indices, distances)."""
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
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    budget = float(args[0]) if len(args) > 0 else 120.0
    seed = int(args[1]) if len(args) > 1 else 1
    rs = np.random.RandomState(seed)
    orc, eng = Oracle(), E.Engine(0)
    t0, n_som, n_lvq, n_win = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        kind = rs.randint(0, 3)
        if '--verbose' in sys.argv:
            print('case kind', kind, flush=True)
        d = int(rs.choice([3, 5, 16, 20, 31, 32, 33, 64, 100, 128, 130, 256]))
        if kind == 0:                                            # SOM training, online or mini-batch
            xdim, ydim = [int(v) for v in rs.choice([5, 7, 8, 9, 12, 16, 24, 32, 40], 2)]
            topol, neigh = int(rs.choice([3, 4])), int(rs.choice([1, 1, 2]))
            nvec = int(rs.choice([150, 400, 1000]))
            x, _ = synth(int(rs.randint(1, 10000)), nvec, d, k=int(rs.randint(1, 9)), spread=float(rs.uniform(0.5, 5)))
            batch = int(rs.choice([1, 1, 7, 32, 33, 100, 256, 1000]))
            length = int(rs.choice([300, 700, 1500])) if batch > 1 else int(rs.choice([200, 500]))
            alpha, radius = float(rs.uniform(0.01, 0.3)), float(rs.uniform(0.5, max(xdim, ydim)))
            at = int(rs.choice([1, 2]))
            mask = (rs.random_sample(x.shape) < 0.1).astype(np.uint8) if rs.rand() < 0.25 else None
            weight = rs.randint(0, 4, nvec).astype(np.int16) if rs.rand() < 0.2 else None
            fixed = None
            if rs.rand() < 0.2:
                fixed = -np.ones((nvec, 2), dtype=np.int16)
                sel = rs.rand(nvec) < 0.1
                fixed[sel, 0] = rs.randint(0, xdim, sel.sum()); fixed[sel, 1] = rs.randint(0, ydim, sel.sum())
            ini = orc.randinit(x, xdim, ydim, int(rs.randint(1, 1000)))
            want, wi, wd = orc.som_train(ini, xdim, ydim, topol, neigh, x, length, alpha, radius, alpha_type=at, weight=weight,
                                         fixed_xy=fixed, mask=mask, fixed_on=int(fixed is not None),
                                         weights_on=int(weight is not None), batch=batch)
            cb = E.Codebook(eng, ini, topol, neigh, xdim, ydim)
            ds = E.Dataset(eng, x, mask=mask, weight=weight, fixed_xy=fixed)
            ti, td = E.som_train(cb, ds, length, alpha, radius, alpha_type=at, use_fixed=int(fixed is not None),
                                 use_weights=int(weight is not None), batch=batch)
            got = cb.download()
            ok = np.array_equal(bits(got), bits(want)) and np.array_equal(ti, wi) and np.array_equal(bits(td), bits(wd))
            desc = ("som", xdim, ydim, d, topol, neigh, nvec, batch, length, alpha, radius, at, mask is not None,
                    weight is not None, fixed is not None)
            n_som += 1
        elif kind == 1:                                          # LVQ training (exact batched engine)
            n, nvec = int(rs.choice([4, 9, 30, 200, 900])), int(rs.choice([120, 500, 1500]))
            x, lab = synth(int(rs.randint(1, 10000)), nvec, d, k=int(rs.randint(1, 8)), spread=float(rs.uniform(0.3, 4)))
            pick = rs.randint(0, nvec, n)
            codes = (x[pick] + 0.1 * rs.standard_normal((n, d))).astype(np.float32)
            clab = lab[pick].copy()
            lk = int(rs.choice([1, 2, 3, 4]))
            kw = {}
            if lk >= 3:
                kw["winlen"] = float(rs.uniform(0.1, 0.5))
            if lk == 4:
                kw["epsilon"] = float(rs.uniform(0.05, 0.5))
            length, alpha = int(rs.choice([300, 1000, 2500])), float(rs.uniform(0.01, 0.3))
            try:
                oc, ol, oi, od = orc.lvq_train(lk, codes, clab, x, lab, length, alpha, **kw)
            except FloatingPointError:                     # rate too large for this data: codes diverge
                continue
            cb = E.Codebook(eng, codes, labels=clab)
            ds = E.Dataset(eng, x, labels=lab)
            tal, ti, td = E.lvq_train(cb, ds, lk, length, alpha, **kw)
            ok = np.array_equal(bits(cb.download()), bits(oc)) and np.array_equal(ti, oi) and np.array_equal(bits(td), bits(od))
            if lk == 2:
                ok = ok and np.array_equal(bits(tal), bits(ol))
            desc = ("lvq", lk, n, d, nvec, length, alpha, kw)
            n_lvq += 1
        else:                                                    # winner scans, k-NN and masks
            n, nvec = int(rs.choice([10, 64, 65, 500, 3000, 9000])), int(rs.choice([31, 32, 200, 1000, 4100]))
            x, _ = synth(int(rs.randint(1, 10000)), nvec, d, k=int(rs.randint(1, 8)))
            codes = (x[rs.randint(0, nvec, n)] + rs.choice([0.0, 0.5]) * rs.standard_normal((n, d))).astype(np.float32)
            knn = int(rs.choice([1, 1, 2, 3, 5, 8]))
            knn = min(knn, n)
            mask = (rs.random_sample(x.shape) < 0.1).astype(np.uint8) if (knn == 1 and rs.rand() < 0.3) else None
            wi, wd, wr = orc.winners(codes, x, knn, knn > 1, mask)
            cb, ds = E.Codebook(eng, codes), E.Dataset(eng, x, mask=mask)
            gi, gd, gr = E.find_winners(cb, ds, knn=knn, tie=E.TIE_KNN if knn > 1 else E.TIE_FIRST)
            live = wr != 0 if wr is not None else slice(None)
            ok = np.array_equal(gi[live], wi[live]) and np.array_equal(bits(gd[live]), bits(wd[live]))
            desc = ("win", n, d, nvec, knn, mask is not None)
            n_win += 1
        if '--verbose' in sys.argv:
            print(desc, ok, flush=True)
        if not ok:
            print("MISMATCH", desc)
            sys.exit(1)
        if '--verbose' in sys.argv:
            print('closing', flush=True)
        cb.close(); ds.close()
        if '--verbose' in sys.argv:
            print('closed', flush=True)
    print("fuzz ok: %d som, %d lvq, %d winner cases in %.0f s (seed %d)" % (n_som, n_lvq, n_win, time.time() - t0, seed))


if __name__ == "__main__":
    main()
