#!/usr/bin/env python3
"""Randomised sweep of update mode gemm against the exact kernels (which equal the batch oracle bit for bit): random map
shapes (8x8-patch order and linear), dims in whole 128s, batch lengths that are no multiple of a chunk, bubble and gaussian
neighbourhoods, several batches per run with the rate decaying to zero inside the schedule.  Same winners in the first batch;
the codebooks within fp32 rounding of each other (later batches see slightly different codebooks, so their winners may differ
at ties -- the runs are therefore compared batch by batch from the SAME start).   python tools/fuzz_gemm.py [seconds] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from som_lvq_pak_amd import engine as E  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    eng = E.Engine(0)
    t0, n, worst, worst_pair = time.time(), 0, 0.0, 0.0
    while time.time() - t0 < budget:
        xd, yd = [int(v) for v in rs.choice([6, 8, 12, 16, 24, 32, 40, 64], 2)]
        d = int(rs.choice([128, 256, 384, 512]))
        topol, neigh = int(rs.choice([3, 4])), int(rs.choice([1, 2]))
        B = int(rs.choice([17, 100, 255, 256, 600, 1000, 2049, 4096]))
        nvec = int(rs.choice([B, 2 * B + 3, 5000]))
        radius = float(rs.uniform(0.6, max(xd, yd)))
        alpha = float(rs.choice([0.02, 0.05, 0.3, 0.9]))
        length = int(rs.choice([B, 4 * B, 64 * B]))
        ds = E.Dataset(eng, generate=(int(rs.randint(1, 1 << 30)), int(rs.randint(1, 12)), d, 0, nvec))
        lo, hi, cnt = E.column_minmax(ds)
        init = E.randinit_from_bbox(lo, hi, cnt, xd, yd, int(rs.randint(1, 1000)))
        ds_rows = ds.rows(0, nvec).astype(np.float64)
        it0 = int(rs.randint(0, max(1, length - B + 1)))
        got = {}
        for mode in ("exact", "gemm"):
            eng.set_update_mode(mode)
            cb = E.Codebook(eng, init, topol, neigh, xd, yd)
            ti, _ = E.som_train(cb, ds, length, alpha, radius, batch=B, start_iter=it0, count=min(B - it0 % B, length - it0))
            got[mode] = (cb.download(), ti)
            cb.close()
        eng.set_update_mode("exact")
        ds.close()
        a, b = got["exact"][0], got["gemm"][0]
        case = (xd, yd, d, topol, neigh, B, nvec, radius, alpha, length, it0)
        assert np.array_equal(got["exact"][1], got["gemm"][1]), case
        assert np.isfinite(b).all(), case
        scale = float(np.abs(a).max())
        err = float(np.abs(a - b).max()) / scale
        # the yardstick: a float64 replay of the batch on a few units (the exact kernels' fp32 chain of thousands of
        # hits has an error of its own, as large as the GEMM's: the two may differ by their sum)
        units = rs.choice(xd * yd, min(48, xd * yd), replace=False)
        win = got["exact"][1]
        cnt_it = len(win)
        c = init[units].astype(np.float64)
        ux, uy = (units % xd).astype(np.int64), (units // xd).astype(np.int64)
        rows = ds_rows
        for j in range(cnt_it):
            it = it0 + j
            al = np.float32(np.float32(alpha) * np.float32(length - it) / np.float32(length))
            rad = np.float32(1.0) + np.float32(radius - 1.0) * np.float32(length - it) / np.float32(length)
            if win[j] < 0:
                continue
            bx, by = int(win[j]) % xd, int(win[j]) // xd
            dx = (bx - ux).astype(np.float64)
            dy = (by - uy).astype(np.float64)
            if topol == 3:
                odd = ((by - uy) % 2) != 0
                dx = np.where(odd, dx + (0.5 if by % 2 else -0.5), dx)
                lat = dx * dx + 0.75 * dy * dy
            else:
                lat = dx * dx + dy * dy
            if neigh == 1:
                r = np.where(np.sqrt(lat).astype(np.float32) <= rad, np.float64(al), 0.0)
            else:
                dd = np.sqrt(lat).astype(np.float32)
                neg = -(dd * dd)
                r = np.float64(al) * np.exp(neg.astype(np.float64) / (2.0 * float(rad) * float(rad))).astype(np.float32).astype(np.float64)
            c += r[:, None] * (rows[(it % nvec)][None, :] - c)
        e_exact = float(np.abs(a[units] - c).max()) / scale
        e_gemm = float(np.abs(b[units] - c).max()) / scale
        # (a unit that sits on the rim of a neighbourhood to within an ulp can fall on the other side in the float64
        # replay's own radius arithmetic: then BOTH kernels differ from the replay by a whole hit -- the replay's doing;
        # the largest difference between the two kernels is reported beside it)
        worst = max(worst, e_gemm)
        worst_pair = max(worst_pair, err)
        # a sum of k products accumulated in fp32 carries ~ sqrt(k) 2^-24 of the magnitude of the sum (the exact kernels'
        # chain is contractive -- every step damps the earlier roundings by 1 - a -- and stays near 1e-6)
        # (e_gemm is the LARGEST of ~20 000 element errors: about four standard deviations of a random walk of k roundings
        # of a sum a few times the scale -- hence the factor 12)
        allow = max(2.0 * e_exact, 12.0 * np.sqrt(cnt_it) * 5.96e-8, 3e-6)
        assert e_gemm <= allow, (case, e_exact, e_gemm, allow)
        assert err <= max(2.0 * allow, 10.0 * (e_exact + e_gemm), 2e-5), (case, err, e_exact, e_gemm)   # all units: the sample may miss the worst
        n += 1
        if n % 20 == 0:
            print("%d cases, worst error of the gemm result against the float64 replay %.2e of the scale, against the exact kernels %.2e" % (n, worst, worst_pair), flush=True)
    print("fuzz gemm ok: %d cases in %.0f s, worst error of the gemm result against a float64 replay %.2e of the scale (bound per case: 12 sqrt(hits) 2^-24, or twice the exact kernels' own error); largest difference between the gemm and the exact kernels %.2e of the scale" % (n, time.time() - t0, worst, worst_pair))


if __name__ == "__main__":
    main()
