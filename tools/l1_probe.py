#!/usr/bin/env python3
"""l1_probe.py -- level 1 of the winner search, ring kernel (k_dist_mfma_bf16_l1r) against the two-buffer kernel of
round 2 (k_dist_mfma_bf16_l1w16, SOMHIP_L1_NORING=1): same packed winner keys bit for bit on a set of shapes (edge
tiles, group counts that are not multiples of 4, short K), and the HIP-event time of the level-1 launch and of the whole
search at the configs[3] shape for several batch lengths.

    python tools/l1_probe.py [--quick]
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from som_lvq_pak_amd import engine as E
from som_lvq_pak_amd._lib import check


def keys_of(eng, cb, ds, first, count, buf):
    check(eng.lib.somhip_batch_winner_keys(cb.h, ds.h, first, count, C.c_void_p(buf.data_ptr())))
    eng.sync()
    return buf[:count].cpu().numpy().copy()


def main():
    quick = "--quick" in sys.argv
    eng = E.Engine(0)
    rs = np.random.RandomState(5)
    buf = torch.empty(65536, dtype=torch.int64, device="cuda")
    bad = 0
    # ---- parity on shapes with edges: (rows, dim, samples) ----
    shapes = [(512 * 64, 64, 5000), (513 * 64 + 17, 128, 4100), (600 * 64, 192, 300), (515 * 64, 512, 9000), (1024 * 64, 512, 33000)]
    if quick:
        shapes = shapes[:3]
    for n, d, m in shapes:
        cent = rs.standard_normal((32, d)).astype(np.float32) * 3
        codes = (cent[rs.randint(0, 32, n)] + 0.3 * rs.standard_normal((n, d))).astype(np.float32)
        x = (cent[rs.randint(0, 32, m)] + rs.standard_normal((m, d))).astype(np.float32)
        codes[7] = codes[n - 1]                           # ties across the codebook's ends
        cb = E.Codebook(eng, codes)
        ds = E.Dataset(eng, x)
        os.environ["SOMHIP_L1_NORING"] = "1"; os.environ["SOMHIP_NO_ROWMAJOR"] = "1"
        k_old = keys_of(eng, cb, ds, 0, m, buf)
        os.environ.pop("SOMHIP_L1_NORING"); os.environ.pop("SOMHIP_NO_ROWMAJOR")
        k_new = keys_of(eng, cb, ds, 0, m, buf)
        os.environ["SOMHIP_L1_RING_NOGMIN"] = "1"
        k_new2 = keys_of(eng, cb, ds, 0, m, buf)
        os.environ.pop("SOMHIP_L1_RING_NOGMIN")
        eng.set_scan_mode("direct")
        k_dir = keys_of(eng, cb, ds, 0, min(m, 2048), buf)
        eng.set_scan_mode("mfma_bf16")
        same = np.array_equal(k_old, k_new) and np.array_equal(k_old, k_new2) and np.array_equal(k_old[:len(k_dir)], k_dir)
        print("parity rows %6d dim %4d samples %5d: %s" % (n, d, m, "equal" if same else "DIFFERENT (%d / %d / %d)" % (
            int((k_old != k_new).sum()), int((k_old != k_new2).sum()), int((k_old[:len(k_dir)] != k_dir).sum()))), flush=True)
        bad += 0 if same else 1
        cb.close(); ds.close()
    # ---- timing at the configs[3] shape ----
    n, d = 65536, 512
    L = 65536
    ds = E.Dataset(eng, generate=(3456, 256, d, 0, L))
    lo, hi, cnt = E.column_minmax(ds)
    init = E.randinit_from_bbox(lo, hi, cnt, 256, 256, 7)
    cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, 256, 256)
    # a partly organised map (so that the later stages of the search see realistic survivor counts)
    eng.set_update_mode("gemm")
    E.som_train(cb, ds, 10_000_000, 0.05, 128.0, batch=32768, start_iter=0, count=65536, trace=False)
    eng.sync()
    cb_rows = cb.download()
    for B in ((32768,) if quick else (4096, 8192, 32768)):
        for mode in ("old", "ring, pairs from tiles", "ring"):
            os.environ.pop("SOMHIP_L1_NORING", None); os.environ.pop("SOMHIP_NO_ROWMAJOR", None)
            if mode == "old":
                os.environ["SOMHIP_L1_NORING"] = "1"
            if mode != "ring":
                os.environ["SOMHIP_NO_ROWMAJOR"] = "1"
            cb.upload(cb_rows)                               # (the codebook's prepared copies are made again under this mode)
            eng.timing(True); eng.timing_reset()
            ref = keys_of(eng, cb, ds, 0, B, buf)             # (the call that prepares the codebook's copies: bf16 tiles, norms, row-major rows)
            prep_us = 1e3 * eng.timing_table()["k_norms_tau"][1]
            eng.timing_reset()
            reps = 8
            t0 = time.perf_counter()
            for r in range(reps):
                check(eng.lib.somhip_batch_winner_keys(cb.h, ds.h, (r * B) % (L - B + 1), B, C.c_void_p(buf.data_ptr())))
            eng.sync()
            wall = (time.perf_counter() - t0) / reps
            tab = eng.timing_table()
            eng.timing(False)
            l1 = tab["k_dist_mfma_bf16"]
            tot = sum(v[1] for v in tab.values()) / reps
            print("B %5d %-24s level 1 %8.1f us/launch (%.3f of 2.5 Pflop/s)  all kernels %8.1f us  wall %8.1f us  prep+tau (first call) %6.1f us   %s" % (
                B, mode, 1e3 * l1[1] / max(l1[0], 1), 2.0 * n * d * B / (l1[1] / max(l1[0], 1) * 1e-3) / 2.5e15,
                1e3 * tot, 1e6 * wall, prep_us, " ".join("%s=%.0f" % (k, 1e3 * v[1] / reps) for k, v in tab.items() if v[0] and k != "k_dist_mfma_bf16")), flush=True)
            if mode == "old":
                ref_old = ref
            elif not np.array_equal(ref, ref_old):
                print("   keys DIFFER from the two-buffer kernel's (%d of %d)" % (int((ref != ref_old).sum()), B))
                bad += 1
    os.environ.pop("SOMHIP_L1_NORING", None); os.environ.pop("SOMHIP_NO_ROWMAJOR", None)
    print("l1_probe:", "OK" if not bad else "%d FAILURES" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
