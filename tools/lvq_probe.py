#!/usr/bin/env python3
"""Throughput probe of the online LVQ engine at BASELINE configs[2] shape (OLVQ1, 10 000 codes,
dim 256, synthetic 100-class mixture) + the CPU reference on a short prefix.
  python tools/lvq_probe.py [ncodes dim nvec iters]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from som_lvq_pak_amd import engine as E  # noqa: E402


def main():
    ncodes, dim, nvec, iters = (10000, 256, 200000, 100000)
    if len(sys.argv) >= 5:
        ncodes, dim, nvec, iters = [int(a) for a in sys.argv[1:5]]
    rs = np.random.RandomState(2345)
    k = 100
    cent = (4.0 * rs.standard_normal((k, dim))).astype(np.float32)
    lab = rs.randint(0, k, nvec)
    x = (cent[lab] + rs.standard_normal((nvec, dim)).astype(np.float32)).astype(np.float32)
    lab = (lab + 1).astype(np.int32)
    pick = np.concatenate([np.where(lab == c + 1)[0][:ncodes // k] for c in range(k)])
    codes, clab = x[pick].copy(), lab[pick].copy()
    eng = E.Engine(0)
    ds = E.Dataset(eng, x, labels=lab)
    for kind, name, kw in ((E.OLVQ1, "olvq1", dict(alpha=0.3)), (E.LVQ1, "lvq1", dict(alpha=0.05)),
                           (E.LVQ3, "lvq3", dict(alpha=0.05, winlen=0.3, epsilon=0.1))):
        cb = E.Codebook(eng, codes, labels=clab)
        s0 = eng.lvq_stats()
        eng.timing_reset()
        eng.timing(True)
        t0 = time.time()
        E.lvq_train(cb, ds, kind, iters, trace=False, **kw)
        eng.sync()
        dt = time.time() - t0
        eng.timing(False)
        s1 = eng.lvq_stats()
        nb = s1["batches"] - s0["batches"]
        if nb:
            print("   exact batches: %d (%.1f samples each; %d ended by the candidate list, %d by the cache; %.1f independent components "
                  "per batch, longest walk %.1f samples); kernels ms: %s"
                  % (nb, iters / nb, s1["stop_list"] - s0["stop_list"], s1["stop_cache"] - s0["stop_cache"],
                     (s1["components"] - s0["components"]) / nb, (s1["largest"] - s0["largest"]) / nb,
                     {k: round(v[1], 1) for k, v in eng.timing_table().items() if v[0]}))
            print("   in-order kernel, us per sample by phase (inputs, distances, decision, correction):",
                  [round((a - b) / iters, 2) for a, b in zip(s1["phase_us"], s0["phase_us"])])
        ds2 = E.Dataset(eng, x[:20000])
        wi, _, _ = E.find_winners(cb, ds2)
        acc = float((clab[wi[:, 0]] == lab[:20000]).mean())
        print("%-6s %d codes x %d: %8.0f vectors/s (%.2f us/iter), accuracy on 20k train vectors %.4f"
              % (name, len(codes), dim, iters / dt, 1e6 * dt / iters, acc))
        cb.close()
    try:
        from oracle import RefHarness, ref_available
        if ref_available():
            ref = RefHarness()
            n = 300
            ref.lvq_train(2, codes, clab, x, lab, n, 0.3, trace=False)
            print("reference olvq1_training, 1 core: %.1f vectors/s" % (n / ref.last_seconds))
    except Exception as exc:       # pragma: no cover
        print("no reference:", exc)


if __name__ == "__main__":
    main()
