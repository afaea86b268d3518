#!/usr/bin/env python3
"""Randomised check of the sharded winner search with exchanged bounds (somhip_shard_winner_begin / _refine / _finish)
against the CPU oracle's find_winner_euc over the whole codebook, and against the plain per-shard search + MIN.
Random shard cuts (uneven, some tiny), dims, sample counts, codebooks whose shards differ wildly in norm and distance
from the data (far shards that must abstain, shards of tiny norm, duplicated rows across shards).
Run on a GPU box:  python tools/fuzz_exchange.py [seconds] [seed]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import Oracle  # noqa: E402
from som_lvq_pak_amd import engine as E  # noqa: E402

NONE = np.uint64(0x7FFFFFFFFFFFFFFF)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def host(e, dev, n, dt):
    out = np.empty(n, dtype=dt)
    assert e.lib.somhip_copy_to_host(e.h, out.ctypes.data_as(C.c_void_p), dev, out.nbytes) == 0
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = np.random.RandomState(seed)
    orc = Oracle()
    S_MAX = 4
    engs = [E.Engine(0) for _ in range(S_MAX)]
    t0, cases, abstained, samples = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        d = int(rs.choice([32, 64, 96, 128, 256, 512, 1024]))
        m = int(rs.choice([225, 256, 300, 480, 1000, 2049]))
        S = int(rs.randint(2, S_MAX + 1))
        n = int(rs.choice([300, 700, 1500, 5000]))
        scale = float(rs.choice([1e-3, 1.0, 1.0, 30.0]))
        x = (scale * rs.standard_normal((m, d))).astype(np.float32)
        codes = (x[rs.randint(0, m, n)] + scale * float(rs.uniform(0.05, 1.5)) * rs.standard_normal((n, d))).astype(np.float32)
        cuts = sorted(set([0, n] + [int(v) for v in rs.randint(64, n - 64, S - 1)]))
        cuts = [c for i, c in enumerate(cuts) if i == 0 or c - cuts[i - 1] >= 64 or c == n]
        if cuts[-1] - cuts[-2] < 64:
            cuts.pop(-2)
        S = len(cuts) - 1
        for a, b in zip(cuts, cuts[1:]):                          # what a shard may look like
            r = rs.rand()
            if r < 0.2:
                codes[a:b] += np.float32(scale * rs.uniform(5, 60))          # nowhere near the data
            elif r < 0.3:
                codes[a:b] *= np.float32(1e-4)                     # tiny norm
            elif r < 0.4:
                codes[a:b] *= np.float32(rs.uniform(2, 20))        # large norm
        for _ in range(int(rs.randint(0, 6))):                    # the same row in two places: the lower index wins
            i, j = rs.randint(0, n, 2)
            codes[i] = codes[j] = x[rs.randint(0, m)]
        want_i, want_d, _ = orc.winners(codes, x)
        want_i, want_d = np.asarray(want_i).reshape(-1), np.asarray(want_d).reshape(-1)
        first = int(rs.randint(0, m))
        count = int(rs.randint(225, m + 1))
        sel = (first + np.arange(count)) % m
        dss = [E.Dataset(engs[s], x) for s in range(S)]
        cbs = [E.Codebook(engs[s], codes[a:b], row_offset=a, n_global=n) for s, (a, b) in enumerate(zip(cuts, cuts[1:]))]
        kb = [engs[s].device_alloc(8 * count) for s in range(S)]
        bb = [engs[s].device_alloc(4 * count) for s in range(S)]

        def exchange():
            hb = [host(engs[s], bb[s], count, np.float32) for s in range(S)]
            mn = np.ascontiguousarray(np.minimum.reduce(hb))
            for s in range(S):
                assert engs[s].lib.somhip_copy_to_device(engs[s].h, bb[s], mn.ctypes.data_as(C.c_void_p), 4 * count) == 0

        ok = all(engs[s].lib.somhip_shard_exchange_available(cbs[s].h, dss[s].h, count) == 1 for s in range(S))
        assert ok, ("not available", d, m, n, cuts, count)
        for s in range(S):
            assert engs[s].lib.somhip_shard_winner_begin(cbs[s].h, dss[s].h, first, count, kb[s], bb[s]) == 0
        exchange()
        for s in range(S):
            assert engs[s].lib.somhip_shard_winner_refine(cbs[s].h, dss[s].h, first, count, bb[s]) == 0
        exchange()
        own = []
        for s in range(S):
            assert engs[s].lib.somhip_shard_winner_finish(cbs[s].h, dss[s].h, first, count, bb[s], kb[s]) == 0
            own.append(host(engs[s], kb[s], count, np.uint64))
        merged = np.minimum.reduce(own)
        plain = []
        for s in range(S):
            assert engs[s].lib.somhip_batch_winner_keys(cbs[s].h, dss[s].h, first, count, kb[s]) == 0
            plain.append(host(engs[s], kb[s], count, np.uint64))
        good = (np.array_equal((merged & np.uint64(0xFFFFFFFF)).astype(np.int64), want_i[sel]) and
                np.array_equal((merged >> np.uint64(32)).astype(np.uint32), bits(want_d[sel])) and
                np.array_equal(np.minimum.reduce(plain), merged))
        if not good:
            print("MISMATCH", dict(d=d, m=m, n=n, cuts=cuts, first=first, count=count, scale=scale, seed=seed, case=cases))
            sys.exit(1)
        abstained += int(sum((o == NONE).sum() for o in own))
        samples += count * S
        cases += 1
        for s in range(S):
            engs[s].device_free(kb[s])
            engs[s].device_free(bb[s])
            cbs[s].close()
            dss[s].close()
    print("fuzz_exchange: %d cases ok in %.0f s (seed %d): %d (shard, sample) searches, %d of them abstained"
          % (cases, time.time() - t0, seed, samples, abstained))


if __name__ == "__main__":
    main()
