"""The N > 1 path on CPU ranks: world_size 2, gloo.

What is exercised is the product's orchestration (som_lvq_pak_amd/sharded.py: shard ranges,
packed keys, the signed-vs-unsigned MIN fix, all-reduce, owner-only updates); the per-shard
compute -- a GPU kernel in production -- is supplied here by the CPU checker so the result
can be compared bit for bit with the unsharded oracle run."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class CheckerShard:
    """winner_keys/update for one shard, computed with the CPU oracle.  `units`: the global unit index of
    every local row (a contiguous block, or the interleaved 8x8 patches of somhip_shard_units)."""

    def __init__(self, orc, rows, units, n_global, xdim, topol, neigh, data, length, alpha, radius):
        import torch
        self.torch = torch
        self.orc, self.rows, self.units = orc, rows.copy(), np.asarray(units, dtype=np.int64)
        self.xdim, self.topol, self.neigh = xdim, topol, neigh
        self.data, self.length, self.alpha, self.radius = data, length, alpha, radius

    def winner_keys(self, first, count):
        from som_lvq_pak_amd.sharded import pack_keys
        idx = [(first + j) % self.data.shape[0] for j in range(count)]
        wi, wd, _ = self.orc.winners(self.rows, self.data[idx])
        keys = pack_keys(wd[:, 0], self.units[wi[:, 0]])
        return self.torch.from_numpy(keys.view(np.int64).copy())

    def update(self, it0, count, first, keys):
        from som_lvq_pak_amd.sharded import unpack_keys
        _, widx = unpack_keys(keys.numpy())
        for j in range(count):
            le = it0 + j
            x = self.data[(first + j) % self.data.shape[0]]
            trad = self.orc.som_radius(le, self.length, self.radius)
            talp = self.orc.alpha(1, le, self.length, self.alpha)
            bx, by = int(widx[j] % self.xdim), int(widx[j] // self.xdim)
            for k in range(self.rows.shape[0]):
                g = int(self.units[k])
                dd = self.orc.mapdist(self.topol, bx, by, g % self.xdim, g // self.xdim)
                if self.neigh == 2:
                    self.rows[k] = self.orc.adapt_vector(self.rows[k], x, self.orc.gaussian_h(dd, trad, talp))
                elif dd <= trad:
                    self.rows[k] = self.orc.adapt_vector(self.rows[k], x, talp)

    def collective_scope(self):
        import contextlib
        return contextlib.nullcontext()


class CheckerExchangeShard(CheckerShard):
    """The three-call winner search of sharded.ShardedSom (GpuShard.winner_begin / _refine / _finish in production:
    somhip_shard_winner_*) on the CPU checker: exact distances stand in for the pre-filter's values, a wide and a narrow
    slack for its two error bounds.  What is under test is the orchestration: the two float MIN all-reduces, the order
    of the calls, a shard that abstains."""

    D1, D3 = np.float32(2.0), np.float32(0.05)             # the "error bounds" of the two levels

    def exchange_available(self, count):
        return True

    def _s(self, first, count):
        idx = [(first + j) % self.data.shape[0] for j in range(count)]
        x = self.data[idx]
        n = self.rows.shape[0]
        wi, wd, _ = self.orc.winners(self.rows, x, n, True)              # every row's exact distance (find_winner_knn, k = n) ...
        order = np.argsort((wd.view(np.uint32).astype(np.uint64) << np.uint64(32)) | wi.astype(np.uint64), axis=1)
        wi, wd = np.take_along_axis(wi, order, 1), np.take_along_axis(wd, order, 1)   # ... in (distance, index) order
        xx = (x.astype(np.float64) ** 2).sum(1)[:, None]
        return wi, wd, (wd.astype(np.float64) - xx).astype(np.float32)   # s = distance - ||x||^2, the same on every shard

    def winner_begin(self, first, count):
        self.wi, self.wd, self.s = self._s(first, count)
        self.bound = self.torch.from_numpy((self.s[:, 0] + self.D1).copy())
        self.phase = 1
        return self.bound

    def winner_refine(self, first, count):
        assert self.phase == 1
        ub = self.bound.numpy().copy()                                   # the MIN over the shards
        assert (ub <= self.s[:, 0] + self.D1).all()
        self.kept = self.s <= (ub + self.D1)[:, None]                    # rows this shard still has to look at
        best = np.where(self.kept.any(1), np.where(self.kept, self.s, np.float32(3.4e38)).min(1) + self.D3, np.float32(3.4e38))
        self.bound.copy_(self.torch.from_numpy(best.astype(np.float32)))
        self.phase = 2
        return self.bound

    def winner_finish(self, first, count):
        from som_lvq_pak_amd.sharded import pack_keys
        assert self.phase == 2
        ub = self.bound.numpy()
        cand = self.kept & (self.s <= (ub + self.D3)[:, None])
        keys = np.full(count, np.uint64(0x7FFFFFFFFFFFFFFF), dtype=np.uint64)
        self.abstained = getattr(self, "abstained", 0) + int((~cand.any(1)).sum())
        for b in np.where(cand.any(1))[0]:
            j = int(np.argmax(cand[b]))                                  # first candidate in (distance, index) order
            keys[b] = pack_keys(self.wd[b, j:j + 1], self.units[self.wi[b, j:j + 1]])[0]
        self.phase = 0
        return self.torch.from_numpy(keys.view(np.int64).copy())


class CheckerLvqShard:
    """topk_keys / merge / candidates / apply of sharded.ShardedLvq for one shard, on the CPU checker.  Serves only ITS
    rows to the exchange; for `apply` (replicated on every rank in production) it keeps a private replica of the whole
    codebook, checks that the exchanged candidate rows equal the replica's, and walks the batch one sample at a time."""

    def __init__(self, orc, codes, clab, r0, r1, data, dlab, kind, length, alpha, winlen=0.0, epsilon=0.0):
        import torch
        self.torch, self.orc = torch, orc
        self.full, self.clab, self.r0, self.r1 = codes.copy(), clab, r0, r1
        self.data, self.dlab, self.kind, self.length, self.alpha = data, dlab, kind, length, alpha
        self.winlen, self.epsilon = winlen, epsilon
        self.knn = 2 if kind in (3, 4) else 1
        self.talpha = np.full(codes.shape[0], alpha, dtype=np.float32)
        self.d4 = (codes.shape[1] + 3) // 4

    def _keys(self, rows, base, x):
        """every row's exact key for every sample, ascending (tag = global row, complemented for the k-NN tie rule)"""
        n = rows.shape[0]
        wi, wd, _ = self.orc.winners(rows, x, n, True)
        tag = (wi + base).astype(np.uint64)
        if self.knn == 2:
            tag = (~tag) & np.uint64(0xFFFFFFFF)
        keys = (wd.view(np.uint32).astype(np.uint64) << np.uint64(32)) | tag
        return np.sort(keys, axis=1)

    def _samples(self, first, count):
        return self.data[[(first + j) % self.data.shape[0] for j in range(count)]]

    def topk_keys(self, first, count):
        keys = self._keys(self.full[self.r0:self.r1], self.r0, self._samples(first, count))
        out = np.full((count, 8), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
        k = min(8, keys.shape[1])
        out[:, :k] = keys[:, :k]
        return self.torch.from_numpy(out.view(np.int64).copy())

    def merge(self, gathered, count):
        g = gathered.numpy().view(np.uint64)                       # [world, count, 8]
        allk = np.sort(np.concatenate(list(g), axis=1), axis=1)[:, :8]
        return self.torch.from_numpy(np.ascontiguousarray(allk).view(np.int64).copy())

    def _rows_of(self, keys):
        k = keys.numpy().view(np.uint64)
        none = k == np.uint64(0xFFFFFFFFFFFFFFFF)
        tag = k & np.uint64(0xFFFFFFFF)
        row = ((~tag) & np.uint64(0xFFFFFFFF) if self.knn == 2 else tag).astype(np.int64)
        row[none] = -1
        return row

    def candidates(self, keys, count, xrows):
        row = self._rows_of(keys)
        mine = (row >= self.r0) & (row < self.r1)
        lab = np.where(mine, self.clab[np.clip(row, 0, None)], 0).astype(np.int32)
        ta = np.where(mine, self.talpha[np.clip(row, 0, None)], 0).astype(np.float32) if self.kind == 2 else None
        rows = np.zeros((count, xrows, 4 * self.d4), dtype=np.float32)
        d = self.full.shape[1]
        for j in range(count):
            for c in range(xrows):
                if mine[j, c]:
                    rows[j, c, :d] = self.full[row[j, c]]
        t = self.torch.from_numpy
        return t(lab), (t(ta) if ta is not None else None), t(rows)

    def apply(self, it0, count, first, keys, lab, ta, rows, xrows):
        row = self._rows_of(keys)
        d = self.full.shape[1]
        got_rows, got_lab = rows.numpy(), lab.numpy()
        for j in range(count):                                  # the exchange delivered every listed row it promised
            for c in range(8):
                if row[j, c] >= 0:
                    assert got_lab[j, c] == self.clab[row[j, c]]
                    if c < xrows:
                        assert np.array_equal(got_rows[j, c, :d].view(np.uint32), self.full[row[j, c]].view(np.uint32))
        ti = np.zeros(count * self.knn, dtype=np.int32)
        td = np.zeros(count * self.knn, dtype=np.float32)
        orc = self.orc
        ratio = np.float32((np.float32(1) - np.float32(self.winlen)) / (np.float32(1) + np.float32(self.winlen)))
        for j in range(count):                                  # lvq_rout.c:542-555, 650-673, 855-896, one sample at a time
            r = (first + j) % self.data.shape[0]
            x, xl = self.data[r], self.dlab[r]
            wi, wd, _ = orc.winners(self.full, x[None, :], self.knn, self.knn == 2)
            ti[j * self.knn:(j + 1) * self.knn], td[j * self.knn:(j + 1) * self.knn] = wi[0], wd[0]
            a = orc.alpha(1, it0 + j, self.length, self.alpha)
            b = int(wi[0, 0])
            if self.kind == 1:
                self.full[b] = orc.adapt_vector(self.full[b], x, a if self.clab[b] == xl else -a)
            elif self.kind == 2:
                t = self.talpha[b]
                if self.clab[b] == xl:
                    self.full[b] = orc.adapt_vector(self.full[b], x, t)
                    self.talpha[b] = np.float32(t / np.float32(1 + t))
                else:
                    self.full[b] = orc.adapt_vector(self.full[b], x, -t)
                    self.talpha[b] = min(np.float32(t / np.float32(1 - t)), np.float32(self.alpha))
            else:
                nb = int(wi[0, 1])
                if self.clab[b] != self.clab[nb]:
                    if (self.clab[b] == xl or self.clab[nb] == xl) and np.float32(wd[0, 0] / wd[0, 1]) > ratio:
                        if self.clab[nb] == xl:
                            b, nb = nb, b
                        self.full[b] = orc.adapt_vector(self.full[b], x, a)
                        self.full[nb] = orc.adapt_vector(self.full[nb], x, -a)
                elif self.kind == 4 and self.clab[b] == xl:
                    ae = np.float32(np.float32(a) * np.float32(self.epsilon))
                    self.full[b] = orc.adapt_vector(self.full[b], x, ae)
                    self.full[nb] = orc.adapt_vector(self.full[nb], x, ae)
        return count, ti, td

    def collective_scope(self):
        import contextlib
        return contextlib.nullcontext()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import Oracle
    from som_lvq_pak_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = Oracle()
        out = {}
        cases = [(8, 6, 3, 1, 16, 160, False), (7, 5, 4, 2, 8, 96, False), (24, 8, 3, 1, 16, 96, True)]
        for case, (xdim, ydim, topol, neigh, B, length, interleaved) in enumerate(cases):
            x, _ = synth(50 + case, 70, 6)
            ini = orc.randinit(x, xdim, ydim, 2 + case)
            n = xdim * ydim
            if interleaved:          # the layout bench.py uses for N > 1: 8x8 patches dealt round-robin (3 patches, 2 ranks)
                from som_lvq_pak_amd import engine as E
                units = E.shard_units(xdim, ydim, rank, world)
            else:
                r0, r1 = sharded.shard_rows(n, world, rank)
                units = np.arange(r0, r1)
            sh = CheckerShard(orc, ini[units], units, n, xdim, topol, neigh, x, length, 0.09, 3.0)
            som = sharded.ShardedSom(sh, B, x.shape[0])
            winners = som.train(length)
            widx = np.concatenate([sharded.unpack_keys(w.numpy())[1] for w in winners])
            full = sharded.gather_codebook(sh.rows, units, n)          # X3: uneven shards, both layouts
            if rank == 0:
                want, wi, _ = orc.som_train(ini, xdim, ydim, topol, neigh, x, length, 0.09, 3.0, batch=B)
                out[case] = (bool(np.array_equal(full.view(np.uint32), want.view(np.uint32))),
                             bool(np.array_equal(widx, wi)))
        # the winner search with the bounds exchanged between its levels (what 8 ranks and more do; asked for here)
        os.environ["SOMHIP_SHARD_EXCHANGE"] = "1"
        x, _ = synth(58, 80, 6)
        ini = orc.randinit(x, 9, 8, 4)
        ini[36:] += np.float32(3.0)                          # rank 1's rows lie far off: it abstains for most samples
        r0, r1 = sharded.shard_rows(72, world, rank)
        sh = CheckerExchangeShard(orc, ini[r0:r1], np.arange(r0, r1), 72, 9, 3, 1, x, 128, 0.09, 3.0)
        som = sharded.ShardedSom(sh, 16, x.shape[0])
        winners = som.train(128)
        widx = np.concatenate([sharded.unpack_keys(w.numpy())[1] for w in winners])
        full = sharded.gather_codebook(sh.rows, np.arange(r0, r1), 72)
        ab = torch.tensor([getattr(sh, "abstained", 0)], dtype=torch.int64)
        dist.all_reduce(ab)
        if rank == 0:
            want, wi, _ = orc.som_train(ini, 9, 8, 3, 1, x, 128, 0.09, 3.0, batch=16)
            out["exchange"] = (bool(np.array_equal(full.view(np.uint32), want.view(np.uint32))), bool(np.array_equal(widx, wi)),
                               som._exch.get(16), int(ab.item()) > 0)
        del os.environ["SOMHIP_SHARD_EXCHANGE"]
        # X2: every rank contributes the k best keys of its shard; the sorted union is the global k-NN
        x, _ = synth(60, 90, 7)
        codes = np.concatenate([x[:50], x[:15]]).astype(np.float32)        # duplicates: knn tie order matters
        n = codes.shape[0]
        r0, r1 = sharded.shard_rows(n, world, rank)
        wi, wd, _ = orc.winners(codes[r0:r1], x, 2, True)
        tag = (~(wi + r0).astype(np.uint64)) & np.uint64(0xFFFFFFFF)
        keys = (wd.view(np.uint32).astype(np.uint64) << np.uint64(32)) | tag
        keys[wi < 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
        merged = sharded.allgather_topk_keys(keys, 2)
        gd, gi = sharded.unpack_knn_keys(merged)
        fi, fd, _ = orc.winners(codes, x, 2, True)
        if rank == 0:
            out["knn"] = (bool(np.array_equal(gi, fi)), bool(np.array_equal(gd.view(np.uint32), fd.view(np.uint32))))
        # the LVQ loops over a row-sharded codebook: all-gather of candidate lists, integer all-reduce of their rows
        x, lab = synth(77, 160, 9, k=4)
        pick = np.random.RandomState(3).choice(160, 37, replace=False)
        codes, clab = x[pick].copy(), lab[pick].copy()
        r0, r1 = sharded.shard_rows(37, world, rank)
        for kind, kw in ((1, {}), (2, {}), (4, {"winlen": 0.3, "epsilon": 0.2})):
            sh = CheckerLvqShard(orc, codes, clab, r0, r1, x, lab, kind, 300, 0.07, **kw)
            lv = sharded.ShardedLvq(sh, kind, x.shape[0], xrows=3, max_batch=64)
            ti, td = lv.train(300)
            got = sharded.gather_codebook(sh.full[r0:r1], np.arange(r0, r1), 37)
            if rank == 0:
                wc, wt, wi, wd = orc.lvq_train(kind, codes, clab, x, lab, 300, 0.07, **kw)
                out["lvq%d" % kind] = (bool(np.array_equal(got.view(np.uint32), wc.view(np.uint32))), bool(np.array_equal(ti, wi)),
                                       bool(np.array_equal(td.view(np.uint32), wd.view(np.uint32))), lv.batches)
        # the all-ones "no winner" key must lose a signed MIN
        k = torch.tensor([-1 if rank == 0 else 5, 7 + rank], dtype=torch.int64)
        sharded.allreduce_min_keys(k)
        if rank == 0:
            out["none_key"] = k.tolist()
            q.put(out)
    finally:
        dist.destroy_process_group()


def test_sharded_training_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0] == (True, True)
    assert out[1] == (True, True)
    assert out[2] == (True, True)          # interleaved patches
    assert out["none_key"] == [5, 7]
    assert out["exchange"] == (True, True, True, True)      # same bits as the unsharded run; the path was taken; a shard abstained
    assert out["knn"] == (True, True)
    for kind in (1, 2, 4):
        assert out["lvq%d" % kind][:3] == (True, True, True), kind
        assert out["lvq%d" % kind][3] >= 5                       # 300 iterations in batches of at most 64


def test_shard_rows_and_keys():
    from som_lvq_pak_amd import sharded
    for n, w in ((65536, 8), (1024, 3), (10, 4), (5, 8)):
        spans = [sharded.shard_rows(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    d = np.array([0.0, 1.5, 3.25e7, 1e-30], dtype=np.float32)
    i = np.array([0, 7, 65535, 4000000000], dtype=np.uint64)
    k = sharded.pack_keys(d, i)
    dd, ii = sharded.unpack_keys(k)
    assert np.array_equal(dd, d) and np.array_equal(ii, i.astype(np.int64))
    # order of keys == (distance, index) order
    order = np.argsort(k)
    assert list(order) == [0, 3, 1, 2]
    assert (k.view(np.int64) >= 0).all()
