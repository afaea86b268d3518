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
    assert out["knn"] == (True, True)


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
