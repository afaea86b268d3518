"""Row-sharded SOM and LVQ training across ranks (one process per GPU, torch.distributed).

The codebook is split into contiguous row blocks; every rank sees the same batch.  Per batch:
  1. each rank finds the best row of ITS shard for every sample   -> packed keys [B]
  2. one all-reduce(MIN) of the keys                              -> global winners
  3. each rank applies the batch's neighbourhood updates to its own rows, in iteration order
The update of a row needs only the sample, the winner's lattice coordinates and the
per-iteration scalars, never another row, so there is no halo and no other collective.

Keys are uint64 = (fp32 bits of the squared distance << 32) | global row index.  Distances
are >= 0, so unsigned order == (distance, index) order and MIN reproduces find_winner_euc
over the whole codebook, lowest index winning ties (reference lvq_pak.c:79).  torch has no
uint64 reductions; int64 order agrees with uint64 order for every key whose top bit is clear,
i.e. every real key -- only the all-ones "no winner" key must be mapped first.

The local work is behind a small interface so the same orchestration runs on GPUs (bench.py:
`GpuShard`, RCCL) and in the CPU tests (gloo, a checker-backed shard).
"""
import numpy as np

KEY_NONE_I64 = -1                       # 0xFFFF_FFFF_FFFF_FFFF seen as int64
KEY_MAX_I64 = 0x7FFFFFFFFFFFFFFF


def shard_rows(n_global, world, rank):
    """Contiguous block [r0, r1) of rank `rank`; blocks differ by at most one row group."""
    per = (n_global + world - 1) // world
    r0 = min(n_global, rank * per)
    return r0, min(n_global, r0 + per)


def pack_keys(diff, index):
    """numpy: (float32 squared distance >= 0, row index) -> uint64 keys."""
    bits = np.ascontiguousarray(diff, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (bits << np.uint64(32)) | np.asarray(index, dtype=np.uint64)


def unpack_keys(keys):
    keys = np.asarray(keys).view(np.uint64)
    diff = (keys >> np.uint64(32)).astype(np.uint32).view(np.float32)
    index = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    return diff, index


def allreduce_min_keys(keys, group=None, nonnegative=False):
    """In-place global MIN of packed keys held in an int64 torch tensor (any device).
    nonnegative=True: the producer already mapped the all-ones key (somhip_batch_winner_keys does)."""
    import torch
    import torch.distributed as dist
    if not nonnegative:
        keys.copy_(torch.where(keys < 0, torch.full_like(keys, KEY_MAX_I64), keys))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if keys.is_cuda and dist.get_backend(group) == "gloo":     # rehearsal path: stage through the host
            host = keys.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MIN, group=group)
            keys.copy_(host)
        else:
            dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=group)
    return keys


def allreduce_min_floats(t, group=None):
    """Element-wise MIN of a float32 tensor over all ranks, in place (the exchanged pre-filter bounds)."""
    import torch.distributed as dist
    if dist.is_initialized():                             # (also with one rank: the collective is then a stream-ordered no-op)
        if dist.get_backend(group) == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        elif dist.get_world_size(group) > 1:
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MIN, group=group)
            t.copy_(host)
    return t


def allgather_topk_keys(keys, knn, group=None):
    """X2: `keys` = this rank's [count, knn] packed keys (uint64 numpy or int64 torch tensor, ascending);
    returns the knn smallest per sample over all ranks as a uint64 numpy array [count, knn].  Keys are
    unique per row (tag = global row, complemented for the k-NN tie order), so sorting the union IS
    find_winner_knn over the whole codebook (reference lvq_pak.c:152-221)."""
    import torch
    import torch.distributed as dist
    t = keys if isinstance(keys, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(keys).view(np.int64))
    t = t.reshape(-1, knn)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        host_staged = t.is_cuda and dist.get_backend(group) == "gloo"
        src = t.cpu() if host_staged else t
        parts = [torch.empty_like(src) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, src, group=group)
        allk = torch.cat(parts, dim=1)
    else:
        allk = t
    u = allk.cpu().numpy().view(np.uint64)
    u = np.sort(u, axis=1)[:, :knn]                  # unsigned order = (distance, tag) order
    return np.ascontiguousarray(u)


def unpack_knn_keys(keys):
    """(diff, index) of keys written with SOMHIP_TIE_KNN (tag = ~row); all-ones -> (-1, -1)."""
    keys = np.asarray(keys).view(np.uint64)
    none = keys == np.uint64(0xFFFFFFFFFFFFFFFF)
    diff = (keys >> np.uint64(32)).astype(np.uint32).view(np.float32).copy()
    index = ((~keys) & np.uint64(0xFFFFFFFF)).astype(np.int64)
    diff[none] = -1.0
    index[none] = -1
    return diff, index


def gather_codebook(local_rows, units, n_global, group=None):
    """X3 (save / snapshot): every rank contributes the rows it owns and the global unit index of each
    (contiguous block: arange(r0, r1); interleaved shard: engine.shard_units) and gets the whole codebook
    back in the reference's row order (datafile.c:781,836).  Shards may differ in size (padded to the largest
    for the all-gather).  numpy in, numpy out; the exchange runs on the process group's backend."""
    import torch
    import torch.distributed as dist
    rows = np.ascontiguousarray(local_rows, dtype=np.float32)
    units = np.ascontiguousarray(units, dtype=np.int64)
    assert rows.shape[0] == units.shape[0]
    full = np.empty((n_global, rows.shape[1]), dtype=np.float32)
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        full[units] = rows
        return full
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    count = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    most = int(max(int(c.item()) for c in counts))
    pad_rows = torch.zeros((most, rows.shape[1]), dtype=torch.float32, device=dev)
    pad_units = torch.full((most,), -1, dtype=torch.int64, device=dev)
    pad_rows[:rows.shape[0]] = torch.from_numpy(rows).to(dev)
    pad_units[:units.shape[0]] = torch.from_numpy(units).to(dev)
    all_rows = [torch.empty_like(pad_rows) for _ in range(world)]
    all_units = [torch.empty_like(pad_units) for _ in range(world)]
    dist.all_gather(all_rows, pad_rows, group=group)
    dist.all_gather(all_units, pad_units, group=group)
    for r, u, c in zip(all_rows, all_units, counts):
        k = int(c.item())
        full[u[:k].cpu().numpy()] = r[:k].cpu().numpy()
    return full


def _several_ranks():
    import torch.distributed as dist
    return dist.is_initialized() and (dist.get_world_size() > 1 or dist.get_backend() == "nccl")   # (a 1-rank RCCL group: the stream-ordered test)


class ShardedSom:
    """Mini-batch SOM training over a row-sharded codebook.

    `shard` provides:
        winner_keys(first, count) -> int64 torch tensor [count] of this shard's packed keys
        update(it0, count, first, keys)  apply iterations [it0, it0+count) to the local rows
        collective_scope()               context in which torch collectives are ordered after
                                         winner_keys and before update (same stream, or host syncs)
    """

    def __init__(self, shard, batch, n_data):
        self.shard = shard
        self.batch = batch
        self.n_data = n_data
        self._exch = {}

    def step(self, it0, data_first, count):
        if self._exchange(count):
            # the pre-filter's bounds go round between its levels (somhip.h, somhip_shard_winner_*): three small
            # all-reduces in place of one, and every shard re-ranks only what the whole codebook's search would
            bound = self.shard.winner_begin(data_first, count)
            with self.shard.collective_scope():
                allreduce_min_floats(bound)
            self.shard.winner_refine(data_first, count)
            with self.shard.collective_scope():
                allreduce_min_floats(bound)
            keys = self.shard.winner_finish(data_first, count)
        else:
            keys = self.shard.winner_keys(data_first, count)
        if _several_ranks():                              # (one rank: no collective, so nothing to order it against -- and no host sync)
            with self.shard.collective_scope():
                allreduce_min_keys(keys, nonnegative=getattr(self.shard, "keys_nonnegative", False))
        self.shard.update(it0, count, data_first, keys)
        return keys

    def _exchange(self, count):
        """Every rank has to take the same path: the shards' answers are MIN-reduced once per batch length."""
        if not hasattr(self.shard, "exchange_available"):
            return False
        import torch
        import torch.distributed as dist
        import os
        asked = os.environ.get("SOMHIP_SHARD_EXCHANGE")
        if not dist.is_initialized() or (dist.get_world_size() == 1 and asked != "force"):
            return False                                  # one shard: its own minimum IS the whole codebook's ("force": a 1-rank test of the path)
        # the exchange saves kernel time that grows with the number of shards (tools/shard_rehearsal.py: 33 us per
        # 32768 vectors at 4 shards, 96 at 8) and costs two more small all-reduces: from 8 ranks on, or when asked for
        if dist.get_world_size() < 8 and not asked:
            return False
        if count not in self._exch:
            ok = 1 if self.shard.exchange_available(count) else 0
            if dist.is_initialized():
                dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
                t = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                ok = int(t.item())
            self._exch[count] = bool(ok)
        return self._exch[count]

    def train(self, length, start_iter=0, count=None, data_first=None):
        count = length - start_iter if count is None else count
        data_first = start_iter % self.n_data if data_first is None else data_first
        off = 0
        winners = []
        while off < count:
            it0 = start_iter + off
            c = min(self.batch - (it0 % self.batch), count - off)
            keys = self.step(it0, (data_first + off) % self.n_data, c)
            winners.append(keys.clone())
            off += c
        return winners


class GpuShard:
    """The local half on an MI355X, through the C ABI (somhip_batch_winner_keys /
    somhip_som_batch_update).  `keys` is a torch int64 tensor on the same device whose
    storage the engine writes directly."""

    keys_nonnegative = True          # somhip_batch_winner_keys never returns the all-ones key

    def __init__(self, engine, codebook, dataset, params_factory, max_batch):
        import torch
        self.e, self.cb, self.ds = engine, codebook, dataset
        self.params_factory = params_factory
        self.keys = torch.empty(max_batch, dtype=torch.int64, device=torch.device("cuda", engine.device))
        self.bound = torch.empty(max_batch, dtype=torch.float32, device=torch.device("cuda", engine.device))

    def winner_keys(self, first, count):
        import ctypes as C
        from ._lib import check
        check(self.e.lib.somhip_batch_winner_keys(self.cb.h, self.ds.h, first, count,
                                                  C.c_void_p(self.keys.data_ptr())))
        return self.keys[:count]

    # the same search with the pre-filter's bounds exchanged between the shards (include/somhip.h)
    def exchange_available(self, count):
        import os
        if os.environ.get("SOMHIP_NO_SHARD_EXCHANGE"):
            return False
        return bool(self.e.lib.somhip_shard_exchange_available(self.cb.h, self.ds.h, count))

    def winner_begin(self, first, count):
        import ctypes as C
        from ._lib import check
        check(self.e.lib.somhip_shard_winner_begin(self.cb.h, self.ds.h, first, count, C.c_void_p(self.keys.data_ptr()),
                                                   C.c_void_p(self.bound.data_ptr())))
        return self.bound[:count]

    def winner_refine(self, first, count):
        import ctypes as C
        from ._lib import check
        check(self.e.lib.somhip_shard_winner_refine(self.cb.h, self.ds.h, first, count, C.c_void_p(self.bound.data_ptr())))
        return self.bound[:count]

    def winner_finish(self, first, count):
        import ctypes as C
        from ._lib import check
        check(self.e.lib.somhip_shard_winner_finish(self.cb.h, self.ds.h, first, count, C.c_void_p(self.bound.data_ptr()),
                                                    C.c_void_p(self.keys.data_ptr())))
        return self.keys[:count]

    def update(self, it0, count, first, keys):
        import ctypes as C
        from ._lib import check
        p = self.params_factory()
        check(self.e.lib.somhip_som_batch_update(self.cb.h, self.ds.h, C.byref(p), it0, count, first,
                                                 C.c_void_p(keys.data_ptr())))

    def collective_scope(self):
        """RCCL: enqueue the collective on the engine's own HIP stream (torch.cuda.ExternalStream), so
        scan -> all-reduce -> update are stream-ordered with no host synchronisation.  gloo
        rehearsal (host-staged): plain host syncs around it."""
        import contextlib
        import torch
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_backend() == "nccl":
            if not hasattr(self, "_ext"):
                self._ext = torch.cuda.ExternalStream(self.e.stream, device=torch.device("cuda", self.e.device))
            return torch.cuda.stream(self._ext)

        @contextlib.contextmanager
        def host_synced():
            self.e.sync()
            yield
            torch.cuda.current_stream().synchronize()
        return host_synced()



# ------------------------------------------------------------------------------------------------
# lvq1 / olvq1 / lvq2 / lvq3_training (reference lvq_rout.c:498-916) over a row-sharded codebook
# ------------------------------------------------------------------------------------------------
def allgather_tensor(t, group=None):
    """[...] -> [world, ...] (a copy with one leading axis in a single process)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return t.unsqueeze(0).contiguous()
    staged = t.is_cuda and dist.get_backend(group) == "gloo"
    src = (t.cpu() if staged else t).contiguous()
    parts = [torch.empty_like(src) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, src, group=group)
    out = torch.stack(parts, dim=0)
    return out.to(t.device) if staged else out


def allreduce_sum_bits(t, group=None):
    """In-place SUM of a buffer seen as 32-bit integers: exact when, for every element, all ranks but one hold 0
    (a float sum would turn -0.0 into +0.0)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return t
    v = t.view(torch.int32)
    if v.is_cuda and dist.get_backend(group) == "gloo":
        host = v.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        v.copy_(host)
    else:
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return t


class ShardedLvq:
    """The LVQ loops over a row-sharded codebook, exact (bit-identical to the reference's online loop).

    Per batch of <= 1024 iterations (include/somhip.h, "lvq*_training over a ROW-SHARDED codebook"):
      1. every rank: the 8 nearest rows of ITS shard for every sample        -> all-gather, keep the 8 smallest
      2. every rank: labels / rates / rows of the listed candidates it owns   -> all-reduce(SUM) as integers
      3. every rank walks the batch (deterministic: all take the same decisions) and commits the rows it owns
    `shard` provides topk_keys / merge / candidates / apply (GpuLvqShard on an MI355X; a checker-backed one in
    the CPU tests).  xrows: how many of the 8 candidates' rows are exchanged (a winner beyond them ends the batch)."""

    def __init__(self, shard, kind, n_data, xrows=4, max_batch=1024, group=None):
        self.shard, self.kind, self.n_data, self.xrows, self.group = shard, kind, n_data, xrows, group
        self.max_batch = max_batch
        self.knn = 2 if kind in (3, 4) else 1
        self.batches = 0

    def batch(self, it0, data_first, count):
        keys = self.shard.merge(allgather_tensor(self.shard.topk_keys(data_first, count), self.group), count)
        lab, ta, rows = self.shard.candidates(keys, count, self.xrows)
        with self.shard.collective_scope():
            allreduce_sum_bits(lab, self.group)
            if ta is not None:
                allreduce_sum_bits(ta, self.group)
            allreduce_sum_bits(rows, self.group)
        self.batches += 1
        return self.shard.apply(it0, count, data_first, keys, lab, ta, rows, self.xrows)

    def train(self, length, start_iter=0, count=None, data_first=None):
        """iterations [start_iter, start_iter + count) of a schedule of `length`; returns (trace_index, trace_diff)"""
        count = length - start_iter if count is None else count
        data_first = start_iter % self.n_data if data_first is None else data_first
        ti, td = [], []
        off, B = 0, min(256, self.max_batch)
        while off < count:
            c = min(B, count - off)
            done, i, d = self.batch(start_iter + off, (data_first + off) % self.n_data, c)
            assert 0 < done <= c
            ti.append(i)
            td.append(d)
            off += done
            B = min(self.max_batch, 2 * B) if done == c else min(self.max_batch, max(32, done + done // 4 + 8))
        return np.concatenate(ti), np.concatenate(td)


class GpuLvqShard:
    """The local half on an MI355X, through the C ABI (somhip_batch_topk_keys, somhip_merge_topk_keys,
    somhip_lvq_batch_candidates, somhip_lvq_batch_apply).  All exchanged buffers are torch tensors on the GPU."""

    def __init__(self, engine, codebook, dataset, params_factory, kind):
        import torch
        self.e, self.cb, self.ds, self.kind = engine, codebook, dataset, kind
        self.params_factory = params_factory          # -> LvqParams (length, alpha, ... ; the per-batch fields are ignored)
        self.knn = 2 if kind in (3, 4) else 1
        self.dev = torch.device("cuda", engine.device)
        self.d4 = (codebook.dim + 3) // 4

    def topk_keys(self, first, count):
        import ctypes as C
        import torch
        from ._lib import check
        keys = torch.empty((count, 8), dtype=torch.int64, device=self.dev)
        check(self.e.lib.somhip_batch_topk_keys(self.cb.h, self.ds.h, first, count, 8, 1 if self.knn == 2 else 0,
                                                C.c_void_p(keys.data_ptr())))
        self.e.sync()
        return keys

    def merge(self, gathered, count):
        import ctypes as C
        import torch
        from ._lib import check
        gathered = gathered.contiguous()
        out = torch.empty((count, 8), dtype=torch.int64, device=self.dev)
        check(self.e.lib.somhip_merge_topk_keys(self.e.h, C.c_void_p(gathered.data_ptr()), gathered.shape[0], count, 8,
                                                C.c_void_p(out.data_ptr())))
        self.e.sync()
        return out

    def candidates(self, keys, count, xrows):
        import ctypes as C
        import torch
        from ._lib import check
        lab = torch.empty((count, 8), dtype=torch.int32, device=self.dev)
        ta = torch.empty((count, 8), dtype=torch.float32, device=self.dev) if self.kind == 2 else None
        rows = torch.empty((count, xrows, 4 * self.d4), dtype=torch.float32, device=self.dev)
        check(self.e.lib.somhip_lvq_batch_candidates(self.cb.h, count, self.kind, C.c_void_p(keys.data_ptr()), xrows,
                                                     C.c_void_p(lab.data_ptr()), C.c_void_p(ta.data_ptr()) if ta is not None else None,
                                                     C.c_void_p(rows.data_ptr())))
        self.e.sync()
        return lab, ta, rows

    def apply(self, it0, count, first, keys, lab, ta, rows, xrows):
        import ctypes as C
        from . import _lib
        from ._lib import check
        p = self.params_factory()
        done = C.c_int64(0)
        ti = np.empty(count * self.knn, dtype=np.int32)
        td = np.empty(count * self.knn, dtype=np.float32)
        check(self.e.lib.somhip_lvq_batch_apply(self.cb.h, self.ds.h, C.byref(p), it0, count, first, C.c_void_p(keys.data_ptr()),
                                                C.c_void_p(lab.data_ptr()), C.c_void_p(ta.data_ptr()) if ta is not None else None,
                                                C.c_void_p(rows.data_ptr()), xrows, C.byref(done),
                                                ti.ctypes.data_as(_lib.c_i32_p), td.ctypes.data_as(_lib.c_float_p)))
        n = done.value
        return n, ti[:n * self.knn], td[:n * self.knn]

    def collective_scope(self):
        """host-synchronous calls on both sides (each C call returns with its work complete): torch's own stream"""
        import contextlib
        import torch

        @contextlib.contextmanager
        def synced():
            yield
            torch.cuda.current_stream().synchronize()
        return synced()
