"""Python host mirror of the engine's C ABI (include/somhip.h): thin objects over ctypes,
numpy in / numpy out.  Used by the parity tests and bench.py; the command-line tools are
C (som_lvq_pak_amd/host/).  Everything here runs on the GPU through libsomhip.so -- there
is no CPU path."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import LvqParams, SomParams, check

TOPOL_HEXA, TOPOL_RECT, TOPOL_LVQ = 3, 4, 2
NEIGH_BUBBLE, NEIGH_GAUSSIAN = 1, 2
ALPHA_LINEAR, ALPHA_INVERSE_T = 1, 2
LVQ1, OLVQ1, LVQ2, LVQ3 = 1, 2, 3, 4
TIE_FIRST, TIE_KNN = 0, 1


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _arr(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


class Engine:
    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.somhip_engine_create(device, C.byref(h)))
        self.h = h
        self.device = device
        self._children = []          # weakrefs: mirrors must be destroyed before their engine

    def _adopt(self, child):
        import weakref
        self._children.append(weakref.ref(child))

    def close(self):
        if self.h:
            for ref in self._children:
                child = ref()
                if child is not None:
                    child.close()
            self._children = []
            self.lib.somhip_engine_destroy(self.h)
            self.h = None

    def sync(self):
        check(self.lib.somhip_engine_sync(self.h))

    @property
    def stream(self):
        return self.lib.somhip_engine_stream(self.h)

    def set_scan_mode(self, mode):
        """'direct', 'mfma' or 'mfma_bf16' (see include/somhip.h)"""
        check(self.lib.somhip_engine_set_scan_mode(self.h, {"direct": 0, "mfma": 1, "mfma_bf16": 2}[mode]))

    def set_update_mode(self, mode):
        """'exact' (adapt_vector's arithmetic, bit-identical to the batch oracle) or 'gemm' (matrix-pipe form)"""
        check(self.lib.somhip_engine_set_update_mode(self.h, {"exact": 0, "gemm": 1}[mode]))

    def scan_stats(self):
        out = (C.c_uint64 * 8)()
        check(self.lib.somhip_scan_stats(self.h, out))
        return {"groups": out[0], "rows": out[1], "max_groups_per_sample": out[2], "samples": out[3],
                "row_updates": out[4], "group_updates": out[5], "gemm_entries": out[6], "l2_pairs": out[7]}

    def lvq_stats(self):
        """exact batched LVQ: codebook rescans (batches) and samples so far"""
        out = (C.c_uint64 * 12)()
        check(self.lib.somhip_lvq_stats(self.h, out))
        return {"batches": out[0], "samples": out[1], "stop_list": out[2], "stop_cache": out[3],
                "phase_us": [out[4 + k] / 100.0 for k in range(4)], "components": out[8], "largest": out[9],
                "topk_pairs": out[10]}

    # --- timing table (HIP events on the engine's stream) ---
    def timing(self, on=True):
        check(self.lib.somhip_timing_enable(self.h, int(on)))

    def timing_select(self, names=None):
        """time only the named kernels (None = all): fewer HIP events on the stream"""
        mask = 0
        for i in range(self.lib.somhip_kernel_count()):
            if names is None or self.lib.somhip_kernel_name(i).decode() in names:
                mask |= 1 << i
        check(self.lib.somhip_timing_select(self.h, mask))

    def timing_reset(self):
        check(self.lib.somhip_timing_reset(self.h))

    def timing_table(self):
        out = {}
        for i in range(self.lib.somhip_kernel_count()):
            n = C.c_int64(0)
            ms = C.c_double(0)
            check(self.lib.somhip_timing_get(self.h, i, C.byref(n), C.byref(ms)))
            out[self.lib.somhip_kernel_name(i).decode()] = (n.value, ms.value)
        return out

    def device_alloc(self, nbytes):
        p = C.c_void_p()
        check(self.lib.somhip_device_alloc(self.h, nbytes, C.byref(p)))
        return p

    def device_free(self, p):
        check(self.lib.somhip_device_free(self.h, p))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_units(xdim, ydim, shard_index, shard_count, lib=None):
    """Global unit indices of the rows of interleaved shard `shard_index` of `shard_count` (somhip_shard_units)."""
    lib = lib or _lib.load()
    n = C.c_int64()
    check(lib.somhip_shard_units(xdim, ydim, shard_index, shard_count, None, C.byref(n)))
    units = np.empty(n.value, dtype=np.int64)
    check(lib.somhip_shard_units(xdim, ydim, shard_index, shard_count, _p(units, _lib.c_i64_p), C.byref(n)))
    return units


class Codebook:
    """Device mirror of the `codes` list (reference lvq_pak.h:89-113)."""

    def __init__(self, engine, rows, topol=TOPOL_LVQ, neigh=0, xdim=0, ydim=0, labels=None,
                 row_offset=0, n_global=None, interleave=None):
        """interleave=(shard_index, shard_count): `rows` are the units shard_units(...) lists, in that order
        (somhip_codebook_create_interleaved); otherwise the contiguous rows [row_offset, row_offset + n)."""
        self.e = engine
        rows = _arr(rows, np.float32)
        self.n, self.dim = rows.shape
        labels = _arr(labels, np.int32)
        self.labels = labels
        self.topol, self.neigh, self.xdim, self.ydim = topol, neigh, xdim, ydim
        self.row_offset = row_offset
        self.n_global = self.n if n_global is None else n_global
        h = C.c_void_p()
        self.interleave = interleave
        if interleave is not None:
            self.n_global = xdim * ydim
            check(engine.lib.somhip_codebook_create_interleaved(engine.h, _p(rows, _lib.c_float_p), self.n, self.dim,
                                                                topol, neigh, xdim, ydim, interleave[0],
                                                                interleave[1], C.byref(h)))
        else:
            check(engine.lib.somhip_codebook_create(engine.h, _p(rows, _lib.c_float_p), _p(labels, _lib.c_i32_p),
                                                    self.n, self.dim, topol, neigh, xdim, ydim, row_offset,
                                                    self.n_global, C.byref(h)))
        self.h = h
        engine._adopt(self)

    def download(self):
        out = np.empty((self.n, self.dim), dtype=np.float32)
        check(self.e.lib.somhip_codebook_download(self.h, _p(out, _lib.c_float_p)))
        return out

    def upload(self, rows):
        rows = _arr(rows, np.float32)
        assert rows.shape == (self.n, self.dim)
        check(self.e.lib.somhip_codebook_upload(self.h, _p(rows, _lib.c_float_p)))

    def close(self):
        if self.h and self.e.h:
            self.e.lib.somhip_codebook_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dataset:
    """Device mirror of the `data` list (one -buffer worth of rows)."""

    def __init__(self, engine, rows=None, mask=None, labels=None, weight=None, fixed_xy=None,
                 device_ptr=None, n=None, dim=None, generate=None):
        """generate=(seed, k_centres, dim, first_row, n_rows): the seeded mixture stream made in HBM
        (somhip_dataset_generate); self.centres then holds each row's mixture id."""
        self.e = engine
        h = C.c_void_p()
        if generate is not None:
            seed, k, gdim, first, gn = generate
            self.n, self.dim = gn, gdim
            self.centres = np.empty(gn, dtype=np.int32)
            check(engine.lib.somhip_dataset_generate(engine.h, seed, k, gdim, first, gn,
                                                     _p(self.centres, _lib.c_i32_p), C.byref(h)))
        elif device_ptr is not None:
            self.n, self.dim = n, dim
            check(engine.lib.somhip_dataset_wrap_device(engine.h, C.c_void_p(device_ptr), n, dim, C.byref(h)))
        else:
            rows = _arr(rows, np.float32)
            self.n, self.dim = rows.shape
            mask = _arr(mask, np.uint8)
            labels = _arr(labels, np.int32)
            weight = _arr(weight, np.int16)
            fixed_xy = _arr(fixed_xy, np.int16)
            check(engine.lib.somhip_dataset_create(engine.h, _p(rows, _lib.c_float_p), self.n, self.dim,
                                                   _p(mask, _lib.c_u8_p), _p(labels, _lib.c_i32_p),
                                                   _p(weight, _lib.c_i16_p), _p(fixed_xy, _lib.c_i16_p),
                                                   C.byref(h)))
        self.h = h
        engine._adopt(self)

    def rows(self, first, count):
        """host copy of rows [first, first + count) (somhip_dataset_download_rows)"""
        out = np.empty((count, self.dim), dtype=np.float32)
        check(self.e.lib.somhip_dataset_download_rows(self.h, first, count, _p(out, _lib.c_float_p)))
        return out

    def close(self):
        if self.h and self.e.h:
            self.e.lib.somhip_dataset_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gen_rows(seed, k_centres, dim, first_row, n_rows):
    """Host form of the seeded mixture stream (numpy restatement of pak_gen_row / k_gen_mixture): (rows, centres)."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)

    def mix(x):
        with np.errstate(over="ignore"):
            x = (x + np.uint64(0x9E3779B97F4A7C15)) & M
            x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M
            x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M
            return x ^ (x >> np.uint64(31))

    def z(sd, counter):
        total = np.zeros(counter.shape, dtype=np.int64)
        with np.errstate(over="ignore"):
            for w in range(3):
                v = mix(np.uint64(sd) ^ (np.uint64(3) * counter + np.uint64(w)))
                for sh in (0, 16, 32, 48):
                    total += ((v >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
        return ((total - 6 * 65535).astype(np.float32) / np.float32(65536.0)).astype(np.float32)

    rows = np.arange(first_row, first_row + n_rows, dtype=np.uint64)
    cen = (mix(np.uint64(seed ^ 0xB492B66FBE98F273) ^ rows) % np.uint64(k_centres)).astype(np.int64)
    cols = np.arange(dim, dtype=np.uint64)[None, :]
    mu = np.float32(4.0) * z(seed ^ 0xC3A5C85C97CB3127, cen.astype(np.uint64)[:, None] * np.uint64(dim) + cols)
    x = mu + z(seed, rows[:, None] * np.uint64(dim) + cols)
    return x.astype(np.float32), cen.astype(np.int32)


def column_minmax(ds):
    """Per-component (lo, hi, count) of the unmasked data on the device (somhip_column_minmax)."""
    lo = np.empty(ds.dim, dtype=np.float32)
    hi = np.empty(ds.dim, dtype=np.float32)
    cnt = np.empty(ds.dim, dtype=np.int64)
    check(ds.e.lib.somhip_column_minmax(ds.h, _p(lo, _lib.c_float_p), _p(hi, _lib.c_float_p), _p(cnt, _lib.c_i64_p)))
    return lo, hi, cnt


def orand_stream(seed, count):
    """The reference's LCG (lvq_pak.c:459-473: next = next * 23 % 100000001, value = next % 32767) as a numpy
    array of `count` draws after init_random(seed): state_k = seed * 23^k mod M, evaluated blockwise."""
    M = np.uint64(100000001)
    blk = 1 << 12
    pw = np.empty(blk, dtype=np.uint64)                 # 23^(i+1) mod M
    v = 1
    for i in range(blk):
        v = v * 23 % 100000001
        pw[i] = v
    nblk = (count + blk - 1) // blk
    base = np.empty(nblk, dtype=np.uint64)              # state before block j
    s = int(seed) % 100000001
    step = int(pw[-1])
    for j in range(nblk):
        base[j] = s
        s = s * step % 100000001
    st = (base[:, None] * pw[None, :]) % M
    return (st.reshape(-1)[:count] % np.uint64(32767)).astype(np.int64)


def randinit_from_bbox(lo, hi, cnt, xdim, ydim, seed):
    """randinit_codes (som_rout.c:98-150) from the data's bounding box: maximum seeded with FLT_MIN, minimum with
    FLT_MAX (:108-111), then unit by unit, component by component lo + (hi - lo) * ((float)orand() / 32768.0)
    evaluated in double and stored as float (:140-150); components without data get 0."""
    lo = np.minimum(np.asarray(lo, dtype=np.float32), np.float32(3.402823466e+38))
    hi = np.maximum(np.asarray(hi, dtype=np.float32), np.float32(1.17549435e-38))
    d = lo.shape[0]
    n = xdim * ydim
    r = orand_stream(seed, n * d).reshape(n, d).astype(np.float32).astype(np.float64) / 32768.0
    span = (hi - lo).astype(np.float32).astype(np.float64)
    out = (lo.astype(np.float64)[None, :] + span[None, :] * r).astype(np.float32)
    out[:, np.asarray(cnt) == 0] = 0.0
    return out


def find_winners(cb, ds, first=0, count=None, knn=1, tie=TIE_FIRST):
    """WINNER_FUNCTION over data rows [first, first+count): (index, diff, ret)."""
    count = ds.n if count is None else count
    idx = np.empty((count, knn), dtype=np.int32)
    diff = np.empty((count, knn), dtype=np.float32)
    ret = np.empty(count, dtype=np.int32)
    check(cb.e.lib.somhip_find_winners(cb.h, ds.h, first, count, knn, tie, _p(idx, _lib.c_i32_p),
                                       _p(diff, _lib.c_float_p), _p(ret, _lib.c_i32_p)))
    return idx, diff, ret


def som_train(cb, ds, length, alpha, radius, alpha_type=ALPHA_LINEAR, use_fixed=0, use_weights=0,
              batch=1, start_iter=0, count=None, data_first=None, trace=True):
    """som_training (reference som_rout.c:556); returns (trace_index, trace_diff)."""
    count = length - start_iter if count is None else count
    data_first = start_iter % ds.n if data_first is None else data_first
    p = SomParams(length, alpha, radius, alpha_type, use_fixed, use_weights, batch, start_iter, count,
                  data_first)
    ti = np.empty(count, dtype=np.int32) if trace else None
    td = np.empty(count, dtype=np.float32) if trace else None
    check(cb.e.lib.somhip_som_train(cb.h, ds.h, C.byref(p), _p(ti, _lib.c_i32_p), _p(td, _lib.c_float_p)))
    return ti, td


BATCH_AUTO = -1          # somhip.h SOMHIP_BATCH_AUTO: the engine's own mini-batch sizes along the schedule


def som_auto_batch(lib, length, it, alpha=0.05, radius=128.0, n_units=65536, topol=TOPOL_HEXA, neigh=NEIGH_BUBBLE,
                   alpha_type=ALPHA_LINEAR):
    """(start, length) of the SOMHIP_BATCH_AUTO batch that holds iteration `it` of a run of `length` iterations with these
    parameters on a map of n_units units (somhip_som_auto_batch: a rule in (units, radius(t), alpha(t)); (it, 1) where the
    rule does not vouch for mini-batches); defaults = configs[3]"""
    a, b = C.c_int64(0), C.c_int64(0)
    p = SomParams(length, alpha, radius, alpha_type, 0, 0, BATCH_AUTO, 0, 0, 0)
    check(lib.somhip_som_auto_batch(C.byref(p), n_units, topol, neigh, it, C.byref(a), C.byref(b)))
    return a.value, b.value


def lvq_train(cb, ds, kind, length, alpha, alpha_type=ALPHA_LINEAR, winlen=0.0, epsilon=0.0,
              talpha=None, start_iter=0, count=None, data_first=None, trace=True):
    """lvq1/olvq1/lvq2/lvq3_training (reference lvq_rout.c:498-916)."""
    count = length - start_iter if count is None else count
    data_first = start_iter % ds.n if data_first is None else data_first
    knn = 2 if kind in (LVQ2, LVQ3) else 1
    p = LvqParams(kind, length, alpha, alpha_type, winlen, epsilon, start_iter, count, data_first)
    if kind == OLVQ1:
        talpha = (np.full(cb.n, alpha, dtype=np.float32) if talpha is None
                  else np.ascontiguousarray(talpha, dtype=np.float32).copy())
    ti = np.empty(count * knn, dtype=np.int32) if trace else None
    td = np.empty(count * knn, dtype=np.float32) if trace else None
    check(cb.e.lib.somhip_lvq_train(cb.h, ds.h, C.byref(p), _p(talpha, _lib.c_float_p),
                                    _p(ti, _lib.c_i32_p), _p(td, _lib.c_float_p)))
    return talpha, ti, td


def qerror_sum(diff, ret=None):
    """find_qerror's accumulation (reference som_rout.c:698-715): a float32 running sum of
    double square roots in data order -- O(n) host work on the winners the GPU returned."""
    q = np.float32(0.0)
    d = np.asarray(diff, dtype=np.float32).reshape(-1)
    for i in range(d.shape[0]):
        if ret is not None and not ret[i]:
            continue
        q = np.float32(np.float64(q) + np.sqrt(np.float64(d[i])))
    return q


def qerror2_sum(cb, ds, radius, first=0, count=None):
    """find_qerror2 (reference som_rout.c:823-885): per-sample neighbourhood-weighted errors from
    the GPU, added in data order into a float32 accumulator like the reference's."""
    n = ds.n if count is None else count
    out = np.zeros(n, dtype=np.float32)
    ret = np.zeros(n, dtype=np.int32)
    check(cb.e.lib.somhip_qerror2(cb.h, ds.h, C.c_float(radius), first, n, _p(out, _lib.c_float_p),
                                  _p(ret, _lib.c_i32_p)))
    q = np.float32(0.0)
    for i in range(n):
        if ret[i]:
            q = np.float32(q + out[i])
    return q
