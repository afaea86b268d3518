"""ctypes bindings for the checker libraries (TEST INFRASTRUCTURE ONLY).

`Oracle`     -> oracle/liboracle.so            (our CPU restatement)
`RefHarness` -> oracle/_ref/libref_harness.so  (the real reference, flat-array driver)

Both expose the same method names so a test can run one against the other.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")

c_float_p = C.POINTER(C.c_float)
c_long_p = C.POINTER(C.c_long)
c_int_p = C.POINTER(C.c_int)
c_short_p = C.POINTER(C.c_short)
c_ubyte_p = C.POINTER(C.c_ubyte)
c_double_p = C.POINTER(C.c_double)


def build(ref=True):
    """(Re)build liboracle.so and, if /root/reference is present, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "-j4", "ref"])


def ref_available():
    return os.path.exists(os.path.join(REF_DIR, "libref_harness.so"))


def ref_tool(name):
    """Path of a reference CLI binary built into oracle/_ref (or None)."""
    p = os.path.join(REF_DIR, name)
    return p if os.path.exists(p) else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _opt(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


class _Base:
    prefix = ""

    def _fn(self, name, restype, argtypes):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        f.argtypes = argtypes
        return f


class Oracle(_Base):
    prefix = "orc_"

    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        L = self._fn
        self._som = L("som_training", C.c_int, [
            c_float_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            c_float_p, C.c_long, c_short_p, c_short_p, c_ubyte_p,
            C.c_long, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_long,
            c_long_p, c_float_p])
        self._lvq = L("lvq_training", C.c_int, [
            C.c_int, c_float_p, c_int_p, C.c_long, C.c_int, c_float_p, c_int_p, C.c_long,
            C.c_long, C.c_float, C.c_int, C.c_float, C.c_float, c_float_p, c_long_p, c_float_p])
        self._winners = L("winners", C.c_int, [
            c_float_p, C.c_long, C.c_int, c_float_p, C.c_long, c_ubyte_p, C.c_int, C.c_int,
            c_long_p, c_float_p, c_int_p])
        self._qe_from = L("qerror_from_diffs", C.c_float, [c_float_p, c_int_p, C.c_long])
        self._qe2 = L("find_qerror2", C.c_float, [
            c_float_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, C.c_long,
            c_ubyte_p, C.c_float])
        self._alpha = L("alpha", C.c_float, [C.c_int, C.c_long, C.c_long, C.c_float])
        self._radius = L("som_radius", C.c_float, [C.c_long, C.c_long, C.c_float])
        self._walpha = L("weighted_alpha", C.c_float, [C.c_float, C.c_float])
        self._gauss = L("gaussian_h", C.c_float, [C.c_float, C.c_float, C.c_float])
        self._mapdist = L("mapdist", C.c_float, [C.c_int] * 5)
        self._vdist = L("vector_dist_euc", C.c_float, [c_float_p, c_ubyte_p, c_float_p, c_ubyte_p, C.c_int])
        self._adapt = L("adapt_vector", None, [c_float_p, c_float_p, c_ubyte_p, C.c_int, C.c_float])
        self._shuffle = L("shuffle_perm", None, [C.c_long, C.c_int, c_long_p])
        self._srand = L("srand", None, [C.c_int])
        self._rand = L("rand", C.c_long, [])
        self._randinit = L("randinit", C.c_int, [c_float_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p])

    # --- training ---
    def som_train(self, codes, xdim, ydim, topol, neigh, data, length, alpha, radius,
                  alpha_type=1, weight=None, fixed_xy=None, mask=None, fixed_on=0, weights_on=0,
                  batch=1, trace=True):
        codes = _f32(codes).copy()
        data = _f32(data)
        n, d = codes.shape
        weight = _opt(weight, np.int16)
        fixed_xy = _opt(fixed_xy, np.int16)
        mask = _opt(mask, np.uint8)
        ti = np.zeros(length, dtype=np.int64) if trace else None
        td = np.zeros(length, dtype=np.float32) if trace else None
        rc = self._som(_ptr(codes, c_float_p), n, d, xdim, ydim, topol, neigh,
                       _ptr(data, c_float_p), data.shape[0], _ptr(weight, c_short_p),
                       _ptr(fixed_xy, c_short_p), _ptr(mask, c_ubyte_p), length, alpha, radius,
                       alpha_type, fixed_on, weights_on, batch, _ptr(ti, c_long_p), _ptr(td, c_float_p))
        assert rc == 0
        return codes, ti, td

    def lvq_train(self, kind, codes, clabels, data, dlabels, length, alpha, alpha_type=1,
                  winlen=0.0, epsilon=0.0, talpha=None, trace=True):
        codes = _f32(codes).copy()
        data = _f32(data)
        n, d = codes.shape
        clabels = np.ascontiguousarray(clabels, dtype=np.int32)
        dlabels = np.ascontiguousarray(dlabels, dtype=np.int32)
        knn = 2 if kind in (3, 4) else 1
        if kind == 2:
            talpha = (np.full(n, alpha, dtype=np.float32) if talpha is None
                      else _f32(talpha).copy())
        ti = np.zeros(length * knn, dtype=np.int64) if trace else None
        td = np.zeros(length * knn, dtype=np.float32) if trace else None
        rc = self._lvq(kind, _ptr(codes, c_float_p), _ptr(clabels, c_int_p), n, d,
                       _ptr(data, c_float_p), _ptr(dlabels, c_int_p), data.shape[0], length,
                       alpha, alpha_type, winlen, epsilon, _ptr(talpha, c_float_p),
                       _ptr(ti, c_long_p), _ptr(td, c_float_p))
        if rc == 2:
            raise FloatingPointError("no winner: the codes diverged (the reference would dereference NULL here)")
        assert rc == 0
        return codes, talpha, ti, td

    # --- scans ---
    def winners(self, codes, data, knn=1, use_knn_fn=False, mask=None):
        codes = _f32(codes)
        data = _f32(data)
        n, d = codes.shape
        m = data.shape[0]
        mask = _opt(mask, np.uint8)
        idx = np.zeros((m, knn), dtype=np.int64)
        diff = np.zeros((m, knn), dtype=np.float32)
        ret = np.zeros(m, dtype=np.int32)
        self._winners(_ptr(codes, c_float_p), n, d, _ptr(data, c_float_p), m, _ptr(mask, c_ubyte_p),
                      knn, int(use_knn_fn), _ptr(idx, c_long_p), _ptr(diff, c_float_p),
                      _ptr(ret, c_int_p))
        return idx, diff, ret

    def qerror_from_diffs(self, diff, ret=None):
        diff = _f32(diff).reshape(-1)
        ret = _opt(ret, np.int32)
        return float(self._qe_from(_ptr(diff, c_float_p), _ptr(ret, c_int_p), diff.shape[0]))

    def find_qerror(self, codes, data, mask=None):
        idx, diff, ret = self.winners(codes, data, 1, False, mask)
        return self.qerror_from_diffs(diff[:, 0], ret), idx[:, 0], diff[:, 0]

    def find_qerror2(self, codes, xdim, topol, neigh, data, radius, mask=None):
        codes = _f32(codes)
        data = _f32(data)
        mask = _opt(mask, np.uint8)
        return float(self._qe2(_ptr(codes, c_float_p), codes.shape[0], codes.shape[1], xdim, topol,
                               neigh, _ptr(data, c_float_p), data.shape[0], _ptr(mask, c_ubyte_p),
                               radius))

    # --- scalars ---
    def alpha(self, typ, it, length, alpha):
        return float(self._alpha(typ, it, length, alpha))

    def som_radius(self, it, length, radius):
        return float(self._radius(it, length, radius))

    def weighted_alpha(self, talp, weight):
        return float(self._walpha(talp, weight))

    def gaussian_h(self, dd, radius, alpha):
        return float(self._gauss(dd, radius, alpha))

    def mapdist(self, topol, bx, by, tx, ty):
        return float(self._mapdist(topol, bx, by, tx, ty))

    def vector_dist(self, a, b, ma=None, mb=None):
        a = _f32(a); b = _f32(b)
        ma = _opt(ma, np.uint8); mb = _opt(mb, np.uint8)
        return float(self._vdist(_ptr(a, c_float_p), _ptr(ma, c_ubyte_p), _ptr(b, c_float_p),
                                 _ptr(mb, c_ubyte_p), a.shape[0]))

    def adapt_vector(self, c, x, alpha, mask=None):
        c = _f32(c).copy(); x = _f32(x)
        mask = _opt(mask, np.uint8)
        self._adapt(_ptr(c, c_float_p), _ptr(x, c_float_p), _ptr(mask, c_ubyte_p), c.shape[0], alpha)
        return c

    def shuffle_perm(self, n, seed):
        perm = np.zeros(n, dtype=np.int64)
        self._shuffle(n, seed, _ptr(perm, c_long_p))
        return perm

    def rand_seq(self, seed, count):
        self._srand(seed)
        return np.array([self._rand() for _ in range(count)], dtype=np.int64)

    def randinit(self, data, xdim, ydim, seed):
        data = _f32(data)
        out = np.zeros((xdim * ydim, data.shape[1]), dtype=np.float32)
        rc = self._randinit(_ptr(data, c_float_p), data.shape[0], data.shape[1], xdim, ydim, seed,
                            _ptr(out, c_float_p))
        assert rc == 0
        return out


class RefHarness(_Base):
    """The real reference (oracle/_ref/libref_harness.so).  Same call shapes as Oracle."""
    prefix = "ref_"

    def __init__(self):
        path = os.path.join(REF_DIR, "libref_harness.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        L = self._fn
        self._som = L("som_train", C.c_int, [
            c_float_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
            c_float_p, C.c_long, c_short_p, c_short_p, c_ubyte_p,
            C.c_long, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
            c_long_p, c_float_p, c_double_p])
        self._qe = L("find_qerror", C.c_float, [
            c_float_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_float_p, C.c_long,
            c_ubyte_p, C.c_int, C.c_float, c_long_p, c_float_p])
        self._lvq = L("lvq_train", C.c_int, [
            C.c_int, c_float_p, c_int_p, C.c_long, C.c_int, c_float_p, c_int_p, C.c_long,
            C.c_long, C.c_float, C.c_int, C.c_float, C.c_float, C.c_char_p, C.c_char_p,
            c_long_p, c_float_p, c_double_p])
        self._winners = L("winners", C.c_int, [
            c_float_p, C.c_long, C.c_int, c_float_p, C.c_long, c_ubyte_p, C.c_int, C.c_int,
            c_long_p, c_float_p, c_int_p])
        self._alpha = L("alpha", C.c_float, [C.c_int, C.c_long, C.c_long, C.c_float])
        self._mapdist = L("mapdist", C.c_float, [C.c_int] * 5)
        self._vdist = L("vector_dist", C.c_float, [c_float_p, c_ubyte_p, c_float_p, c_ubyte_p, C.c_int])
        self._adapt = L("adapt_vector", None, [c_float_p, c_float_p, c_ubyte_p, C.c_int, C.c_float])
        self._shuffle = L("shuffle_perm", C.c_int, [C.c_long, C.c_int, c_long_p])
        self._randseq = L("rand_seq", None, [C.c_int, C.c_long, c_long_p])
        self._randinit = L("randinit", C.c_int, [c_float_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_int, c_float_p])
        self.last_seconds = 0.0

    def som_train(self, codes, xdim, ydim, topol, neigh, data, length, alpha, radius,
                  alpha_type=1, weight=None, fixed_xy=None, mask=None, fixed_on=0, weights_on=0,
                  batch=1, trace=True):
        assert batch == 1, "the reference is strictly online"
        codes = _f32(codes).copy()
        data = _f32(data)
        n, d = codes.shape
        weight = _opt(weight, np.int16)
        fixed_xy = _opt(fixed_xy, np.int16)
        mask = _opt(mask, np.uint8)
        ti = np.zeros(length, dtype=np.int64) if trace else None
        td = np.zeros(length, dtype=np.float32) if trace else None
        secs = C.c_double(0)
        rc = self._som(_ptr(codes, c_float_p), n, d, xdim, ydim, topol, neigh,
                       _ptr(data, c_float_p), data.shape[0], _ptr(weight, c_short_p),
                       _ptr(fixed_xy, c_short_p), _ptr(mask, c_ubyte_p), length, alpha, radius,
                       alpha_type, fixed_on, weights_on, _ptr(ti, c_long_p), _ptr(td, c_float_p),
                       C.byref(secs))
        assert rc == 0
        self.last_seconds = secs.value
        if trace and fixed_on and fixed_xy is not None:
            # the reference skips the winner call for fixed-point samples, so its trace
            # has fewer entries; re-expand to one entry per iteration (-3 = fixed)
            out_i = np.full(length, -3, dtype=np.int64)
            out_d = np.full(length, -1.0, dtype=np.float32)
            pos = 0
            nd = data.shape[0]
            for le in range(length):
                if fixed_xy[le % nd, 0] < 0:
                    out_i[le] = ti[pos]; out_d[le] = td[pos]; pos += 1
            ti, td = out_i, out_d
        return codes, ti, td

    def lvq_train(self, kind, codes, clabels, data, dlabels, length, alpha, alpha_type=1,
                  winlen=0.0, epsilon=0.0, talpha=None, trace=True):
        assert talpha is None, "the reference takes OLVQ1 rates from -alpha or a .lra file"
        codes = _f32(codes).copy()
        data = _f32(data)
        n, d = codes.shape
        clabels = np.ascontiguousarray(clabels, dtype=np.int32)
        dlabels = np.ascontiguousarray(dlabels, dtype=np.int32)
        knn = 2 if kind in (3, 4) else 1
        ti = np.zeros(length * knn, dtype=np.int64) if trace else None
        td = np.zeros(length * knn, dtype=np.float32) if trace else None
        secs = C.c_double(0)
        tal = None
        with tempfile.TemporaryDirectory(prefix="orclra") as tmp:
            assert "." not in tmp
            lin = os.path.join(tmp, "in.cod").encode()
            lout = os.path.join(tmp, "out.cod").encode()
            rc = self._lvq(kind, _ptr(codes, c_float_p), _ptr(clabels, c_int_p), n, d,
                           _ptr(data, c_float_p), _ptr(dlabels, c_int_p), data.shape[0], length,
                           alpha, alpha_type, winlen, epsilon, lin, lout,
                           _ptr(ti, c_long_p), _ptr(td, c_float_p), C.byref(secs))
            assert rc == 0
            lra = os.path.join(tmp, "out.lra")
            if kind == 2 and os.path.exists(lra):
                tal = open(lra).read().split()      # "%g" strings, datafile.c:1081
        self.last_seconds = secs.value
        return codes, tal, ti, td

    def winners(self, codes, data, knn=1, use_knn_fn=False, mask=None):
        codes = _f32(codes)
        data = _f32(data)
        n, d = codes.shape
        m = data.shape[0]
        mask = _opt(mask, np.uint8)
        idx = np.zeros((m, knn), dtype=np.int64)
        diff = np.zeros((m, knn), dtype=np.float32)
        ret = np.zeros(m, dtype=np.int32)
        self._winners(_ptr(codes, c_float_p), n, d, _ptr(data, c_float_p), m, _ptr(mask, c_ubyte_p),
                      knn, int(use_knn_fn), _ptr(idx, c_long_p), _ptr(diff, c_float_p),
                      _ptr(ret, c_int_p))
        return idx, diff, ret

    def find_qerror(self, codes, data, mask=None, xdim=1, ydim=1, topol=3, neigh=1):
        codes = _f32(codes)
        data = _f32(data)
        mask = _opt(mask, np.uint8)
        m = data.shape[0]
        ti = np.zeros(m, dtype=np.int64)
        td = np.zeros(m, dtype=np.float32)
        q = self._qe(_ptr(codes, c_float_p), codes.shape[0], codes.shape[1], xdim, ydim, topol, neigh,
                     _ptr(data, c_float_p), m, _ptr(mask, c_ubyte_p), 0, 0.0,
                     _ptr(ti, c_long_p), _ptr(td, c_float_p))
        return float(q), ti, td

    def find_qerror2(self, codes, xdim, topol, neigh, data, radius, mask=None):
        codes = _f32(codes)
        data = _f32(data)
        mask = _opt(mask, np.uint8)
        ydim = codes.shape[0] // xdim
        return float(self._qe(_ptr(codes, c_float_p), codes.shape[0], codes.shape[1], xdim, ydim,
                              topol, neigh, _ptr(data, c_float_p), data.shape[0],
                              _ptr(mask, c_ubyte_p), 1, radius, None, None))

    def alpha(self, typ, it, length, alpha):
        return float(self._alpha(typ, it, length, alpha))

    def mapdist(self, topol, bx, by, tx, ty):
        return float(self._mapdist(topol, bx, by, tx, ty))

    def vector_dist(self, a, b, ma=None, mb=None):
        a = _f32(a); b = _f32(b)
        ma = _opt(ma, np.uint8); mb = _opt(mb, np.uint8)
        return float(self._vdist(_ptr(a, c_float_p), _ptr(ma, c_ubyte_p), _ptr(b, c_float_p),
                                 _ptr(mb, c_ubyte_p), a.shape[0]))

    def adapt_vector(self, c, x, alpha, mask=None):
        c = _f32(c).copy(); x = _f32(x)
        mask = _opt(mask, np.uint8)
        self._adapt(_ptr(c, c_float_p), _ptr(x, c_float_p), _ptr(mask, c_ubyte_p), c.shape[0], alpha)
        return c

    def shuffle_perm(self, n, seed):
        perm = np.zeros(n, dtype=np.int64)
        self._shuffle(n, seed, _ptr(perm, c_long_p))
        return perm

    def rand_seq(self, seed, count):
        out = np.zeros(count, dtype=np.int64)
        self._randseq(seed, count, _ptr(out, c_long_p))
        return out

    def randinit(self, data, xdim, ydim, seed, topol=3, neigh=1):
        data = _f32(data)
        out = np.zeros((xdim * ydim, data.shape[1]), dtype=np.float32)
        rc = self._randinit(_ptr(data, c_float_p), data.shape[0], data.shape[1], topol, neigh,
                            xdim, ydim, seed, _ptr(out, c_float_p))
        assert rc == 0
        return out
