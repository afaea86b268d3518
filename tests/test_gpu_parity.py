"""Parity of the HIP path (through the C ABI, libsomhip.so) with the oracle and with the
fixtures the real reference produced.  Bit-exact: indices equal, fp32 values compared as
bit patterns.  Needs an MI355X:  pytest -m gpu."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import load_trace, read_cod, synth

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def eng():
    from som_lvq_pak_amd import engine as E
    e = E.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def E():
    from som_lvq_pak_amd import engine
    return engine


# --------------------------------------------------------------------------- layout
@pytest.mark.parametrize("n,d", [(1, 1), (63, 3), (64, 4), (65, 5), (200, 20), (1000, 130)])
def test_codebook_roundtrip(eng, E, n, d):
    rs = np.random.RandomState(n * 7 + d)
    rows = rs.standard_normal((n, d)).astype(np.float32)
    cb = E.Codebook(eng, rows)
    assert np.array_equal(bits(cb.download()), bits(rows))
    rows2 = rows[::-1].copy()
    cb.upload(rows2)
    assert np.array_equal(bits(cb.download()), bits(rows2))
    cb.close()


# --------------------------------------------------------------------------- winner scans
@pytest.mark.parametrize("n,d,m", [(5, 3, 7), (96, 5, 333), (200, 20, 1962), (1030, 130, 77), (4096, 64, 100)])
def test_find_winner_euc_random(eng, E, oracle, n, d, m):
    x, _ = synth(n + d, m, d)
    rs = np.random.RandomState(n)
    codes = (x[rs.randint(0, m, n)] + 0.25 * rs.standard_normal((n, d))).astype(np.float32)
    cb, ds = E.Codebook(eng, codes), E.Dataset(eng, x)
    gi, gd, gr = E.find_winners(cb, ds)
    oi, od, _ = oracle.winners(codes, x)
    assert np.array_equal(gi, oi)
    assert np.array_equal(bits(gd), bits(od))
    assert (gr == 1).all()
    # sub-range with wrap-around start
    gi2, gd2, _ = E.find_winners(cb, ds, first=m - 3, count=10)
    want = np.concatenate([oi[m - 3:], oi[:7]])
    assert np.array_equal(gi2, want)


@pytest.mark.parametrize("knn", [1, 2, 5])
def test_knn_ties_golden(eng, E, exdata, knn):
    """duplicated rows force exact ties; expected values come from the real reference."""
    ci, e2 = exdata["lvq_init"], exdata["ex2"]
    cbrows = np.concatenate([ci.points[:40], ci.points[:40]], axis=0)
    cb, ds = E.Codebook(eng, cbrows), E.Dataset(eng, e2.points[:300])
    g = load_trace("knn_ties_%d" % knn)
    gi, gd, gr = E.find_winners(cb, ds, knn=knn, tie=E.TIE_KNN)
    assert np.array_equal(gi, g["index"])
    assert np.array_equal(bits(gd), bits(g["diff"]))
    g = load_trace("euc_ties")
    gi, gd, _ = E.find_winners(cb, ds, knn=1, tie=E.TIE_FIRST)
    assert np.array_equal(gi, g["index"]) and np.array_equal(bits(gd), bits(g["diff"]))


@pytest.mark.parametrize("knn", [2, 3, 8])
def test_knn_random(eng, E, oracle, knn):
    x, _ = synth(77, 500, 24)
    codes = x[:300:2] + 0.5
    cb, ds = E.Codebook(eng, codes), E.Dataset(eng, x)
    gi, gd, _ = E.find_winners(cb, ds, knn=knn, tie=E.TIE_KNN)
    oi, od, _ = oracle.winners(codes, x, knn=knn, use_knn_fn=True)
    assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od))


def test_masked_winners(eng, E, oracle):
    rs = np.random.RandomState(5)
    x, _ = synth(6, 120, 9)
    codes = x[:25] + 0.1
    mask = (rs.rand(120, 9) < 0.3).astype(np.uint8)
    mask[3] = 1
    cb, ds = E.Codebook(eng, codes), E.Dataset(eng, x, mask=mask)
    gi, gd, gr = E.find_winners(cb, ds)
    oi, od, orr = oracle.winners(codes, x, mask=mask)
    assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od)) and np.array_equal(gr, orr)
    assert gr[3] == 0 and gi[3, 0] == -2


def test_qerror_golden(eng, E, exdata):
    """qerror = find_qerror (som_rout.c:678): GPU winners + the reference's fp32 running sum."""
    g = load_trace("som_ex_hexa_bubble_linear")
    cb, ds = E.Codebook(eng, g["codes"], E.TOPOL_HEXA, E.NEIGH_BUBBLE, 12, 8), E.Dataset(eng, exdata["ex"].points)
    gi, gd, gr = E.find_winners(cb, ds)
    assert np.array_equal(gi[:, 0], g["q_index"])
    assert np.array_equal(bits(gd[:, 0]), bits(g["q_diff"]))
    assert E.qerror_sum(gd, gr) == g["qerror_sum"]
    fin = read_cod("somexample.cod")
    cb2 = E.Codebook(eng, fin.points, E.TOPOL_HEXA, E.NEIGH_BUBBLE, 12, 8)
    _, gd2, gr2 = E.find_winners(cb2, ds)
    assert "%f" % float(E.qerror_sum(gd2, gr2) / np.float32(3840)) == "3.571006"


def test_qerror2_golden(eng, E, exdata):
    """qerror -qetype 1 = find_qerror2 (som_rout.c:823): values the real reference produced."""
    g = load_trace("som_qerror2")
    ds = E.Dataset(eng, exdata["ex"].points)
    hb, hg = read_cod("som_hexa_bubble.cod"), read_cod("som_hexa_gaussian.cod")
    cb = E.Codebook(eng, hb.points, E.TOPOL_HEXA, E.NEIGH_BUBBLE, 12, 8)
    assert E.qerror2_sum(cb, ds, 2.0) == g["bubble_r2"]
    cg = E.Codebook(eng, hg.points, E.TOPOL_HEXA, E.NEIGH_GAUSSIAN, 12, 8)
    assert E.qerror2_sum(cg, ds, 2.0) == g["gaussian_r2"]


@pytest.mark.parametrize("xdim,ydim,d,topol,neigh,radius", [
    (7, 5, 6, 3, 1, 2.5), (7, 5, 6, 4, 2, 1.7), (16, 8, 9, 3, 1, 3.0),      # 16x8: 8x8-patch storage order
    (24, 16, 33, 3, 2, 4.2), (40, 24, 5, 4, 1, 6.0), (9, 30, 4, 3, 1, 0.0), (5, 4, 3, 4, 1, 100.0)])
def test_qerror2_random_vs_oracle(eng, E, oracle, xdim, ydim, d, topol, neigh, radius):
    """bubble and gaussian, both lattices, masked samples, radius 0 and radius > map: the
    per-sample sums are fp32 sums in unit order, so the totals must agree bit for bit."""
    x, _ = synth(70 + xdim, 150, d)
    rs = np.random.RandomState(xdim * ydim)
    codes = (x[rs.randint(0, 150, xdim * ydim)] + rs.standard_normal((xdim * ydim, d))).astype(np.float32)
    mask = (rs.random_sample(x.shape) < 0.15).astype(np.uint8)
    mask[5] = 1                                               # one sample entirely masked: skipped
    want = np.float32(oracle.find_qerror2(codes, xdim, topol, neigh, x, radius, mask=mask))
    cb = E.Codebook(eng, codes, topol, neigh, xdim, ydim)
    ds = E.Dataset(eng, x, mask=mask)
    assert bits(E.qerror2_sum(cb, ds, radius)) == bits(want)
    want2 = np.float32(oracle.find_qerror2(codes, xdim, topol, neigh, x, radius))
    assert bits(E.qerror2_sum(cb, E.Dataset(eng, x), radius)) == bits(want2)


# --------------------------------------------------------------------------- som_training, online
TOPOL = {"hexa": 3, "rect": 4}
NEIGH = {"bubble": 1, "gaussian": 2}


@pytest.mark.parametrize("topol", ["hexa", "rect"])
@pytest.mark.parametrize("neigh", ["bubble", "gaussian"])
def test_som_online_golden(eng, E, exdata, topol, neigh):
    g = load_trace("som_ex_%s_%s_linear" % (topol, neigh))
    ini = read_cod("som_init_%s_%s.cod" % (topol, neigh))
    cb = E.Codebook(eng, ini.points, TOPOL[topol], NEIGH[neigh], 12, 8)
    ds = E.Dataset(eng, exdata["ex"].points)
    ti, td = E.som_train(cb, ds, 5000, 0.05, 10.0)
    assert np.array_equal(ti, g["trace_index"])
    assert np.array_equal(bits(td), bits(g["trace_diff"]))
    assert np.array_equal(bits(cb.download()), bits(g["codes"]))


def test_som_online_inverse_t_and_segments(eng, E, exdata):
    g = load_trace("som_ex_hexa_bubble_inverse_t")
    ini = read_cod("som_init_hexa_bubble.cod")
    cb = E.Codebook(eng, ini.points, 3, 1, 12, 8)
    ds = E.Dataset(eng, exdata["ex"].points)
    # the same schedule run in three pieces (what -snapinterval / -buffer need)
    t1, _ = E.som_train(cb, ds, 5000, 0.05, 10.0, alpha_type=2, start_iter=0, count=1234)
    t2, _ = E.som_train(cb, ds, 5000, 0.05, 10.0, alpha_type=2, start_iter=1234, count=2766)
    t3, _ = E.som_train(cb, ds, 5000, 0.05, 10.0, alpha_type=2, start_iter=4000, count=1000)
    assert np.array_equal(np.concatenate([t1, t2, t3]), g["trace_index"])
    assert np.array_equal(bits(cb.download()), bits(g["codes"]))


@pytest.mark.parametrize("neigh", [1, 2])
def test_som_online_masks_weights_fixed(eng, E, neigh):
    g = load_trace("som_masked_%d" % neigh)
    cb = E.Codebook(eng, g["init"], 3, neigh, 7, 5)
    ds = E.Dataset(eng, g["x"], mask=g["mask"], weight=g["weight"], fixed_xy=g["fixed"])
    ti, td = E.som_train(cb, ds, 1500, 0.08, 4.0, use_fixed=1, use_weights=1)
    assert np.array_equal(ti, g["trace_index"])
    assert np.array_equal(bits(td), bits(g["trace_diff"]))
    assert np.array_equal(bits(cb.download()), bits(g["codes"]))


@pytest.mark.parametrize("topol,neigh", [(3, 1), (4, 2)])
def test_som_online_synthetic_midsize(eng, E, oracle, topol, neigh):
    g = load_trace("som_synth_%d_%d" % (topol, neigh))
    x, _ = synth(21, 4000, 48, k=8)
    ini = oracle.randinit(x, 24, 16, 5)
    cb, ds = E.Codebook(eng, ini, topol, neigh, 24, 16), E.Dataset(eng, x)
    ti, td = E.som_train(cb, ds, 6000, 0.05, 8.0)
    assert np.array_equal(ti, g["trace_index"])
    assert hashlib.sha256(td.tobytes()).hexdigest() == str(g["trace_diff_sha"])
    assert hashlib.sha256(cb.download().tobytes()).hexdigest() == str(g["codes_sha"])


# --------------------------------------------------------------------------- som_training, mini-batch
@pytest.mark.parametrize("batch", [2, 16, 100, 4096])
@pytest.mark.parametrize("topol,neigh", [(3, 1), (4, 2)])
def test_som_minibatch_vs_batch_oracle(eng, E, oracle, batch, topol, neigh):
    x, _ = synth(31, 700, 20)
    ini = oracle.randinit(x, 10, 9, 4)
    length = 1500                                  # wraps the data set twice
    oc, oi, od = oracle.som_train(ini, 10, 9, topol, neigh, x, length, 0.07, 5.0, batch=batch)
    cb, ds = E.Codebook(eng, ini, topol, neigh, 10, 9), E.Dataset(eng, x)
    ti, td = E.som_train(cb, ds, length, 0.07, 5.0, batch=batch)
    assert np.array_equal(ti, oi)
    assert np.array_equal(bits(td), bits(od))
    assert np.array_equal(bits(cb.download()), bits(oc))


def test_som_minibatch_masks_weights_fixed(eng, E, oracle):
    g = load_trace("som_masked_1")
    oc, oi, od = oracle.som_train(g["init"], 7, 5, 3, 1, g["x"], 1500, 0.08, 4.0, weight=g["weight"],
                                  fixed_xy=g["fixed"], mask=g["mask"], fixed_on=1, weights_on=1, batch=32)
    cb = E.Codebook(eng, g["init"], 3, 1, 7, 5)
    ds = E.Dataset(eng, g["x"], mask=g["mask"], weight=g["weight"], fixed_xy=g["fixed"])
    ti, td = E.som_train(cb, ds, 1500, 0.08, 4.0, use_fixed=1, use_weights=1, batch=32)
    assert np.array_equal(ti, oi) and np.array_equal(bits(td), bits(od))
    assert np.array_equal(bits(cb.download()), bits(oc))


def test_som_row_sharded_two_phase(eng, E, oracle):
    """The multi-GPU decomposition on one GPU: two row shards, element-wise MIN of their
    packed keys (what the RCCL all-reduce does), each shard updating only its own rows --
    must equal the unsharded mini-batch run bit for bit."""
    from som_lvq_pak_amd._lib import SomParams
    x, _ = synth(41, 900, 16)
    xdim, ydim, B, length = 16, 12, 64, 1280
    ini = oracle.randinit(x, xdim, ydim, 8)
    oc, oi, _ = oracle.som_train(ini, xdim, ydim, 3, 1, x, length, 0.05, 6.0, batch=B)
    n = xdim * ydim
    cut = 5 * xdim + 3                                       # deliberately not on a map-row edge
    shards = [E.Codebook(eng, ini[:cut], 3, 1, xdim, ydim, row_offset=0, n_global=n),
              E.Codebook(eng, ini[cut:], 3, 1, xdim, ydim, row_offset=cut, n_global=n)]
    ds = E.Dataset(eng, x)
    lib = eng.lib
    kb = [eng.device_alloc(8 * B) for _ in range(3)]
    hk = [np.empty(B, dtype=np.uint64) for _ in range(2)]
    p = SomParams(length, 0.05, 6.0, 1, 0, 0, B, 0, length, 0)
    got_idx = []
    for it0 in range(0, length, B):
        first = it0 % ds.n
        for s in range(2):
            assert lib.somhip_batch_winner_keys(shards[s].h, ds.h, first, B, kb[s]) == 0
            assert lib.somhip_copy_to_host(eng.h, hk[s].ctypes.data_as(C.c_void_p), kb[s], 8 * B) == 0
        merged = np.minimum(hk[0], hk[1])
        got_idx.append((merged & np.uint64(0xFFFFFFFF)).astype(np.int64))
        assert lib.somhip_copy_to_device(eng.h, kb[2], merged.ctypes.data_as(C.c_void_p), 8 * B) == 0
        for s in range(2):
            assert lib.somhip_som_batch_update(shards[s].h, ds.h, C.byref(p), it0, B, first, kb[2]) == 0
    eng.sync()
    assert np.array_equal(np.concatenate(got_idx), oi)
    got = np.concatenate([shards[0].download(), shards[1].download()], axis=0)
    assert np.array_equal(bits(got), bits(oc))
    for k in kb:
        eng.device_free(k)


@pytest.mark.parametrize("shape", [(16, 16, 3, 3, 1), (24, 16, 4, 4, 2), (32, 8, 2, 3, 1)])
def test_som_interleaved_shards_two_phase(eng, E, oracle, shape):
    """Interleaved shards (somhip_codebook_create_interleaved: 8x8 patches dealt round-robin to the ranks,
    uneven counts included): per-shard winners, element-wise MIN, per-shard updates -- bit-equal to the
    unsharded mini-batch run, hexa/rect and bubble/gaussian."""
    from som_lvq_pak_amd._lib import SomParams
    xdim, ydim, S, topol, neigh = shape
    x, _ = synth(43 + S, 700, 12)
    B, length = 96, 1152
    ini = oracle.randinit(x, xdim, ydim, 5)
    oc, oi, _ = oracle.som_train(ini, xdim, ydim, topol, neigh, x, length, 0.05, 7.0, batch=B)
    units = [E.shard_units(xdim, ydim, r, S) for r in range(S)]
    assert sorted(np.concatenate(units).tolist()) == list(range(xdim * ydim))
    shards = [E.Codebook(eng, ini[units[r]], topol, neigh, xdim, ydim, interleave=(r, S)) for r in range(S)]
    for r in range(S):                                        # layout round trip
        assert np.array_equal(bits(shards[r].download()), bits(ini[units[r]]))
    ds = E.Dataset(eng, x)
    lib = eng.lib
    kb = [eng.device_alloc(8 * B) for _ in range(S + 1)]
    hk = [np.empty(B, dtype=np.uint64) for _ in range(S)]
    p = SomParams(length, 0.05, 7.0, 1, 0, 0, B, 0, length, 0)
    got_idx = []
    for it0 in range(0, length, B):
        first = it0 % ds.n
        for s in range(S):
            assert lib.somhip_batch_winner_keys(shards[s].h, ds.h, first, B, kb[s]) == 0
            assert lib.somhip_copy_to_host(eng.h, hk[s].ctypes.data_as(C.c_void_p), kb[s], 8 * B) == 0
        merged = hk[0]
        for s in range(1, S):
            merged = np.minimum(merged, hk[s])
        merged = np.ascontiguousarray(merged)
        got_idx.append((merged & np.uint64(0xFFFFFFFF)).astype(np.int64))
        assert lib.somhip_copy_to_device(eng.h, kb[S], merged.ctypes.data_as(C.c_void_p), 8 * B) == 0
        for s in range(S):
            assert lib.somhip_som_batch_update(shards[s].h, ds.h, C.byref(p), it0, B, first, kb[S]) == 0
    eng.sync()
    assert np.array_equal(np.concatenate(got_idx), oi)
    got = np.empty_like(oc)
    for r in range(S):
        got[units[r]] = shards[r].download()
    assert np.array_equal(bits(got), bits(oc))
    for k in kb:
        eng.device_free(k)


def _exchange_search(E, engs, shards, dss, first, count):
    """begin -> MIN -> refine -> MIN -> finish -> MIN over shards that each live on their own engine (the three phases
    of a search own their engine's scratch); returns the merged keys and every shard's own keys."""
    S = len(shards)
    kb = [engs[s].device_alloc(8 * count) for s in range(S)]
    bb = [engs[s].device_alloc(4 * count) for s in range(S)]
    hb = [np.empty(count, dtype=np.float32) for _ in range(S)]
    hk = [np.empty(count, dtype=np.uint64) for _ in range(S)]

    def exchange_bounds():
        for s in range(S):
            assert engs[s].lib.somhip_copy_to_host(engs[s].h, hb[s].ctypes.data_as(C.c_void_p), bb[s], 4 * count) == 0
        m = np.ascontiguousarray(np.minimum.reduce(hb))
        for s in range(S):
            assert engs[s].lib.somhip_copy_to_device(engs[s].h, bb[s], m.ctypes.data_as(C.c_void_p), 4 * count) == 0
        return m

    for s in range(S):
        assert engs[s].lib.somhip_shard_exchange_available(shards[s].h, dss[s].h, count) == 1
        assert engs[s].lib.somhip_shard_winner_begin(shards[s].h, dss[s].h, first, count, kb[s], bb[s]) == 0
    b1 = exchange_bounds()
    for s in range(S):
        assert engs[s].lib.somhip_shard_winner_refine(shards[s].h, dss[s].h, first, count, bb[s]) == 0
    b2 = exchange_bounds()
    assert (b2 <= b1).all()                                   # the three-product bound is the tighter one
    for s in range(S):
        assert engs[s].lib.somhip_shard_winner_finish(shards[s].h, dss[s].h, first, count, bb[s], kb[s]) == 0
        assert engs[s].lib.somhip_copy_to_host(engs[s].h, hk[s].ctypes.data_as(C.c_void_p), kb[s], 8 * count) == 0
    for s in range(S):
        engs[s].device_free(kb[s])
        engs[s].device_free(bb[s])
    return np.ascontiguousarray(np.minimum.reduce(hk)), hk, b2


def test_shard_winner_search_with_exchanged_bounds(E, oracle):
    """somhip_shard_winner_begin/refine/finish over three uneven row shards, one of them nowhere near the samples (it
    keeps no row group at all and has to abstain), exact ties across shards: after the MIN the keys are those of
    find_winner_euc over the whole codebook, and the far shard re-ranks nothing."""
    rs = np.random.RandomState(11)
    d, m = 64, 480
    x = rs.standard_normal((m, d)).astype(np.float32)
    codes = (x[rs.randint(0, m, 1500)] + 0.5 * rs.standard_normal((1500, d))).astype(np.float32)
    codes[600:1100] += 40.0
    codes[1300] = codes[100] = x[7]                           # the same row in two shards: the lower index wins
    codes[1250] = x[9]
    want_i, want_d, _ = oracle.winners(codes, x)
    want_i, want_d = np.asarray(want_i).reshape(-1), np.asarray(want_d).reshape(-1)
    cuts = [0, 600, 1100, 1500]
    engs = [E.Engine(0) for _ in cuts[1:]]
    try:
        dss = [E.Dataset(e, x) for e in engs]
        shards = [E.Codebook(e, codes[a:b], row_offset=a, n_global=1500) for e, a, b in zip(engs, cuts, cuts[1:])]
        for first, count in ((0, m), (m - 100, 300)):          # the second run wraps round the end of the data
            merged, own, bound = _exchange_search(E, engs, shards, dss, first, count)
            sel = (first + np.arange(count)) % m
            assert np.array_equal((merged & np.uint64(0xFFFFFFFF)).astype(np.int64), want_i[sel])
            assert np.array_equal((merged >> np.uint64(32)).astype(np.uint32), bits(want_d[sel]))
            assert (own[1] == np.uint64(0x7FFFFFFFFFFFFFFF)).all()      # the far shard abstains for every sample
            # the plain per-shard search + MIN gives the same keys
            plain = []
            for e, cb, ds in zip(engs, shards, dss):
                kb = e.device_alloc(8 * count)
                hk = np.empty(count, dtype=np.uint64)
                assert e.lib.somhip_batch_winner_keys(cb.h, ds.h, first, count, kb) == 0
                assert e.lib.somhip_copy_to_host(e.h, hk.ctypes.data_as(C.c_void_p), kb, 8 * count) == 0
                e.device_free(kb)
                plain.append(hk)
            assert np.array_equal(np.minimum.reduce(plain), merged)
            assert (plain[1] != np.uint64(0x7FFFFFFFFFFFFFFF)).all()    # (on its own the far shard does name its best row)
        # shapes the two-level pre-filter does not take: refused, and the availability call says so
        small = E.Codebook(engs[0], codes[:600, :40].copy(), row_offset=0, n_global=1500)
        ds40 = E.Dataset(engs[0], x[:, :40].copy())
        assert engs[0].lib.somhip_shard_exchange_available(small.h, ds40.h, m) == 0
        assert engs[0].lib.somhip_shard_exchange_available(shards[0].h, dss[0].h, 100) == 0
        kb, bb = engs[0].device_alloc(8 * m), engs[0].device_alloc(4 * m)
        assert engs[0].lib.somhip_shard_winner_begin(small.h, ds40.h, 0, m, kb, bb) != 0
        # the three calls are one search: out of order, or for another range, they are refused
        lib0, cb0, ds0 = engs[0].lib, shards[0].h, dss[0].h
        assert lib0.somhip_shard_winner_refine(cb0, ds0, 0, m, bb) != 0                   # nothing begun
        assert lib0.somhip_shard_winner_begin(cb0, ds0, 0, m, kb, bb) == 0
        assert lib0.somhip_shard_winner_finish(cb0, ds0, 0, m, bb, kb) != 0               # refine left out
        assert lib0.somhip_shard_winner_begin(cb0, ds0, 0, m, kb, bb) == 0
        assert lib0.somhip_shard_winner_refine(cb0, ds0, 1, m, bb) != 0                   # another range
        assert lib0.somhip_shard_winner_begin(cb0, ds0, 0, m, kb, bb) == 0
        assert lib0.somhip_batch_winner_keys(cb0, ds0, 0, m, kb) == 0                     # a whole search in between
        assert lib0.somhip_shard_winner_refine(cb0, ds0, 0, m, bb) != 0
        assert b"continuation" in lib0.somhip_last_error()
        engs[0].device_free(kb)
        engs[0].device_free(bb)
    finally:
        for e in engs:
            e.close()


@pytest.mark.parametrize("neigh", ["bubble", "gaussian"])
def test_som_interleaved_shards_with_exchanged_bounds(E, oracle, neigh):
    """Mini-batch SOM training over three interleaved shards with the bounds exchanged inside every winner search:
    bit-equal to the unsharded run of the batch oracle (winners and final codebook)."""
    from som_lvq_pak_amd._lib import SomParams
    xdim, ydim, S, d = 24, 40, 3, 32
    topol, nb = E.TOPOL_HEXA, (E.NEIGH_GAUSSIAN if neigh == "gaussian" else E.NEIGH_BUBBLE)
    x, _ = synth(91, 900, d)
    B, length = 256, 1536
    ini = oracle.randinit(x, xdim, ydim, 5)
    oc, oi, _ = oracle.som_train(ini, xdim, ydim, topol, nb, x, length, 0.05, 9.0, batch=B)
    units = [E.shard_units(xdim, ydim, r, S) for r in range(S)]
    engs = [E.Engine(0) for _ in range(S)]
    try:
        for e in engs:
            e.set_update_mode("exact")
        dss = [E.Dataset(e, x) for e in engs]
        shards = [E.Codebook(engs[r], ini[units[r]], topol, nb, xdim, ydim, interleave=(r, S)) for r in range(S)]
        p = SomParams(length, 0.05, 9.0, 1, 0, 0, B, 0, length, 0)
        got_idx = []
        kb = [engs[s].device_alloc(8 * B) for s in range(S)]
        for it0 in range(0, length, B):
            first = it0 % x.shape[0]
            merged, _, _ = _exchange_search(E, engs, shards, dss, first, B)
            got_idx.append((merged & np.uint64(0xFFFFFFFF)).astype(np.int64))
            for s in range(S):
                assert engs[s].lib.somhip_copy_to_device(engs[s].h, kb[s], merged.ctypes.data_as(C.c_void_p), 8 * B) == 0
                assert engs[s].lib.somhip_som_batch_update(shards[s].h, dss[s].h, C.byref(p), it0, B, first, kb[s]) == 0
            for e in engs:
                e.sync()
        assert np.array_equal(np.concatenate(got_idx), oi)
        got = np.empty_like(oc)
        for r in range(S):
            got[units[r]] = shards[r].download()
        assert np.array_equal(bits(got), bits(oc))
    finally:
        for e in engs:
            e.close()


@pytest.mark.parametrize("knn", [2, 4, 8])
def test_knn_row_sharded_merge(eng, E, oracle, knn):
    """X2 on one GPU: three uneven row shards, each shard's k best keys per sample, union sorted --
    must equal find_winner_knn over the whole codebook (duplicated rows included: on equal distances
    the later row comes first, lvq_pak.c:197)."""
    from som_lvq_pak_amd import sharded
    x, _ = synth(77, 300, 24)
    rs = np.random.RandomState(knn)
    codes = np.concatenate([x[rs.randint(0, 300, 400)], x[:40], x[:40]]).astype(np.float32)   # exact ties
    n = codes.shape[0]
    want_i, want_d, _ = oracle.winners(codes, x, knn, True)
    cuts = [0, 130, 131, n]
    ds = E.Dataset(eng, x)
    parts = []
    for a, b in zip(cuts, cuts[1:]):
        cb = E.Codebook(eng, codes[a:b], row_offset=a, n_global=n)
        kb = eng.device_alloc(8 * 300 * knn)
        assert eng.lib.somhip_batch_topk_keys(cb.h, ds.h, 0, 300, knn, E.TIE_KNN, kb) == 0
        hk = np.empty((300, knn), dtype=np.uint64)
        assert eng.lib.somhip_copy_to_host(eng.h, hk.ctypes.data_as(C.c_void_p), kb, 8 * 300 * knn) == 0
        eng.device_free(kb)
        parts.append(hk)
    merged = np.sort(np.concatenate(parts, axis=1), axis=1)[:, :knn]
    gd, gi = sharded.unpack_knn_keys(merged)
    assert np.array_equal(gi, want_i)
    assert np.array_equal(bits(gd), bits(want_d))


@pytest.mark.parametrize("knn,rule", [(2, "knn"), (4, "knn"), (8, "knn")])
@pytest.mark.parametrize("n,d,m", [(300, 24, 100), (5000, 64, 257), (4096, 130, 64), (70, 512, 33)])
def test_topk_behind_the_mfma_prefilter(eng, E, oracle, knn, rule, n, d, m):
    """k nearest rows through the bf16 pre-filter + group re-rank (K2k) = find_winner_knn: random codes,
    duplicated rows (exact ties: later row first), tight clusters that put many rows within tau"""
    x, _ = synth(n + d, m, d, k=4, spread=2.0)
    rs = np.random.RandomState(n)
    codes = (x[rs.randint(0, m, n)] + 0.01 * rs.standard_normal((n, d))).astype(np.float32)
    codes[n // 2:n // 2 + 20] = codes[:20]                     # exact duplicates
    want_i, want_d, _ = oracle.winners(codes, x, knn, True)
    cb, ds = E.Codebook(eng, codes), E.Dataset(eng, x)
    os.environ["SOMHIP_TOPK_MFMA"] = "1"
    try:
        gi, gd, _ = E.find_winners(cb, ds, knn=knn, tie=E.TIE_KNN)
    finally:
        os.environ.pop("SOMHIP_TOPK_MFMA", None)
    assert np.array_equal(gi, want_i)
    assert np.array_equal(bits(gd), bits(want_d))


def test_top8_two_level_and_by_group_equal_the_plain_paths(eng, E, oracle):
    """The top-8 search of a big codebook (>= 512 row groups): two-level pre-filter + re-rank by row group (the
    defaults there) against the one-level pre-filter, the per-pair re-rank, and find_winner_knn itself; clustered
    codes with exact duplicates, so that ties and crowded groups occur."""
    n, d, m = 40000, 64, 1024
    x, _ = synth(77, m, d, k=6, spread=2.0)
    rs = np.random.RandomState(5)
    codes = (x[rs.randint(0, m, n)] + 0.05 * rs.standard_normal((n, d))).astype(np.float32)
    codes[n // 2:n // 2 + 50] = codes[:50]
    want_i, want_d, _ = oracle.winners(codes, x, 8, True)
    cb, ds = E.Codebook(eng, codes), E.Dataset(eng, x)
    for env in ({}, {"SOMHIP_TOPK_ONE_LEVEL": "1"}, {"SOMHIP_TOPK_BYPAIR": "1"}, {"SOMHIP_TOPK_ONE_LEVEL": "1", "SOMHIP_TOPK_BYPAIR": "1"}):
        os.environ.update(env)
        try:
            gi, gd, _ = E.find_winners(cb, ds, knn=8, tie=E.TIE_KNN)
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert np.array_equal(gi, want_i), env
        assert np.array_equal(bits(gd), bits(want_d)), env


def test_engine_chosen_batch_schedule_is_the_documented_one(eng, E):
    """SOMHIP_BATCH_AUTO (somhip_som_auto_batch): at configs[3]'s parameters 32768-iteration batches up to iteration
    8 486 912 (whole batches), 8192 after; a map the rule does not vouch for: batch 1.  A run with it equals the same run
    made in two explicit segments (maps it vouches for) or the online engine's (small maps), bit for bit."""
    lib = eng.lib
    BL = 32768
    L10 = 10_000_000
    t10 = 259 * BL
    assert E.som_auto_batch(lib, L10, 0) == (0, BL)
    assert E.som_auto_batch(lib, L10, t10 - 1) == (t10 - BL, BL)
    assert E.som_auto_batch(lib, L10, t10) == (t10, 8192)
    last = t10 + (L10 - 1 - t10) // 8192 * 8192
    assert E.som_auto_batch(lib, L10, L10 - 1) == (last, L10 - last)
    assert E.som_auto_batch(lib, 100000, 5000, radius=10.0, n_units=1024) == (5000, 1)
    # 128 x 128 map, 100 long batches' worth of iterations over a 4096-vector data set
    x, _ = synth(31, 4096, 16, k=5, spread=2.0)
    rs = np.random.RandomState(3)
    n = 128 * 128
    init = (x[rs.randint(0, 4096, n)] + 0.1 * rs.standard_normal((n, 16))).astype(np.float32)
    L = 100 * BL
    kw = dict(alpha=0.05, radius=64.0, n_units=n)
    t1 = next(t for t in range(0, L, BL) if E.som_auto_batch(lib, L, t, **kw)[1] != BL)
    bt = E.som_auto_batch(lib, L, t1, **kw)[1]
    assert t1 >= 64 * BL and bt < BL and E.som_auto_batch(lib, L, L - 1, **kw)[1] <= bt
    ds = E.Dataset(eng, x)
    a = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, 128, 128)
    E.som_train(a, ds, L, 0.05, 64.0, batch=E.BATCH_AUTO, trace=False)
    b = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, 128, 128)
    E.som_train(b, ds, L, 0.05, 64.0, batch=BL, count=t1, trace=False)
    E.som_train(b, ds, L, 0.05, 64.0, batch=bt, start_iter=t1, trace=False)
    assert np.array_equal(bits(a.download()), bits(b.download()))
    a.close(); b.close()
    # a small map: auto == online, traces included
    init = init[:256]
    a = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, 16, 16)
    ta = E.som_train(a, ds, 6000, 0.05, 8.0, batch=E.BATCH_AUTO)
    b = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, 16, 16)
    tb = E.som_train(b, ds, 6000, 0.05, 8.0, batch=1)
    assert np.array_equal(ta[0], tb[0]) and np.array_equal(bits(ta[1]), bits(tb[1]))
    assert np.array_equal(bits(a.download()), bits(b.download()))


# --------------------------------------------------------------------------- lvq*_training
LVQ_CASES = [("lvq1", 1, {}), ("olvq1", 2, {}), ("lvq2", 3, {"winlen": 0.3}),
             ("lvq3", 4, {"winlen": 0.3, "epsilon": 0.1}), ("lvq1_invt", 1, {"alpha_type": 2})]


@pytest.mark.parametrize("tag,kind,kw", LVQ_CASES)
def test_lvq_golden(eng, E, exdata, tag, kind, kw):
    from som_lvq_pak_amd import textio
    g = load_trace("lvq_ex1_%s" % tag)
    ci, e1, e2 = exdata["lvq_init"], exdata["ex1"], exdata["ex2"]
    cb = E.Codebook(eng, ci.points, labels=ci.first_label)
    ds = E.Dataset(eng, e1.points, labels=e1.first_label)
    tal, ti, td = E.lvq_train(cb, ds, kind, 5000, 0.05, **kw)
    assert np.array_equal(ti, g["trace_index"])
    assert np.array_equal(bits(td), bits(g["trace_diff"]))
    codes = cb.download()
    assert np.array_equal(bits(codes), bits(g["codes"]))
    if kind == 2:
        assert [textio.fmt_g(a) for a in tal] == list(g["lra"])
    ds2 = E.Dataset(eng, e2.points)
    wi, _, _ = E.find_winners(cb, ds2)
    assert int((ci.first_label[wi[:, 0]] == e2.first_label).sum()) == int(g["correct_on_ex2"])


@pytest.mark.parametrize("kind", [1, 2, 3, 4])
def test_lvq_random_vs_oracle_and_segments(eng, E, oracle, kind):
    x, lab = synth(400 + kind, 640, 33, k=5, spread=2.0)
    rs = np.random.RandomState(kind)
    pick = rs.choice(640, 300, replace=False)
    codes, clab = x[pick].copy(), lab[pick].copy()
    kw = {"winlen": 0.25} if kind >= 3 else {}
    if kind == 4:
        kw["epsilon"] = 0.2
    oc, ol, oi, od = oracle.lvq_train(kind, codes, clab, x, lab, 2000, 0.1, **kw)
    cb = E.Codebook(eng, codes, labels=clab)
    ds = E.Dataset(eng, x, labels=lab)
    tal, t1, d1 = E.lvq_train(cb, ds, kind, 2000, 0.1, count=777, **kw)
    tal, t2, d2 = E.lvq_train(cb, ds, kind, 2000, 0.1, start_iter=777, talpha=tal, **kw)
    assert np.array_equal(np.concatenate([t1, t2]), oi)
    assert np.array_equal(bits(np.concatenate([d1, d2])), bits(od))
    assert np.array_equal(bits(cb.download()), bits(oc))
    if kind == 2:
        assert np.array_equal(bits(tal), bits(ol))


def _lvq_both_engines(eng, E, oracle, kind, codes, clab, x, lab, length, alpha, **kw):
    """exact batched engine (default) and one-launch-per-iteration engine against the oracle"""
    oc, ol, oi, od = oracle.lvq_train(kind, codes, clab, x, lab, length, alpha, **kw)
    stats = {}
    for mode in ("batched", "batched_serial", "batched_mfma_topk", "batched_sync", "batched_pairs_valu", "online"):
        if mode == "online":
            os.environ["SOMHIP_LVQ_ONLINE"] = "1"
        if mode == "batched_sync":
            os.environ["SOMHIP_LVQ_SYNC"] = "1"              # every batch's verdict read back before the next (round 2's first loop)
        if mode == "batched_pairs_valu":
            os.environ["SOMHIP_LVQ_PAIRS_VALU"] = "1"        # relation (*) from direct-form distances instead of the Gram form
        if mode == "batched_mfma_topk":
            os.environ["SOMHIP_TOPK_MFMA"] = "1"
        if mode == "batched_serial":
            os.environ["SOMHIP_LVQ_SERIAL"] = "1"            # one component: the serial walk of round 1
        try:
            cb = E.Codebook(eng, codes, labels=clab)
            ds = E.Dataset(eng, x, labels=lab)
            before = eng.lvq_stats()
            tal, ti, td = E.lvq_train(cb, ds, kind, length, alpha, **kw)
            after = eng.lvq_stats()
        finally:
            os.environ.pop("SOMHIP_LVQ_ONLINE", None)
            os.environ.pop("SOMHIP_TOPK_MFMA", None)
            os.environ.pop("SOMHIP_LVQ_SERIAL", None)
            os.environ.pop("SOMHIP_LVQ_SYNC", None)
            os.environ.pop("SOMHIP_LVQ_PAIRS_VALU", None)
        assert np.array_equal(ti, oi), mode
        assert np.array_equal(bits(td), bits(od)), mode
        assert np.array_equal(bits(cb.download()), bits(oc)), mode
        if kind == 2:
            assert np.array_equal(bits(tal), bits(ol)), mode
        stats[mode] = {k: after[k] - before[k] for k in after if k != "phase_us"}
    assert stats["online"]["batches"] == 0 and stats["online"]["samples"] == 0
    assert stats["batched"]["samples"] == length
    assert stats["batched_serial"]["components"] == stats["batched_serial"]["batches"]      # one walk per batch
    assert stats["batched"]["components"] >= stats["batched"]["batches"]
    # (the two loops size their batches differently after a stop, so only the totals must agree)
    assert stats["batched"]["samples"] == stats["batched_sync"]["samples"] == length
    assert (stats["batched"]["stop_list"] > 0) == (stats["batched_sync"]["stop_list"] > 0)
    return stats["batched_serial"], stats["batched"]


@pytest.mark.parametrize("kind", [1, 2, 3, 4])
@pytest.mark.parametrize("shape", ["tiny_codebook", "one_cluster", "wide_rows", "odd_dim", "many_codes"])
def test_lvq_exact_batches_stop_conditions(eng, E, oracle, kind, shape):
    """The batched LVQ engine must give the online result whatever ends its batches: a codebook
    smaller than the candidate list, every candidate already corrected (all samples in one
    cluster of a dozen codes), the row cache full (dim 1000 -> 32 slots), a dim that is not
    a multiple of 4, and the easy case (many codes, few collisions)."""
    n, d, m, k, spread, length = {"tiny_codebook": (6, 7, 200, 3, 1.0, 600),
                                  "one_cluster": (12, 16, 400, 1, 0.3, 1500),
                                  "wide_rows": (300, 1000, 260, 4, 2.0, 700),
                                  "odd_dim": (90, 13, 300, 4, 1.5, 1200),
                                  "many_codes": (3000, 48, 2000, 20, 3.0, 3000)}[shape]
    x, lab = synth(900 + kind, m, d, k=k, spread=spread)
    rs = np.random.RandomState(7 * kind + len(shape))
    if shape == "one_cluster":
        lab = rs.randint(1, 4, m).astype(lab.dtype)         # mixed labels inside one blob: pushes and pulls
    if n <= m:
        pick = rs.choice(m, n, replace=False)
        codes, clab = x[pick].copy(), lab[pick].copy()
    else:
        pick = rs.randint(0, m, n)
        codes = (x[pick] + 0.05 * rs.randn(n, d)).astype(np.float32)
        clab = lab[pick].copy()
    kw = {"winlen": 0.3} if kind >= 3 else {}
    if kind == 4:
        kw["epsilon"] = 0.15
    st, stc = _lvq_both_engines(eng, E, oracle, kind, codes, clab, x, lab, length, 0.08, **kw)
    assert 1 <= st["batches"] <= length
    if shape == "one_cluster" and kind <= 2:
        assert st["stop_list"] > 0              # the case was built to exhaust candidate lists
    if shape == "wide_rows" and kind <= 2:
        assert st["stop_cache"] > 0             # ... and this one to fill the 32-slot cache (serial walk)
    if shape == "tiny_codebook":
        assert st["stop_list"] == 0             # all rows are listed: nothing can be missed
        assert stc["components"] == stc["batches"]      # lists shorter than 8 rows: everything interacts
    if shape == "many_codes":
        assert st["batches"] <= length // 20    # the speculation has to pay off somewhere
        assert stc["components"] > 4 * stc["batches"] and stc["batches"] <= st["batches"]   # 20 clusters walk side by side


# --------------------------------------------------------------------------- error behaviour
def test_errors(eng, E):
    rows = np.zeros((4, 3), dtype=np.float32)
    cb = E.Codebook(eng, rows, 3, 1, 2, 2)
    ds = E.Dataset(eng, np.zeros((5, 4), dtype=np.float32))
    with pytest.raises(Exception, match="code dimension"):
        E.som_train(cb, ds, 10, 0.1, 1.0)
    cb2 = E.Codebook(eng, rows)           # not a map
    ds2 = E.Dataset(eng, np.zeros((5, 3), dtype=np.float32))
    with pytest.raises(Exception, match="SOM parameters"):
        E.som_train(cb2, ds2, 10, 0.1, 1.0)
    with pytest.raises(Exception, match="labels"):
        E.lvq_train(cb2, ds2, 1, 10, 0.1)


# --------------------------------------------------------------------------- MFMA pre-filter + exact re-rank
@pytest.mark.parametrize("n,d,m", [(64, 4, 32), (200, 20, 333), (1000, 130, 257), (4096, 64, 512),
                                    (777, 33, 129), (130, 512, 64)])
def test_mfma_prefilter_equals_direct_scan(eng, E, oracle, n, d, m):
    x, _ = synth(n + d, m, d)
    rs = np.random.RandomState(n)
    codes = (x[rs.randint(0, m, n)] + 0.25 * rs.standard_normal((n, d))).astype(np.float32)
    cb, ds = E.Codebook(eng, codes), E.Dataset(eng, x)
    oi, od, _ = oracle.winners(codes, x)
    out = {}
    for mode in ("direct", "mfma", "mfma_bf16"):
        eng.set_scan_mode(mode)
        out[mode] = E.find_winners(cb, ds)
    eng.set_scan_mode("mfma_bf16")
    for mode in ("direct", "mfma", "mfma_bf16"):
        gi, gd, _ = out[mode]
        assert np.array_equal(gi, oi), mode
        assert np.array_equal(bits(gd), bits(od)), mode


@pytest.mark.parametrize("n,d,m,env", [(4096, 64, 512, "SOMHIP_L2_GLOBAL"), (2048, 1024, 300, None),
                                        (4096, 64, 512, "SOMHIP_NO_FUSED_GMIN"), (4096, 64, 512, None)])
def test_two_level_level2_variants(eng, E, oracle, monkeypatch, n, d, m, env):
    """Level 2 of the pre-filter in its two forms -- the group's tiles in LDS (dim <= 512) and operand A from global memory
    (dim > 512, or SOMHIP_L2_GLOBAL) --, each folding its minima into the per-sample minimum itself, and the separate
    pass over the matrix (SOMHIP_NO_FUSED_GMIN): the same exact winners."""
    if env:
        monkeypatch.setenv(env, "1")
    x, _ = synth(n + d, m, d)
    rs = np.random.RandomState(n + d)
    codes = (x[rs.randint(0, m, n)] + 0.25 * rs.standard_normal((n, d))).astype(np.float32)
    codes[17] = codes[n - 5] = x[3]                           # an exact tie far apart: the lower index wins
    cb, ds = E.Codebook(eng, codes), E.Dataset(eng, x)
    oi, od, _ = oracle.winners(codes, x)
    before = eng.scan_stats()
    gi, gd, _ = E.find_winners(cb, ds)
    after = eng.scan_stats()
    assert np.array_equal(gi, oi)
    assert np.array_equal(bits(gd), bits(od))
    assert after["l2_pairs"] > before["l2_pairs"]             # the two-level form did run


def test_mfma_prefilter_adversarial(eng, E, oracle):
    """cases built to break a GEMM-form argmin: exact ties (duplicated rows -> lowest index
    must win), rows one ulp apart, a large common offset (cancellation in ||c||^2 - 2xc),
    a codebook of a few tight clusters (many near-candidates)."""
    rs = np.random.RandomState(9)
    d = 48
    base, _ = synth(3, 300, d)
    cases = []
    # exact duplicates, in shuffled positions
    dup = np.concatenate([base[:100], base[:100], base[:100]], axis=0)[rs.permutation(300)]
    cases.append(("duplicates", dup, base[100:260]))
    # one-ulp neighbours
    near = base[:128].copy()
    near2 = near.copy()
    near2[:, 7] = np.nextafter(near2[:, 7], np.float32(np.inf))
    cases.append(("one_ulp", np.concatenate([near2, near], axis=0), base[128:288] * np.float32(1.0)))
    # large offset: norms ~ 1e4, distances ~ 1e2
    off = (base + np.float32(1500.0)).astype(np.float32)
    cases.append(("offset", off[:200], off[100:300]))
    # tight clusters
    cent = base[:5]
    cl = (cent[rs.randint(0, 5, 640)] + 1e-3 * rs.standard_normal((640, d))).astype(np.float32)
    cases.append(("clusters", cl, (cent[rs.randint(0, 5, 192)] + 1e-3 * rs.standard_normal((192, d))).astype(np.float32)))
    # samples that ARE code rows (distance exactly 0, several zero ties)
    cases.append(("zeros", dup, dup[:96]))
    for mode in ("mfma", "mfma_bf16"):
        eng.set_scan_mode(mode)
        for name, codes, x in cases:
            cb, ds = E.Codebook(eng, codes), E.Dataset(eng, x)
            gi, gd, _ = E.find_winners(cb, ds)
            oi, od, _ = oracle.winners(codes, x)
            assert np.array_equal(gi, oi), (mode, name)
            assert np.array_equal(bits(gd), bits(od)), (mode, name)
    eng.set_scan_mode("mfma_bf16")
    st = eng.scan_stats()
    assert st["rows"] >= st["groups"] >= 1


def test_som_minibatch_mfma_and_direct_same_run(eng, E, oracle):
    x, _ = synth(61, 1500, 40)
    ini = oracle.randinit(x, 20, 13, 6)
    oc, oi, od = oracle.som_train(ini, 20, 13, 3, 1, x, 3000, 0.05, 7.0, batch=256)
    for mode in ("direct", "mfma", "mfma_bf16"):
        eng.set_scan_mode(mode)
        cb, ds = E.Codebook(eng, ini, 3, 1, 20, 13), E.Dataset(eng, x)
        ti, td = E.som_train(cb, ds, 3000, 0.05, 7.0, batch=256)
        assert np.array_equal(ti, oi), mode
        assert np.array_equal(bits(td), bits(od)), mode
        assert np.array_equal(bits(cb.download()), bits(oc)), mode
    eng.set_scan_mode("mfma_bf16")


@pytest.mark.parametrize("mode", ["mfma", "mfma_bf16"])
@pytest.mark.parametrize("d,offset,scale", [(32, 0.0, 1.0), (512, 0.0, 1.0), (512, 40.0, 1.0), (130, 1500.0, 3.0),
                                            (1024, 5.0, 0.01)])
def test_prefilter_error_is_inside_its_bound(eng, E, mode, d, offset, scale):
    """the bound behind tau, measured: |s~ + ||x||^2 - d_direct| must stay below tau/2 (= delta) for
    every (code, sample).  Each row group holds 64 copies of one vector, so the group minimum the
    kernel returns IS that vector's s~."""
    import ctypes as C
    rs = np.random.RandomState(d)
    G, B = 8, 96
    base = (offset + scale * rs.standard_normal((G, d)) * rs.uniform(0.2, 3.0, size=(G, 1))).astype(np.float32)
    codes = np.repeat(base, 64, axis=0)
    x = (offset + scale * rs.standard_normal((B, d))).astype(np.float32)
    cb, ds = E.Codebook(eng, codes), E.Dataset(eng, x)
    eng.set_scan_mode(mode)
    wmin = np.empty((G, B), dtype=np.float32)
    tau = np.empty(B, dtype=np.float32)
    bpad = C.c_int64(0)
    E.check(eng.lib.somhip_debug_prefilter(cb.h, ds.h, 0, B, wmin.ctypes.data_as(C.POINTER(C.c_float)),
                                           tau.ctypes.data_as(C.POINTER(C.c_float)), C.byref(bpad)))
    eng.set_scan_mode("mfma_bf16")
    assert bpad.value == B
    # the reference's direct-form value, in its own arithmetic (fp32, left to right)
    direct = np.zeros((G, B), dtype=np.float32)
    for i in range(d):
        t = (base[:, None, i] - x[None, :, i]).astype(np.float32)
        direct = (direct + (t * t).astype(np.float32)).astype(np.float32)
    xn = (x.astype(np.float64) ** 2).sum(1)
    err = np.abs(wmin.astype(np.float64) + xn[None, :] - direct.astype(np.float64))
    ratio = err / (0.5 * tau.astype(np.float64))[None, :]
    assert ratio.max() < 0.5, ratio.max()            # inside the bound with a factor 2 to spare


@pytest.mark.parametrize("cscale", [1e-3, 1e-6, 0.0])
def test_two_level_prefilter_with_a_codebook_of_tiny_norm(eng, E, oracle, cscale):
    """||c|| << ||x|| (a map initialised near the origin): the level-1 error bound shrinks with ||c|| while the
    three-product bound does not, so the level-1 window has to be widened to delta1 + 3 delta3 for the groups left out
    of level 2 to stay beyond the re-rank's reach.  Winners through the two-level pre-filter = the direct scan's."""
    rs = np.random.RandomState(17)
    n, d, m = 4096, 64, 512
    codes = (cscale * rs.standard_normal((n, d))).astype(np.float32)
    x = rs.standard_normal((m, d)).astype(np.float32)
    ini = codes.copy()
    want_i, want_d, _ = oracle.winners(ini, x)
    cb, ds = E.Codebook(eng, ini, E.TOPOL_HEXA, E.NEIGH_BUBBLE, 64, 64), E.Dataset(eng, x)
    got = {}
    for mode in ("direct", "mfma_bf16"):
        eng.set_scan_mode(mode)
        try:
            keys = E.batch_winner_keys(cb, ds, 0, m) if hasattr(E, "batch_winner_keys") else None
            ti, td = E.som_train(cb, ds, 64 * m, 0.0, 1.0, batch=m, count=m)     # alpha 0: the search of one batch, no change
        finally:
            eng.set_scan_mode("mfma_bf16")
        assert np.array_equal(ti, want_i[:, 0]), mode
        assert np.array_equal(bits(td), bits(want_d[:, 0])), mode


# --------------------------------------------------------------------------- 8x8-patch row order
@pytest.mark.parametrize("topol,neigh", [(3, 1), (3, 2), (4, 1), (4, 2)])
@pytest.mark.parametrize("batch", [1, 48])
def test_patch_row_order_equals_oracle(eng, E, oracle, topol, neigh, batch):
    """maps with sides multiple of 8 are stored as 8x8 patches of units; every result must still be
    in the reference's unit order (trace indices, downloaded rows)"""
    x, _ = synth(71 + topol + neigh, 800, 12)
    ini = oracle.randinit(x, 24, 16, 9)
    oc, oi, od = oracle.som_train(ini, 24, 16, topol, neigh, x, 1200, 0.06, 7.0, batch=batch)
    cb, ds = E.Codebook(eng, ini, topol, neigh, 24, 16), E.Dataset(eng, x)
    assert np.array_equal(bits(cb.download()), bits(ini))
    ti, td = E.som_train(cb, ds, 1200, 0.06, 7.0, batch=batch)
    assert np.array_equal(ti, oi)
    assert np.array_equal(bits(td), bits(od))
    assert np.array_equal(bits(cb.download()), bits(oc))
    wi, wd, _ = E.find_winners(cb, ds)
    qi, qd, _ = oracle.winners(oc, x)
    assert np.array_equal(wi, qi) and np.array_equal(bits(wd), bits(qd))


@pytest.mark.parametrize("neigh", [1, 2])
def test_fixed_points_beyond_the_edge_of_the_map(eng, E, oracle, neigh):
    """som_rout.c:628-632 hands xfix / yfix to the neighbourhood function as they are: a fixed point beyond the map's
    edge teaches the units within the radius of it (the engine used to refuse such a data file in the
    middle of training).  Online and mini-batch, 8x8-patch and linear row order, against the oracle's bits."""
    rs = np.random.RandomState(17)
    x, _ = synth(82, 400, 12)
    fixed = np.full((400, 2), -1, dtype=np.int16)
    for r in rs.choice(400, 40, replace=False):
        fixed[r] = (rs.randint(0, 30), rs.randint(0, 26))           # map is 16 x 8 (and 13 x 9): most of these lie outside
    fixed[5] = (16, 3); fixed[6] = (3, 8); fixed[7] = (200, 300); fixed[8] = (32767, 32767)
    for xd, yd in ((16, 8), (13, 9)):
        ini = oracle.randinit(x, xd, yd, 4)
        for batch in (1, 48):
            oc, oi, od = oracle.som_train(ini, xd, yd, 3, neigh, x, 700, 0.07, 5.0, fixed_xy=fixed, fixed_on=1, batch=batch)
            cb = E.Codebook(eng, ini, 3, neigh, xd, yd)
            ds = E.Dataset(eng, x, fixed_xy=fixed)
            ti, td = E.som_train(cb, ds, 700, 0.07, 5.0, use_fixed=1, batch=batch)
            assert np.array_equal(ti, oi), (xd, batch)
            assert np.array_equal(bits(td), bits(od)), (xd, batch)
            assert np.array_equal(bits(cb.download()), bits(oc)), (xd, batch)
            cb.close(); ds.close()


@pytest.mark.parametrize("radius,topol", [(24.0, 3), (6.0, 3), (9.0, 4)])
def test_long_batches_take_the_decoded_winner_path_and_equal_the_oracle(eng, E, oracle, radius, topol):
    """Runs of >= 16384 iterations decode the winners once (k_decode_winners) and k_som_members queues the samples near a
    row group before it decides membership (two phases per trip, round 3): member lists in iteration order with the exact
    lattice test -- checked the only way that does not share the code under test: the exact update kernels on a
    20 000-iteration batch against the batch oracle, codebook bits and winner traces (hexa and rect, the neighbourhood
    a few patches or a good part of the 64 x 48 map wide, a run that wraps around the data and is cut by fixed points)."""
    rs = np.random.RandomState(11)
    x, _ = synth(83, 6000, 8, k=9, spread=3.0)
    fixed = np.full((6000, 2), -1, dtype=np.int16)
    for r in rs.choice(6000, 30, replace=False):
        fixed[r] = (rs.randint(0, 64), rs.randint(0, 48))
    ini = oracle.randinit(x, 64, 48, 4)
    L, B = 40000, 20000
    oc, oi, od = oracle.som_train(ini, 64, 48, topol, 1, x, L, 0.06, radius, fixed_xy=fixed, fixed_on=1, batch=B)
    cb = E.Codebook(eng, ini, topol, 1, 64, 48)
    ds = E.Dataset(eng, x, fixed_xy=fixed)
    ti, td = E.som_train(cb, ds, L, 0.06, radius, use_fixed=1, batch=B)
    assert np.array_equal(ti, oi) and np.array_equal(bits(td), bits(od))
    assert np.array_equal(bits(cb.download()), bits(oc))
    cb.close(); ds.close()


def test_long_batches_gemm_form_tail_lists_equal_full_lists(eng, E, monkeypatch):
    """The same path in update mode gemm (dims in whole 128s): k_som_members in tail mode -- only the end of every list
    is made, the trips run from the end of the batch -- must give k_som_update_gemm what whole lists give it, bit for
    bit (SOMHIP_GEMM_FULL_LISTS=1), at a radius where the tail is cut short and at one where it is the whole list; and
    the result stays within fp32 rounding of the exact kernels'."""
    x, _ = synth(84, 20000, 128, k=7, spread=3.0)
    rs = np.random.RandomState(5)
    ini = (x[rs.randint(0, 20000, 64 * 48)] + 0.2 * rs.standard_normal((64 * 48, 128))).astype(np.float32)
    ds = E.Dataset(eng, x)
    scale = float(np.abs(x).max())
    for radius, it0 in ((24.0, 0), (5.0, 60000)):
        out = {}
        for tag, mode, env in (("tail", "gemm", None), ("full", "gemm", "1"), ("exact", "exact", None)):
            if env:
                monkeypatch.setenv("SOMHIP_GEMM_FULL_LISTS", env)
            else:
                monkeypatch.delenv("SOMHIP_GEMM_FULL_LISTS", raising=False)
            eng.set_update_mode(mode)
            cb = E.Codebook(eng, ini, 3, 1, 64, 48)
            ti, _ = E.som_train(cb, ds, 100000, 0.05, radius, batch=20000, start_iter=it0, count=20000, data_first=0)
            out[tag] = (ti, cb.download())
            cb.close()
        eng.set_update_mode("exact")
        monkeypatch.delenv("SOMHIP_GEMM_FULL_LISTS", raising=False)
        assert np.array_equal(out["tail"][0], out["exact"][0])
        assert np.array_equal(bits(out["tail"][1]), bits(out["full"][1])), radius
        assert float(np.abs(out["tail"][1] - out["exact"][1]).max()) <= 8e-6 * scale, radius


def test_patch_row_order_masks_fixed_weights_and_shards(eng, E, oracle):
    import ctypes as C
    from som_lvq_pak_amd._lib import SomParams
    rs = np.random.RandomState(3)
    x, _ = synth(81, 500, 10)
    mask = (rs.rand(500, 10) < 0.1).astype(np.uint8)
    mask[11] = 1
    weight = rs.randint(0, 3, size=500).astype(np.int16)
    fixed = np.full((500, 2), -1, dtype=np.int16)
    for r in rs.choice(500, 25, replace=False):
        fixed[r] = (rs.randint(0, 16), rs.randint(0, 16))
    ini = oracle.randinit(x, 16, 16, 4)
    for batch in (1, 40):
        oc, oi, od = oracle.som_train(ini, 16, 16, 3, 1, x, 900, 0.07, 5.0, weight=weight, fixed_xy=fixed,
                                      mask=mask, fixed_on=1, weights_on=1, batch=batch)
        cb = E.Codebook(eng, ini, 3, 1, 16, 16)
        ds = E.Dataset(eng, x, mask=mask, weight=weight, fixed_xy=fixed)
        ti, td = E.som_train(cb, ds, 900, 0.07, 5.0, use_fixed=1, use_weights=1, batch=batch)
        assert np.array_equal(ti, oi), batch
        assert np.array_equal(bits(td), bits(od)), batch
        assert np.array_equal(bits(cb.download()), bits(oc)), batch
    # two shards on an 8-row boundary (both in patch order) == the whole map
    oc, oi, _ = oracle.som_train(ini, 16, 16, 3, 1, x, 512, 0.05, 6.0, batch=64)
    shards = [E.Codebook(eng, ini[:128], 3, 1, 16, 16, row_offset=0, n_global=256),
              E.Codebook(eng, ini[128:], 3, 1, 16, 16, row_offset=128, n_global=256)]
    ds = E.Dataset(eng, x)
    kb = [eng.device_alloc(8 * 64) for _ in range(3)]
    hk = [np.empty(64, dtype=np.uint64) for _ in range(2)]
    p = SomParams(512, 0.05, 6.0, 1, 0, 0, 64, 0, 512, 0)
    for it0 in range(0, 512, 64):
        first = it0 % 500
        for s in range(2):
            E.check(eng.lib.somhip_batch_winner_keys(shards[s].h, ds.h, first, 64, kb[s]))
            E.check(eng.lib.somhip_copy_to_host(eng.h, hk[s].ctypes.data_as(C.c_void_p), kb[s], 8 * 64))
        merged = np.minimum(hk[0], hk[1])
        E.check(eng.lib.somhip_copy_to_device(eng.h, kb[2], merged.ctypes.data_as(C.c_void_p), 8 * 64))
        for s in range(2):
            E.check(eng.lib.somhip_som_batch_update(shards[s].h, ds.h, C.byref(p), it0, 64, first, kb[2]))
    eng.sync()
    got = np.concatenate([shards[0].download(), shards[1].download()], axis=0)
    assert np.array_equal(bits(got), bits(oc))
    for k in kb:
        eng.device_free(k)


@pytest.mark.parametrize("batch", [1, 32])
def test_map_wider_than_1024_uses_the_general_lattice_path(eng, E, oracle, batch):
    """sides > 1024: the fp32-only / integer lattice shortcuts do not apply (values no longer exact
    multiples of 1/4 below 2^22 in general), the per-unit mixed fp32/fp64 form must be used"""
    x, _ = synth(91, 300, 3)
    ini = oracle.randinit(x, 1032, 8, 2)
    oc, oi, od = oracle.som_train(ini, 1032, 8, 3, 1, x, 160, 0.2, 300.0, batch=batch)
    cb, ds = E.Codebook(eng, ini, 3, 1, 1032, 8), E.Dataset(eng, x)
    ti, td = E.som_train(cb, ds, 160, 0.2, 300.0, batch=batch)
    assert np.array_equal(ti, oi) and np.array_equal(bits(td), bits(od))
    assert np.array_equal(bits(cb.download()), bits(oc))


# --------------------------------------------------------------------------- randomised sweep
def test_random_parity_sweep_short():
    """tools/fuzz_parity.py for 20 s: generated map / codebook shapes, ragged dims, batch sizes, masks,
    weights, fixed points, all LVQ kinds, k-NN scans -- each case bit for bit against the oracle
    (profiles/r01_fuzz_parity.txt holds a 7-minute run: 15 601 cases)."""
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "20", "4242"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0 and "fuzz ok" in p.stdout, p.stdout[-2000:]


def test_device_generator_equals_host_stream(eng, E, oracle):
    """somhip_dataset_generate (k_gen_mixture) against the host form of the stream (engine.gen_rows, itself pinned to
    paklib.c's pak_gen_row on the CPU): mixture ids equal, and the winners of the generated rows on a codebook --
    index and distance bits -- equal those of the host rows uploaded the ordinary way, windows included."""
    seed, k, dim, n = 20251, 9, 24, 3000
    hx, hc = E.gen_rows(seed, k, dim, 0, n)
    dsg = E.Dataset(eng, generate=(seed, k, dim, 0, n))
    assert np.array_equal(dsg.centres, hc)
    cb = E.Codebook(eng, hx[::13][:150].copy())
    gi, gd, _ = E.find_winners(cb, dsg)
    hi, hd, _ = E.find_winners(cb, E.Dataset(eng, hx))
    assert np.array_equal(gi, hi) and np.array_equal(bits(gd), bits(hd))
    wi, wd, _ = oracle.winners(hx[::13][:150].copy(), hx[:400])
    assert np.array_equal(gi[:400], wi) and np.array_equal(bits(gd[:400]), bits(wd))
    dsw = E.Dataset(eng, generate=(seed, k, dim, 1234, 500))               # a window of the same stream
    assert np.array_equal(dsw.centres, hc[1234:1734])
    wi2, wd2, _ = E.find_winners(cb, dsw)
    assert np.array_equal(wi2, hi[1234:1734]) and np.array_equal(bits(wd2), bits(hd[1234:1734]))


@pytest.mark.parametrize("shape", [(16, 16, 64, 3, 1, 64), (24, 8, 64, 4, 1, 100), (16, 24, 128, 3, 1, 333), (10, 7, 64, 3, 1, 30),
                                   (16, 16, 64, 3, 2, 64), (8, 24, 16, 4, 2, 37), (13, 9, 32, 3, 2, 50)])
def test_scalar_operand_update_kernels_vs_oracle(eng, E, oracle, shape, monkeypatch):
    """The scalar-operand update kernels (k_som_update_bubble_s: scalar loads of member entries and sample
    slices, exec mask from the entry, packed fp32; k_som_update_gauss_s: rates once per workgroup through LDS)
    are picked by size on big maps; here they are forced on small ones (SOMHIP_UPD_QW=4) and compared with the
    batch oracle bit for bit -- data set wrapped, batch not a multiple of the tile, patch and linear row order,
    and against the LDS-tile kernel (SOMHIP_UPD_LDS=1) on the same run."""
    xdim, ydim, dim, topol, neigh, batch = shape
    x, _ = synth(61 + dim, 900, dim)
    ini = oracle.randinit(x, xdim, ydim, 3)
    length = 1400                                                   # wraps the data set
    oc, oi, od = oracle.som_train(ini, xdim, ydim, topol, neigh, x, length, 0.06, 6.0, batch=batch)
    ds = E.Dataset(eng, x)
    got = {}
    for form in ("scalar", "lds"):
        monkeypatch.setenv("SOMHIP_UPD_QW", "4")
        if form == "lds":
            monkeypatch.setenv("SOMHIP_UPD_LDS", "1")
        cb = E.Codebook(eng, ini, topol, neigh, xdim, ydim)
        eng.timing(True)
        eng.timing_reset()
        ti, td = E.som_train(cb, ds, length, 0.06, 6.0, batch=batch)
        table = eng.timing_table()
        eng.timing(False)
        monkeypatch.delenv("SOMHIP_UPD_LDS", raising=False)
        assert np.array_equal(ti, oi) and np.array_equal(bits(td), bits(od)), form
        got[form] = cb.download()
        assert np.array_equal(bits(got[form]), bits(oc)), form
        if form == "scalar" and neigh == 1:
            assert table["k_som_update_bubble_s"][0] > 0               # the kernel under test did run (dim % 64 == 0)
        if form == "lds":
            assert table["k_som_update_bubble_s"][0] == 0 and table["k_som_update_run"][0] > 0


@pytest.mark.parametrize("batch", [64, 61, 4])
def test_scalar_update_kernel_full_lists(eng, E, oracle, batch, monkeypatch):
    """Radius larger than the map: every sample is in every row group's member list, so the lists use their whole
    capacity and the eight null entries behind them sit at the very end of each group's segment (the scalar-operand
    kernel prefetches three entries past the last real one and walks four per trip: batch sizes 4k, 4k+1, tiny)."""
    monkeypatch.setenv("SOMHIP_UPD_QW", "4")
    xdim, ydim, dim = 16, 16, 64
    x, _ = synth(97, 500, dim)
    ini = oracle.randinit(x, xdim, ydim, 11)
    length = 61 * 12
    oc, oi, od = oracle.som_train(ini, xdim, ydim, 3, 1, x, length, 0.04, 60.0, batch=batch)
    cb, ds = E.Codebook(eng, ini, 3, 1, xdim, ydim), E.Dataset(eng, x)
    eng.timing(True)
    eng.timing_reset()
    ti, td = E.som_train(cb, ds, length, 0.04, 60.0, batch=batch)
    ran = eng.timing_table()["k_som_update_bubble_s"][0]
    eng.timing(False)
    assert ran > 0
    assert np.array_equal(ti, oi) and np.array_equal(bits(td), bits(od))
    assert np.array_equal(bits(cb.download()), bits(oc))


# --------------------------------------------------------------------------- round 2: state flags, randinit pass
def test_prefilter_copies_follow_every_writer(eng, E, oracle):
    """The bf16 tiles / norms of the MFMA pre-filter are reused while the codebook is unchanged.  Every writer
    must invalidate them: MFMA scan (full split at S0) -> upload (S1) -> lvq_train on a codebook below 4096 rows
    (direct top-k: no split; only the corrected rows are re-split) -> MFMA scan.  A stale reuse filters with S0's
    tiles and can lose the exact winner silently."""
    n, d, m = 640, 48, 1500
    x, lab = synth(91, m, d, k=5)
    rs = np.random.RandomState(5)
    s0 = (x[rs.randint(0, m, n)] + 0.2 * rs.standard_normal((n, d))).astype(np.float32)
    s1 = (x[rs.randint(0, m, n)][::-1] * 1.7 - 3.0).astype(np.float32)       # far from s0
    clab = lab[rs.randint(0, m, n)].astype(np.int32)
    eng.set_scan_mode("mfma_bf16")
    cb = E.Codebook(eng, s0, labels=clab)
    ds = E.Dataset(eng, x, labels=lab)
    gi, _, _ = E.find_winners(cb, ds)
    assert np.array_equal(gi, oracle.winners(s0, x)[0])
    cb.upload(s1)
    want, _, _, _ = oracle.lvq_train(1, s1, clab, x, lab, 400, 0.05)
    E.lvq_train(cb, ds, E.LVQ1, 400, 0.05, trace=False)
    assert np.array_equal(bits(cb.download()), bits(want))
    gi, gd, _ = E.find_winners(cb, ds)
    oi, od, _ = oracle.winners(want, x)
    assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od))
    # and after SOM-style writers: online and mini-batch training on a map
    cb.close()
    ini = oracle.randinit(x, 32, 20, 3)
    cbm = E.Codebook(eng, ini, E.TOPOL_HEXA, E.NEIGH_BUBBLE, 32, 20)
    E.find_winners(cbm, ds)
    for batch in (1, 64):
        E.som_train(cbm, ds, 300, 0.05, 6.0, batch=batch, trace=False)
        now = cbm.download()
        gi, gd, _ = E.find_winners(cbm, ds)
        oi, od, _ = oracle.winners(now, x)
        assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od)), batch
    cbm.close()
    ds.close()


def test_column_minmax_and_randinit_from_device_bbox(eng, E, oracle):
    """randinit_codes' data pass on the device (som_rout.c:98-131) + the reference's LCG fill: equals the oracle's
    randinit on host rows, masked components and an all-negative column (FLT_MIN seed of the maximum) included"""
    rs = np.random.RandomState(17)
    x = (3.0 * rs.standard_normal((3000, 37)) - 1.0).astype(np.float32)
    x[:, 5] = -np.abs(x[:, 5]) - 0.5
    ds = E.Dataset(eng, x)
    lo, hi, cnt = E.column_minmax(ds)
    assert np.array_equal(bits(lo), bits(x.min(0))) and np.array_equal(bits(hi), bits(x.max(0))) and (cnt == 3000).all()
    got = E.randinit_from_bbox(lo, hi, cnt, 11, 7, 123)
    assert np.array_equal(bits(got), bits(oracle.randinit(x, 11, 7, 123)))
    mask = (rs.rand(3000, 37) < 0.2).astype(np.uint8)
    mask[:, 9] = 1
    dm = E.Dataset(eng, x, mask=mask)
    lo, hi, cnt = E.column_minmax(dm)
    xm = np.where(mask == 0, x, np.nan)
    assert cnt[9] == 0 and np.array_equal(cnt, (mask == 0).sum(0))
    ok = cnt > 0
    assert np.array_equal(bits(lo[ok]), bits(np.nanmin(xm[:, ok], 0).astype(np.float32)))
    assert np.array_equal(bits(hi[ok]), bits(np.nanmax(xm[:, ok], 0).astype(np.float32)))
    # the generated stream: device rows == host rows, so the box must agree too
    g = E.Dataset(eng, generate=(3456, 8, 24, 0, 5000))
    hx, _ = E.gen_rows(3456, 8, 24, 0, 5000)
    lo, hi, cnt = E.column_minmax(g)
    assert np.array_equal(bits(lo), bits(hx.min(0))) and np.array_equal(bits(hi), bits(hx.max(0)))
    for h in (ds, dm, g):
        h.close()


def test_c_host_may_destroy_the_engine_first(tmp_path):
    """include/somhip.h: "nothing here aborts".  A C host that destroys the engine before its codebook and data set
    (round 1 aborted with std::bad_variant_access out of hipStreamSynchronize on the freed stream) must run through;
    calls on the orphans fail with a message (tests/helpers/abi_order.c)."""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "abi_order")
    lib = os.path.join(ROOT, "som_lvq_pak_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "helpers", "abi_order.c"), "-L", lib, "-lsomhip", "-Wl,-rpath," + lib])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0 and "abi_order ok" in p.stdout, (p.returncode, p.stdout, p.stderr)


def test_python_mirrors_survive_engine_close_first(E):
    e2 = E.Engine(0)
    rows = np.arange(40, dtype=np.float32).reshape(10, 4)
    cb, ds = E.Codebook(e2, rows), E.Dataset(e2, rows)
    e2.lib.somhip_engine_destroy(e2.h)           # behind the wrapper's back: the C ABI itself must cope
    with pytest.raises(Exception, match="destroyed"):
        E.find_winners(cb, ds)
    e2.lib.somhip_codebook_destroy(cb.h)
    e2.lib.somhip_dataset_destroy(ds.h)
    cb.h = ds.h = e2.h = None
    e2._children = []


# --------------------------------------------------------------------------- LVQ over a row-sharded codebook
_LVQ_SHARDED_CHILD = r'''
import os, sys, ctypes as C
import numpy as np, torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda")        # torch's HIP runtime first, then the engine's
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import synth
from oracle import Oracle
from som_lvq_pak_amd import engine as E, sharded
from som_lvq_pak_amd._lib import LvqParams


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


class LocalLvqGroup:
    """S shards of one codebook held in ONE process: the shard object sharded.ShardedLvq drives, with the two
    collectives done by hand (stacking for the all-gather, an integer sum for the all-reduce) -- what N ranks do
    over RCCL, on one GPU, so that the real kernels of every step are exercised."""

    def __init__(self, eng, shards, kind, mk_params):
        self.parts = [sharded.GpuLvqShard(eng, cb, ds, mk_params, kind) for cb, ds in shards]

    def topk_keys(self, first, count):
        return torch.stack([p.topk_keys(first, count) for p in self.parts], dim=0)      # already "gathered"

    def merge(self, gathered, count):
        g = gathered[0] if gathered.dim() == 4 else gathered                            # allgather_tensor adds an axis
        outs = [p.merge(g, count) for p in self.parts]
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        return outs[0]

    def candidates(self, keys, count, xrows):
        got = [p.candidates(keys, count, xrows) for p in self.parts]
        lab = sum(g[0].view(torch.int32) for g in got)
        ta = None if got[0][1] is None else sum(g[1].view(torch.int32) for g in got).view(torch.float32)
        rows = sum(g[2].view(torch.int32) for g in got).view(torch.float32)
        return lab, ta, rows

    def apply(self, it0, count, first, keys, lab, ta, rows, xrows):
        res = [p.apply(it0, count, first, keys, lab, ta, rows, xrows) for p in self.parts]
        assert all(r[0] == res[0][0] for r in res)                                      # every rank consumed the same
        assert all(np.array_equal(r[1], res[0][1]) and np.array_equal(bits(r[2]), bits(res[0][2])) for r in res)
        return res[0]

    def collective_scope(self):
        import contextlib
        return contextlib.nullcontext()


orc, eng = Oracle(), E.Engine(0)
ok = True
for kind in (1, 2, 3, 4):
    n, d, m, length = 900, 50, 1500, 2500
    x, lab = synth(300 + kind, m, d, k=12, spread=2.5)
    rs = np.random.RandomState(kind)
    pick = rs.randint(0, m, n)
    codes = (x[pick] + 0.05 * rs.randn(n, d)).astype(np.float32)
    clab = lab[pick].copy()
    kw = {"winlen": 0.3} if kind >= 3 else {}
    if kind == 4:
        kw["epsilon"] = 0.15
    oc, ol, oi, od = orc.lvq_train(kind, codes, clab, x, lab, length, 0.06, **kw)
    cuts = [0, 260, 700, n]
    for xrows in (2, 8):
        ds = E.Dataset(eng, x, labels=lab)
        shards = []
        for a, b in zip(cuts, cuts[1:]):
            cb = E.Codebook(eng, codes[a:b], labels=clab[a:b], row_offset=a, n_global=n)
            if kind == 2:
                ta0 = np.full(b - a, 0.06, dtype=np.float32)
                E.check(eng.lib.somhip_lvq_rates_upload(cb.h, ta0.ctypes.data_as(C.POINTER(C.c_float))))
            shards.append((cb, ds))
        mk = lambda: LvqParams(kind, length, 0.06, 1, kw.get("winlen", 0.0), kw.get("epsilon", 0.0), 0, 0, 0)
        lv = sharded.ShardedLvq(LocalLvqGroup(eng, shards, kind, mk), kind, m, xrows=xrows, max_batch=512)
        ti, td = lv.train(length)
        got = np.concatenate([cb.download() for cb, _ in shards])
        good = np.array_equal(ti, oi) and np.array_equal(bits(td), bits(od)) and np.array_equal(bits(got), bits(oc))
        if kind == 2:
            tal = []
            for cb, _ in shards:
                t = np.empty(cb.n, dtype=np.float32)
                E.check(eng.lib.somhip_lvq_rates_download(cb.h, t.ctypes.data_as(C.POINTER(C.c_float))))
                tal.append(t)
            good = good and np.array_equal(bits(np.concatenate(tal)), bits(ol))
        print("kind", kind, "xrows", xrows, "batches", lv.batches, "ok", good)
        ok = ok and good
        for cb, _ in shards:
            cb.close()
        ds.close()
print("RESULT", ok)
'''


def test_lvq_row_sharded_equals_online():
    """lvq1 / olvq1 / lvq2 / lvq3 over a codebook cut into three uneven row shards: per-shard top-8 -> merge ->
    candidate rows by integer all-reduce -> replicated walk -> owners commit.  Winners, distances, codebook and
    OLVQ1 rates must be the unsharded online result, bit for bit; with 2 exchanged rows (batches end early when a
    deeper candidate wins) as with all 8.  In a child process: torch's HIP runtime has to come up before the engine's."""
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, "-c", _LVQ_SHARDED_CHILD % (ROOT, ROOT)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert "RESULT True" in p.stdout, (p.stdout[-3000:], p.stderr[-3000:])


# --------------------------------------------------------------------------- update mode "gemm"
@pytest.mark.parametrize("xd,yd,d,n,B,radius,alpha", [(64, 64, 128, 4096, 2048, 30.0, 0.05), (24, 40, 256, 3000, 1500, 9.0, 0.03),
                                                       (16, 16, 512, 4096, 4096, 30.0, 0.05), (48, 32, 384, 2048, 1024, 2.0, 0.02)])
def test_gemm_update_mode_equals_exact_within_rounding(eng, E, oracle, xd, yd, d, n, B, radius, alpha):
    """SOMHIP_UPDATE_GEMM evaluates the batch's updates of every unit as one affine map c' = P c + sum w_j x_j on the fp32
    matrix pipe (kernels/som_update_gemm.hpp).  Same winners (they are found before the update), and a codebook that
    differs from the exact kernels' -- which equal the batch oracle bit for bit -- only by the rounding of a sum instead
    of a chain: a few fp32 ulps of the data's scale.  The 16x16 case puts all 4096 samples into every unit's list: the
    walk stops where the weights have decayed below 2^-24 (~330 hits at alpha 0.05) and must still agree."""
    ds = E.Dataset(eng, generate=(11, 16, d, 0, n))
    lo, hi, cnt = E.column_minmax(ds)
    init = E.randinit_from_bbox(lo, hi, cnt, xd, yd, 5)
    x = ds.rows(0, n)
    # the 16x16 case runs the first batch of a LONG schedule (rate and radius stay at their initial values inside the
    # batch, as in a real run); the oracle replays whole schedules only, so there the exact kernels are the yardstick
    length = 64 * B if (xd, yd) == (16, 16) else B
    want, wi = None, None
    if length == B:
        want, wi, _ = oracle.som_train(init, xd, yd, 3, 1, x, B, alpha, radius, batch=B)
    got = {}
    for mode in ("exact", "gemm", "gemm_full_lists"):
        eng.set_update_mode(mode.split("_")[0])
        if mode == "gemm_full_lists":
            os.environ["SOMHIP_GEMM_FULL_LISTS"] = "1"       # K4b makes every list whole instead of the tail the walk can reach
        try:
            cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xd, yd)
            s0 = eng.scan_stats()
            ti, _ = E.som_train(cb, ds, length, alpha, radius, batch=B, count=B)
            s1 = eng.scan_stats()
            got[mode] = (cb.download(), ti, s1["gemm_entries"] - s0["gemm_entries"], s1["group_updates"] - s0["group_updates"])
            cb.close()
        finally:
            eng.set_update_mode("exact")
            os.environ.pop("SOMHIP_GEMM_FULL_LISTS", None)
    # the tail of a list is all the backward walk ever reaches: same bits as with whole lists, same entries walked
    assert np.array_equal(bits(got["gemm"][0]), bits(got["gemm_full_lists"][0]))
    assert got["gemm"][2] == got["gemm_full_lists"][2] and got["gemm"][3] <= got["gemm_full_lists"][3]
    if want is None:
        want, wi = got["exact"][0], got["exact"][1]
    assert np.array_equal(got["exact"][1], wi) and np.array_equal(got["gemm"][1], wi)
    assert np.array_equal(bits(got["exact"][0]), bits(want))
    assert got["exact"][2] == 0 and got["gemm"][2] > 0                       # the matrix-pipe kernel really ran
    scale = float(np.abs(want).max())
    assert float(np.abs(got["gemm"][0] - want).max()) <= 8e-6 * scale
    if (xd, yd) == (16, 16):
        assert got["gemm"][2] < 0.3 * got["gemm_full_lists"][3]               # most of every 4096-entry list skipped
    ds.close()


def test_gemm_update_makes_only_the_reachable_tail_of_a_list(eng, E):
    """Update mode gemm on a map with many row groups, a big radius and a long schedule (rate ~ constant inside the batch):
    K4b makes only the tail of every list that the backward walk can reach before every unit's decay is below the cut --
    far fewer entries than the whole lists, the same entries walked, the same bits."""
    xd, yd, d, B = 256, 128, 128, 4096
    ds = E.Dataset(eng, generate=(21, 16, d, 0, B))
    lo, hi, cnt = E.column_minmax(ds)
    init = E.randinit_from_bbox(lo, hi, cnt, xd, yd, 9)
    got = {}
    eng.set_update_mode("gemm")
    try:
        for mode in ("tail", "full"):
            if mode == "full":
                os.environ["SOMHIP_GEMM_FULL_LISTS"] = "1"
            try:
                cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xd, yd)
                s0 = eng.scan_stats()
                E.som_train(cb, ds, 64 * B, 0.05, 64.0, batch=B, count=B, trace=False)
                s1 = eng.scan_stats()
                got[mode] = (cb.download(), s1["gemm_entries"] - s0["gemm_entries"], s1["group_updates"] - s0["group_updates"])
                cb.close()
            finally:
                os.environ.pop("SOMHIP_GEMM_FULL_LISTS", None)
    finally:
        eng.set_update_mode("exact")
    assert np.array_equal(bits(got["tail"][0]), bits(got["full"][0]))
    assert got["tail"][1] == got["full"][1] > 0
    assert got["tail"][2] < 0.6 * got["full"][2]
    ds.close()


@pytest.mark.parametrize("xd,yd,d,B,radius,alpha,topol", [(32, 24, 128, 1024, 6.0, 0.05, 3), (40, 16, 256, 600, 2.5, 0.03, 4),
                                                           (16, 16, 512, 2048, 12.0, 0.05, 3)])
def test_gemm_update_mode_gaussian(eng, E, oracle, xd, yd, d, B, radius, alpha, topol):
    """gaussian neighbourhoods in update mode gemm: every unit's rate alpha exp(-lattice_sq / (2 radius^2)) per sample as a dense
    weight matrix on the fp32 matrix pipe, against the exact kernels (= the batch oracle, bit for bit): same winners, the
    codebook within fp32 rounding of a sum of B products (the rates themselves go through v_exp_f32 instead of a double exp)."""
    ds = E.Dataset(eng, generate=(13, 12, d, 0, B))
    lo, hi, cnt = E.column_minmax(ds)
    init = E.randinit_from_bbox(lo, hi, cnt, xd, yd, 3)
    x = ds.rows(0, B)
    want, wi, _ = oracle.som_train(init, xd, yd, topol, 2, x, 8 * B, alpha, radius, batch=B) if False else (None, None, None)
    got = {}
    for mode in ("exact", "gemm"):
        eng.set_update_mode(mode)
        try:
            cb = E.Codebook(eng, init, topol, E.NEIGH_GAUSSIAN, xd, yd)
            s0 = eng.scan_stats()
            ti, _ = E.som_train(cb, ds, 8 * B, alpha, radius, batch=B, count=B)
            s1 = eng.scan_stats()
            got[mode] = (cb.download(), ti, s1["gemm_entries"] - s0["gemm_entries"])
            cb.close()
        finally:
            eng.set_update_mode("exact")
    assert np.array_equal(got["exact"][1], got["gemm"][1])
    assert got["exact"][2] == 0 and got["gemm"][2] > 0                       # the matrix-pipe kernel really ran
    scale = float(np.abs(got["exact"][0]).max())
    assert float(np.abs(got["gemm"][0] - got["exact"][0]).max()) <= 2e-5 * scale
    ds.close()


def test_gemm_update_mode_falls_back_and_shards_identically(eng, E, oracle):
    """gemm mode is taken only where it applies (no masks, dim % 128 == 0): dim-48 maps (bubble and gaussian) must give the
    exact kernels' bits; interleaved shards in gemm mode -- bubble and gaussian -- give the unsharded gemm bits."""
    from som_lvq_pak_amd._lib import SomParams
    eng.set_update_mode("gemm")
    try:
        x, _ = synth(7, 900, 48)
        ini = oracle.randinit(x, 16, 8, 3)
        for neigh in (1, 2):
            want, _, _ = oracle.som_train(ini, 16, 8, 3, neigh, x, 600, 0.05, 5.0, batch=100, trace=False)
            cb, ds = E.Codebook(eng, ini, 3, neigh, 16, 8), E.Dataset(eng, x)
            E.som_train(cb, ds, 600, 0.05, 5.0, batch=100, trace=False)
            assert np.array_equal(bits(cb.download()), bits(want)), neigh
            cb.close(); ds.close()
        # sharded == unsharded, gemm kernels on both sides (bubble, and gaussian: the dense-weight form)
        d, n, B, xd, yd = 128, 2048, 512, 32, 24
        ds = E.Dataset(eng, generate=(5, 8, d, 0, n))
        lo, hi, cnt = E.column_minmax(ds)
        ini = E.randinit_from_bbox(lo, hi, cnt, xd, yd, 9)
        for neigh in (1, 2):
            cb = E.Codebook(eng, ini, 3, neigh, xd, yd)
            s0 = eng.scan_stats()
            E.som_train(cb, ds, 2 * B, 0.05, 10.0, batch=B, trace=False)
            assert eng.scan_stats()["gemm_entries"] > s0["gemm_entries"], neigh
            whole = cb.download()
            shards = []
            for r in range(3):
                units = E.shard_units(xd, yd, r, 3, eng.lib)
                shards.append((units, E.Codebook(eng, ini[units], 3, neigh, xd, yd, interleave=(r, 3))))
            kb = eng.device_alloc(8 * B)
            p = SomParams(2 * B, 0.05, 10.0, 1, 0, 0, B, 0, 2 * B, 0)
            for it0 in range(0, 2 * B, B):
                ks = []
                for _, s in shards:
                    E.check(eng.lib.somhip_batch_winner_keys(s.h, ds.h, it0, B, kb))
                    k = np.empty(B, dtype=np.uint64)
                    E.check(eng.lib.somhip_copy_to_host(eng.h, k.ctypes.data_as(C.c_void_p), kb, 8 * B))
                    ks.append(k)
                merged = np.minimum(np.minimum(ks[0], ks[1]), ks[2])
                E.check(eng.lib.somhip_copy_to_device(eng.h, kb, merged.ctypes.data_as(C.c_void_p), 8 * B))
                for _, s in shards:
                    E.check(eng.lib.somhip_som_batch_update(s.h, ds.h, C.byref(p), it0, B, it0, kb))
            eng.sync()
            eng.device_free(kb)
            full = np.empty_like(whole)
            for units, s in shards:
                full[units] = s.download()
            assert np.array_equal(bits(full), bits(whole)), neigh
    finally:
        eng.set_update_mode("exact")
