"""The CPU oracle against the LIVE reference (oracle/_ref/libref_harness.so = the
unmodified reference objects behind oracle/ref_harness.c) on seeded random cases.
Skipped where oracle/_ref is not built (it needs /root/reference; the GPU box has the
prebuilt files).  Bit-exact."""
import numpy as np
import pytest

from conftest import synth


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("seed", range(6))
def test_som_random_cases(oracle, ref, seed):
    rs = np.random.RandomState(100 + seed)
    d = int(rs.choice([3, 8, 17, 64]))
    xdim, ydim = int(rs.randint(2, 12)), int(rs.randint(2, 9))
    n = int(rs.randint(50, 400))
    x, _ = synth(200 + seed, n, d)
    topol = int(rs.choice([3, 4]))
    neigh = int(rs.choice([1, 2]))
    at = int(rs.choice([1, 2]))
    length = int(rs.randint(n // 2, 3 * n))
    radius = float(rs.uniform(1.0, max(xdim, ydim)))
    alpha = float(rs.uniform(0.01, 0.5))
    ini = ref.randinit(x, xdim, ydim, 7 + seed)
    assert np.array_equal(_bits(oracle.randinit(x, xdim, ydim, 7 + seed)), _bits(ini))
    rc, ri, rd = ref.som_train(ini, xdim, ydim, topol, neigh, x, length, alpha, radius, alpha_type=at)
    oc, oi, od = oracle.som_train(ini, xdim, ydim, topol, neigh, x, length, alpha, radius, alpha_type=at)
    assert np.array_equal(oi, ri)
    assert np.array_equal(_bits(od), _bits(rd))
    assert np.array_equal(_bits(oc), _bits(rc))
    rq, _, _ = ref.find_qerror(rc, x)
    oq, _, _ = oracle.find_qerror(oc, x)
    assert np.float32(rq) == np.float32(oq)
    assert np.float32(ref.find_qerror2(rc, xdim, topol, neigh, x, 2.5)) == \
        np.float32(oracle.find_qerror2(oc, xdim, topol, neigh, x, 2.5))


@pytest.mark.parametrize("kind", [1, 2, 3, 4])
@pytest.mark.parametrize("seed", range(3))
def test_lvq_random_cases(oracle, ref, kind, seed):
    rs = np.random.RandomState(300 + seed)
    d = int(rs.choice([4, 20, 33]))
    n = int(rs.randint(100, 500))
    x, lab = synth(400 + seed, n, d, k=5, spread=2.0)
    ncodes = int(rs.randint(10, 40))
    pick = rs.choice(n, ncodes, replace=False)
    codes, clab = x[pick].copy(), lab[pick].copy()
    length = int(rs.randint(n, 4 * n))
    kw = {}
    if kind >= 3:
        kw["winlen"] = float(rs.uniform(0.1, 0.5))
    if kind == 4:
        kw["epsilon"] = float(rs.uniform(0.05, 0.5))
    at = int(rs.choice([1, 2]))
    rc, rl, ri, rd = ref.lvq_train(kind, codes, clab, x, lab, length, 0.1, alpha_type=at, **kw)
    oc, ol, oi, od = oracle.lvq_train(kind, codes, clab, x, lab, length, 0.1, alpha_type=at, **kw)
    assert np.array_equal(oi, ri)
    assert np.array_equal(_bits(od), _bits(rd))
    assert np.array_equal(_bits(oc), _bits(rc))
    if kind == 2:
        assert ["%g" % float(a) for a in ol] == rl


def test_masked_winners_and_dist(oracle, ref):
    rs = np.random.RandomState(5)
    x, _ = synth(6, 120, 9)
    codes = x[:25] + 0.1
    mask = (rs.rand(120, 9) < 0.3).astype(np.uint8)
    mask[3] = 1
    for knn, use in ((1, False), (1, True), (3, True)):
        ri, rd, rr = ref.winners(codes, x, knn=knn, use_knn_fn=use, mask=mask)
        oi, od, orr = oracle.winners(codes, x, knn=knn, use_knn_fn=use, mask=mask)
        assert np.array_equal(ri, oi) and np.array_equal(_bits(rd), _bits(od)) and np.array_equal(rr, orr)
    assert rr[3] == 0
    for r in range(0, 120, 7):
        a, b = x[r], codes[r % 25]
        assert ref.vector_dist(a, b, mask[r], None) == oracle.vector_dist(a, b, mask[r], None)
        assert ref.vector_dist(a, b, mask[r], mask[(r + 1) % 120]) == \
            oracle.vector_dist(a, b, mask[r], mask[(r + 1) % 120])
        assert np.array_equal(_bits(ref.adapt_vector(b, a, -0.3, mask[r])),
                              _bits(oracle.adapt_vector(b, a, -0.3, mask[r])))


def test_batch_schedule_is_reference_at_b1_only(oracle, ref):
    """batch=1 IS the reference; batch>1 (the HIP throughput mode's schedule) is a
    different, separately specified algorithm -- show that it really differs."""
    x, _ = synth(9, 500, 12)
    ini = ref.randinit(x, 8, 6, 3)
    rc, ri, _ = ref.som_train(ini, 8, 6, 3, 1, x, 1000, 0.05, 4.0)
    o1, i1, _ = oracle.som_train(ini, 8, 6, 3, 1, x, 1000, 0.05, 4.0, batch=1)
    o64, i64, _ = oracle.som_train(ini, 8, 6, 3, 1, x, 1000, 0.05, 4.0, batch=64)
    assert np.array_equal(_bits(o1), _bits(rc)) and np.array_equal(i1, ri)
    assert not np.array_equal(i64, ri)
    assert np.array_equal(i64[:64], ri[:1].repeat(1).tolist() + i64[1:64].tolist())  # first winner equal
    q1, _, _ = oracle.find_qerror(o1, x)
    q64, _, _ = oracle.find_qerror(o64, x)
    assert abs(q1 - q64) / q1 < 0.05
