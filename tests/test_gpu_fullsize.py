"""Full BASELINE-size checks (C4: 256x256 map, dim 512 -- far beyond what the CPU oracle can
replay) through size-independent properties of the path:
  * the three winner-search implementations (direct scan, fp32-MFMA and split-bf16 pre-filter +
    exact re-rank) return identical keys;
  * a sample that IS a code row wins that row (or an identical earlier one) at distance exactly 0;
  * row-sharded == unsharded, bit for bit;
  * alpha = 0 leaves every bit of the codebook unchanged;
  * the online engine and the mini-batch engine with batch 1-sized runs agree.
A reduced-size replay against the oracle sits beside them."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

XDIM = YDIM = 256
DIM = 512
B = 4096


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def world():
    from som_lvq_pak_amd import engine as E
    eng = E.Engine(0)
    rs = np.random.RandomState(2024)
    cent = (4.0 * rs.standard_normal((64, DIM))).astype(np.float32)
    x = (cent[rs.randint(0, 64, 3 * B)] + rs.standard_normal((3 * B, DIM)).astype(np.float32)).astype(np.float32)
    lo, hi = x.min(0), x.max(0)
    init = (lo + (hi - lo) * rs.rand(XDIM * YDIM, DIM)).astype(np.float32)
    init[1000] = init[77]                     # an exact duplicate row: ties must go to the lower index
    ds = E.Dataset(eng, x)
    yield E, eng, x, init, ds
    eng.close()


def _keys(E, eng, cb, ds, first, count):
    buf = eng.device_alloc(8 * count)
    out = np.empty(count, dtype=np.uint64)
    E.check(eng.lib.somhip_batch_winner_keys(cb.h, ds.h, first, count, buf))
    E.check(eng.lib.somhip_copy_to_host(eng.h, out.ctypes.data_as(C.c_void_p), buf, 8 * count))
    eng.device_free(buf)
    return out


def test_c4_scan_modes_agree_and_exact_hits(world):
    E, eng, x, init, ds = world
    cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, XDIM, YDIM)
    keys = {}
    for mode in ("direct", "mfma", "mfma_bf16"):
        eng.set_scan_mode(mode)
        keys[mode] = _keys(E, eng, cb, ds, 0, B)
    eng.set_scan_mode("mfma_bf16")
    assert np.array_equal(keys["direct"], keys["mfma"])
    assert np.array_equal(keys["direct"], keys["mfma_bf16"])
    # samples equal to code rows: distance exactly 0, lowest identical row wins
    probe = np.array([77, 1000, 5, 65535, 31337])
    ds2 = E.Dataset(eng, np.concatenate([init[probe], x[:59]], axis=0))
    idx, diff, _ = E.find_winners(cb, ds2)
    assert list(idx[:5, 0]) == [77, 77, 5, 65535, 31337]
    assert (bits(diff[:5, 0]) == 0).all()


def test_c4_benched_mode_at_the_benched_shape():
    """What bench.py times, at the size it times it (missing 5): 256 x 256 x 512, one 32768-vector batch of the
    generator stream through the persistent level-1 ring kernel, level 2, the exact re-rank from the row-major copy and the
    GEMM-form update.  Winners: the keys of the direct fp32 scan, bit for bit.  Update: the exact kernels' codebook within
    8e-6 of the data's scale (fp32 rounding of a sum instead of a chain), at the head of the schedule (radius 128: every
    unit is hit about 29 000 times in the batch) and near its end (radius 7)."""
    from som_lvq_pak_amd import engine as E
    eng = E.Engine(0)
    BIG, L = 32768, 10_000_000
    ds = E.Dataset(eng, generate=(3456, 256, DIM, 0, 2 * BIG))
    lo, hi, cnt = E.column_minmax(ds)
    init = E.randinit_from_bbox(lo, hi, cnt, XDIM, YDIM, 7)
    scale = float(max(abs(lo).max(), abs(hi).max()))
    cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, XDIM, YDIM)
    try:
        # a map that has seen one batch (so that the winners are spread over the map), then the batch under test
        eng.set_update_mode("gemm")
        E.som_train(cb, ds, L, 0.05, 128.0, batch=BIG, start_iter=0, count=BIG, trace=False)
        start = cb.download()
        eng.set_scan_mode("direct")
        kd = _keys(E, eng, cb, ds, BIG, BIG)
        eng.set_scan_mode("mfma_bf16")
        km = _keys(E, eng, cb, ds, BIG, BIG)
        assert np.array_equal(kd, km)
        for it0 in (BIG, 9_500_000 // BIG * BIG):
            res = {}
            for mode in ("gemm", "exact"):
                eng.set_update_mode(mode)
                cb.upload(start)
                ti, td = E.som_train(cb, ds, L, 0.05, 128.0, batch=BIG, start_iter=it0, count=BIG, data_first=BIG)
                res[mode] = (ti, td, cb.download())
            assert np.array_equal(res["gemm"][0], res["exact"][0]) and np.array_equal(bits(res["gemm"][1]), bits(res["exact"][1]))
            assert np.array_equal(res["gemm"][0], (km & np.uint64(0xFFFFFFFF)).astype(np.int64))   # the traced winners are the keys' rows
            err = float(np.abs(res["gemm"][2] - res["exact"][2]).max())
            assert err <= 8e-6 * scale, (it0, err, scale)
            assert not np.array_equal(res["exact"][2], start)
    finally:
        eng.set_update_mode("exact")
        eng.close()


def test_c4_sharded_equals_unsharded_and_alpha_zero(world):
    from som_lvq_pak_amd._lib import SomParams
    E, eng, x, init, ds = world
    n = XDIM * YDIM
    length = 2 * B
    # unsharded, two mini-batches
    cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, XDIM, YDIM)
    ti, _ = E.som_train(cb, ds, length, 0.05, 100.0, batch=B)
    whole = cb.download()
    assert not np.array_equal(bits(whole), bits(init))
    # three uneven shards, explicit key merge
    cuts = [0, 20000, 20000 + 64 * 300 + 17, n]
    shards = [E.Codebook(eng, init[cuts[i]:cuts[i + 1]], E.TOPOL_HEXA, E.NEIGH_BUBBLE, XDIM, YDIM,
                         row_offset=cuts[i], n_global=n) for i in range(3)]
    kb = eng.device_alloc(8 * B)
    p = SomParams(length, 0.05, 100.0, 1, 0, 0, B, 0, length, 0)
    got_idx = []
    for it0 in range(0, length, B):
        ks = [_keys(E, eng, s, ds, it0, B) for s in shards]
        merged = np.minimum(np.minimum(ks[0], ks[1]), ks[2])
        got_idx.append((merged & np.uint64(0xFFFFFFFF)).astype(np.int32))
        E.check(eng.lib.somhip_copy_to_device(eng.h, kb, merged.ctypes.data_as(C.c_void_p), 8 * B))
        for s in shards:
            E.check(eng.lib.somhip_som_batch_update(s.h, ds.h, C.byref(p), it0, B, it0, kb))
    eng.sync()
    eng.device_free(kb)
    assert np.array_equal(np.concatenate(got_idx), ti)
    parts = np.concatenate([s.download() for s in shards], axis=0)
    assert np.array_equal(bits(parts), bits(whole))
    # alpha = 0: c + 0*(x-c) == c for every element, whatever the neighbourhoods
    cb0 = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, XDIM, YDIM)
    E.som_train(cb0, ds, B, 0.0, 100.0, batch=B, trace=False)
    assert np.array_equal(bits(cb0.download()), bits(init))


def test_c4_online_equals_minibatch_of_one_and_reduced_oracle(world, oracle):
    E, eng, x, init, ds = world
    # online engine vs mini-batch machinery driven one sample per run (batch=1 is routed to the
    # online kernels, so drive the two-phase primitives directly)
    from som_lvq_pak_amd._lib import SomParams
    steps = 24
    cb1 = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, XDIM, YDIM)
    t1, d1 = E.som_train(cb1, ds, steps, 0.05, 90.0, batch=1)
    cb2 = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, XDIM, YDIM)
    kb = eng.device_alloc(8)
    p = SomParams(steps, 0.05, 90.0, 1, 0, 0, 1, 0, steps, 0)
    t2 = []
    for it in range(steps):
        E.check(eng.lib.somhip_batch_winner_keys(cb2.h, ds.h, it, 1, kb))
        k = np.empty(1, dtype=np.uint64)
        E.check(eng.lib.somhip_copy_to_host(eng.h, k.ctypes.data_as(C.c_void_p), kb, 8))
        t2.append(int(k[0] & np.uint64(0xFFFFFFFF)))
        E.check(eng.lib.somhip_som_batch_update(cb2.h, ds.h, C.byref(p), it, 1, it, kb))
    eng.device_free(kb)
    assert list(t1) == t2
    assert np.array_equal(bits(cb1.download()), bits(cb2.download()))
    # reduced replay against the oracle: same dim, 48x32 map, 600 iterations, batch 128
    rs = np.random.RandomState(7)
    small = (x[:4000].min(0) + (x[:4000].max(0) - x[:4000].min(0)) * rs.rand(48 * 32, DIM)).astype(np.float32)
    oc, oi, od = oracle.som_train(small, 48, 32, 3, 1, x[:600], 600, 0.05, 20.0, batch=128)
    cbs, dss = E.Codebook(eng, small, 3, 1, 48, 32), E.Dataset(eng, x[:600])
    ti, td = E.som_train(cbs, dss, 600, 0.05, 20.0, batch=128)
    assert np.array_equal(ti, oi) and np.array_equal(bits(td), bits(od))
    assert np.array_equal(bits(cbs.download()), bits(oc))


@pytest.mark.parametrize("exchange", [False, True])
def test_rccl_stream_ordered_step_single_rank(oracle, exchange):
    """The production multi-GPU step (scan -> RCCL all-reduce on the engine's own stream via
    torch.cuda.ExternalStream -> update, no host sync) exercised with a 1-rank NCCL group in a
    child process: must equal the plain mini-batch run.  exchange: the step of 8 ranks and more -- the pre-filter's
    bounds through two float MIN all-reduces inside the winner search (SOMHIP_SHARD_EXCHANGE=force: also with one rank)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import os, sys, ctypes as C
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import synth
from oracle import Oracle
from som_lvq_pak_amd import engine as E, sharded
from som_lvq_pak_amd._lib import SomParams
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
orc = Oracle()
x, _ = synth(5, 900, %d)
ini = orc.randinit(x, 16, 16, 3)
want, wi, _ = orc.som_train(ini, 16, 16, 3, 1, x, 1536, 0.05, 6.0, batch=256)
eng = E.Engine(0)
ds = E.Dataset(eng, x)
cb = E.Codebook(eng, ini, 3, 1, 16, 16)
sh = sharded.GpuShard(eng, cb, ds, lambda: SomParams(1536, 0.05, 6.0, 1, 0, 0, 256, 0, 0, 0), 256)
som = sharded.ShardedSom(sh, 256, 900)
winners = som.train(1536)
eng.sync(); torch.cuda.synchronize()
assert bool(som._exch.get(256)) == %r, som._exch
idx = np.concatenate([sharded.unpack_keys(w.cpu().numpy())[1] for w in winners])
ok = np.array_equal(idx, wi) and np.array_equal(cb.download().view(np.uint32), want.view(np.uint32))
dist.destroy_process_group()
print("RESULT", ok)
''' % (ROOT, ROOT, 32 if exchange else 24, exchange)
    env = dict(os.environ)
    env.pop("SOMHIP_SHARD_EXCHANGE", None)
    if exchange:
        env["SOMHIP_SHARD_EXCHANGE"] = "force"
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert "RESULT True" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


# ------------------------------------------------------------------ BASELINE configs[2] and configs[4] at full size
def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_c3_olvq1_full_size_against_reference_prefix():
    """BASELINE.json configs[2]: OLVQ1, 10 000 codes x 256, 1 M labelled vectors of gen:k=100,dim=256,seed=2345, alpha0 0.3.
    The REAL reference's olvq1_training ran the first 20 000 vectors here (tests/golden/make_golden.py --c3; OLVQ1 has no
    global schedule, so that is the state of the full run after 20 000 iterations): winners, distances, codebook and the
    "%g" text of the rates must match its hashes; then the run goes on to the full 1 M vectors."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    from som_lvq_pak_amd import engine as E
    g = json.load(open(os.path.join(GOLDEN, "cli", "expected.json")))["c3_prefix"]
    eng = E.Engine(0)
    n = 1000000
    ds = E.Dataset(eng, generate=(g["seed"], g["k"], g["dim"], 0, n))
    head = ds.rows(0, g["head"])
    cen = ds.centres
    pick = np.concatenate([np.where(cen[:g["head"]] == c)[0][:g["codes_per_class"]] for c in range(g["k"])])
    codes, clab = head[pick].copy(), cen[pick].astype(np.int32)
    assert _sha(codes) == g["init_sha256"]                    # device stream == host stream, same picking rule
    cb = E.Codebook(eng, codes, labels=clab)
    tal, ti, td = E.lvq_train(cb, ds, E.OLVQ1, n, g["alpha"], count=g["prefix"])
    assert _sha(ti.astype(np.int32)) == g["winners_sha256"]
    assert _sha(td.astype(np.float32)) == g["diffs_sha256"]
    assert _sha(cb.download()) == g["codes_sha256"]
    assert hashlib.sha256(" ".join("%g" % v for v in tal).encode()).hexdigest() == g["rates_g_sha256"]
    before = eng.lvq_stats()
    tal, _, _ = E.lvq_train(cb, ds, E.OLVQ1, n, g["alpha"], start_iter=g["prefix"], talpha=tal, trace=False)
    after = eng.lvq_stats()
    assert after["samples"] - before["samples"] == n - g["prefix"]
    assert (after["samples"] - before["samples"]) / (after["batches"] - before["batches"]) > 900     # ~1024 per rescan
    dse = E.Dataset(eng, head[:20000])
    wi, _, _ = E.find_winners(cb, dse)
    assert (clab[wi[:, 0]] == cen[:20000]).mean() > 0.999
    assert np.isfinite(tal).all() and (tal > 0).all() and (tal <= np.float32(g["alpha"])).all()
    eng.close()


def test_c5_shape_lvq3_batched_equals_online_kernel():
    """BASELINE.json configs[4] shape on one GPU: LVQ3 (alpha 0.05, win 0.3, eps 0.1) on a 100 000 x 1024 codebook, stream
    gen:k=1000,dim=1024,seed=4567.  No CPU can replay this; the exact batched engine (components walked side by side)
    and the one-launch-per-iteration kernel are two independent implementations of the reference's online loop and must
    give the same winners, distances and codebook bits.  (Reduced shapes against the oracle: test_gpu_parity.py.)"""
    import os
    from som_lvq_pak_amd import engine as E
    eng = E.Engine(0)
    ncodes, d, iters = 100000, 1024, 3000
    ds = E.Dataset(eng, generate=(4567, 1000, d, 0, ncodes + 60000))
    codes = ds.rows(0, ncodes)
    clab = ds.centres[:ncodes].astype(np.int32)
    res = {}
    for mode in ("batched", "online"):
        if mode == "online":
            os.environ["SOMHIP_LVQ_ONLINE"] = "1"
        try:
            cb = E.Codebook(eng, codes, labels=clab)
            s0 = eng.lvq_stats()
            _, ti, td = E.lvq_train(cb, ds, E.LVQ3, 100000000, 0.05, winlen=0.3, epsilon=0.1, start_iter=0, count=iters,
                                    data_first=ncodes)
            s1 = eng.lvq_stats()
            res[mode] = (ti, td, cb.download(), s1["components"] - s0["components"], s1["batches"] - s0["batches"])
            cb.close()
        finally:
            os.environ.pop("SOMHIP_LVQ_ONLINE", None)
    assert np.array_equal(res["batched"][0], res["online"][0])
    assert np.array_equal(bits(res["batched"][1]), bits(res["online"][1]))
    assert np.array_equal(bits(res["batched"][2]), bits(res["online"][2]))
    assert not np.array_equal(bits(res["batched"][2]), bits(codes))          # something was learnt
    assert res["batched"][3] > 50 * res["batched"][4]                        # hundreds of independent components per batch
    assert res["online"][4] == 0
    eng.close()
