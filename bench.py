#!/usr/bin/env python3
"""bench.py -- training vectors/sec of the SOM hot path on MI355X.

Workload (BASELINE.json configs[3], the configuration the metric is quoted on; it fits one
GPU): `vsom` on a 256x256 hexagonal map with bubble neighbourhood, dim = 512 (codebook
65 536 x 512 fp32 = 128 MiB), alpha 0.05 linear, radius 128 -> 1, synthetic 256-component
Gaussian mixture generated on the device (no data sets / no network here).

A "step" is one mini-batch of --batch vectors through the whole hot path: exact
best-matching-unit search for every vector + the in-order neighbourhood update.  The K
timed steps form one complete training run of K*batch vectors (the radius sweeps its whole
range inside the timed region).  With --batch 1 the engine runs the reference's strictly
online algorithm instead (bit-exact with the CPU reference, much slower).

N > 1 (one process per GPU, torch.distributed/RCCL): the codebook is row-sharded, every
rank scans its shard for the same batch, one all-reduce(MIN) of packed (distance, index)
keys gives the global winners, each rank updates its own rows.  Total work is fixed as N
grows -> "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3     # MI355X fp32 matrix == fp32 vector FMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--xdim", type=int, default=256)
    ap.add_argument("--ydim", type=int, default=256)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--alpha", type=float, default=0.05)
    ap.add_argument("--radius", type=float, default=None)
    ap.add_argument("--cpu-vectors", type=int, default=400,
                    help="vectors the CPU reference trains on for cpu_baseline (0 = skip)")
    ap.add_argument("--eval-vectors", type=int, default=8192)
    ap.add_argument("--scan", default="auto", choices=["auto", "direct", "mfma", "mfma_bf16"],
                    help="winner-search implementation (all bit-identical); auto = the engine's default")
    ap.add_argument("--force-sharded-path", action="store_true",
                    help="N=1 only: drive the two-phase step from Python as the N>1 path does (host-overhead check)")
    ap.add_argument("--neigh", default="bubble", choices=["bubble", "gaussian"],
                    help="neighbourhood kernel (the headline workload is bubble; gaussian updates every unit for every vector)")
    ap.add_argument("--shards", default="interleaved", choices=["interleaved", "contiguous"],
                    help="N > 1: how the map's units are dealt to the ranks")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path with ranks sharing GPUs (keys staged through the host)")
    ap.add_argument("--online-vectors", type=int, default=-1,
                    help="also run the reference-exact online engine (batch 1) over the same schedule; "
                         "-1 = the whole K*batch run when it is <= 600k vectors, 0 = skip")
    return ap.parse_args()


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from som_lvq_pak_amd import engine as E
    from som_lvq_pak_amd._lib import SomParams

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (a.gpus, a.gpus))
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU path")
    local = local % torch.cuda.device_count()          # gloo rehearsal: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    B, K, W = a.batch, a.steps, a.warmup
    xdim, ydim, d = a.xdim, a.ydim, a.dim
    N = xdim * ydim
    radius = a.radius if a.radius is not None else max(xdim, ydim) / 2.0
    length = K * B

    # ---- synthetic data, identical on every rank, resident in HBM before timing.  Generated in
    # fixed chunks with per-chunk seeds so that the stream does not depend on K, W or the batch size.
    ncent, chunk = 256, 65536
    g = torch.Generator(device=dev)
    g.manual_seed(3456)
    centres = 4.0 * torch.randn(ncent, d, generator=g, device=dev)
    nvec = K * B
    data = torch.empty(nvec, d, device=dev)
    for c0 in range(0, nvec, chunk):
        g.manual_seed(3456 + 1 + c0 // chunk)
        m = min(chunk, nvec - c0)
        assign = torch.randint(0, ncent, (chunk,), generator=g, device=dev)[:m]
        data[c0:c0 + m] = centres[assign] + torch.randn(chunk, d, generator=g, device=dev)[:m]
    g.manual_seed(3455)
    first = data[:min(nvec, chunk)]
    lo, hi = first.min(0).values, first.max(0).values
    init = (lo + (hi - lo) * torch.rand(N, d, generator=g, device=dev)).cpu().numpy()   # randinit-like
    del centres, assign, first
    torch.cuda.synchronize()

    eng = E.Engine(local)
    if a.scan != "auto":
        eng.set_scan_mode(a.scan)
    ds = E.Dataset(eng, device_ptr=data.data_ptr(), n=nvec, dim=d)
    from som_lvq_pak_amd.sharded import shard_rows
    neigh = E.NEIGH_GAUSSIAN if a.neigh == "gaussian" else E.NEIGH_BUBBLE
    if a.shards == "interleaved" and world > 1 and xdim % 8 == 0 and ydim % 8 == 0:
        # 8x8-unit patches dealt round-robin to the ranks: every rank sees every region of the map, so the
        # neighbourhood updates of a batch are spread evenly whatever the radius (include/somhip.h)
        mine = E.shard_units(xdim, ydim, rank, world, eng.lib)
        cb = E.Codebook(eng, init[mine], E.TOPOL_HEXA, neigh, xdim, ydim, interleave=(rank, world))
        layout = "8x8-unit patches interleaved over %d ranks" % world
    else:
        r0, r1 = shard_rows(N, world, rank)
        mine = np.arange(r0, r1)
        cb = E.Codebook(eng, init[r0:r1], E.TOPOL_HEXA, neigh, xdim, ydim, row_offset=r0, n_global=N)
        layout = "contiguous row blocks /%d" % world
    lib = eng.lib

    from som_lvq_pak_amd import sharded
    cur_len = [length]
    gshard = sharded.GpuShard(eng, cb, ds, lambda: SomParams(cur_len[0], a.alpha, radius, E.ALPHA_LINEAR,
                                                             0, 0, max(B, 1), 0, 0, 0), max(B, a.eval_vectors))
    ssom = sharded.ShardedSom(gshard, max(B, 1), nvec)

    def step(it0, data_first, count, length_):
        if world == 1 and not a.force_sharded_path:
            p = SomParams(length_, a.alpha, radius, E.ALPHA_LINEAR, 0, 0, max(B, 1), it0, count, data_first)
            E.check(lib.somhip_som_train(cb.h, ds.h, C.byref(p), None, None))
            return
        # local shard winners -> all-reduce(MIN) of packed keys (RCCL) -> local update
        cur_len[0] = length_
        ssom.step(it0, data_first, count)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup on the first W batches, then restore the initial codebook ----
    for k in range(min(W, K)):
        step(k * B, k * B, B, max(W, 1) * B)
    eng.sync()
    cb.upload(init[mine])

    # ---- timed region: one complete training run of K*B vectors ----
    if world > 1:      # N > 1: steps are short; event only the two kernels the roofline lines need
        eng.timing_select({"k_som_update_run", "k_som_update_bubble_s", "k_dist_mfma_bf16", "k_dist_mfma", "k_scan_exact"})
    eng.timing(True)
    eng.timing_reset()
    stats_before = eng.scan_stats()
    barrier()
    t0 = time.perf_counter()
    if world == 1 and not a.force_sharded_path:
        # the K steps (mini-batches) in one call of the epoch-level entry point, as a host tool makes it
        step(0, 0, K * B, length)
    else:
        for k in range(K):
            step(k * B, k * B, B, length)
    barrier()
    t1 = time.perf_counter()
    eng.timing(False)
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    table = eng.timing_table()
    stats_after = eng.scan_stats()

    # ---- final qerror on the first eval vectors of the training stream ----
    ne = min(a.eval_vectors, nvec)
    ek = gshard.winner_keys(0, ne)
    eng.sync()
    sharded.allreduce_min_keys(ek)
    diffs, _ = sharded.unpack_keys(ek.cpu().numpy())
    qerr = float(E.qerror_sum(diffs) / np.float32(ne))

    # ---- the reference-exact online engine on the same stream and schedule (N = 1 only) ----
    online = None
    nonl = a.online_vectors if a.online_vectors >= 0 else (length if length <= 600000 else 0)
    if world == 1 and B > 1 and nonl > 0:
        nonl = min(nonl, length)
        cb.upload(init[mine])
        eng.sync()
        p = SomParams(length, a.alpha, radius, E.ALPHA_LINEAR, 0, 0, 1, 0, nonl, 0)
        t2 = time.perf_counter()
        E.check(lib.somhip_som_train(cb.h, ds.h, C.byref(p), None, None))
        eng.sync()
        t3 = time.perf_counter()
        ek = gshard.winner_keys(0, ne)
        eng.sync()
        d2, _ = sharded.unpack_keys(ek.cpu().numpy())
        q2 = float(E.qerror_sum(d2) / np.float32(ne))
        online = {"schedule": "online, batch 1: the reference's algorithm, bit-exact with the CPU reference",
                  "vectors": nonl, "value": nonl / (t3 - t2), "unit": "vectors/s",
                  "final_qerror": q2 if nonl == length else None,
                  "minibatch_qerror_rel_delta": (qerr - q2) / q2 if nonl == length else None}

    out = None
    if rank == 0:
        value = K * B / elapsed
        # rooflines of the kernels of the timed region; the dominant one (by total time) is reported
        n_local = len(mine)
        bpad = ((B + 31) // 32) * 32
        rows_upd = stats_after["row_updates"] - stats_before["row_updates"]

        def roof_of(kname):
            kl, kms = table[kname]
            avg_s = (kms / max(kl, 1)) * 1e-3
            base = {"kernel": kname, "launches": kl, "avg_launch_ms": avg_s * 1e3, "traffic": None}
            if kname == "k_dist_mfma_bf16":
                alg = 2.0 * n_local * d * bpad                  # algorithmic flops: 2*N*d per vector
                base.update({"bound": "mfma", "achieved": alg / avg_s / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                             "frac": alg / avg_s / 1e12 / 2500.0, "executed_tflops": 3 * alg / avg_s / 1e12,
                             "note": "split-bf16 distance GEMM (v_mfma_f32_32x32x16_bf16, 3 MFMAs per K-step for "
                                     "hi*hi + hi*lo + lo*hi); achieved = ALGORITHMIC 2*N*d flop per vector over "
                                     "the dense bf16 peak; the kernel executes 3x that"})
                return base
            if kname == "k_dist_mfma":
                alg = 2.0 * n_local * d * bpad                  # 2*N*d per vector, GEMM form
                note = "fp32 MFMA (v_mfma_f32_32x32x2_f32) distance GEMM, 2*N*d flop per vector"
            elif kname == "k_scan_exact":
                alg = 3.0 * n_local * d * B                     # direct form: sub, mul, add
                note = ("direct-form fp32 scan on the vector ALU, 3*N*d flop per vector; no FMA allowed, so its "
                        "ceiling is half the fp32 peak")
            elif kname in ("k_som_update_run", "k_som_update_bubble_s"):
                alg = 3.0 * d * rows_upd / max(kl, 1)           # c += a*(x-c): sub, mul, add per element
                note = ("in-order neighbourhood update on the vector ALU: 3*d flop per (row, iteration) update, "
                        "%.0f row updates per launch counted by the kernel; no FMA allowed (ceiling = half the "
                        "fp32 peak); priced against the fp32 matrix/vector peak" % (rows_upd / max(kl, 1)))
            elif kname == "k_som_members":
                pairs = (stats_after["group_updates"] - stats_before["group_updates"]) / max(kl, 1)
                alg = 16.0 * pairs + 24.0 * B                   # member entries written + winners/scalars read
                base.update({"bound": "hbm", "achieved": alg / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": alg / avg_s / 1e9 / PEAK_HBM_GBS,
                             "note": "neighbourhood membership lists: 16 B per (row group, sample) entry written "
                                     "(%.0f per launch) + 24 B per sample read; integer lattice arithmetic and "
                                     "ordered compaction, latency-bound" % pairs})
                return base
            else:
                alg = 4.0 * n_local * d                         # streaming: one read of the shard per launch
                base.update({"bound": "hbm", "achieved": alg / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": alg / avg_s / 1e9 / PEAK_HBM_GBS,
                             "note": "algorithmic bytes = one read of the local codebook shard per launch"})
                return base
            base.update({"bound": "mfma", "achieved": alg / avg_s / 1e12, "peak": PEAK_F32_TFLOPS,
                         "unit": "TFLOP/s", "frac": alg / avg_s / 1e12 / PEAK_F32_TFLOPS, "note": note})
            return base

        # HBM traffic per launch from the committed rocprofv3 PMC passes (bench.py cannot run the
        # profiler on itself); null when the file or the kernel is missing
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        except Exception:
            pass

        def with_traffic(r):
            ks = pmc.get("kernels", {})
            k = ks.get(r["kernel"])
            if k is None:                                       # template / variant suffixes (k_dist_mfma_bf16_wide)
                hits = [v for n, v in ks.items() if n.startswith(r["kernel"] + "_")]
                k = hits[0] if len(hits) == 1 else None
            if k and (xdim, ydim, d, world) == (256, 256, 512, 1):
                r["traffic"] = {"bytes_per_launch": k["bytes"], "read": k["read_bytes"], "write": k["write_bytes"],
                                "source": "profiles/r01_pmc_traffic.json: " + pmc.get("source", "")}
            return r

        ranked = sorted((k for k in table if table[k][0]), key=lambda k: -table[k][1])
        roof = with_traffic(roof_of(ranked[0]))
        roof_other = [with_traffic(roof_of(k)) for k in ranked[1:3]]
        cpu = cpu_baseline(a, init, data, xdim, ydim, d, radius) if (world == 1 and a.cpu_vectors > 0) else None
        out = {
            "metric": "training_vectors_per_sec", "value": value, "unit": "vectors/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "final_qerror": qerr,
            "config": {"workload": "vsom 256x256 hexa bubble SOM, dim=512 (BASELINE.json configs[3])"
                       if (xdim, ydim, d, a.neigh) == (256, 256, 512, "bubble")
                       else "vsom %dx%d hexa %s SOM, dim=%d" % (xdim, ydim, a.neigh, d),
                       "dim": d, "codebook_rows": N, "batch": B, "vectors": K * B, "alpha": a.alpha,
                       "radius": radius, "alpha_type": "linear",
                       "schedule": "mini-batch (winners per batch, in-order updates)" if B > 1 else "online (reference-exact)",
                       "parallelism": "codebook sharded (%s), all-reduce(MIN) of (dist,idx) keys" % layout
                       if world > 1 else "single GPU"},
            "roofline": roof,
            "roofline_other": roof_other,
            "cpu_baseline": cpu,
            "online_exact": online,
            "rerank_stats": {"samples": stats_after["samples"] - stats_before["samples"],
                             "groups_per_sample": (stats_after["groups"] - stats_before["groups"])
                             / max(stats_after["samples"] - stats_before["samples"], 1),
                             "rows_per_sample": (stats_after["rows"] - stats_before["rows"])
                             / max(stats_after["samples"] - stats_before["samples"], 1),
                             "max_groups_per_sample": stats_after["max_groups_per_sample"]},
            "update_stats": {"row_updates": stats_after["row_updates"] - stats_before["row_updates"],
                             "lane_efficiency": (stats_after["row_updates"] - stats_before["row_updates"])
                             / max(64 * (stats_after["group_updates"] - stats_before["group_updates"]), 1)},
            "kernels_ms": {k: {"launches": v[0], "total_ms": round(v[1], 3)} for k, v in table.items() if v[0]},
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def cpu_baseline(a, init, data, xdim, ydim, d, radius):
    """The reference's own som_training (oracle/_ref, built from /root/reference) -- or, if that
    is absent, our CPU restatement -- timed on this host, 1 core, on the first --cpu-vectors
    vectors of the same stream with the same (compressed) schedule.  A reported baseline only."""
    try:
        import oracle
    except Exception as exc:                                  # pragma: no cover
        return {"error": "oracle package not importable: %s" % exc}
    n = a.cpu_vectors
    x = data[:n].cpu().numpy()
    t0 = time.perf_counter()
    if oracle.ref_available():
        ref = oracle.RefHarness()
        ref.som_train(init, xdim, ydim, 3, 2 if a.neigh == "gaussian" else 1, x, n, a.alpha, radius, trace=False)
        secs, kind = ref.last_seconds, "reference"
    else:
        orc = oracle.Oracle()
        t0 = time.perf_counter()
        orc.som_train(init, xdim, ydim, 3, 2 if a.neigh == "gaussian" else 1, x, n, a.alpha, radius, trace=False)
        secs, kind = time.perf_counter() - t0, "port"
    return {"value": n / secs, "unit": "vectors/s", "cores": 1, "kind": kind,
            "host_cores": os.cpu_count(),
            "sample": "som_training on the first %d vectors of the same stream, same map, radius %g->1 and "
                      "alpha over those %d iterations; epoch loop only (%.1f s)" % (n, radius, n, secs)}


if __name__ == "__main__":
    main()
