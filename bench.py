#!/usr/bin/env python3
"""bench.py -- training vectors/sec of the SOM / LVQ hot path on MI355X.

Default workload = BASELINE.json configs[3], the configuration the metric is quoted on (it fits one GPU):
`vsom` on a 256x256 hexagonal map, bubble neighbourhood, dim 512 (codebook 65 536 x 512 fp32 = 128 MiB),
alpha 0.05 linear, radius 128 -> 1, over 10 000 000 vectors of the seeded Gaussian-mixture stream
(`-din gen:k=256,dim=512,n=10000000,seed=3456`, made in HBM by somhip_dataset_generate; initial map =
`randinit -rand 7` from the stream's bounding box).  One stream for the GPU leg, the CPU baseline and the C tools.

A "step" is one mini-batch through the whole hot path: exact best-matching-unit search for every vector of the
batch + the in-order neighbourhood update.  Batch sizes are the engine's own schedule (somhip.h SOMHIP_BATCH_AUTO,
somhip_som_auto_batch: a rule in (units, radius(t), alpha(t)) -- at this workload 32768 vectors up to iteration
8 486 912, 8192 after; `--batch B` fixes one size).  The timed region is EXACTLY --steps such batches of the real
10 M-iteration schedule, evenly spread over it (step k = the batch that holds iteration k * length / steps), so the
radius sweeps its whole range inside the timed region; `value_timed_steps` = vectors of those batches / time.  `value`
is the rate of the COMPLETE 10 M-vector run made after the timed region (`full_run`): the timed steps start from the
initial map, the complete run is what they are samples of.

Conformity (north_star: "qerror within 1e-4 of the CPU reference").  The reference is strictly online
(som_rout.c:600-662); the engine's batch = 1 path is bit-exact with it but HBM-bound (27 k vectors/s).  The
mini-batch schedule is a different algorithm, so its result is CHECKED: the final qerror of the complete run is
compared with the online engine's on the same stream, same initial map (a 380 s run, recorded with its per-sample
statistics in profiles/r03_c4_online_golden.json -- both engines are bit-deterministic, so that number is a constant
of the workload; `--online-full` re-measures it live).  `qerror_check` reports the difference both ways -- `rel_delta`
(north_star's reading) and `abs_delta` (BASELINE.md section 4's) -- for find_qerror's own float accumulator and for the
same mean accumulated in double, next to the resolution of that statistic measured on three seed pairs with exact-
arithmetic controls (any batch > 1, the exact update kernels at batch 256 included, ends on a map
whose qerror differs from the online one by a chaotic +-4e-4 -- the float accumulator alone carries 2e-4 of rounding at
262 144 vectors of size 22.6).  `qerror_check.pass` (|rel| <= 1e-4) gates `value`: if it fails, `value` falls back to
the online engine's rate; `pass_abs` says whether this run also landed inside 1e-4 absolute.

N > 1 (one process per GPU, torch.distributed/RCCL): the codebook is row-sharded (8x8-unit patches dealt
round-robin), every rank scans its shard for the same batch, one all-reduce(MIN) of packed (distance, index)
keys gives the global winners, each rank updates its own rows.  Total work is fixed as N grows -> "strong".

`--config c3` / `--config c5`: the LVQ configurations (BASELINE.json configs[2] / configs[4] shape), see bench_lvq().

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3     # MI355X fp32 matrix == fp32 vector FMA peak (MI355X_MICROARCH.md)
PEAK_F32_NOFMA_TFLOPS = 78.6   # the reference's arithmetic forbids FMA: one flop per lane-cycle
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
GOLDEN = os.path.join(ROOT, "profiles", "r03_c4_online_golden.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", default="c4", choices=["c4", "c2", "c3", "c5"],
                    help="c4 = BASELINE configs[3] (SOM 256x256x512, the headline); c2 = configs[1] (SOM 32x32x128, 100k vectors: the "
                         "online engine, checked against the unmodified reference over the whole run); c3 = configs[2] (OLVQ1 10k x 256); "
                         "c5 = configs[4] shape (LVQ3 100k x 1024)")
    ap.add_argument("--batch", type=int, default=-1,
                    help="vectors per mini-batch; -1 (default) = the engine's own schedule (SOMHIP_BATCH_AUTO: somhip_som_auto_batch)")
    ap.add_argument("--xdim", type=int, default=256)
    ap.add_argument("--ydim", type=int, default=256)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--alpha", type=float, default=0.05)
    ap.add_argument("--radius", type=float, default=None)
    ap.add_argument("--length", type=int, default=10000000, help="schedule length = vectors of the real run (-rlen)")
    ap.add_argument("--cpu-vectors", type=int, default=400,
                    help="vectors the CPU reference trains on for cpu_baseline (0 = skip)")
    ap.add_argument("--eval-vectors", type=int, default=262144)
    ap.add_argument("--eval-wide", type=int, default=2097152, help="second, larger evaluation set (double accumulation only)")
    ap.add_argument("--seed", type=int, default=3456, help="seed of the generator stream")
    ap.add_argument("--init-seed", type=int, default=7, help="randinit -rand")
    ap.add_argument("--scan", default="auto", choices=["auto", "direct", "mfma", "mfma_bf16"],
                    help="winner-search implementation (all bit-identical); auto = the engine's default")
    ap.add_argument("--update", default="gemm", choices=["gemm", "exact"],
                    help="how a mini-batch applies its updates: gemm = affine map of every unit on the fp32 matrix pipe "
                         "(kernels/som_update_gemm.hpp), exact = adapt_vector's arithmetic hit by hit (bit-identical to the batch oracle)")
    ap.add_argument("--neigh", default="bubble", choices=["bubble", "gaussian"],
                    help="neighbourhood kernel (the headline workload is bubble; gaussian updates every unit for every vector)")
    ap.add_argument("--shards", default="interleaved", choices=["interleaved", "contiguous"],
                    help="N > 1: how the map's units are dealt to the ranks")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path with ranks sharing GPUs (keys staged through the host)")
    ap.add_argument("--events-all", action="store_true", help="HIP events around every launch of the timed region (default: only the roofline kernels')")
    ap.add_argument("--no-full-run", action="store_true", help="skip the complete run + qerror check after the timed region")
    ap.add_argument("--online-vectors", type=int, default=65536,
                    help="N = 1: iterations of the real schedule the reference-exact online engine runs live (its rate); 0 = skip")
    ap.add_argument("--online-full", action="store_true",
                    help="N = 1: run the online engine over the WHOLE schedule live (minutes) instead of using the recorded qerror")
    return ap.parse_args()


def setup_dist(a):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:                                  # (a bare `--gpus N` never gets here: main() starts the ranks itself)
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
    return rank, world, local, dev


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True).stdout.strip() or None
    except Exception:
        return None


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (bench.py cannot run the profiler on itself)."""
    for name in ("r03_pmc_traffic.json",):
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", name)))
            j["file"] = "profiles/" + name
            return j
        except Exception:
            continue
    return {}


def launcher_command(argv, gpus, port):
    """The driver's N > 1 command (task contract): one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1;
    `argv` = this script's own arguments, handed on unchanged."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(a, argv):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as a FRESH child (this process has not touched a
    GPU and never will: no exec of a process that initialised HIP), pass its output through -- rank 0 prints the JSON
    line -- and leave with its exit code."""
    cmd = launcher_command(argv, a.gpus, free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print("bench.py: starting %d ranks: %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    try:
        for line in child.stdout:                        # rank 0's JSON line to stdout; whatever else the ranks' libraries
            if line.lstrip().startswith("{"):            # print there (gloo's connection notes) to stderr: ONE line on stdout
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write(line)
        rc = child.wait()
    except KeyboardInterrupt:
        child.terminate()
        rc = child.wait()
    raise SystemExit(rc if rc >= 0 else 128 - rc)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a, sys.argv[1:])
    if a.config == "c2":
        return bench_c2(a)
    if a.config != "c4":
        return bench_lvq(a)
    return bench_som(a)


def bench_som(a):
    import torch
    import torch.distributed as dist
    from som_lvq_pak_amd import engine as E
    from som_lvq_pak_amd import sharded
    from som_lvq_pak_amd._lib import SomParams

    rank, world, local, dev = setup_dist(a)
    B, K, W, L = a.batch, a.steps, a.warmup, a.length
    xdim, ydim, d = a.xdim, a.ydim, a.dim
    N = xdim * ydim
    radius = a.radius if a.radius is not None else max(xdim, ydim) / 2.0
    seed, kcent, init_seed = a.seed, 256, a.init_seed
    auto_b = B == -1
    if not auto_b:
        nbatches = L // B
        if nbatches < 1:
            raise SystemExit("--length must be at least one batch")
        stride = max(1, nbatches // max(K, 1)) * B        # timed step k starts at iteration k * stride
        if (K - 1) * stride + B > L:
            raise SystemExit("--steps %d x --batch %d does not fit a schedule of %d iterations" % (K, B, L))

    # ---- the stream, resident in HBM before any timing; identical on every rank (counter-based generator) ----
    eng = E.Engine(local)
    if a.scan != "auto":
        eng.set_scan_mode(a.scan)
    eng.set_update_mode(a.update)
    ds = E.Dataset(eng, generate=(seed, kcent, d, 0, L))
    lo, hi, cnt = E.column_minmax(ds)
    init = E.randinit_from_bbox(lo, hi, cnt, xdim, ydim, init_seed)     # = randinit -rand 7 on the same source
    neigh = E.NEIGH_GAUSSIAN if a.neigh == "gaussian" else E.NEIGH_BUBBLE
    if a.shards == "interleaved" and world > 1 and xdim % 8 == 0 and ydim % 8 == 0:
        mine = E.shard_units(xdim, ydim, rank, world, eng.lib)
        cb = E.Codebook(eng, init[mine], E.TOPOL_HEXA, neigh, xdim, ydim, interleave=(rank, world))
        layout = "8x8-unit patches interleaved over %d ranks" % world
    else:
        r0, r1 = sharded.shard_rows(N, world, rank)
        mine = np.arange(r0, r1)
        cb = E.Codebook(eng, init[r0:r1], E.TOPOL_HEXA, neigh, xdim, ydim, row_offset=r0, n_global=N)
        layout = "contiguous row blocks /%d" % world
    lib = eng.lib
    gshard = sharded.GpuShard(eng, cb, ds, lambda: SomParams(L, a.alpha, radius, E.ALPHA_LINEAR, 0, 0, B, 0, 0, 0),
                              max(B, 32768))
    ssom = sharded.ShardedSom(gshard, B, L)
    def auto_at(it):
        return E.som_auto_batch(lib, L, it, alpha=a.alpha, radius=radius, n_units=N, topol=E.TOPOL_HEXA, neigh=neigh)

    # the timed steps: K batches of the schedule, evenly spread over it -- (first iteration, vectors) each
    if auto_b and auto_at(0)[1] == 1:
        raise SystemExit("the engine's schedule for this map / run length is batch 1 (the rule does not vouch for mini-batches "
                         "here: somhip_som_auto_batch); use --batch B, or --config c2 for the online engine's bench line")
    if auto_b:
        steps_at = []
        for k in range(K):
            st, ln = auto_at((k * L) // max(K, 1))
            if not steps_at or st != steps_at[-1][0]:
                steps_at.append((st, ln))
        if len(steps_at) != K:
            raise SystemExit("--steps %d: more steps than batches in a schedule of %d iterations" % (K, L))
    else:
        steps_at = [(k * stride, B) for k in range(K)]
    vectors_timed = sum(ln for _, ln in steps_at)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def final_qerror():
        """(find_qerror's float-accumulated mean over the first --eval-vectors vectors, their number, the same mean
        accumulated in double, the double mean over the first --eval-wide vectors)"""
        ne, nw = min(a.eval_vectors, L), min(max(a.eval_wide, a.eval_vectors), L)
        parts = []
        for f in range(0, nw, 8192):                            # runs the MFMA pre-filter path (one call handles <= 8192)
            ek = gshard.winner_keys(f, min(8192, nw - f))
            eng.sync()
            sharded.allreduce_min_keys(ek, nonnegative=True)
            parts.append(ek.cpu().numpy().copy())
        diffs, _ = sharded.unpack_keys(np.concatenate(parts))
        r = np.sqrt(diffs.astype(np.float64))
        return float(E.qerror_sum(diffs[:ne]) / np.float32(ne)), ne, float(r[:ne].mean()), float(r[:nw].mean()), nw

    # ---- warmup on the first W steps of the timed sequence, then the initial map again ----
    for ln in sorted({ln for _, ln in steps_at}):        # (which winner-search path the ranks take is agreed once per batch length: not in the timed region)
        ssom._exchange(ln)
    for st, ln in steps_at[:min(W, K)]:
        ssom.step(st, st, ln)
    eng.sync()
    cb.upload(init[mine])

    # ---- timed region: exactly K steps, scan -> all-reduce -> update enqueued back to back on the engine's stream ----
    # HIP events in the timed region only around the kernels the roofline lines are made from (every evented launch
    # costs two event records on the stream: with all 16 launches of a step evented the step was 3 % longer); the table
    # of ALL kernels comes from a replay of the same steps after the timed region (same start state, same batches)
    roofline_kernels = {"k_som_update_run", "k_som_update_bubble_s", "k_som_update_gemm", "k_dist_mfma_bf16", "k_dist_mfma",
                        "k_scan_exact", "k_dist_l2", "k_rerank_pairs"}
    if not a.events_all:
        eng.timing_select(roofline_kernels)
    eng.timing(True)
    eng.timing_reset()
    stats_before = eng.scan_stats()
    barrier()
    t0 = time.perf_counter()
    for st, ln in steps_at:
        ssom.step(st, st, ln)
    barrier()
    t1 = time.perf_counter()
    eng.timing(False)
    elapsed = max_over_ranks(t1 - t0)
    table = eng.timing_table()
    stats_after = eng.scan_stats()
    table_all = table
    if not a.events_all and world == 1:                  # the same K steps again, every launch evented, outside the timed region
        cb.upload(init[mine])
        eng.timing_select(None)
        eng.timing(True)
        eng.timing_reset()
        for st, ln in steps_at:
            ssom.step(st, st, ln)
        eng.sync()
        eng.timing(False)
        table_all = eng.timing_table()

    # ---- the complete run with the same schedule: what the timed steps were samples of, and the conformity check ----
    full = None
    if not a.no_full_run:
        cb.upload(init[mine])
        barrier()
        t2 = time.perf_counter()
        if world == 1:
            # one call of the epoch-level entry point, as a host tool makes it
            p = SomParams(L, a.alpha, radius, E.ALPHA_LINEAR, 0, 0, B, 0, L, 0)
            E.check(lib.somhip_som_train(cb.h, ds.h, C.byref(p), None, None))
        else:
            it0 = 0
            while it0 < L:
                ln = auto_at(it0)[1] if auto_b else min(B, L - it0)
                ssom.step(it0, it0, ln)
                it0 += ln
        barrier()
        secs = max_over_ranks(time.perf_counter() - t2)
        q, ne, q64, q64w, nw = final_qerror()
        full = {"vectors": L, "seconds": secs, "value": L / secs, "unit": "vectors/s", "final_qerror": q,
                "eval_vectors": ne, "mean_f64": q64, "mean_f64_wide": q64w, "eval_wide": nw}

    # ---- the reference-exact online engine, live, on the head of the same schedule (N = 1) ----
    online = None
    if world == 1 and B != 1 and (a.online_vectors > 0 or a.online_full):
        nonl = L if a.online_full else min(a.online_vectors, L)
        cb.upload(init[mine])
        eng.sync()
        t2 = time.perf_counter()
        seg = 1 << 18
        for s in range(0, nonl, seg):
            p = SomParams(L, a.alpha, radius, E.ALPHA_LINEAR, 0, 0, 1, s, min(seg, nonl - s), s)
            E.check(lib.somhip_som_train(cb.h, ds.h, C.byref(p), None, None))
            if a.online_full:
                print("online %d / %d (%.0f s)" % (min(s + seg, nonl), nonl, time.perf_counter() - t2), file=sys.stderr, flush=True)
        eng.sync()
        t3 = time.perf_counter()
        online = {"schedule": "online, batch 1: the reference's algorithm (som_rout.c:600-662), bit-exact with the CPU reference",
                  "vectors": nonl, "value": nonl / (t3 - t2), "unit": "vectors/s",
                  "sample": "whole schedule" if nonl == L else "iterations [0, %d) of the %d-iteration schedule (radius ~ %g: the costliest part)" % (nonl, L, radius)}
        if nonl == L:
            fq = final_qerror()
            online.update({"final_qerror": fq[0], "mean_f64": fq[2], "mean_f64_wide": fq[3]})

    out = None
    if rank == 0:
        value_steps = vectors_timed / elapsed
        if auto_b:
            t1 = next((st for st in range(0, L, 32768) if auto_at(st)[1] != 32768), L)
            bdesc = "engine-chosen (somhip_som_auto_batch: 32768 up to iteration %d, %d after)" % (t1, auto_at(min(t1, L - 1))[1])
        else:
            bdesc = str(B)
        # ---- conformity: the complete mini-batch run against the online engine's result on the same stream ----
        check = None
        if full is not None:
            ref, src = None, None
            if online is not None and "final_qerror" in online:
                ref = {"qerror": online["final_qerror"], "mean_f64": online["mean_f64"], "mean_f64_wide": online["mean_f64_wide"]}
                src = "measured live in this run (--online-full)"
            else:
                try:
                    g = json.load(open(GOLDEN))
                    for run in g["runs"]:
                        same = ((xdim, ydim, d, a.neigh, a.alpha, radius, L, seed, init_seed, full["eval_vectors"], full["eval_wide"]) ==
                                (256, 256, 512, "bubble", 0.05, 128.0, run["length"], run["seed"], run["init_seed"], g["eval_vectors"], g["eval_wide"]))
                        if same:
                            ref = run["online"]
                            src = ("profiles/r03_c4_online_golden.json: the online engine over the same 10 M-vector stream and initial map, "
                                   "%.0f s on one MI355X at commit %s (bit-deterministic; bench.py --online-full re-measures it)"
                                   % (ref["seconds"], g.get("commit")))
                except Exception as exc:
                    src = "no recorded online result readable: %s" % exc
            if ref is not None:
                d32 = full["final_qerror"] - ref["qerror"]
                check = {"online": ref["qerror"], "value": full["final_qerror"], "abs_delta": d32, "rel_delta": d32 / ref["qerror"],
                         "online_f64": ref["mean_f64"], "value_f64": full["mean_f64"], "abs_delta_f64": full["mean_f64"] - ref["mean_f64"],
                         "abs_delta_f64_wide": full["mean_f64_wide"] - ref["mean_f64_wide"], "eval_wide": full["eval_wide"],
                         "tol": 1e-4, "pass": bool(abs(d32 / ref["qerror"]) <= 1e-4), "pass_abs": bool(abs(d32) <= 1e-4),
                         "vectors": L, "batch": bdesc, "online_source": src,
                         "resolution": "three seed pairs at full length, online vs batch > 1 (profiles/r03_conformity_*.jsonl): per-sample distances "
                                       "decorrelate (rms 0.042) under ANY batch > 1 -- batch 256 with the exact update kernels: abs_delta "
                                       "-4.6e-4 / -9.7e-5, double-accumulated -3.3e-4 / -2.7e-5; find_qerror's float accumulator alone carries "
                                       "~2e-4 of rounding at 262144 vectors of size 22.6 (online: float 22.571381 vs double 22.571050): an "
                                       "absolute 1e-4 is below what this statistic resolves, `pass` is the relative reading"}
            elif src:
                check = {"pass": False, "error": src}
        conforming = check is not None and check.get("pass", False)
        unverified = full is None or check is None
        if conforming:                                       # the complete run IS the measurement
            value, sched = full["value"], "mini-batch %s (winners per batch against the codebook before the batch, updates in iteration order)" % bdesc
        elif unverified:                                     # no complete run / nothing to check it against: say so
            value, sched = value_steps, "mini-batch %s -- UNVERIFIED: no complete run or no online result to check its qerror against" % bdesc
        else:                                                # the mini-batch schedule missed the tolerance: only batch 1 conforms
            value, sched = (online["value"] if online else 0.0), "online (reference-exact); the mini-batch schedule FAILED the qerror check"

        n_local = len(mine)
        Bavg = vectors_timed / max(K, 1)                     # vectors per timed step (per launch of every per-batch kernel)
        bpad = ((int(Bavg) + 31) // 32) * 32
        rows_upd = stats_after["row_updates"] - stats_before["row_updates"]
        pmc = pmc_traffic()

        def roof_of(kname):
            kl, kms = table[kname]
            avg_s = (kms / max(kl, 1)) * 1e-3
            base = {"kernel": kname, "launches": kl, "avg_launch_ms": avg_s * 1e3, "traffic": None}
            if kname == "k_dist_mfma_bf16":
                alg = 2.0 * n_local * d * Bavg                  # algorithmic flops: 2*N*d per vector
                two_level = table.get("k_dist_l2", (0, 0.0))[0] > 0
                base.update({"bound": "mfma", "achieved": alg / avg_s / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                             "frac": alg / avg_s / 1e12 / PEAK_BF16_TFLOPS})
                if two_level:
                    base["note"] = ("level 1 of the two-level pre-filter: ||c||^2 - 2<c_hi, x_hi> for every (code, vector) pair, one "
                                    "bf16 MFMA per K-step (k_dist_mfma_bf16_l1r: v_mfma_f32_16x16x32_bf16; executed = ALGORITHMIC 2*N*d flop "
                                    "per vector) over the dense bf16 peak; 8 32x32x16 MFMAs back to back take 146 ns on this part "
                                    "(tools/micro/mfma_indep.hip -DBF16): "
                                    "1.84 Pflop/s is what the instruction sustains; the level-2 three-product GEMM on the survivors and "
                                    "the exact re-rank are separate kernels (k_dist_l2, k_rerank_*)")
                    base["frac_of_sustained_mfma_rate"] = alg / avg_s / 1e12 / 1840.0
                else:
                    base.update({"executed_tflops": 3 * alg / avg_s / 1e12, "executed_frac": 3 * alg / avg_s / 1e12 / PEAK_BF16_TFLOPS,
                                 "note": "split-bf16 distance GEMM (v_mfma_f32_32x32x16_bf16, 3 MFMAs per K-step for "
                                         "hi*hi + hi*lo + lo*hi); achieved = ALGORITHMIC 2*N*d flop per vector over "
                                         "the dense bf16 peak; the kernel executes 3x that"})
                return base
            if kname == "k_dist_mfma":
                alg = 2.0 * n_local * d * bpad                  # 2*N*d per vector, GEMM form
                base.update({"bound": "mfma", "achieved": alg / avg_s / 1e12, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                             "frac": alg / avg_s / 1e12 / PEAK_F32_TFLOPS,
                             "note": "fp32 MFMA (v_mfma_f32_32x32x2_f32) distance GEMM, 2*N*d flop per vector"})
                return base
            if kname == "k_som_update_gemm":
                walked = (stats_after["gemm_entries"] - stats_before["gemm_entries"]) / max(kl, 1)
                alg = 3.0 * d * rows_upd / max(kl, 1)           # what the reference's arithmetic spends on these updates
                exe = 2.0 * d * 64.0 * walked                   # what the matrix pipe executed (64 units per walked entry)
                base.update({"bound": "mfma", "achieved": exe / avg_s / 1e12, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                             "frac": exe / avg_s / 1e12 / PEAK_F32_TFLOPS, "algorithmic_tflops": alg / avg_s / 1e12,
                             "note": "neighbourhood update of a batch as c' = P c + W X on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 "
                                     "accumulate): achieved = EXECUTED 2*d*64 flop per walked list entry (%.0f per launch; hits whose weight "
                                     "decayed below 2^-24 are skipped) over the fp32 matrix peak; algorithmic_tflops = the reference's "
                                     "3*d flop per (row, iteration) update of the list entries K4b made (%.0f per launch: only the tails of "
                                     "the lists that the walk can reach are made, so most of the reference's updates are not even counted) "
                                     "at this kernel's time"
                                     % (walked, rows_upd / max(kl, 1))})
                return base
            if kname in ("k_scan_exact", "k_som_update_run", "k_som_update_bubble_s"):
                if kname == "k_scan_exact":
                    alg = 3.0 * n_local * d * Bavg              # direct form: sub, mul, add
                    note = "direct-form fp32 scan on the vector ALU, 3*N*d flop per vector"
                else:
                    alg = 3.0 * d * rows_upd / max(kl, 1)       # c += a*(x-c): sub, mul, add per element
                    note = ("in-order neighbourhood update on the vector ALU: 3*d flop per (row, iteration) update, "
                            "%.0f row updates per launch counted by the kernel" % (rows_upd / max(kl, 1)))
                ach = alg / avg_s / 1e12
                base.update({"bound": "valu", "achieved": ach, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                             "frac": ach / PEAK_F32_TFLOPS, "peak_no_fma": PEAK_F32_NOFMA_TFLOPS,
                             "frac_of_no_fma_ceiling": ach / PEAK_F32_NOFMA_TFLOPS,
                             "note": note + "; three separately rounded ops per element (the reference's arithmetic: no FMA "
                                            "allowed), so the ceiling of this form is half the fp32 vector peak; `frac` is "
                                            "against the full 157.3 TFLOP/s"})
                return base
            if kname == "k_som_members":
                pairs = (stats_after["group_updates"] - stats_before["group_updates"]) / max(kl, 1)
                alg = 16.0 * pairs + 24.0 * Bavg                # member entries written + winners/scalars read
                note = ("neighbourhood membership lists: 16 B per (row group, sample) entry written (%.0f per launch) + "
                        "24 B per sample read; integer lattice arithmetic and ordered compaction, latency-bound" % pairs)
            else:
                alg = 4.0 * n_local * d                         # streaming: one read of the shard per launch
                note = "algorithmic bytes = one read of the local codebook shard per launch"
            base.update({"bound": "hbm", "achieved": alg / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": alg / avg_s / 1e9 / PEAK_HBM_GBS, "note": note})
            return base

        def with_traffic(r):
            ks = pmc.get("kernels", {})
            k = ks.get(r["kernel"])
            if k is None:                                       # template / variant suffixes (k_dist_mfma_bf16_wide)
                hits = [v for n, v in ks.items() if n.startswith(r["kernel"] + "_")]
                k = hits[0] if len(hits) == 1 else None
            if k and (xdim, ydim, d, world) == (256, 256, 512, 1) and auto_b and pmc.get("schedule") == "auto":
                r["traffic"] = {"bytes_per_launch": k["bytes"], "read": k["read_bytes"], "write": k["write_bytes"],
                                "source": "%s (commit %s): %s" % (pmc.get("file"), pmc.get("commit", "round 1"), pmc.get("source", ""))}
            return r

        ranked = sorted((k for k in table if table[k][0]), key=lambda k: -table[k][1])
        roof = with_traffic(roof_of(ranked[0]))
        roof_other = [with_traffic(roof_of(k)) for k in ranked[1:3]]
        cpu = cpu_baseline_som(a, init, xdim, ydim, d, radius, seed, kcent) if (world == 1 and a.cpu_vectors > 0) else None
        out = {
            "metric": "training_vectors_per_sec", "value": value, "unit": "vectors/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "final_qerror": full["final_qerror"] if full else None,
            "qerror_check": check,
            "value_unverified": bool(unverified),
            "config": {"workload": "vsom 256x256 hexa bubble SOM, dim=512, 10M vectors (BASELINE.json configs[3])"
                       if (xdim, ydim, d, a.neigh, L) == (256, 256, 512, "bubble", 10000000)
                       else "vsom %dx%d hexa %s SOM, dim=%d, %d vectors" % (xdim, ydim, a.neigh, d, L),
                       "dim": d, "codebook_rows": N, "batch": bdesc, "schedule_length": L, "vectors_timed": vectors_timed,
                       "timed_steps": "%d batches of the schedule, %s vectors each, starting at iterations %s ... of the %d-iteration schedule (radius %g -> 1 across them)"
                                      % (K, "/".join(str(x) for x in sorted({ln for _, ln in steps_at}, reverse=True)),
                                         ", ".join(str(st) for st, _ in steps_at[:3]), L, radius),
                       "alpha": a.alpha, "radius": radius, "alpha_type": "linear",
                       "stream": "gen:k=%d,dim=%d,n=%d,seed=%d (somhip_dataset_generate); randinit -rand %d" % (kcent, d, L, seed, init_seed),
                       "schedule": sched, "update_mode": a.update,
                       "parallelism": "codebook sharded (%s), %sall-reduce(MIN) of (dist,idx) keys"
                                      % (layout, "pre-filter bounds MIN-reduced between the levels of the winner search, "
                                         if any(ssom._exch.values()) else "")
                       if world > 1 else "single GPU",
                       "commit": git_head()},
            "value_timed_steps": value_steps,
            "full_run": full,
            "roofline": roof,
            "roofline_other": roof_other,
            "cpu_baseline": cpu,
            "online_exact": online,
            "rerank_stats": {"samples": stats_after["samples"] - stats_before["samples"],
                             "groups_per_sample": (stats_after["groups"] - stats_before["groups"])
                             / max(stats_after["samples"] - stats_before["samples"], 1),
                             "rows_per_sample": (stats_after["rows"] - stats_before["rows"])
                             / max(stats_after["samples"] - stats_before["samples"], 1),
                             "max_groups_per_sample": stats_after["max_groups_per_sample"],
                             "level2_groups_per_sample": (stats_after["l2_pairs"] - stats_before["l2_pairs"])
                             / max(stats_after["samples"] - stats_before["samples"], 1)},
            "update_stats": {"row_updates": rows_upd,
                             "lane_efficiency": rows_upd / max(64 * (stats_after["group_updates"] - stats_before["group_updates"]), 1)},
            "kernels_ms": {k: {"launches": v[0], "total_ms": round(v[1], 3)} for k, v in table_all.items() if v[0]},
            "kernels_ms_note": "per-launch HIP-event durations of every kernel of the K timed batches; measured in a replay of the same batches "
                               "after the timed region unless --events-all (in the timed region only the roofline kernels carry events)",
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def cpu_baseline_som(a, init, xdim, ydim, d, radius, seed, kcent):
    """The reference's own som_training (oracle/_ref, built from /root/reference) -- or, if that is absent, our CPU
    restatement -- timed on this host, 1 core, on the first --cpu-vectors vectors of the same generator stream and
    the same initial map, with the radius and alpha schedules run over those vectors (the whole radius range, as
    in the GPU leg's timed region).  A reported baseline only."""
    try:
        import oracle
    except Exception as exc:                                  # pragma: no cover
        return {"error": "oracle package not importable: %s" % exc}
    from som_lvq_pak_amd import engine as E
    n = a.cpu_vectors
    x, _ = E.gen_rows(seed, kcent, d, 0, n)
    neigh = 2 if a.neigh == "gaussian" else 1
    if oracle.ref_available():
        ref = oracle.RefHarness()
        ref.som_train(init, xdim, ydim, 3, neigh, x, n, a.alpha, radius, trace=False)
        secs, kind = ref.last_seconds, "reference"
    else:
        orc = oracle.Oracle()
        t0 = time.perf_counter()
        orc.som_train(init, xdim, ydim, 3, neigh, x, n, a.alpha, radius, trace=False)
        secs, kind = time.perf_counter() - t0, "port"
    return {"value": n / secs, "unit": "vectors/s", "cores": 1, "kind": kind,
            "host_cores": os.cpu_count(),
            "sample": "som_training (the reference's own code, gcc -O3 -ffp-contract=off) on the first %d vectors of the same "
                      "stream, same initial map, radius %g->1 and alpha over those %d iterations; epoch loop only (%.1f s)"
                      % (n, radius, n, secs)}


def bench_c2(a):
    """BASELINE.json configs[1]: vsom 32x32 hexa bubble map, dim 128, 100 000 vectors of `gen:k=16,dim=128,seed=1234`, alpha
    0.05 linear, radius 10 -> 1, `randinit -rand 7`.  The map is 512 KiB: nothing here is bandwidth- or matrix-bound, and
    the engine's schedule for it is the reference's own (somhip_som_auto_batch answers batch 1) -- the ONLINE engine, one
    launch per iteration replayed from hipGraphs of 1024.  A "step" is 1024 consecutive iterations (one graph); the timed
    region is --steps of them from the head of the schedule (the widest neighbourhoods); `value` is the rate of the
    complete 100 000-vector run, whose final qerror is checked against the UNMODIFIED REFERENCE's over the same
    complete run (cpu_baseline, ~26 s on one host core) and against the reference CLI's recorded output
    (tests/golden/cli/expected.json: 11.168620).  Single GPU: small maps are "replicas only" (DESIGN section 6)."""
    import torch
    from som_lvq_pak_amd import engine as E
    from som_lvq_pak_amd._lib import SomParams
    if a.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--config c2 is a one-GPU line: a 512 KiB map does not shard (replicas only)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU path")
    xdim, ydim, d, L, seed, kcent, init_seed, alpha, radius = 32, 32, 128, 100000, 1234, 16, 7, 0.05, 10.0
    K, W, STEP = a.steps, a.warmup, 1024
    if K * STEP > L:                                      # (the default --steps is sized for configs[3]: as many steps as the schedule holds)
        K = L // STEP
    eng = E.Engine(0)
    lib = eng.lib
    ds = E.Dataset(eng, generate=(seed, kcent, d, 0, L))
    lo, hi, cnt = E.column_minmax(ds)
    init = E.randinit_from_bbox(lo, hi, cnt, xdim, ydim, init_seed)
    cb = E.Codebook(eng, init, E.TOPOL_HEXA, E.NEIGH_BUBBLE, xdim, ydim)
    auto = E.som_auto_batch(lib, L, 0, alpha=alpha, radius=radius, n_units=xdim * ydim)

    def run(it0, count, batch=1):
        p = SomParams(L, alpha, radius, E.ALPHA_LINEAR, 0, 0, batch, it0, count, it0)
        E.check(lib.somhip_som_train(cb.h, ds.h, C.byref(p), None, None))

    for k in range(W):
        run(k * STEP, STEP)
    eng.sync()
    cb.upload(init)
    eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        run(k * STEP, STEP)
    eng.sync(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # per-launch duration of the step kernel: the same steps again with HIP events (no graph while events are on)
    cb.upload(init)
    eng.timing(True); eng.timing_reset()
    run(0, min(4, K) * STEP)
    eng.sync()
    eng.timing(False)
    kl, kms = eng.timing_table()["k_som_online_step"]
    # the complete run through the engine's own schedule (SOMHIP_BATCH_AUTO = batch 1 here), then its qerror over the whole data set
    cb.upload(init)
    eng.sync()
    t2 = time.perf_counter()
    run(0, L, batch=E.BATCH_AUTO)
    eng.sync()
    secs = time.perf_counter() - t2
    _, diff, ret = E.find_winners(cb, ds, 0, L)
    q_gpu = float(E.qerror_sum(diff, ret) / np.float32(L))
    final = cb.download()
    full = {"vectors": L, "seconds": secs, "value": L / secs, "unit": "vectors/s", "final_qerror": q_gpu, "eval_vectors": L,
            "schedule": "SOMHIP_BATCH_AUTO -> batch %d" % auto[1]}
    cpu = check = None
    if a.cpu_vectors > 0:
        try:
            import oracle
            x = ds.rows(0, L)
            if oracle.ref_available():
                ref = oracle.RefHarness()
                want, _, _ = ref.som_train(init, xdim, ydim, 3, 1, x, L, alpha, radius, trace=False)
                csecs, kind = ref.last_seconds, "reference"
                q_cpu = ref.find_qerror(want, x, xdim=xdim, ydim=ydim)[0] / np.float32(L)
            else:
                orc = oracle.Oracle()
                t3 = time.perf_counter()
                want, _, _ = orc.som_train(init, xdim, ydim, 3, 1, x, L, alpha, radius, trace=False)
                csecs, kind = time.perf_counter() - t3, "port"
                q_cpu = orc.find_qerror(want, x)[0] / np.float32(L)
            cpu = {"value": L / csecs, "unit": "vectors/s", "cores": 1, "kind": kind, "host_cores": os.cpu_count(),
                   "sample": "som_training over the WHOLE run: all %d vectors of the same stream, same initial map (%.1f s); "
                             "gcc -O3 -ffp-contract=off" % (L, csecs), "final_qerror": float(q_cpu)}
            check = {"cpu": float(q_cpu), "value": q_gpu, "abs_delta": q_gpu - float(q_cpu), "rel_delta": (q_gpu - float(q_cpu)) / float(q_cpu),
                     "tol": 1e-4, "codebook_bits_equal": bool(np.array_equal(final.view(np.uint32), want.view(np.uint32))),
                     "pass": bool(abs(q_gpu - float(q_cpu)) <= 1e-4)}
        except Exception as exc:                              # pragma: no cover
            cpu = {"error": "%s" % exc}
    golden_cli = None
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "cli", "expected.json")))["c2_full"]
        golden_cli = {"reference_cli_qerror_stdout": g["qerror_stdout"].strip(), "this_run": "%f" % q_gpu,
                      "abs_delta": q_gpu - float(g["qerror_stdout"]), "agree": bool(abs(q_gpu - float(g["qerror_stdout"])) <= 2e-6),
                      "note": "the reference's `qerror` tool evaluates the map as vsom wrote it to the .cod file (\"%g\": six significant "
                              "digits per value), this run the map in memory: the same map to the file format's precision "
                              "(tests/test_cli_tools.py::test_c2_full_size_matches_reference_cli compares the .cod bytes and the tool's line)"}
    except Exception:
        pass
    avg_us = 1e3 * kms / max(kl, 1)
    cbytes = 4.0 * xdim * ydim * d
    out = {
        "metric": "training_vectors_per_sec", "value": full["value"], "unit": "vectors/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "final_qerror": q_gpu, "qerror_check": check, "reference_cli": golden_cli,
        "value_unverified": check is None,
        "config": {"workload": "vsom 32x32 hexa bubble SOM, dim=128, 100k vectors (BASELINE.json configs[1])", "dim": d,
                   "codebook_rows": xdim * ydim, "batch": "1 (online: the reference's schedule; somhip_som_auto_batch answers batch %d for this map)" % auto[1],
                   "schedule_length": L, "vectors_timed": K * STEP, "step": "1024 consecutive iterations = one hipGraph of k_som_online_step launches",
                   "alpha": alpha, "radius": radius, "alpha_type": "linear",
                   "stream": "gen:k=%d,dim=%d,n=%d,seed=%d (somhip_dataset_generate); randinit -rand %d" % (kcent, d, L, seed, init_seed),
                   "parallelism": "single GPU (replicas only: a 512 KiB map does not shard)", "commit": git_head()},
        "value_timed_steps": K * STEP / elapsed,
        "full_run": full,
        "roofline": {"kernel": "k_som_online_step", "bound": "hbm", "achieved": cbytes / (avg_us * 1e-6) / 1e9, "peak": PEAK_HBM_GBS,
                     "unit": "GB/s", "frac": cbytes / (avg_us * 1e-6) / 1e9 / PEAK_HBM_GBS, "launches": kl, "avg_launch_ms": avg_us * 1e-3,
                     "traffic": None,
                     "note": "one launch per iteration: update(t-1) + distance(t) in one pass over the 512 KiB map (algorithmic bytes = one "
                             "read of the map per iteration; it lives in L2). Latency-bound, not bandwidth-bound: %.2f us per launch with "
                             "events, %.2f us per iteration inside the graph replays of the timed region (launch-to-launch floor of "
                             "dependent kernels ~1.5 us, MI355X_MICROARCH.md 'boundary')" % (avg_us, 1e6 * elapsed / (K * STEP))},
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    return out



LVQ_CONFIGS = {
    # BASELINE.json configs[2]: OLVQ1, 10 000 codes x 256, 1 M labelled vectors, per-code alpha (K = 100 classes =
    # mixture ids, seed 2345, initial codes = the first 100 samples of each class, alpha0 = 0.3)
    "c3": dict(kind=2, name="olvq1", codes=10000, dim=256, classes=100, seed=2345, length=1000000, window=1000000,
               alpha=0.3, winlen=0.0, epsilon=0.0, cpu_vectors=16000,
               workload="OLVQ1 10k codebook, dim=256, 1M labelled vectors (BASELINE.json configs[2])"),
    # BASELINE.json configs[4]: LVQ3, 100 000 codes x 1024, 100 M vectors (seed 4567, K = 1000, alpha 0.05, win 0.3, eps 0.1).  The
    # 410 GB stream cannot be stored: the run cycles over a window of it held in HBM.
    "c5": dict(kind=4, name="lvq3", codes=100000, dim=1024, classes=1000, seed=4567, length=100000000, window=2097152,
               alpha=0.05, winlen=0.3, epsilon=0.1, cpu_vectors=400,
               workload="LVQ3 100k codebook, dim=1024, 100M-iteration schedule over a 2M-vector window (BASELINE.json configs[4] shape)"),
}


def bench_lvq(a):
    """The LVQ configurations.  A "step" is 1024 iterations of the reference's online loop (lvq_rout.c:584-697 / 808-916),
    run by the exact batched engine (kernels/lvq_batch.hpp): bit-identical to one-sample-at-a-time training, so there is no
    tolerance to check -- `exact_check` replays a prefix with the one-launch-per-iteration kernel and compares codebook bits.
    N > 1: the codebook is row-sharded (sharded.ShardedLvq: per-shard top-8, all-gather, candidate rows by integer
    all-reduce, replicated walk, owners commit)."""
    import torch
    import torch.distributed as dist
    from som_lvq_pak_amd import engine as E
    from som_lvq_pak_amd import sharded
    from som_lvq_pak_amd._lib import LvqParams

    cfg = LVQ_CONFIGS[a.config]
    rank, world, local, dev = setup_dist(a)
    K, W, STEP = a.steps, a.warmup, 1024
    kind, N, d, L, nwin = cfg["kind"], cfg["codes"], cfg["dim"], cfg["length"], cfg["window"]
    knn = 2 if kind in (3, 4) else 1
    eng = E.Engine(local)
    ds = E.Dataset(eng, generate=(cfg["seed"], cfg["classes"], d, 0, nwin))
    lab = ds.centres.astype(np.int32) + 1                         # class = mixture id (1-based: 0 is LABEL_EMPTY, labels.h)
    # the data set's labels are the mixture ids as generated (0-based); codes get the same numbering
    per = N // cfg["classes"]
    head = min(nwin, max(4 * N, 65536))
    hx = ds.rows(0, head)
    pick = np.concatenate([np.where(ds.centres[:head] == c)[0][:per] for c in range(cfg["classes"])])
    if len(pick) != per * cfg["classes"]:
        raise SystemExit("not enough samples of every class in the first %d rows" % head)
    codes, clab = hx[pick].copy(), ds.centres[pick].astype(np.int32)
    del hx
    r0, r1 = sharded.shard_rows(len(codes), world, rank)
    cb = E.Codebook(eng, codes[r0:r1], labels=clab[r0:r1], row_offset=r0, n_global=len(codes))
    talpha0 = np.full(r1 - r0, cfg["alpha"], dtype=np.float32)
    mk = lambda: LvqParams(kind, L, cfg["alpha"], 1, cfg["winlen"], cfg["epsilon"], 0, 0, 0)
    lv = None
    if world > 1:
        if kind == 2:
            E.check(eng.lib.somhip_lvq_rates_upload(cb.h, talpha0.ctypes.data_as(C.POINTER(C.c_float))))
        lv = sharded.ShardedLvq(sharded.GpuLvqShard(eng, cb, ds, mk, kind), kind, nwin, xrows=4, max_batch=STEP)
    state = {"talpha": talpha0.copy()}

    def run(it0, count):
        if world == 1:
            state["talpha"], _, _ = E.lvq_train(cb, ds, kind, L, cfg["alpha"], winlen=cfg["winlen"], epsilon=cfg["epsilon"],
                                                talpha=state["talpha"], start_iter=it0, count=count, data_first=it0 % nwin, trace=False)
        else:
            lv.train(L, start_iter=it0, count=count, data_first=it0 % nwin)

    def reset():
        cb.upload(codes[r0:r1])
        state["talpha"] = talpha0.copy()
        if world > 1 and kind == 2:
            E.check(eng.lib.somhip_lvq_rates_upload(cb.h, talpha0.ctypes.data_as(C.POINTER(C.c_float))))

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for k in range(W):
        run(k * STEP, STEP)
    reset()
    eng.timing(True)
    if not a.events_all:                 # events only around the kernels the roofline lines are made from (see bench_som)
        eng.timing_select({"k_rerank", "k_dist_mfma_bf16", "k_lvq_components", "k_dist_l2"})
    eng.timing_reset()
    st0 = eng.lvq_stats()
    barrier()
    t0 = time.perf_counter()
    if world == 1:
        run(0, K * STEP)                 # the K steps in one call: the engine's loop launches them without waiting for the host
    else:
        for k in range(K):
            run(k * STEP, STEP)
    barrier()
    elapsed = time.perf_counter() - t0
    eng.timing(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    table = eng.timing_table()
    st1 = eng.lvq_stats()
    table_all = table
    if not a.events_all and world == 1:  # the same K steps again with every kernel evented, outside the timed region
        reset()
        eng.timing_select(None)
        eng.timing(True)
        eng.timing_reset()
        run(0, K * STEP)
        eng.sync()
        eng.timing(False)
        table_all = eng.timing_table()

    full = exact = None
    if world == 1 and not a.no_full_run:
        nfull = min(L, 2 * nwin)
        reset()
        eng.sync()
        t2 = time.perf_counter()
        run(0, nfull)
        eng.sync()
        secs = time.perf_counter() - t2
        ne = min(20000, nwin)
        dse = E.Dataset(eng, ds.rows(0, ne))
        wi, _, _ = E.find_winners(cb, dse)
        acc = float((clab[wi[:, 0]] == ds.centres[:ne]).mean())
        dse.close()
        full = {"iterations": nfull, "seconds": secs, "value": nfull / secs, "unit": "vectors/s",
                "accuracy_on_first_%d_training_vectors" % ne: acc,
                "note": "the whole schedule" if nfull == L else "the first %d iterations of the %d-iteration schedule" % (nfull, L)}
        # exactness: the batched engine against the one-launch-per-iteration kernel on a prefix, codebook bits
        npre = 4096
        reset(); run(0, npre); got = cb.download(); ta_b = state["talpha"].copy()
        os.environ["SOMHIP_LVQ_ONLINE"] = "1"
        try:
            reset(); run(0, npre); want = cb.download(); ta_o = state["talpha"].copy()
        finally:
            os.environ.pop("SOMHIP_LVQ_ONLINE", None)
        exact = {"iterations": npre, "pass": bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)) and
                                                   np.array_equal(ta_b.view(np.uint32), ta_o.view(np.uint32))),
                 "what": "batched engine vs one launch per iteration (SOMHIP_LVQ_ONLINE): codebook%s bits after %d iterations"
                         % (" and OLVQ1 rate" if kind == 2 else "", npre)}

    out = None
    if rank == 0:
        n_local = r1 - r0
        nb = max(st1["batches"] - st0["batches"], 1)
        pairs = (st1["topk_pairs"] - st0["topk_pairs"])

        def roof_of(kname):
            kl, kms = table[kname]
            avg_s = (kms / max(kl, 1)) * 1e-3
            base = {"kernel": kname, "launches": kl, "avg_launch_ms": avg_s * 1e3, "traffic": None}
            per_launch = K * STEP / max(kl, 1)                       # samples one launch serves
            if kname == "k_dist_mfma_bf16":
                alg = 2.0 * n_local * d * per_launch
                base.update({"bound": "mfma", "achieved": alg / avg_s / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                             "frac": alg / avg_s / 1e12 / PEAK_BF16_TFLOPS, "executed_tflops": 3 * alg / avg_s / 1e12,
                             "note": "split-bf16 distance GEMM (pre-filter of the top-8 search): ALGORITHMIC 2*N*d flop per vector "
                                     "over the dense bf16 peak; the kernel executes 3x that"})
            elif kname == "k_lvq_components":
                alg = 3.0 * d * per_launch * (per_launch + 64) / 2   # pairwise sample distances, lower-triangle tiles
                base.update({"bound": "valu", "achieved": alg / avg_s / 1e12, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                             "frac": alg / avg_s / 1e12 / PEAK_F32_TFLOPS,
                             "note": "sample-to-sample distances of a batch (3*d flop per pair, B(B+64)/2 pairs) + component labelling; "
                                     "fp32 vector ALU, LDS-tiled"})
            elif kname == "k_rerank":
                spl = (K * STEP / max(st1["samples"] - st0["samples"], 1))
                alg = 4.0 * d * 8.0 * per_launch                          # ALGORITHMIC: the 8 rows a sample's list ends up holding, read once
                exe = 4.0 * d * 64.0 * pairs / max(kl, 1) * spl           # executed: one whole 64-row group per (sample, group) pair
                base.update({"bound": "hbm", "achieved": alg / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": alg / avg_s / 1e9 / PEAK_HBM_GBS, "executed_gbs": exe / avg_s / 1e9,
                             "note": "exact re-rank of the row groups the pre-filter kept; achieved = ALGORITHMIC bytes (8 rows of 4*d bytes per "
                                     "sample); executed_gbs counts one 64-row group per (sample, group) pair, %.1f pairs per sample, mostly "
                                     "served by L2 / Infinity Cache -- gather-bound" % (pairs / max(st1["samples"] - st0["samples"], 1))})
            else:
                alg = (4.0 * d * (1 + 2 * knn)) * per_launch        # sample + the rows it corrects, read and written once
                base.update({"bound": "hbm", "achieved": alg / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": alg / avg_s / 1e9 / PEAK_HBM_GBS,
                             "note": "in-order walk: algorithmic bytes = each sample and the <= %d rows it corrects; serial "
                                     "dependency chains inside a component, latency-bound" % knn})
            return base

        ranked = sorted((k for k in table if table[k][0]), key=lambda k: -table[k][1])
        cpu = cpu_baseline_lvq(a, cfg, codes, clab, ds, lab) if (world == 1 and a.cpu_vectors > 0) else None
        value = K * STEP / elapsed
        out = {
            "metric": "training_vectors_per_sec", "value": value, "unit": "vectors/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["workload"], "dim": d, "codebook_rows": N, "step_vectors": STEP, "schedule_length": L,
                       "vectors_timed": K * STEP, "algorithm": cfg["name"], "alpha": cfg["alpha"], "winlen": cfg["winlen"],
                       "epsilon": cfg["epsilon"],
                       "stream": "gen:k=%d,dim=%d,n=%d,seed=%d,labels=1; initial codes = the first %d samples of each class"
                                 % (cfg["classes"], d, nwin, cfg["seed"], per),
                       "schedule": "the reference's online loop, exact (speculative batches of <= 1024 iterations, independent components walked side by side)",
                       "parallelism": "codebook row-sharded over %d ranks (top-8 all-gather, candidate rows all-reduce, replicated walk)" % world
                       if world > 1 else "single GPU", "commit": git_head()},
            # the line's roofline is the WHOLE step's: SURVEY 8(d)'s algorithmic 2*N*d flop per vector at the measured rate over the
            # dense bf16 peak (a per-kernel line that counts executed, cache-served re-reads flatters); the
            # kernels' own lines follow in roofline_kernels
            "roofline": {"bound": "mfma", "achieved": 2.0 * N * d * value / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": 2.0 * N * d * value / 1e12 / PEAK_BF16_TFLOPS, "traffic": None, "kernel": "whole step (all kernels of a batch)",
                         "note": "algorithmic 2*N*d flop per training vector (winner search in GEMM form; the update is 3*d flop per corrected row: "
                                 "nothing) x vectors/s over the dense bf16 peak; the step is NOT matrix-bound: exact top-8 re-rank, the "
                                 "independent-component walk and launch gaps are most of it (roofline_kernels, kernels_ms)"},
            "roofline_kernels": [roof_of(k) for k in ranked[:3]],
            "cpu_baseline": cpu,
            "full_run": full,
            "exact_check": exact,
            "lvq_stats": {"batches": nb, "samples_per_batch": (st1["samples"] - st0["samples"]) / nb,
                          "components_per_batch": (st1["components"] - st0["components"]) / nb,
                          "longest_walk": (st1["largest"] - st0["largest"]) / nb,
                          "stop_list": st1["stop_list"] - st0["stop_list"], "stop_cache": st1["stop_cache"] - st0["stop_cache"]},
            "kernels_ms": {k: {"launches": v[0], "total_ms": round(v[1], 3)} for k, v in table_all.items() if v[0]},
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def cpu_baseline_lvq(a, cfg, codes, clab, ds, lab):
    """The reference's own olvq1_training / lvq3_training (oracle/_ref) on the first vectors of the same stream, 1 core."""
    try:
        import oracle
    except Exception as exc:                                  # pragma: no cover
        return {"error": "oracle package not importable: %s" % exc}
    n = cfg["cpu_vectors"] if a.cpu_vectors == 400 else a.cpu_vectors
    x = ds.rows(0, n)
    dl = ds.centres[:n].astype(np.int32)
    kw = dict(winlen=cfg["winlen"], epsilon=cfg["epsilon"]) if cfg["kind"] >= 3 else {}
    if oracle.ref_available():
        ref = oracle.RefHarness()
        ref.lvq_train(cfg["kind"], codes, clab, x, dl, n, cfg["alpha"], trace=False, **kw)
        secs, kind = ref.last_seconds, "reference"
    else:
        orc = oracle.Oracle()
        t0 = time.perf_counter()
        orc.lvq_train(cfg["kind"], codes, clab, x, dl, n, cfg["alpha"], trace=False, **kw)
        secs, kind = time.perf_counter() - t0, "port"
    return {"value": n / secs, "unit": "vectors/s", "cores": 1, "kind": kind, "host_cores": os.cpu_count(),
            "sample": "%s_training (the reference's own code, gcc -O3 -ffp-contract=off) on the first %d vectors of the same stream, "
                      "same initial codes; epoch loop only (%.1f s)" % (cfg["name"], n, secs)}


if __name__ == "__main__":
    main()
