#!/usr/bin/env python3
"""shard_rehearsal.py -- what ONE rank of an N-GPU run does per step, measured on one GPU.

bench.py --gpus N needs N GPUs; the development box has one.  This tool holds all V row shards of
the bench workload (256x256x512 SOM, batch 4096) in ONE process on ONE engine stream and runs the
sharded step exactly as sharded.ShardedSom does, shard after shard:

    for every shard:  somhip_batch_winner_keys      (local scan)
    element-wise MIN of the V key arrays            (stands in for the all-reduce)
    for every shard:  somhip_som_batch_update       (local update with the global winners)

Everything is stream-ordered on the engine's stream, so wall time / V is the time one rank of a
V-GPU run spends per step outside the collective (kernels + launch gaps, or the host's launch
rate if that is the limit), and the per-kernel table (HIP events, second pass) shows which
kernels stop shrinking as the shard gets smaller.  The result must equal the unsharded run bit
for bit; the tool checks the final qerror against V = 1.

    python tools/shard_rehearsal.py --shards 1 2 4 8 --steps 32
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--xdim", type=int, default=256)
    ap.add_argument("--ydim", type=int, default=256)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--layout", default="interleaved", choices=["interleaved", "contiguous"])
    ap.add_argument("--exchange", action="store_true",
                    help="the pre-filter's bounds exchanged between the shards (somhip_shard_winner_begin/refine/finish): one engine "
                         "per shard (the phases of a search own their engine's scratch) and a host sync after every phase, so "
                         "the kernel sums (HIP events) are the measure here, not the wall time")
    ap.add_argument("--update", default="gemm", choices=["gemm", "exact"], help="update mode (bench.py's default: gemm)")
    a = ap.parse_args()
    import torch
    from som_lvq_pak_amd import engine as E
    from som_lvq_pak_amd import sharded
    from som_lvq_pak_amd._lib import SomParams

    dev = torch.device("cuda", 0)
    B, K, d = a.batch, a.steps, a.dim
    N = a.xdim * a.ydim
    radius = max(a.xdim, a.ydim) / 2.0
    length = K * B
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    centres = 4.0 * torch.randn(256, d, generator=g, device=dev)
    data = centres[torch.randint(0, 256, (length,), generator=g, device=dev)] + torch.randn(length, d, generator=g, device=dev)
    lo, hi = data.min(0).values, data.max(0).values
    init = (lo + (hi - lo) * torch.rand(N, d, generator=g, device=dev)).cpu().numpy()
    torch.cuda.synchronize()

    eng = E.Engine(0)
    eng.set_update_mode(a.update)
    ds = E.Dataset(eng, device_ptr=data.data_ptr(), n=length, dim=d)
    ext = torch.cuda.ExternalStream(eng.stream, device=dev)
    ref_q = None
    eng0, ds0 = eng, ds
    for V in a.shards:
        shards = []
        engs = [eng0]
        for r in range(V):
            if a.exchange:
                eng = E.Engine(0) if r else eng0
                if r:
                    eng.set_update_mode(a.update)
                    engs.append(eng)
                ds = E.Dataset(eng, device_ptr=data.data_ptr(), n=length, dim=d) if r else ds0
            if a.layout == "interleaved":
                mine = E.shard_units(a.xdim, a.ydim, r, V, eng.lib)
                cb = E.Codebook(eng, init[mine], E.TOPOL_HEXA, E.NEIGH_BUBBLE, a.xdim, a.ydim, interleave=(r, V))
            else:
                r0, r1 = sharded.shard_rows(N, V, r)
                mine = np.arange(r0, r1)
                cb = E.Codebook(eng, init[r0:r1], E.TOPOL_HEXA, E.NEIGH_BUBBLE, a.xdim, a.ydim, row_offset=r0, n_global=N)
            shards.append((mine, None, cb, sharded.GpuShard(
                eng, cb, ds, lambda: SomParams(length, 0.05, radius, E.ALPHA_LINEAR, 0, 0, B, 0, 0, 0), B)))

        def sync_all():
            for e_ in engs:
                e_.sync()
            torch.cuda.synchronize()

        def min_into_all(ts):
            sync_all()
            m = ts[0].clone()
            for o in ts[1:]:
                torch.minimum(m, o, out=m)
            for t_ in ts:
                t_.copy_(m)
            torch.cuda.synchronize()

        def one_by_one(call):                             # shard after shard: kernels of different engines must not overlap,
            out = []                                      # or the HIP events of one would time the others' work too
            for s in shards:
                out.append(call(s[3]))
                s[3].e.sync()
            return out

        def step_exchange(k):
            min_into_all(one_by_one(lambda g: g.winner_begin(k * B, B)))
            min_into_all(one_by_one(lambda g: g.winner_refine(k * B, B)))
            keys = one_by_one(lambda g: g.winner_finish(k * B, B))
            min_into_all(keys)
            one_by_one(lambda g: g.update(k * B, B, k * B, keys[0]))
            return keys[0]

        def step(k):
            if a.exchange:
                return step_exchange(k)
            keys = [s[3].winner_keys(k * B, B) for s in shards]
            with torch.cuda.stream(ext):
                for other in keys[1:]:
                    torch.minimum(keys[0], other, out=keys[0])
            for s in shards:
                s[3].update(k * B, B, k * B, keys[0])
            return keys[0]

        def run(timed):
            for s in shards:
                s[2].upload(init[s[0]])
            for e_ in engs:
                e_.timing(timed)
                e_.timing_reset()
                e_.sync()
            t0 = time.perf_counter()
            for k in range(K):
                step(k)
            for e_ in engs:
                e_.sync()
            return time.perf_counter() - t0

        run(False)                       # warm-up (allocations, code objects)
        wall = run(False)
        run(True)
        table = {}
        for e_ in engs:
            for kname, v in e_.timing_table().items():
                old = table.get(kname, (0, 0.0))
                table[kname] = (old[0] + v[0], old[1] + v[1])
            e_.timing(False)
        ne = min(8192, length)
        ks = []
        for s in shards:
            kk = s[3].winner_keys(0, min(ne, B))
            s[3].e.sync()
            ks.append(kk.clone())
        torch.cuda.synchronize()
        m = ks[0]
        for o in ks[1:]:
            m = torch.minimum(m, o)
        diffs, _ = sharded.unpack_keys(m.cpu().numpy())
        q = float(E.qerror_sum(diffs) / np.float32(len(diffs)))
        if ref_q is None:
            ref_q = q
        per_rank = {k: round(v[1] / (K * V) * 1e3, 2) for k, v in table.items() if v[0]}
        print(json.dumps({"shards": V, "layout": a.layout, "exchange": bool(a.exchange), "rows_per_shard": len(shards[0][0]),
                          "ms_per_step_per_rank": round(1e3 * wall / (K * V), 4),
                          "speedup_excl_collective": None,
                          "kernel_us_per_step_per_rank": dict(sorted(per_rank.items(), key=lambda kv: -kv[1])),
                          "kernel_sum_us": round(sum(per_rank.values()), 1),
                          "qerror": q, "same_bits_as_first": q == ref_q}))
        del shards
        for e_ in engs[1:]:
            e_.close()


if __name__ == "__main__":
    main()
