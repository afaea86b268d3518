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
    for V in a.shards:
        shards = []
        for r in range(V):
            if a.layout == "interleaved":
                mine = E.shard_units(a.xdim, a.ydim, r, V, eng.lib)
                cb = E.Codebook(eng, init[mine], E.TOPOL_HEXA, E.NEIGH_BUBBLE, a.xdim, a.ydim, interleave=(r, V))
            else:
                r0, r1 = sharded.shard_rows(N, V, r)
                mine = np.arange(r0, r1)
                cb = E.Codebook(eng, init[r0:r1], E.TOPOL_HEXA, E.NEIGH_BUBBLE, a.xdim, a.ydim, row_offset=r0, n_global=N)
            shards.append((mine, None, cb, sharded.GpuShard(
                eng, cb, ds, lambda: SomParams(length, 0.05, radius, E.ALPHA_LINEAR, 0, 0, B, 0, 0, 0), B)))

        def step(k):
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
            eng.timing(timed)
            eng.timing_reset()
            eng.sync()
            t0 = time.perf_counter()
            for k in range(K):
                step(k)
            eng.sync()
            return time.perf_counter() - t0

        run(False)                       # warm-up (allocations, code objects)
        wall = run(False)
        run(True)
        table = eng.timing_table()
        eng.timing(False)
        ne = min(8192, length)
        ks = []
        for s in shards:
            kk = s[3].winner_keys(0, min(ne, B))
            eng.sync()
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
        print(json.dumps({"shards": V, "layout": a.layout, "rows_per_shard": len(shards[0][0]),
                          "ms_per_step_per_rank": round(1e3 * wall / (K * V), 4),
                          "speedup_excl_collective": None,
                          "kernel_us_per_step_per_rank": dict(sorted(per_rank.items(), key=lambda kv: -kv[1])),
                          "kernel_sum_us": round(sum(per_rank.values()), 1),
                          "qerror": q, "same_bits_as_first": q == ref_q}))
        del shards


if __name__ == "__main__":
    main()
