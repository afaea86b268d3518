#!/usr/bin/env python3
"""Summaries of the rocprofv3 result databases written by tools/profile_round.sh.

  python tools/rocpd_summary.py gpurun_out/prof_r01 profiles r01

writes  profiles/<tag>_kernel_stats_bench_default.csv   (kernel-trace --stats equivalent)
        profiles/<tag>_pmc_traffic.json + _pmc_hbm_traffic.txt   (FETCH_SIZE / WRITE_SIZE passes)
        profiles/<tag>_pmc_sq_counters.txt                       (SQ pass)
gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE / WRITE_SIZE are KiB;
FETCH_SIZE reports half of the bytes of a wide coalesced stream -> x2; WRITE_SIZE is exact.
"""
import collections
import csv
import json
import sqlite3
import statistics
import sys


def short(name):
    return name.split("somhip::")[1].split("(")[0].split("<")[0] if "somhip::" in name else None


def kernel_stats(db, out):
    c = sqlite3.connect(db)
    agg = collections.defaultdict(list)
    for name, dur in c.execute("select name, duration from kernels"):
        agg[name].append(dur)
    total = sum(sum(v) for v in agg.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), sum(v) / len(v), round(100.0 * sum(v) / total, 2), min(v), max(v),
                        statistics.pstdev(v) if len(v) > 1 else 0.0])


def counters(db):
    c = sqlite3.connect(db)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, ctr, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
        s = short(name)
        if s:
            agg[s][ctr].append(val)
    return {k: {n: sum(v) / len(v) for n, v in d.items()} for k, d in agg.items()}


def main():
    src, dst, tag = sys.argv[1:4]
    commit = sys.argv[4] if len(sys.argv) > 4 else None
    kernel_stats(f"{src}/stats/s_results.db", f"{dst}/{tag}_kernel_stats_bench_default.csv")
    f = counters(f"{src}/pmc_FETCH_SIZE/p_results.db")
    w = counters(f"{src}/pmc_WRITE_SIZE/p_results.db")
    note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), "
            "bench.py --steps 8 --warmup 1 --no-full-run: 8 batches of the engine-chosen schedule (seven of 32768 vectors, one of 8192) "
            "spread over the 10 M-iteration schedule (radius 128 -> 17); FETCH_SIZE x2 (gfx950), per-launch averages")
    bdesc = "engine-chosen (somhip_som_auto_batch: 32768 up to iteration 8486912, 8192 after)"
    res = {}
    for k in sorted(set(f) | set(w)):
        rd = 2.0 * f.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0
        wr = w.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
        res[k] = {"read_bytes": rd, "write_bytes": wr, "bytes": rd + wr}
    json.dump({"source": note, "commit": commit, "batch": bdesc, "schedule": "auto", "kernels": res}, open(f"{dst}/{tag}_pmc_traffic.json", "w"), indent=1, sort_keys=True)
    with open(f"{dst}/{tag}_pmc_hbm_traffic.txt", "w") as o:
        o.write(f"# {note}\n# Calibration: k_rows_to_tiles reads 128 MiB and writes 128 MiB.\n")
        o.write(f"{'kernel':<28}{'read MiB':>14}{'write MiB':>15}\n")
        for k, v in res.items():
            o.write(f"{k:<28}{v['read_bytes'] / 2**20:>14.1f}{v['write_bytes'] / 2**20:>15.1f}\n")
    sq = counters(f"{src}/pmc_SQ/p_results.db")
    with open(f"{dst}/{tag}_pmc_sq_counters.txt", "w") as o:
        o.write("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU "
                "SQ_WAIT_INST_ANY\n#   SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_LDS ; bench.py --steps 8 --warmup 1 --no-full-run; "
                "averages per dispatch\n")
        for k, d in sq.items():
            o.write(k + " " + str({n: f"{v:.4g}" for n, v in sorted(d.items())}) + "\n")
    print("ok")


if __name__ == "__main__":
    main()
