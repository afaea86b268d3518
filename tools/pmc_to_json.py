#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files -> HBM bytes per launch.

  python tools/pmc_to_json.py <fetch_csv> <write_csv> <out.json>

gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3): counters are in KiB; FETCH_SIZE
reports exactly half of the bytes of a wide coalesced stream -> x2; WRITE_SIZE is exact.
"""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "somhip::" not in name:
            continue
        short = name.split("somhip::")[1].split("(")[0].split("<")[0]
        agg[short].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = per_kernel(fetch), per_kernel(write)
    res = {}
    for k in sorted(set(f) | set(w)):
        rd = 2.0 * f.get(k, 0.0) * 1024.0
        wr = w.get(k, 0.0) * 1024.0
        res[k] = {"read_bytes": rd, "write_bytes": wr, "bytes": rd + wr}
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), "
                         "bench.py --steps 4 --warmup 1; FETCH_SIZE x2 (gfx950), per-launch averages",
               "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
