#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace result database.
  python tools/trace_gaps.py <results.db> [first_kernel_substring]
Prints, for the steady-state part of the run, each kernel's average duration and the average idle
time on the GPU before it starts (previous kernel's end -> this kernel's start)."""
import collections
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    short = lambda n: n.split("somhip::")[1].split("(")[0].split("<")[0] if "somhip::" in n else n.split("(")[0][-40:]
    rows = [(short(n), s, e) for n, s, e in rows]
    half = len(rows) // 2
    rows = rows[half:]                       # steady state: second half of the run
    dur = collections.defaultdict(list)
    gap = collections.defaultdict(list)
    for (pn, ps, pe), (n, s, e) in zip(rows, rows[1:]):
        dur[n].append(e - s)
        gap[n].append(max(0, s - pe))
    tot_d = tot_g = 0.0
    print("%-34s %6s %10s %10s" % ("kernel", "calls", "avg us", "gap before us"))
    for n in sorted(dur, key=lambda k: -sum(dur[k])):
        d, g = sum(dur[n]) / len(dur[n]) / 1e3, sum(gap[n]) / len(gap[n]) / 1e3
        tot_d += sum(dur[n]); tot_g += sum(gap[n])
        print("%-34s %6d %10.1f %10.1f" % (n, len(dur[n]), d, g))
    print("busy %.1f ms, idle %.1f ms (%.1f %%)" % (tot_d / 1e6, tot_g / 1e6, 100.0 * tot_g / (tot_d + tot_g)))


if __name__ == "__main__":
    main()
