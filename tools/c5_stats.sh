#!/bin/bash
# rocprofv3 kernel stats of bench.py --config c5 (20 batches): gpurun_out/c5_stats/kernel_stats.csv and the top of it on stdout
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/c5_stats
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/p -o s -- python3 $ROOT/bench.py --config c5 --steps 20 --warmup 5 --cpu-vectors 0 --no-full-run > $OUT/bench.json 2> $OUT/err.log
python3 -c "import sys, glob; sys.path.insert(0, '$ROOT/tools'); import rocpd_summary as r; r.kernel_stats(glob.glob('$OUT/p/**/*_results.db', recursive=True)[0], '$OUT/kernel_stats.csv')"
cd $ROOT
python3 - <<PY
import csv
rows=list(csv.reader(open('$OUT/kernel_stats.csv')))
for r in rows[1:22]:
    name=r[0].split('(')[0].replace('void ','').replace('somhip::','')[:44]
    print(f"  {name:46s} calls {int(float(r[1])):5d} total {float(r[2])/1e6:8.3f} ms avg {float(r[3])/1e3:8.1f} us")
PY
rm -rf $OUT/p
