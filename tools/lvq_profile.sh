#!/bin/bash
# rocprofv3 kernel stats of bench.py --config c3 / c5 (LVQ engines): per-kernel durations behind DESIGN.md's K6 table
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/lvq_profile
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in c3 c5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/$C -o s -- python3 $ROOT/bench.py --config $C --steps 20 --warmup 5 --cpu-vectors 0 --no-full-run > $OUT/bench_$C.json 2> $OUT/err_$C.log
  python3 -c "import sys, glob; sys.path.insert(0, '$ROOT/tools'); import rocpd_summary as r; r.kernel_stats(glob.glob('$OUT/$C/**/*_results.db', recursive=True)[0], '$OUT/kernel_stats_$C.csv')"
done
cd $ROOT
