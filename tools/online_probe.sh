#!/bin/bash
# kernel durations of the online (batch 1) SOM engine at configs[3]: rocprofv3 kernel trace of a short bench run
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/online_probe
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-vectors 0 --no-full-run --online-vectors 16384 > $OUT/bench.json 2> $OUT/err.log
cd $ROOT
python3 -c "import sys; sys.path.insert(0, 'tools'); import rocpd_summary as r; import glob; r.kernel_stats(glob.glob('$OUT/stats/**/*_results.db', recursive=True)[0], '$OUT/kernel_stats.csv')"
ls $OUT
