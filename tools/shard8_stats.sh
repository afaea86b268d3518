#!/bin/bash
# rocprofv3 kernel stats of the 8-shard rehearsal with the bounds exchanged (one process, eight engines):
# gpurun_out/shard8/kernel_stats.csv -- per-kernel calls and average durations of what ONE rank of an 8-GPU run launches
# per step (divide calls by 8 shards x steps).
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/shard8
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $ROOT/tools/shard_rehearsal.py --batch 32768 --steps 6 --shards 8 --exchange > $OUT/rehearsal.json 2> $OUT/err.log
python3 -c "import sys, glob; sys.path.insert(0, '$ROOT/tools'); import rocpd_summary as r; r.kernel_stats(glob.glob('$OUT/stats/**/*_results.db', recursive=True)[0], '$OUT/kernel_stats.csv')"
cd $ROOT
python3 - <<PY
import csv
rows=list(csv.reader(open('$OUT/kernel_stats.csv')))
for r in rows[1:30]:
    name=r[0].split('(')[0].replace('void ','').replace('somhip::','')[:44]
    print(f"  {name:46s} calls {int(float(r[1])):5d} total {float(r[2])/1e6:8.3f} ms avg {float(r[3])/1e3:8.1f} us")
PY
