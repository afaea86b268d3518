#!/bin/bash
# Runs on the GPU box (from the repo root): the bench line, the rocprofv3 kernel stats of the
# same command, and the PMC passes (each counter set in its own run, kernel-trace only).
# Everything lands under gpurun_out/prof_<tag>/; copy what is to be judged into profiles/.
set -e -o pipefail
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench_default_n1.json 2> $OUT/bench_default_n1.err
tail -c 600 $OUT/bench_default_n1.json; echo
cd /tmp
# kernel stats of the default command's timed region (the untimed legs are switched off: they would only add launches)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $ROOT/bench.py --cpu-vectors 0 --online-vectors 0 --no-full-run > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o p -- python3 $ROOT/bench.py --steps 8 --warmup 1 --cpu-vectors 0 --online-vectors 0 --no-full-run > $OUT/pmc_$C.log 2>&1
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -d $OUT/pmc_SQ -o p -- python3 $ROOT/bench.py --steps 8 --warmup 1 --cpu-vectors 0 --online-vectors 0 --no-full-run > $OUT/pmc_SQ.log 2>&1
cd $ROOT
find $OUT -name "*.csv" | head -20
