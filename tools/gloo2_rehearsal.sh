#!/bin/bash
# Two ranks of bench.py on ONE GPU over gloo (RCCL refuses two ranks on one device): the N > 1 code path of the bench --
# shards, the exchanged pre-filter bounds, the key all-reduce, the sharded update -- on real kernels.  The final qerror
# has to equal the N = 1 run's bit for bit.  Timings mean nothing here (host-staged collectives, a shared GPU).
#   bash tools/gloo2_rehearsal.sh [extra bench.py flags]
set -e -o pipefail
OUT=gpurun_out/gloo2
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
FLAGS="--steps 6 --warmup 1 --cpu-vectors 0 --online-vectors 0 --length 400000 $*"
python3 bench.py $FLAGS > $OUT/n1.json 2> $OUT/n1.err
SOMHIP_SHARD_EXCHANGE=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 2 --backend gloo $FLAGS > $OUT/n2.json 2> $OUT/n2.err
SOMHIP_NO_SHARD_EXCHANGE=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 \
    bench.py --gpus 2 --backend gloo $FLAGS > $OUT/n2_plain.json 2> $OUT/n2_plain.err
python3 - <<PY
import json
def last(p):
    return json.loads(open(p).read().strip().splitlines()[-1])
a, b, c = last("$OUT/n1.json"), last("$OUT/n2.json"), last("$OUT/n2_plain.json")
for name, j in (("N=1", a), ("N=2 gloo, bounds exchanged", b), ("N=2 gloo, plain", c)):
    print(name, "final_qerror", j.get("final_qerror"), "full_run", (j.get("full_run") or {}).get("final_qerror"), "rows re-ranked per vector", (j.get("rerank_stats") or {}).get("rows_per_sample"))
assert a.get("final_qerror") == b.get("final_qerror") == c.get("final_qerror"), "qerror differs"
fa, fb = (a.get("full_run") or {}).get("final_qerror"), (b.get("full_run") or {}).get("final_qerror")
assert fa == fb, "full-run qerror differs"
print("same bits")
PY
