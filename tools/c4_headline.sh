#!/bin/bash
# The headline run of bench.py through the command-line tools on one MI355X: BASELINE.json configs[3] at its real length --
# 256x256 hexa bubble map, dim 512, 10 000 000 vectors of the seeded generator stream (made in HBM by every tool that reads a
# gen: source), randinit -rand 7, vsom -alpha 0.05 -radius 128 with the engine's own batch schedule (-batch auto) and the
# GEMM-form update (SOMHIP_UPDATE_MODE=gemm), then qerror over the first 262 144 vectors of the stream: the number
# bench.py prints as full_run.final_qerror.  Wall time per tool (process start and engine creation included).
set -e -o pipefail
N=${1:-10000000}
B=$(dirname $0)/../som_lvq_pak_amd/host/bin
T=${TMPDIR:-/tmp}/c4h_$$; mkdir -p $T
t() { local s=$(date +%s%N); "$@"; local e=$(date +%s%N); printf '   [%d.%02d s] %s\n' $(( (e - s) / 1000000000 )) $(( (e - s) / 10000000 % 100 )) "$(basename $1)"; }
G="gen:k=256,dim=512,n=$N,seed=3456"
echo "configs[3]: 256x256x512 map, $N vectors, -batch auto, update mode gemm"
t $B/randinit -din "$G" -cout $T/init.f32 -xdim 256 -ydim 256 -topol hexa -neigh bubble -rand 7 -v 0
export SOMHIP_UPDATE_MODE=gemm
t $B/vsom -din "$G" -cin $T/init.f32 -cout $T/out.f32 -rlen $N -alpha 0.05 -radius 128 -batch auto -v 0
unset SOMHIP_UPDATE_MODE
t $B/qerror -din "gen:k=256,dim=512,n=262144,seed=3456" -cin $T/out.f32 -v 0
rm -rf $T
