#!/bin/bash
# BASELINE.json configs[3] shape through the command-line tools on one MI355X: 256x256 hexa bubble map, dim 512,
# N vectors of the seeded generator stream (k=256), -alpha 0.05 -radius 128, mini-batch 4096; data and codebooks
# as raw fp32 files (a name ending in .f32), so that no tool spends its time printing or parsing text.
# Wall time per tool (file load, PCIe and engine creation included).   usage: tools/c4_vsom.sh [N]
set -e -o pipefail
N=${1:-524288}
B=$(dirname $0)/../som_lvq_pak_amd/host/bin
T=${TMPDIR:-/tmp}/c4_$$; mkdir -p $T
t() { local s=$(date +%s%N); "$@"; local e=$(date +%s%N); printf '   [%d.%02d s] %s\n' $(( (e - s) / 1000000000 )) $(( (e - s) / 10000000 % 100 )) "$(basename $1)"; }
echo "C4 shape: 256x256x512 map, $N vectors"
t $B/datconv -din "gen:k=256,dim=512,n=$N,seed=3456" -dout $T/c4.f32 -v 0
ls -l $T/c4.f32 | awk '{printf "   raw fp32 data: %.2f GB\n", $5/1e9}'
t $B/randinit -din $T/c4.f32 -cout $T/init.f32 -xdim 256 -ydim 256 -topol hexa -neigh bubble -rand 7 -v 0
t $B/vsom -din $T/c4.f32 -cin $T/init.f32 -cout $T/out.f32 -rlen $N -alpha 0.05 -radius 128 -batch 4096 -v 0
t $B/qerror -din $T/c4.f32 -cin $T/out.f32 -v 0
rm -rf $T
