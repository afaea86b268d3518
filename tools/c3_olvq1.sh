#!/bin/bash
# BASELINE.json configs[2] through the command-line tools, end to end on one MI355X: a 100-class Gaussian
# mixture of N labelled 256-dim vectors from the seeded generator (written once as raw fp32), eveninit
# (all-pairs 5-NN of the data against itself on the GPU) -> 10 000 codes, olvq1 over N iterations, accuracy.
# Wall time per tool (file load and PCIe included).   usage: tools/c3_olvq1.sh [N] [codes]
set -e -o pipefail
N=${1:-1000000}; NOC=${2:-10000}
B=$(dirname $0)/../som_lvq_pak_amd/host/bin
T=${TMPDIR:-/tmp}/c3_$$; mkdir -p $T
t() { local s=$(date +%s%N); "$@"; local e=$(date +%s%N); printf '   [%d.%02d s] %s\n' $(( (e - s) / 1000000000 )) $(( (e - s) / 10000000 % 100 )) "$(basename $1)"; }
echo "C3: $N vectors x 256, 100 classes, $NOC codes"
t $B/datconv -din "gen:k=100,dim=256,n=$N,seed=2345,labels=1" -dout $T/c3.f32 -v 0
ls -l $T/c3.f32 | awk '{printf "   raw fp32 file: %.2f GB\n", $5/1e9}'
t $B/eveninit -din $T/c3.f32 -cout $T/init.cod -noc $NOC -v 0
t $B/accuracy -din $T/c3.f32 -cin $T/init.cod -v 0 | grep -E 'Total|s\]'
t $B/olvq1 -din $T/c3.f32 -cin $T/init.cod -cout $T/out.cod -rlen $N -v 0
t $B/accuracy -din $T/c3.f32 -cin $T/out.cod -v 0 | grep -E 'Total|s\]'
rm -rf $T
