#!/bin/bash
# BASELINE.json configs[1] through the command-line tools on one MI355X: 32x32 hexa bubble map, 100 000 vectors x 128
# of the seeded generator stream, -rlen 100000 -alpha 0.05 -radius 10; the online schedule (the reference's algorithm,
# byte-exact: tests/test_cli_tools.py::test_c2_full_size_matches_reference_cli) and mini-batch schedules.
# Wall time per tool (data generation on the host and PCIe included).
set -e -o pipefail
B=$(dirname $0)/../som_lvq_pak_amd/host/bin
T=${TMPDIR:-/tmp}/c2_$$; mkdir -p $T
SPEC="gen:k=16,dim=128,n=100000,seed=1234"
t() { local s=$(date +%s%N); "$@"; local e=$(date +%s%N); printf '   [%d.%02d s] %s %s\n' $(( (e - s) / 1000000000 )) $(( (e - s) / 10000000 % 100 )) "$(basename $1)" "$TAG"; }
TAG=""; t $B/datconv -din $SPEC -dout $T/c2.f32 -v 0
t $B/randinit -din $T/c2.f32 -cout $T/init.cod -xdim 32 -ydim 32 -topol hexa -neigh bubble -rand 7 -v 0
for BATCH in 1 64 256 1024; do
  TAG="-batch $BATCH"; t $B/vsom -din $T/c2.f32 -cin $T/init.cod -cout $T/out$BATCH.cod -rlen 100000 -alpha 0.05 -radius 10 -batch $BATCH -v 0
  printf '      qerror '; $B/qerror -din $T/c2.f32 -cin $T/out$BATCH.cod -v 0
done
rm -rf $T
