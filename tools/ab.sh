#!/bin/bash
# Same-box A/B of two builds of libkge_hip.so (box-to-box clock variation is +-5 %, larger than most
# kernel-level gains):   bash tools/ab.sh gpurun_ab_old.so gpurun_ab_new.so [rounds] [-- command ...]
# Build the two variants here (e.g. `git stash; build; cp .../libkge_hip.so gpurun_ab_old.so; git stash pop;
# build; cp ... gpurun_ab_new.so`), then run this through gpurun; the files travel with the snapshot.
A=$1; B=$2; ROUNDS=${3:-2}
shift 3 2>/dev/null
[ "$1" == "--" ] && shift
CMD=${*:-python bench.py --no-cpu-baseline --no-secondary --steps 30}
LIB=torchkge_amd/csrc/libkge_hip.so
for r in $(seq 1 $ROUNDS); do
  for v in "$A" "$B"; do
    cp "$v" $LIB
    echo "$v $($CMD 2>/dev/null | tail -1 | grep -o 'ms_per_step[^,]*\|kernel_ms[^,]*' | head -2 | paste -s -d' ')"
  done
done
