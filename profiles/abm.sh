#!/bin/sh
# several builds x several channel counts on one box, two rounds, alternating: gpurun -- 'sh profiles/abm.sh "800 4096 8192" lib1.so lib2.so ...'
CHS=$1; shift
for round in 1 2; do
  for ch in $CHS; do
    for lib in "$@"; do
      printf "%s %s " "$ch" "$lib"
      TETRA_DEMOD_LIB=$lib python profiles/sweep_channels.py --channels $ch --steps 10 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
    done
  done
done
