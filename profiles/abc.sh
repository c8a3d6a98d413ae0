#!/bin/sh
# A/B of builds of the library on the same box, alternating, k_fused launch time at a given channel count (steady clocks):
#   gpurun -- 'sh profiles/abc.sh 800 profiles/dbg/lib_a.so profiles/dbg/lib_b.so ...'
CH=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    printf "%s %s " "$CH" "$lib"
    TETRA_DEMOD_LIB=$lib python profiles/sweep_channels.py --channels $CH --steps 12 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
  done
done
