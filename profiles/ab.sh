#!/bin/sh
# A/B of two builds of the library on the same box, alternating, k_fused launch time at 4096 x 36000 (steady clocks):
#   gpurun -- 'sh profiles/ab.sh profiles/dbg/lib_base.so profiles/dbg/lib_new.so'
for round in 1 2 3; do
  for lib in "$@"; do
    printf "%s " "$lib"
    TETRA_DEMOD_LIB=$lib python profiles/sweep_channels.py --channels 4096 --steps 12 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
  done
done
