#!/bin/sh
# A/B of builds of the library on the same box, k_fused launch time of the 32-channel workgroup shape at 8192 x 36000:
#   gpurun -- 'sh profiles/abw.sh profiles/dbg/lib_a.so profiles/dbg/lib_b.so ...'
for round in 1 2; do
  for lib in "$@"; do
    printf "%s " "$lib"
    TETRA_DEMOD_LIB=$lib python profiles/sweep_channels.py --channels 8192 --shape wide --steps 8 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
  done
done
