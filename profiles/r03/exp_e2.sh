#!/bin/sh
# r03: Costas loop's phase wrap as rint + fma (recurrence loop 67 -> 65 slots per symbol): before (lib_pre_e) / after (product)
cd $GRAFT_REPO_ROOT
for ch in 1024 4096 8192; do
  echo "== $ch x 36000"
  for round in 1 2 3; do
    for lib in profiles/dbg/lib_pre_e.so sdrpp-tetra-demodulator_amd/libtetra_demod_hip.so; do
      printf "%s " $lib
      TETRA_DEMOD_LIB=$lib timeout 120 python profiles/sweep_channels.py --channels $ch --steps 10 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
    done
  done
done
