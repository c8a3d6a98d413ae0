#!/bin/sh
# r03_t: what paces the 4-channel shape now?  (timing only; 1024 x 36000)  base / FLL16 block without its middle taps /
# Costas recurrence without arithmetic / timing wave without arithmetic / AGC without arithmetic / RRC wave with 1 of 9 chunks
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for lib in base nomid e0 d0 a0 c9; do
    printf "%s " $lib
    TETRA_DEMOD_LIB=profiles/dbg/lib_$lib.so timeout 300 python profiles/sweep_channels.py --channels 1024 --steps 8 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
  done
done
