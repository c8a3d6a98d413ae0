#!/bin/sh
# r03_c: two 16-channel workgroups per CU (rings cut to 44 KB AND VGPRs capped at 128; timing only, output garbage)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for spec in "base narrow 8192" "cores2v narrow 8192" "base wide 8192" "cores2v narrow 4096" "base narrow 4096"; do
    set -- $spec
    printf "%s %s %s " $1 $2 $3
    TETRA_DEMOD_LIB=profiles/dbg/lib_$1.so timeout 300 python profiles/sweep_channels.py --channels $3 --shape $2 --steps 8 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
  done
done
