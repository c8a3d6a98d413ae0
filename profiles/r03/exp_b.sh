#!/bin/sh
# r03_b: upper bounds for the matrix-pipe item (FIR work taken out of the waves, timing only) and the co-residency experiment
cd $GRAFT_REPO_ROOT
echo "== 4096 x 36000, alternating (ms): base / RRC wave with 1 of 9 chunks / FLL waves without their middle taps / both"
timeout 600 sh profiles/ab.sh profiles/dbg/lib_base.so profiles/dbg/lib_rrc9.so profiles/dbg/lib_nomid.so profiles/dbg/lib_nofir.so
echo "== 8192 x 36000 (ms): base narrow (two rounds) / rings cut to 44 KB narrow (two workgroups per CU?) / base wide"
for round in 1 2; do
  for spec in "base narrow" "cores2 narrow" "base wide"; do
    set -- $spec
    printf "%s %s " $1 $2
    TETRA_DEMOD_LIB=profiles/dbg/lib_$1.so timeout 300 python profiles/sweep_channels.py --channels 8192 --shape $2 --steps 8 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
  done
done
echo "== 4096 x 36000 cores2 narrow (one workgroup per CU, shorter rings: does the ring depth itself matter?)"
TETRA_DEMOD_LIB=profiles/dbg/lib_cores2.so timeout 300 python profiles/sweep_channels.py --channels 4096 --shape narrow --steps 12 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
