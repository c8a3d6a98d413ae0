#!/bin/sh
# r03: do the FLL waves run faster as the OLDEST waves of the workgroup?  16-channel shape with eight waves, role table
# F0 = wave 0, F1 = 1 (SIMD0 / SIMD1, beside the role-less waves 4, 5), E = 2 + A = 6 on SIMD2, D = 3 + C = 7 on SIMD3 (q_fold),
# against the product (q_base) and against eight waves with the product's table (q_w8); nomid = without the middle taps
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for n in q_base q_w8 q_fold q_fold_nomid; do
    printf "%s " $n
    TETRA_DEMOD_LIB=profiles/dbg/lib_$n.so timeout 120 python profiles/sweep_channels.py --channels 4096 --steps 10 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
  done
done
