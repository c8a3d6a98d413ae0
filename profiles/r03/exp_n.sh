#!/bin/sh
# r03 ablation matrix on the kernel with the {E,A}{D,C} placement, 4096 x 36000 (timing-only builds, EXP_ABLATE_MASK=1):
# FLL waves without their middle taps (nomid) x roles without their arithmetic: e Costas, c RRC (a ninth), a AGC, d timing
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for n in base nomid e c a d nomid_e nomid_c nomid_a nomid_d nomid_ea nomid_dc; do
    printf "%s " $n
    TETRA_DEMOD_LIB=profiles/dbg/lib_n_$n.so timeout 120 python profiles/sweep_channels.py --channels 4096 --steps 10 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
  done
done
