#!/bin/sh
# r03 role placement revisited on the final kernel, 4096 x 36000: {E,A}{D,C} (AGC beside Costas, RRC beside timing) against the
# product's {E,C}{D,A}, with and without the FLL waves' middle taps (timing-only builds)
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for n in m_base p_eadc m_nomid p_nomid_eadc; do
    printf "%s " $n
    TETRA_DEMOD_LIB=profiles/dbg/lib_$n.so timeout 120 python profiles/sweep_channels.py --channels 4096 --steps 10 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'
  done
done
