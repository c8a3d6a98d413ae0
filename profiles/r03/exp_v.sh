#!/bin/sh
# r03_v: round 2's tree (commit 9730847, extracted to profiles/dbg/r02_tree by `git archive`) against this round's, same box,
# alternating: k_fused ms per 36000 samples at 256 / 800 / 4096 / 8192 channels (each tree's own sweep_channels.py and library)
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for tree in profiles/dbg/r02_tree .; do
    printf "%s " "$tree"
    (cd $tree && timeout 400 python profiles/sweep_channels.py --channels 256 800 4096 8192 --steps 8 2>/dev/null | grep '^{' | sed 's/.*"channels": \([0-9]*\).*"kernel_ms": \([0-9.]*\).*/\1:\2/' | tr '\n' ' ')
    echo
  done
done
