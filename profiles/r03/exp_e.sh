#!/bin/sh
# r03_e: the 4-channel workgroup shape (FLL rows of 16 lanes per channel) against the 16-channel one, up to 1024 channels
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for shape in narrow small; do
    timeout 300 python profiles/sweep_channels.py --channels 256 800 1024 --shape $shape --steps 8 2>/dev/null | grep '^{'
  done
done
