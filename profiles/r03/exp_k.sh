#!/bin/sh
# r03_k: timing wave on four lanes per channel (one interpolator row per lane): all shapes
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  timeout 300 python profiles/sweep_channels.py --channels 256 800 1024 4096 8192 --steps 8 2>/dev/null | grep '^{' | sed 's/"workgroups_of_16.*"msamples_s"/"msamples_s"/'
done
