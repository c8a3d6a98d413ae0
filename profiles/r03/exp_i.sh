#!/bin/sh
# r03_i: time-major against channel-major over channel counts (is the 5 % at 4096 channels a power-of-two row stride effect?)
cd $GRAFT_REPO_ROOT
for lay in channel_major time_major; do
  timeout 400 python profiles/sweep_channels.py --channels 800 1024 4000 4096 4112 --layout $lay --steps 8 2>/dev/null | grep '^{' | sed 's/"workgroups_of_16.*"msamples_s"/"msamples_s"/'
done
