#!/bin/sh
# r03_x: AGC output rows padded to 34 samples (the FLL waves' ds_read_b128 of 8 / 16 channels no longer hit the same banks), with and
# without fetching a whole tile's samples at the top of the tile; base = before both
cd $GRAFT_REPO_ROOT
echo "== 4096 x 36000"
timeout 600 sh profiles/ab.sh profiles/dbg/lib_base.so profiles/dbg/lib_padonly.so profiles/dbg/lib_padtile.so
echo "== 8192 x 36000"
timeout 600 sh profiles/abw.sh profiles/dbg/lib_base.so profiles/dbg/lib_padonly.so
echo "== 1024 x 36000"
for round in 1 2; do for lib in base padonly padtile; do printf "%s " $lib; TETRA_DEMOD_LIB=profiles/dbg/lib_$lib.so timeout 300 python profiles/sweep_channels.py --channels 1024 --steps 8 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'; done; done
