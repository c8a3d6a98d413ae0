#!/bin/sh
# r03_y: symbol-ring rows padded by one entry (timing-wave writes / Costas-wave reads of all channels no longer on the same banks)
cd $GRAFT_REPO_ROOT
echo "== 4096 x 36000"
timeout 600 sh profiles/ab.sh profiles/dbg/lib_before.so profiles/dbg/lib_sring.so
echo "== 8192 x 36000"
timeout 600 sh profiles/abw.sh profiles/dbg/lib_before.so profiles/dbg/lib_sring.so
echo "== 1024 x 36000"
for round in 1 2; do for lib in before sring; do printf "%s " $lib; TETRA_DEMOD_LIB=profiles/dbg/lib_$lib.so timeout 300 python profiles/sweep_channels.py --channels 1024 --steps 8 2>/dev/null | grep '^{' | sed 's/.*"kernel_ms": \([0-9.]*\).*/\1/'; done; done
