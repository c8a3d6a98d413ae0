#!/bin/sh
# r03_p: with the FLL stream at 57.06 / 73.19 slots -- does the two-pass Costas wave pay on the 16- / 32-channel shapes now?
cd $GRAFT_REPO_ROOT
echo "== 4096 x 36000"
timeout 600 sh profiles/ab.sh profiles/dbg/lib_base.so profiles/dbg/lib_twopass.so
echo "== 8192 x 36000 (32-channel shape)"
timeout 600 sh profiles/abw.sh profiles/dbg/lib_base.so profiles/dbg/lib_twopass.so
