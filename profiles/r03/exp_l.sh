#!/bin/sh
# r03_l: with the lighter timing wave -- two-pass Costas wave and / or the folded Cody-Waite step on the 16- and 32-channel shapes
cd $GRAFT_REPO_ROOT
echo "== 4096 x 36000"
timeout 600 sh profiles/ab.sh profiles/dbg/lib_base.so profiles/dbg/lib_twopass.so profiles/dbg/lib_fold.so profiles/dbg/lib_twopassfold.so
echo "== 8192 x 36000 (32-channel shape)"
timeout 600 sh profiles/abw.sh profiles/dbg/lib_base.so profiles/dbg/lib_twopass.so profiles/dbg/lib_fold.so profiles/dbg/lib_twopassfold.so
