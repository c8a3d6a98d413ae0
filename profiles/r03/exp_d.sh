#!/bin/sh
# r03_d: what would a faster Costas wave unlock?  (timing only)  base / Costas wave without arithmetic / + FLL waves without middle taps
cd $GRAFT_REPO_ROOT
timeout 600 sh profiles/ab.sh profiles/dbg/lib_base.so profiles/dbg/lib_e0.so profiles/dbg/lib_e0nomid.so profiles/dbg/lib_nomid.so
