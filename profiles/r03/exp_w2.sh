#!/bin/sh
# r03 role placement of the 32-channel shape revisited on the final kernel, 8192 x 36000 (product: {E older, D} SIMD2, {C older, A} SIMD3):
# w_eadc {E,A}{D,C}; w_eacd {E,A}{C older,D}; w_deca {D older,E}{C,A}
cd $GRAFT_REPO_ROOT
timeout 600 sh profiles/abw.sh profiles/dbg/lib_m_base.so profiles/dbg/lib_w_eadc.so profiles/dbg/lib_w_eacd.so profiles/dbg/lib_w_deca.so
