#!/bin/sh
# Builds oracle/_ref/libtetra_burst_ref.so (and libtetra_lmac_ref.so, below) from the REFERENCE's own source files, compiled where it lies under
# /root/reference (nothing is copied, no stand-in headers or stubs are written): src/decoder/src/phy/tetra_burst.c
# holds tetra_find_train_seq() (the training-sequence search, :271-341), the burst builders
# build_sync_c_d_burst() / build_norm_c_d_burst() (:171-269) and tetra_burst_rx_cb() (:343-393); phy/tetra_burst_sync.c holds
# the synchroniser state machine tetra_burst_sync_in() (:54-155); tetra_tdma.c its slot counter.  tetra_burst_rx_cb() calls
# tp_sap_udata_ind() of the lower MAC, which stays undefined in the shared object: the checker loads it with lazy binding
# (RTLD_LAZY) after the test-side recorder below.  Output only into oracle/_ref/ (git-ignored; travels to the GPU box with the snapshot).
# No-op when /root/reference is absent (the GPU box uses the prebuilt file).
set -e
REF=${TETRA_REFERENCE_DIR:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
SRC="$REF/src/decoder/src"
if [ ! -f "$SRC/phy/tetra_burst.c" ]; then
    echo "reference sources not present ($SRC): keeping any prebuilt oracle/_ref"
    exit 0
fi
mkdir -p "$HERE/_ref"
gcc -O2 -std=gnu11 -fPIC -shared -w -I"$SRC" -o "$HERE/_ref/libtetra_burst_ref.so" "$SRC/phy/tetra_burst.c" "$SRC/phy/tetra_burst_sync.c" "$SRC/tetra_tdma.c"
echo "built $HERE/_ref/libtetra_burst_ref.so from $SRC/phy/tetra_burst.c, phy/tetra_burst_sync.c, tetra_tdma.c"
# The one downstream callback of that library, tp_sap_udata_ind, gets a TEST-SIDE recorder (our code, tests/refrec/, compiled
# against the reference's headers; a separate object so that the library above stays reference sources only).  With it the
# reference's own tetra_burst_sync_in() / tetra_burst_rx_cb() run in the tests (tests/test_burst_sync.py).
gcc -O2 -std=gnu11 -fPIC -shared -Wall -I"$SRC" -o "$HERE/_ref/libtetra_tpsap_recorder.so" "$HERE/../tests/refrec/tp_sap_recorder.c"
echo "built $HERE/_ref/libtetra_tpsap_recorder.so (test-side recorder for tp_sap_udata_ind)"
# Lower-MAC channel-coding primitives, again the reference's own files compiled in place:
# scrambler, block (de)interleaver, RCPC (de)puncturer + mother-code encoder, CRC16, and the K=5 rate-1/4 Viterbi decoder
# (viterbi_dec_sb1_wrapper -> conv_cch_decode -> osmo_conv_decode).  tetra_lower_mac.c itself (tp_sap_udata_ind) is NOT
# built: it needs the upper MAC / crypto / codec objects; the checker chains the primitives in the order its lines
# :181-227 call them (oracle/ref_binding.py: lmac_decode).
LM="$SRC/lower_mac"
gcc -O2 -std=gnu11 -fPIC -shared -w -I"$SRC" -o "$HERE/_ref/libtetra_lmac_ref.so" \
    "$LM/tetra_scramb.c" "$LM/tetra_interleave.c" "$LM/tetra_conv_enc.c" "$LM/crc_simple.c" \
    "$LM/viterbi.c" "$LM/viterbi_cch.c" "$LM/osmo_conv.c"
echo "built $HERE/_ref/libtetra_lmac_ref.so from $LM/{tetra_scramb,tetra_interleave,tetra_conv_enc,crc_simple,viterbi,viterbi_cch,osmo_conv}.c"
