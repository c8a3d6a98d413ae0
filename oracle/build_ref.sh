#!/bin/sh
# Builds oracle/_ref/libtetra_burst_ref.so from the REFERENCE's own source file, compiled where it lies under
# /root/reference (nothing is copied, no stand-in headers or stubs are written): src/decoder/src/phy/tetra_burst.c
# holds tetra_find_train_seq() (the training-sequence search, :271-341) and the burst builders
# build_sync_c_d_burst() / build_norm_c_d_burst() (:171-269).  The file also defines tetra_burst_rx_cb(), which calls into
# the lower MAC; those symbols stay undefined in the shared object and are never called -- the checker loads the library
# with lazy binding (RTLD_LAZY).  Output only into oracle/_ref/ (git-ignored; travels to the GPU box with the snapshot).
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
gcc -O2 -std=gnu11 -fPIC -shared -w -I"$SRC" -o "$HERE/_ref/libtetra_burst_ref.so" "$SRC/phy/tetra_burst.c" "$SRC/phy/tetra_burst_sync.c"
echo "built $HERE/_ref/libtetra_burst_ref.so from $SRC/phy/tetra_burst.c"
