#!/bin/sh
# Debug / instrumented build of the library (NOT the product): -DTETRA_DEMOD_DEBUG compiles in the per-role busy-clock
# instrumentation of k_fused and the TETRA_DEMOD_PROFILE=<file> hook that appends one JSON line per launch.  The release
# library (sdrpp-tetra-demodulator_amd/build.py) has neither the getenv nor the instrumented kernel instantiation.
#   sh profiles/build_debug.sh                      -> profiles/dbg/lib_DEBUG.so (git-ignored, travels with gpurun)
#   TETRA_DEMOD_LIB=profiles/dbg/lib_DEBUG.so TETRA_DEMOD_PROFILE=gpurun_out/roles.jsonl python bench.py --no-host-path
# (TETRA_DEMOD_LIB is read by the Python test/bench binding only.)
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
CSRC="$HERE/../sdrpp-tetra-demodulator_amd/csrc"
mkdir -p "$HERE/dbg"
python3 "$CSRC/gen_fll_asm.py" --check
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -DTETRA_DEMOD_DEBUG -fPIC -shared -o "$HERE/dbg/lib_DEBUG.so" \
    "$CSRC/tetra_demod.hip" "$CSRC/tetra_chan.hip" "$CSRC/tetra_burst_scan.hip" "$CSRC/tetra_lmac.hip" "$CSRC/tetra_burst_sync.hip"
echo "built $HERE/dbg/lib_DEBUG.so"
