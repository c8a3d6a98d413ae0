#!/bin/sh
# Timing-only EXPERIMENT builds of the library (never the product: their output is wrong by construction).  A copy of csrc/ is
# made under profiles/dbg/exp_<name>/, the FLL assembly is re-generated there with the experiment's environment, and the copy
# is compiled with the experiment's -D flags.
#   sh profiles/build_exp.sh <name> "<-D flags>" [ENV=VALUE ...]      -> profiles/dbg/lib_<name>.so
# e.g. sh profiles/build_exp.sh rrc9   "-DTETRA_EXP_ABLATE=2"
#      sh profiles/build_exp.sh nomid  ""  TETRA_EXP_FLL_NO_MIDDLE=1
#      sh profiles/build_exp.sh chanexp "-DTETRA_CHAN_EXPERIMENTS"      (the channeliser's ablation switches 0x100 / 0x200 / 0x400 of cfg.reserved)
#      sh profiles/build_exp.sh cores2 "-DTETRA_EXP_XRING=128 -DTETRA_EXP_YRING=64 -DTETRA_EXP_SRING=32" TETRA_EXP_XRING=128
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
NAME=$1; FLAGS=$2; shift 2
D="$HERE/dbg/exp_$NAME"
rm -rf "$D" && mkdir -p "$D/sdrpp-tetra-demodulator_amd" "$HERE/dbg"
cp -r "$HERE/../sdrpp-tetra-demodulator_amd/csrc" "$D/sdrpp-tetra-demodulator_amd/csrc"
cp -r "$HERE/../include" "$D/include"
env "$@" python3 "$D/sdrpp-tetra-demodulator_amd/csrc/gen_fll_asm.py" > /dev/null
C="$D/sdrpp-tetra-demodulator_amd/csrc"
# EXP_ABLATE_MASK=1 (environment of this script): TETRA_EXP_ABLATE becomes a bit mask in the copy, so that several roles can be
# ablated together: 1 Costas (E), 2 RRC (C), 4 AGC (A), 8 timing (D)
if [ -n "$EXP_ABLATE_MASK" ]; then
    sed -i 's/TETRA_EXP_ABLATE == 1/(TETRA_EXP_ABLATE \& 1)/; s/TETRA_EXP_ABLATE == 2/(TETRA_EXP_ABLATE \& 2)/; s/TETRA_EXP_ABLATE == 3/(TETRA_EXP_ABLATE \& 4)/; s/TETRA_EXP_ABLATE == 4/(TETRA_EXP_ABLATE \& 8)/' "$C/kernel_fused.hpp"
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $FLAGS \
    -fPIC -shared -o "$HERE/dbg/lib_$NAME.so" "$C/tetra_demod.hip" "$C/tetra_chan.hip" "$C/tetra_resamp.hip" "$C/tetra_burst_scan.hip" "$C/tetra_lmac.hip" "$C/tetra_burst_sync.hip" "$C/tetra_rx.hip"
rm -rf "$D"
echo "built $HERE/dbg/lib_$NAME.so"
