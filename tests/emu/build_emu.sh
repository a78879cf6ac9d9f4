#!/bin/bash
# TEST HARNESS ONLY: compile the HIP kernel sources for the host (g++ -DVG_EMU) so that index
# arithmetic can be checked (optionally under AddressSanitizer) without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../vae-gam_amd/csrc"
SAN=""
if [ "$1" = "asan" ]; then SAN="-fsanitize=address -fno-omit-frame-pointer"; fi
OUT="$HERE/libvaegam_emu.so"
g++ -std=c++20 -O1 -g -fPIC -shared -DVG_EMU $SAN -I"$HERE" -I"$SRC" \
    -x c++ "$SRC/vg_api.hip" "$SRC/vg_conv.hip" "$SRC/vg_wgrad.hip" "$SRC/vg_bn.hip" "$SRC/vg_gam.hip" "$SRC/vg_chol.hip" "$SRC/vg_latent.hip" "$SRC/vg_gp.hip" "$SRC/vg_conv_mm.hip" "$SRC/vg_fc.hip" \
    -x c++ "$HERE/hip_emu.cpp" -lpthread -o "$OUT"
echo "built $OUT"
