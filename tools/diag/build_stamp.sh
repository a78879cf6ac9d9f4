#!/bin/bash
# Diagnostic build: vg_conv_mm.hip with in-kernel phase stamps (-DVG_STAMP), linked against the product objects -> tools/diag/libvg_stamp.so
set -e
cd "$(dirname "$0")/../.."
O=vae-gam_amd/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DVG_STAMP ${VG_EXTRA} -c vae-gam_amd/csrc/vg_conv_mm.hip -o tools/diag/vg_conv_mm_stamp.o
objs=$(ls $O/*.o | grep -v vg_conv_mm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs tools/diag/vg_conv_mm_stamp.o -o tools/diag/libvg_stamp.so
echo built tools/diag/libvg_stamp.so
