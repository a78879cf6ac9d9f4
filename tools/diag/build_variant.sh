#!/bin/bash
# Diagnostic build of ONE kernel source with extra -D flags, linked against the product objects:  tools/diag/build_variant.sh <name> <source.hip> <flags...>
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift 2
O=vae-gam_amd/_obj
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c vae-gam_amd/csrc/$src -o tools/diag/${base}_${name}.o
objs=$(ls $O/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs tools/diag/${base}_${name}.o -o tools/diag/libvg_${name}.so
echo built tools/diag/libvg_${name}.so
