#!/bin/bash
# Diagnostic build of EVERY kernel source with extra -D flags:  tools/diag/build_all_variant.sh <name> <flags...>  -> tools/diag/libvg_<name>.so
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
mkdir -p tools/diag/_obj_$name
for src in vae-gam_amd/csrc/*.hip; do
  base=$(basename $src .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -Iinclude "$@" -c $src -o tools/diag/_obj_$name/$base.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared tools/diag/_obj_$name/*.o -o tools/diag/libvg_${name}.so
echo built tools/diag/libvg_${name}.so
