#!/bin/bash
# build_variant.sh NAME "EXTRA_FLAGS": builds tools/variants/libsrx_NAME.so from the current sources
set -e
cd "$(dirname "$0")/../super-resolution_amd/csrc"
mkdir -p ../../tools/variants/obj_$1
for f in api conv wgrad elementwise tiles rdb; do
  EX=""; [ $f = wgrad ] && EX="-mllvm -amdgpu-mfma-vgpr-form"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function $EX $2 -c $f.hip -o ../../tools/variants/obj_$1/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 ../../tools/variants/obj_$1/*.o -o ../../tools/variants/libsrx_$1.so
rm -rf ../../tools/variants/obj_$1
echo built $1
