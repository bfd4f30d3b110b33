#!/bin/bash
# A/B builds of the library: one csrc file recompiled with extra -D flags, linked with the other objects of csrc/build.
# Usage: tools/build_variant_lib.sh NAME FILE.hip "-DFOO=1 -DBAR=2"   ->  csrc/build/libflowers_NAME.so  (use with FL_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
C=oxford_102_flower_gan_vae_latent_diffusion_amd/csrc
name=$1; file=$2; flags=$3
base=$(basename $file .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -I include $flags -c $C/$file -o /tmp/${base}_$name.o
objs=$(ls $C/build/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build/libflowers_$name.so $objs /tmp/${base}_$name.o
echo built $C/build/libflowers_$name.so
