#!/bin/bash
# timing-only builds of the library with -DFL_H2_ABLATE=n in conv_h2.hip (WRONG results; tools/conv_h2_bench.py)
set -e
cd "$(dirname "$0")/.."
C=oxford_102_flower_gan_vae_latent_diffusion_amd/csrc
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -I include -DFL_H2_ABLATE=$n -c $C/conv_h2.hip -o /tmp/conv_h2_abl$n.o
  objs=$(ls $C/build/*.o | grep -v conv_h2.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build/libflowers_abl$n.so $objs /tmp/conv_h2_abl$n.o
  echo built $C/build/libflowers_abl$n.so
done
