#!/bin/bash
# timing-only builds of the library with -DFL_PG_ABLATE=n in program.hip (WRONG results; tools/program_ab.py)
set -e
cd "$(dirname "$0")/.."
C=oxford_102_flower_gan_vae_latent_diffusion_amd/csrc
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -I include -DFL_PG_ABLATE=$n -c $C/program.hip -o /tmp/program_abl$n.o
  objs=$(ls $C/build/*.o | grep -v program.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build/libflowers_pabl$n.so $objs /tmp/program_abl$n.o
  echo built $C/build/libflowers_pabl$n.so
done
