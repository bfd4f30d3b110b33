#!/bin/bash
# timing-only build of the library with per-phase stamps in the persistent executor (tools/program_trace.py)
set -e
cd "$(dirname "$0")/.."
C=oxford_102_flower_gan_vae_latent_diffusion_amd/csrc

/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -I include -DFL_PROG_TRACE=1 -c $C/program.hip -o /tmp/program_tr.o
objs=$(ls $C/build/*.o | grep -v program.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o oxford_102_flower_gan_vae_latent_diffusion_amd/csrc/build/libflowers_tr.so $objs /tmp/program_tr.o
echo built oxford_102_flower_gan_vae_latent_diffusion_amd/csrc/build/libflowers_tr.so
