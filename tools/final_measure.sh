set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python3 $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2>/tmp/p1.err) || exit 1
python tools/rocpd_stats.py $(find /tmp/p1 -name "*.db" | head -1) > $O/bench_default_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/p2 -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 30 > $GRAFT_REPO_ROOT/$O/ddpm_step_under_rocprof.json 2>/tmp/p2.err) || exit 1
python tools/rocpd_stats.py $(find /tmp/p2 -name "*.db" | head -1) > $O/ddpm_step_kernel_stats.csv
python tools/rocpd_timeline.py $(find /tmp/p2 -name "*.db" | head -1) > $O/step_timeline.txt
TRAIN=1 FL_LIB_PATH=oxford_102_flower_gan_vae_latent_diffusion_amd/csrc/build/libflowers_tr.so timeout -k 10 200 python tools/program_trace.py > $O/train_program_trace.txt 2>&1
FOLDED=1 FL_LIB_PATH=oxford_102_flower_gan_vae_latent_diffusion_amd/csrc/build/libflowers_tr.so timeout -k 10 200 python tools/program_trace.py > $O/unet_fwd_phase_trace.txt 2>&1
timeout -k 10 120 python tools/program_ab.py > $O/program_ab.json 2>&1
timeout -k 10 200 python tools/conv_h2_bench.py 30 > $O/conv_h2_nine_shapes.txt 2>&1
tail -c 400 $O/bench_n1.json
