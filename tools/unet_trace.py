"""U-Net forward (batch 256) replayed from a hipGraph, for `rocprofv3 --kernel-trace` timelines.
   python tools/unet_trace.py run [reps]           -> replays the captured forward
   python tools/unet_trace.py analyze <trace.csv>  -> per-kernel busy time vs gaps of the LAST replay"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(reps):
    import torch
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    un = M.ConditionalUNet().to(dev).eval()
    B = int(os.environ.get("B", "256"))
    x = torch.empty(B, 256, device=dev); ops.philox_fill(x, 1)
    t = torch.empty(B, dtype=torch.int64, device=dev); ops.philox_randint(t, 2, 1000)
    c = torch.empty(B, dtype=torch.int64, device=dev); ops.philox_randint(c, 3, 102)
    with torch.no_grad():
        un(x, t, c)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            un(x, t, c)
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()


def analyze(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    first = names[-1]
    # the last replay = from the last occurrence of the first kernel of the graph; find the period
    n = len(rows)
    per = None
    for p in range(10, n // 2):
        if names[n - p:] == names[n - 2 * p:n - p]:
            per = p
            break
    if per is None:
        raise SystemExit("no period found")
    last = rows[n - per:]
    t0 = int(last[0]["Start_Timestamp"])
    busy = 0
    prev_end = t0
    print("%-4s %-60s %9s %9s" % ("#", "kernel", "dur us", "gap us"))
    for i, r in enumerate(last):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        print("%-4d %-60s %9.2f %9.2f" % (i, r["Kernel_Name"][:60], (e - s) / 1e3, (s - prev_end) / 1e3))
        prev_end = e
    span = prev_end - t0
    print("kernels %d  span %.1f us  busy %.1f us  gaps %.1f us" % (per, span / 1e3, busy / 1e3, (span - busy) / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    else:
        analyze(sys.argv[2])
