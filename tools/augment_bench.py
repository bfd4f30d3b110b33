"""Throughput of the device-side training transform (fl_augment_batch) against its HBM roofline, with the
Pillow pipeline the reference's DataLoader workers would run timed beside it on one host core.

Algorithmic bytes per image: 64*64*3 read (uint8 cache) + 3*64*64*4 written (float32 NCHW) + 64 B record = 61,504 B.
    python tools/augment_bench.py [--batch 256] [--iters 200]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxford_102_flower_gan_vae_latent_diffusion_amd import data as D, ops  # noqa: E402

BYTES_PER_IMAGE = 64 * 64 * 3 + 3 * 64 * 64 * 4 + 64


def pillow_pipeline(cache, index, params):
    """transform_train's random part with Pillow calls (what torchvision's PIL path executes), draws made explicit."""
    from PIL import Image, ImageEnhance
    out = np.empty((len(index), 3, 64, 64), dtype=np.float32)
    enh = (ImageEnhance.Brightness, ImageEnhance.Contrast, ImageEnhance.Color)
    for b, i in enumerate(index):
        rec = params[b]
        pil = Image.fromarray(cache[i])
        if rec["flip"]:
            pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
        pil = pil.rotate(rec["angle"], Image.NEAREST, False, None, fillcolor=0)
        for op in rec["order"]:
            if op < 3:
                pil = enh[op](pil).enhance(rec["fac"][op])
        out[b] = np.asarray(pil, dtype=np.float32).transpose(2, 0, 1) / np.float32(255)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--cpu-images", type=int, default=2048)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    r = np.random.RandomState(0)
    cache_np = r.randint(0, 256, (1020, 64, 64, 3)).astype(np.uint8)
    cache = torch.from_numpy(cache_np).to(dev)
    labels = (torch.arange(1020) % 102).to(dev)
    g = torch.Generator().manual_seed(1)
    res = {"batch": a.batch, "bytes_per_image": BYTES_PER_IMAGE}
    for name, train in (("train", True), ("test", False)):
        index = torch.randperm(1020, generator=g)[: a.batch].to(dev)
        params = D.draw_train_params(a.batch, g).to(dev) if train else None
        out = torch.empty(a.batch, 3, 64, 64, device=dev)
        for _ in range(10):
            ops.augment_batch(cache, index, params, labels=labels, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.augment_batch(cache, index, params, labels=labels, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        res[name] = {"us_per_batch": round(us, 2), "images_per_s": round(a.batch / us * 1e6),
                     "GBps": round(a.batch * BYTES_PER_IMAGE / us * 1e-3, 1),
                     "frac_of_8TBps": round(a.batch * BYTES_PER_IMAGE / us * 1e-3 / 8000, 4)}
    # host side of one batch: drawing the records (the only per-batch CPU work left)
    t0 = time.perf_counter()
    for _ in range(20):
        D.draw_train_params(a.batch, g)
    res["host_draw_us_per_batch"] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
    # CPU baseline: the Pillow pipeline on one core (the reference uses 2 DataLoader workers, v2:1303)
    n = a.cpu_images
    idx = r.randint(0, 1020, n)
    recs = [dict(flip=bool(r.randint(2)), angle=float(np.float32(r.uniform(-10, 10))), order=list(r.permutation(4)),
                 fac=[float(np.float32(f)) for f in r.uniform(0.8, 1.2, 3)]) for _ in range(n)]
    t0 = time.perf_counter()
    pillow_pipeline(cache_np, idx, recs)
    dt = time.perf_counter() - t0
    res["cpu_pillow"] = {"images": n, "cores": 1, "images_per_s": round(n / dt), "note": "random part only; decode+resize excluded on both sides"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
