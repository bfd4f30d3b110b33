"""Seeded procedural weights for fixtures (TEST INFRASTRUCTURE — see v2_oracle.py header).

No trained checkpoint exists in the reference tree and the autoencoder is 277 MB in fp32, so
fixtures carry only seeds: each tensor is generated from (seed, crc32(key)) with the torch CPU
generator, which is bit-reproducible between the build container and the GPU box (same torch).
The values are scaled so activations stay O(1) through every block and every bias / affine
path is exercised (non-zero biases, non-unit norm weights).
"""
import zlib

import torch


def _gen(seed, key):
    return torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))


def fill(key, shape, seed, dtype=torch.float32):
    g = _gen(seed, key)
    shape = tuple(shape)
    leaf = key.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_mean":
        return (0.1 * torch.randn(shape, generator=g)).to(dtype)
    if leaf == "running_var":
        return (1.0 + 0.2 * torch.rand(shape, generator=g)).to(dtype)
    if leaf == "center_counts":
        return torch.zeros(shape, dtype=dtype)
    if leaf == "class_centers":
        return (0.5 * torch.randn(shape, generator=g)).to(dtype)
    if leaf == "residual_weight":
        return torch.tensor(0.1, dtype=dtype)
    if "embedding" in key and len(shape) == 2:
        return torch.randn(shape, generator=g).to(dtype)
    if len(shape) >= 2:  # conv / linear / in_proj weights
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return (torch.randn(shape, generator=g) * (1.3 / fan_in ** 0.5)).to(dtype)
    if leaf in ("weight",):  # 1-D weight = norm gain
        return (1.0 + 0.1 * torch.randn(shape, generator=g)).to(dtype)
    return (0.05 * torch.randn(shape, generator=g)).to(dtype)  # biases


def procedural_state_dict(template, seed, dtype=torch.float32):
    """template: {key: tensor or shape}. Order-independent."""
    out = {}
    for k, v in template.items():
        shape = tuple(v.shape) if hasattr(v, "shape") else tuple(v)
        out[k] = fill(k, shape, seed, dtype)
    return out


def vgg16_features16_shapes(prefix="feature_extractor."):
    """Parameter shapes of torchvision vgg16().features[:16] (public layout; conv indices 0,2,5,7,10,12,14)."""
    cfg = [(0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256), (12, 256, 256), (14, 256, 256)]
    shapes = {}
    for idx, ci, co in cfg:
        shapes["%s%d.weight" % (prefix, idx)] = (co, ci, 3, 3)
        shapes["%s%d.bias" % (prefix, idx)] = (co,)
    return shapes
