"""CPU restatement of the reference's training-image transform (TEST INFRASTRUCTURE ONLY - only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does).

Reference call site: v2/model_train_test.py:24-30 (transform_train) and :32-35 (transform_test), consumed by
DataLoader(flowers_train, batch_size, shuffle=True) at v2:1303. The arithmetic itself lives in third-party code that
is NOT under /root/reference: torchvision (unpinned by the reference, README:98; not installed in this image) calling
Pillow for PIL images. Its published algorithm for a PIL input is restated here:

  Resize((64,64))           -> done once when the uint8 cache is built (Image.resize, bilinear), not restated
  RandomHorizontalFlip()    -> x -> W-1-x with probability 0.5
  RandomRotation(10)        -> Image.rotate(angle, NEAREST, expand=False, center=(W/2,H/2), fill 0), angle ~ U(-10,10):
                               inverse affine map in Pillow's 16.16 fixed point (Geometry.c affine_fixed)
  ColorJitter(.2,.2,.2)     -> brightness / contrast / saturation in a random order, each factor ~ U(0.8,1.2), each an
                               ImageEnhance blend(degenerate, image, factor) on uint8 with Pillow's float32 arithmetic
                               and truncation (Blend.c); grey = (19595 R + 38470 G + 7471 B + 0x8000) >> 16
  ToTensor()                -> HWC uint8 -> CHW float32 / 255

PARITY PIN: torchvision is absent, so the flip/order/ToTensor glue is "parity unpinned" against torchvision itself;
the rotation and the three enhancers ARE pinned - tests/test_augment_cpu.py checks every function below bit-for-bit
against Pillow 12.2.0 (the library torchvision would call), which is installed in this image.

Parameter record per image (16 x int32, the same table the HIP kernel takes - include/flowers_hip.h fl_augment_batch):
  [0] flip  [1..6] a0 a1 a2 a3 a4 a5 (16.16 fixed inverse map, half-pixel offsets folded into a2/a5)
  [7..10] op order (0 brightness, 1 contrast, 2 saturation, 3 none/hue-disabled)
  [11..13] float32 bits of the brightness / contrast / saturation factors   [14] 1 = skip rotation+jitter  [15] 0
"""
import math

import numpy as np

WORDS = 16


def fix16(v):
    """Pillow's FIX(): floor(v*65536 + 0.5)."""
    return int(math.floor(v * 65536.0 + 0.5))


def rotation_fixed_matrix(angle_deg, w=64, h=64):
    """Image.rotate's inverse affine matrix (double, cos/sin rounded to 15 decimals) -> affine_fixed's six integers."""
    angle = angle_deg % 360.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = w / 2, h / 2
    m[2] = m[0] * -cx + m[1] * -cy + m[2] + cx
    m[5] = m[3] * -cx + m[4] * -cy + m[5] + cy
    return (fix16(m[0]), fix16(m[1]), fix16(m[2] + m[0] * 0.5 + m[1] * 0.5),
            fix16(m[3]), fix16(m[4]), fix16(m[5] + m[3] * 0.5 + m[4] * 0.5))


def pack_params(flip, angle_deg, order, factors, w=64, h=64, plain=False):
    rec = np.zeros(WORDS, dtype=np.int32)
    rec[0] = int(bool(flip))
    rec[1:7] = rotation_fixed_matrix(angle_deg, w, h)
    rec[7:11] = list(order)
    rec[11:14] = np.asarray(factors, dtype=np.float32).view(np.int32)
    rec[14] = int(bool(plain))
    return rec


def hflip(img):
    return img[:, ::-1, :]


def rotate_fixed(img, fm):
    """img (H,W,3) uint8; fm = the six 16.16 integers. Nearest neighbour, zero fill."""
    H, W, _ = img.shape
    a0, a1, a2, a3, a4, a5 = (int(v) for v in fm)
    y, x = np.mgrid[0:H, 0:W].astype(np.int64)
    xin = (a2 + y * a1 + x * a0) >> 16
    yin = (a5 + y * a4 + x * a3) >> 16
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def grey(img):
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.int32)


def blend(in1, in2, alpha):
    """Pillow's ImagingBlend on uint8 bands: float32 arithmetic, truncating store; clipping only when extrapolating."""
    alpha = np.float32(alpha)
    i1, i2 = in1.astype(np.int32), in2.astype(np.int32)
    if alpha == 0.0:
        return i1.astype(np.uint8)
    if alpha == 1.0:
        return i2.astype(np.uint8)
    t = i1.astype(np.float32) + (alpha * (i2 - i1).astype(np.float32)).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    out = np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32)))
    return out.astype(np.uint8)


def brightness(img, f):
    return blend(np.zeros_like(img), img, f)


def contrast(img, f):
    g = grey(img)
    mean = int(float(g.sum()) / float(g.size) + 0.5)
    return blend(np.full_like(img, mean), img, f)


def saturation(img, f):
    g = grey(img).astype(np.uint8)
    return blend(np.repeat(g[..., None], 3, axis=2), img, f)


_OPS = (brightness, contrast, saturation)


def apply_record(img, rec):
    """img (H,W,3) uint8, rec int32[16] -> (3,H,W) float32 in [0,1] (transform_train with the draws made explicit)."""
    rec = np.asarray(rec, dtype=np.int32)
    if not rec[14]:
        if rec[0]:
            img = hflip(img)
        img = rotate_fixed(img, rec[1:7])
        fac = rec[11:14].view(np.float32)
        for op in rec[7:11]:
            if 0 <= op < 3:
                img = _OPS[op](img, fac[op])
    return np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)


def augment_batch(cache, index, params):
    """cache (N,H,W,3) uint8, index (B,) or None, params (B,16) int32 or None (= transform_test) -> (B,3,H,W) f32."""
    B = len(index) if index is not None else (len(params) if params is not None else cache.shape[0])
    plain = pack_params(0, 0.0, (3, 3, 3, 3), (1, 1, 1), plain=True)
    out = np.empty((B, 3) + cache.shape[1:3], dtype=np.float32)
    for b in range(B):
        src = cache[int(index[b])] if index is not None else cache[b]
        out[b] = apply_record(src, params[b] if params is not None else plain)
    return out
