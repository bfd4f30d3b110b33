"""Input pipeline that feeds the training steps.

The reference builds `datasets.Flowers102(..., transform=transform_train)` (v2:1299) and iterates it through
`DataLoader(flowers_train, batch_size=64, shuffle=True, num_workers=2, pin_memory=True)` (v2:1303): PIL decode, resize,
flip, rotate and colour-jitter run per image on two CPU workers, then a pinned H2D copy per batch. Here the deterministic
part (decode + `Resize((64,64))`, v2:25) is done ONCE into a uint8 NHWC cache that lives in HBM (the train split is 1020
images = 12.5 MB), and the random part runs on the device per batch (`fl_augment_batch`, csrc/augment.hip):

    cache = build_cache(pil_images_or_arrays)                     # host, once (PIL does the resize, as torchvision does)
    loader = FlowersCacheLoader(cache, labels, batch_size=64, shuffle=True, train=True, device="cuda:0")
    for data, labels in loader:                                    # float32 (B,3,64,64) in [0,1], int64 (B,)
        ...

Only the per-image draws (flip?, angle, op order, three factors: 16 words per image) are made on the host, with a seeded
`torch.Generator`; the table is one small pinned H2D copy per batch. Image bytes never leave HBM.
There is no CPU path: the transform needs the HIP library (ops.augment_batch raises otherwise)."""
import math

import numpy as np
import torch

IMG = 64            # v2:21 img_size
WORDS = 16          # int32 words per image record (include/flowers_hip.h, fl_augment_batch)
ROT_DEGREES = 10.0  # v2:27 RandomRotation(10)
JITTER = 0.2        # v2:28 ColorJitter(brightness=0.2, contrast=0.2, saturation=0.2)


def build_cache(images, size=IMG):
    """Decode-side, once per dataset: `transforms.Resize((64,64))` (v2:25) on each image -> uint8 (N,64,64,3) NHWC (CPU
    tensor; move it to the device once). `images`: PIL images or HxWx3 uint8 arrays."""
    from PIL import Image

    out = np.empty((len(images), size, size, 3), dtype=np.uint8)
    for i, im in enumerate(images):
        if not isinstance(im, Image.Image):
            im = Image.fromarray(np.asarray(im, dtype=np.uint8))
        out[i] = np.asarray(im.convert("RGB").resize((size, size), Image.BILINEAR))
    return torch.from_numpy(out)


def _fix16(v):
    return math.floor(v * 65536.0 + 0.5)


def _rotation_words(angle_deg, w=IMG, h=IMG):
    """Image.rotate(angle, NEAREST, expand=False) as Pillow sets it up: inverse map about (w/2, h/2) with cos/sin rounded
    to 15 decimals, then the 16.16 fixed-point constants of its nearest-neighbour affine loop (half-pixel folded in)."""
    a = -math.radians(angle_deg % 360.0)
    c, s = round(math.cos(a), 15), round(math.sin(a), 15)
    cx, cy = w / 2, h / 2
    m2 = c * -cx + s * -cy + 0.0 + cx
    m5 = -s * -cx + c * -cy + 0.0 + cy
    return (_fix16(c), _fix16(s), _fix16(m2 + c * 0.5 + s * 0.5), _fix16(-s), _fix16(c), _fix16(m5 + -s * 0.5 + c * 0.5))


def pack_record(flip, angle_deg, order, factors, plain=False):
    """One image's 16-word record from explicit draws (parity tests and callers that bring their own randomness)."""
    rec = np.zeros(WORDS, dtype=np.int32)
    rec[0] = int(bool(flip))
    rec[1:7] = _rotation_words(float(angle_deg))
    rec[7:11] = [int(o) for o in order]
    rec[11:14] = np.asarray(factors, dtype=np.float32).view(np.int32)
    rec[14] = int(bool(plain))
    return rec


def draw_train_params(B, generator=None):
    """The random draws of transform_train for B images -> (B,16) int32 CPU tensor:
    flip ~ Bernoulli(0.5) (RandomHorizontalFlip), angle ~ U(-10,10) float32 (RandomRotation.get_params), the op order
    = randperm(4) and three factors ~ U(0.8,1.2) float32 (ColorJitter.get_params; hue is None -> op 3 is a no-op)."""
    g = generator
    flip = torch.rand(B, generator=g) < 0.5
    angle = torch.empty(B).uniform_(-ROT_DEGREES, ROT_DEGREES, generator=g)
    fac = torch.empty(B, 3).uniform_(1.0 - JITTER, 1.0 + JITTER, generator=g)
    order = torch.argsort(torch.rand(B, 4, generator=g), dim=1)  # a uniform random permutation per row (randperm(4))
    rec = np.zeros((B, WORDS), dtype=np.int32)
    rec[:, 0] = flip.numpy()
    ang = angle.numpy()
    for b in range(B):
        rec[b, 1:7] = _rotation_words(float(ang[b]))
    rec[:, 7:11] = order.numpy()
    rec[:, 11:14] = fac.numpy().view(np.int32)
    return torch.from_numpy(rec)


def epoch_plan(n, batch_size, shuffle=True, generator=None, drop_last=False, rank=0, world=1):
    """Index batches of one epoch, as DataLoader(batch_size, shuffle) walks them (v2:1303); the last batch is short
    unless drop_last. Data parallel: every rank draws the SAME permutation (same generator seed) and keeps the strided
    shard rank::world of it, truncated to a common length, so shards are disjoint and equally long."""
    perm = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
    if world > 1:
        per = n // world
        perm = perm[: per * world][rank::world].contiguous()
    batches = list(torch.split(perm, batch_size))
    if drop_last and batches and len(batches[-1]) < batch_size:
        batches.pop()
    return batches


class FlowersCacheLoader:
    """Iterable with DataLoader's contract for the reference loops (`for data, labels in train_loader`, v2:1060, v2:1254):
    yields device tensors `data` float32 (B,3,64,64) in [0,1] and `labels` int64 (B,).

    cache: uint8 (N,64,64,3) (moved to `device` once and kept there); labels: int64 (N,).
    train=True applies transform_train's random part on the device, train=False is transform_test (ToTensor only).
    seed fixes shuffling and the per-image draws; rank/world shard each epoch for one-process-per-GPU training."""

    def __init__(self, cache, labels, batch_size=64, shuffle=True, train=True, device="cuda:0", seed=0, drop_last=False,
                 rank=0, world=1):
        from . import ops  # loads libflowers_hip.so: fails loudly when it is missing

        self._ops = ops
        self.device = torch.device(device)
        if cache.dtype != torch.uint8 or cache.dim() != 4 or tuple(cache.shape[1:]) != (IMG, IMG, 3):
            raise ValueError("cache must be uint8 (N,%d,%d,3) - see build_cache()" % (IMG, IMG))
        if labels.shape[0] != cache.shape[0]:
            raise ValueError("labels and cache disagree on N")
        self.cache = cache.to(self.device).contiguous()
        self.labels = labels.to(self.device, torch.int64).contiguous()
        self.n = cache.shape[0]
        self.batch_size, self.shuffle, self.train, self.drop_last = int(batch_size), shuffle, train, drop_last
        self.rank, self.world = int(rank), int(world)
        self.epoch = 0
        self.seed = int(seed)
        self._shuffle_gen = torch.Generator().manual_seed(self.seed)               # identical on every rank
        self._draw_gen = torch.Generator().manual_seed(self.seed * 7919 + 1 + self.rank)  # per-rank augmentation draws

    def __len__(self):
        n = (self.n // self.world) if self.world > 1 else self.n
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        plan = epoch_plan(self.n, self.batch_size, self.shuffle, self._shuffle_gen, self.drop_last, self.rank, self.world)
        self.epoch += 1
        for idx in plan:
            index = idx.to(self.device, non_blocking=True)
            params = None
            if self.train:
                params = draw_train_params(len(idx), self._draw_gen).pin_memory().to(self.device, non_blocking=True)
            yield self._ops.augment_batch(self.cache, index, params, labels=self.labels)
