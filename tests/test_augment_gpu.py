"""fl_augment_batch (transform_train / transform_test, v2/model_train_test.py:24-35, on a uint8 cache in
HBM) against oracle/augment_oracle.py - byte/integer work, so the bar is BIT-EXACT - and the cache loader that replaces
DataLoader(v2:1303). The oracle itself is pinned against Pillow in tests/test_augment_cpu.py."""
import numpy as np
import pytest
import torch

from oracle import augment_oracle as AO
from oxford_102_flower_gan_vae_latent_diffusion_amd import data as D, ops

pytestmark = pytest.mark.gpu


def _cache(n, seed, smooth_every=3):
    r = np.random.RandomState(seed)
    c = r.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:64, 0:64]
    for i in range(0, n, smooth_every):  # smooth, partly saturated images: long runs of equal bytes + clipping in the blends
        a = np.stack([(np.sin(xx / (5.0 + i % 7) + ch) * np.cos(yy / 6.0 - ch) * 150 + 128) for ch in range(3)], -1)
        c[i] = np.clip(a, 0, 255).astype(np.uint8)
    return c


def _run(cache, index, params, dev):
    out = ops.augment_batch(torch.from_numpy(cache).to(dev),
                            None if index is None else torch.from_numpy(index).to(dev),
                            None if params is None else torch.from_numpy(params).to(dev))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_random_draws_bit_exact(dev):
    cache = _cache(50, 1)
    g = torch.Generator().manual_seed(3)
    params = D.draw_train_params(37, g).numpy()
    index = torch.randint(0, 50, (37,), generator=g).numpy()  # repeats allowed
    got = _run(cache, index, params, dev)
    ref = AO.augment_batch(cache, index, params)
    assert got.dtype == np.float32 and got.shape == (37, 3, 64, 64)
    assert np.array_equal(got, ref)


def test_edge_records_bit_exact(dev):
    cache = _cache(8, 2, smooth_every=2)
    recs = [
        D.pack_record(False, 0.0, (0, 1, 2, 3), (1.0, 1.0, 1.0)),          # identity: every blend returns the image
        D.pack_record(True, 0.0, (3, 3, 3, 3), (1.0, 1.0, 1.0)),           # flip only
        D.pack_record(False, 10.0, (3, 2, 1, 0), (0.8, 0.8, 0.8)),         # range ends
        D.pack_record(True, -10.0, (1, 0, 3, 2), (1.2, 1.2, 1.2)),
        D.pack_record(False, 3.0, (0, 1, 2, 3), (0.0, 0.0, 0.0)),          # factor 0: the degenerate image itself
        D.pack_record(False, -7.5, (2, 1, 0, 3), (2.0, 1.9, 2.5)),         # extrapolation beyond the jitter range: clipping
        D.pack_record(True, 9.99, (1, 2, 0, 3), (-0.5, 0.3, -1.0)),        # negative factors clip at 0
        D.pack_record(True, 5.0, (0, 1, 2, 3), (0.9, 1.1, 0.95), plain=True),  # plain flag wins: ToTensor only
        D.pack_record(False, 45.0, (3, 3, 3, 3), (1, 1, 1)),               # large angle: corners zero-filled
    ]
    params = np.stack(recs)
    index = (np.arange(len(recs)) % 8).astype(np.int64)
    got = _run(cache, index, params, dev)
    ref = AO.augment_batch(cache, index, params)
    for i in range(len(recs)):
        assert np.array_equal(got[i], ref[i]), "record %d" % i
    assert np.array_equal(got[7], cache[7].transpose(2, 0, 1).astype(np.float32) / np.float32(255))
    assert np.array_equal(got[1], (cache[1][:, ::-1].transpose(2, 0, 1).astype(np.float32) / np.float32(255)))


def test_plain_paths_and_ragged_sizes(dev):
    cache = _cache(5, 4)
    plain = cache.transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255)
    assert np.array_equal(_run(cache, None, None, dev), plain)                         # transform_test on the whole cache
    assert np.array_equal(_run(cache, np.array([4, 0], dtype=np.int64), None, dev), plain[[4, 0]])
    assert np.array_equal(_run(cache, np.array([2], dtype=np.int64), None, dev), plain[[2]])   # B = 1
    out = ops.augment_batch(torch.from_numpy(cache).to(dev), torch.zeros(0, dtype=torch.int64, device=dev), None)
    assert tuple(out.shape) == (0, 3, 64, 64)                                          # empty batch
    # out-of-range indices are clamped (device data cannot be validated on the host)
    got = _run(cache, np.array([-3, 99], dtype=np.int64), None, dev)
    assert np.array_equal(got, plain[[0, 4]])


def test_argument_errors(dev):
    cache = torch.zeros(4, 64, 64, 3, dtype=torch.uint8, device=dev)
    with pytest.raises((RuntimeError, ValueError)):
        ops.augment_batch(cache.cpu())                                                # no CPU path
    with pytest.raises((RuntimeError, ValueError)):
        ops.augment_batch(torch.zeros(4, 32, 32, 3, dtype=torch.uint8, device=dev))   # not the 64x64 cache
    with pytest.raises((RuntimeError, ValueError)):
        ops.augment_batch(cache, torch.zeros(2, dtype=torch.int64, device=dev), torch.zeros(3, 16, dtype=torch.int32, device=dev))
    with pytest.raises((RuntimeError, ValueError)):
        ops.augment_batch(cache, torch.zeros(2, dtype=torch.int32, device=dev))       # index must be int64


def test_full_batch_vs_oracle_and_flip_involution(dev):
    """BASELINE size: one batch of 256 drawn from the 1020-image train split, checked against the oracle, plus a
    size-independent property on the whole split: flipping twice (re-quantised in between) returns the cache."""
    cache = _cache(1020, 5, smooth_every=17)
    g = torch.Generator().manual_seed(11)
    index = torch.randperm(1020, generator=g)[:256].numpy()
    params = D.draw_train_params(256, g).numpy()
    got = _run(cache, index, params, dev)
    assert np.array_equal(got, AO.augment_batch(cache, index, params))
    flip = np.stack([D.pack_record(True, 0.0, (3, 3, 3, 3), (1, 1, 1))] * 1020)
    once = _run(cache, None, flip, dev)
    back = np.ascontiguousarray(np.rint(once * 255.0).astype(np.uint8).transpose(0, 2, 3, 1))
    twice = _run(back, None, flip, dev)
    assert np.array_equal(twice, cache.transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255))


def test_loader_contract(dev):
    cache = torch.from_numpy(_cache(150, 6))
    labels = torch.arange(150) % 102
    ld = D.FlowersCacheLoader(cache, labels, batch_size=64, shuffle=True, train=True, device=dev, seed=9)
    assert len(ld) == 3
    seen, sizes = [], []
    for data, lab in ld:
        assert data.is_cuda and data.dtype == torch.float32 and tuple(data.shape[1:]) == (3, 64, 64)
        assert lab.dtype == torch.int64 and lab.shape[0] == data.shape[0]
        assert 0.0 <= float(data.min()) and float(data.max()) <= 1.0
        sizes.append(data.shape[0])
        seen.append(lab.cpu())
    assert sizes == [64, 64, 22]                                                      # short last batch, like DataLoader
    assert sorted(torch.cat(seen).tolist()) == sorted(labels.tolist())                 # every image exactly once
    # same seed -> same epoch, bit for bit; the next epoch differs
    a = [d.cpu() for d, _ in D.FlowersCacheLoader(cache, labels, 64, True, True, dev, seed=9)]
    b_ld = D.FlowersCacheLoader(cache, labels, 64, True, True, dev, seed=9)
    b = [d.cpu() for d, _ in b_ld]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    c = [d.cpu() for d, _ in b_ld]
    assert not torch.equal(b[0], c[0])
    # eval loader = transform_test: no shuffle, ToTensor only
    ev = D.FlowersCacheLoader(cache, labels, 64, shuffle=False, train=False, device=dev)
    d0, l0 = next(iter(ev))
    assert torch.equal(d0.cpu(), cache[:64].permute(0, 3, 1, 2).float() / 255)
    assert torch.equal(l0.cpu(), labels[:64])
    # data parallel: two ranks walk disjoint halves of the same permutation
    r0 = torch.cat([l.cpu() for _, l in D.FlowersCacheLoader(cache, torch.arange(150), 32, True, True, dev, seed=4, rank=0, world=2)])
    r1 = torch.cat([l.cpu() for _, l in D.FlowersCacheLoader(cache, torch.arange(150), 32, True, True, dev, seed=4, rank=1, world=2)])
    assert len(r0) == len(r1) == 75 and not set(r0.tolist()) & set(r1.tolist())


def test_loader_feeds_the_autoencoder(dev):
    """The loader's batch goes straight into the drop-in module surface (autoencoder(data), v2:1069)."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M
    cache = torch.from_numpy(_cache(8, 7))
    ld = D.FlowersCacheLoader(cache, torch.arange(8), batch_size=4, device=dev, seed=1)
    ae = M.SimpleAutoencoder().to(dev).eval()
    data, lab = next(iter(ld))
    with torch.no_grad():
        recon, mu, logvar, z = ae(data)
    assert tuple(recon.shape) == (4, 3, 64, 64) and tuple(mu.shape) == (4, 256) and torch.isfinite(recon).all()
