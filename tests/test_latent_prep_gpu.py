"""csrc/latent_prep.hip: the encoder heads' tail (v2:210-221 + reparameterize v2:339-343) and the latent-DDPM step's draws +
q_sample (v2:1260-1261, v2:574-578) as two launches. Checked against torch fp64, against the multi-launch path they replace
(same Philox counters -> the same draws, bit for bit) and through the trainer."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import oxford_102_flower_gan_vae_latent_diffusion_amd.ops as ops  # noqa: E402
from oxford_102_flower_gan_vae_latent_diffusion_amd import engine, rng as rngmod  # noqa: E402


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _heads_case(B, K, L, seed=0):
    g2 = rnd(2, B, K, seed=seed + 1, scale=2.0) + 0.3
    ln = [(rnd(K, seed=seed + 2 + h).abs() + 0.5, rnd(K, seed=seed + 4 + h, scale=0.2)) for h in range(2)]
    lin = [(rnd(L, K, seed=seed + 6 + h, scale=K ** -0.5), rnd(L, seed=seed + 8 + h, scale=0.1)) for h in range(2)]
    return g2, ln, lin


def _heads_ref(g2, ln, lin, eps=1e-5):
    outs = []
    for h in range(2):
        y = F.layer_norm(g2[h].double(), (g2.shape[2],), ln[h][0].double(), ln[h][1].double(), eps)
        y = y * torch.sigmoid(y)
        outs.append(y @ lin[h][0].double().t() + lin[h][1].double())
    return outs


@pytest.mark.parametrize("B,K,L", [(256, 512, 256), (40, 512, 256), (1, 512, 64), (33, 256, 32), (64, 256, 256)])
def test_encoder_heads_vs_fp64_and_unfused(dev, B, K, L):
    """LayerNorm -> Swish -> Linear of both heads in one launch: exact-fp32 MFMA, so <= 3e-6 of the output's max against fp64
    (tolerance of the other fp32 GEMM tests: 2e-5), ragged row tiles included; and within 2e-5 of the launches it replaces."""
    g2, ln, lin = _heads_case(B, K, L)
    ref = _heads_ref(g2, ln, lin)
    d = lambda t: t.to(dev).contiguous()
    mu, lv, z = ops.encoder_heads_fwd(d(g2), tuple(map(d, ln[0])), tuple(map(d, ln[1])), tuple(map(d, lin[0])), tuple(map(d, lin[1])))
    assert z is None
    for got, want in ((mu, ref[0]), (lv, ref[1])):
        err = (got.cpu().double() - want).abs().max().item()
        assert err <= 3e-6 * max(1.0, want.abs().max().item()), err
    for h, got in enumerate((mu, lv)):
        gh, _, _ = ops.layernorm_fwd(d(g2[h]), d(ln[h][0]), d(ln[h][1]), eps=1e-5, act=ops.ACT_SWISH, save=False)
        old = ops.linear(gh, d(lin[h][0]), d(lin[h][1]))
        assert (got - old).abs().max().item() <= 2e-5 * max(1.0, old.abs().max().item())


def test_encoder_heads_reparam_draws_are_the_philox_stream(dev):
    """z = mu + eps * exp(0.5 clamp(logvar)) with eps drawn inside the kernel: the numbers rng.normal would have produced at the same
    counter; the kernel moves the device-resident counter on by fill_count(B * L) and leaves its done word at zero, so a second
    call draws the NEXT numbers."""
    B, K, L = 72, 512, 256
    g2, ln, lin = _heads_case(B, K, L, seed=11)
    d = lambda t: t.to(dev).contiguous()
    args = (d(g2), tuple(map(d, ln[0])), tuple(map(d, ln[1])), tuple(map(d, lin[0])), tuple(map(d, lin[1])))
    r1, r2 = rngmod.Philox(77), rngmod.Philox(77)
    for r in (r1, r2):
        r.normal((10,), dev)  # a counter that is not zero
    start = int(r1.state(dev).item())
    mu, lv, z = ops.encoder_heads_fwd(*args, want_stats=True, rng_seed=r1.seed, rng_state=r1.state(dev))
    eps = r2.normal((B, L), dev)
    want, _ = ops.reparameterize(mu, lv, eps)
    assert torch.equal(z, want)
    assert int(r1.state(dev).item()) == start + rngmod.Philox.fill_count(B * L) == int(r2.state(dev).item())
    _, _, z2 = ops.encoder_heads_fwd(*args, want_stats=True, rng_seed=r1.seed, rng_state=r1.state(dev))
    want2, _ = ops.reparameterize(mu, lv, r2.normal((B, L), dev))
    assert torch.equal(z2, want2) and not torch.equal(z2, z)
    # logvar outside [-2, 10] is clamped (v2:341)
    lin[1] = (lin[1][0] * 40.0, lin[1][1])
    args = (d(g2), tuple(map(d, ln[0])), tuple(map(d, ln[1])), tuple(map(d, lin[0])), tuple(map(d, lin[1])))
    mu, lv, z = ops.encoder_heads_fwd(*args, want_stats=True, rng_seed=r1.seed, rng_state=r1.state(dev))
    assert lv.max().item() > 10 and lv.min().item() < -2
    want, _ = ops.reparameterize(mu, lv, r2.normal((B, L), dev))
    assert torch.equal(z, want) and torch.isfinite(z).all()


def test_encoder_heads_rejects_unsupported_shapes(dev):
    assert not ops.encoder_heads_ok(8, 384, 256) and not ops.encoder_heads_ok(8, 512, 48) and ops.encoder_heads_ok(8, 512, 256)
    g2, ln, lin = _heads_case(4, 512, 48)
    d = lambda t: t.to(dev).contiguous()
    with pytest.raises(RuntimeError, match="unsupported shape"):
        ops.encoder_heads_fwd(d(g2), tuple(map(d, ln[0])), tuple(map(d, ln[1])), tuple(map(d, lin[0])), tuple(map(d, lin[1])))


@pytest.mark.parametrize("B,L,nm", [(256, 256, 598016), (3, 64, 0), (17, 256, 1001)])
def test_diffuse_prep_equals_the_separate_draws(dev, B, L, nm):
    """t, noise, x_t and the dropout scales of one step in one launch == philox_randint, philox_fill (normal, offset B), q_sample
    and philox_fill (dropout) at the same counters: the integers and the draws bit for bit, x_t to fp32 rounding."""
    z = rnd(B, L, seed=5).to(dev)
    n_steps = 1000
    ab = torch.cumprod(1.0 - torch.linspace(1e-4, 0.02, n_steps), 0).to(dev)
    ra, rb, ma, mb = rngmod.Philox(3), rngmod.Philox(3), rngmod.Philox(9), rngmod.Philox(9)
    for r in (ra, rb, ma, mb):
        r.normal((6,), dev)
    masks = torch.empty(nm, device=dev) if nm else None
    t, noise, xt = ops.diffuse_prep(z, ab, n_steps, ra.seed, ra.state(dev), masks=masks, mask_seed=ma.seed,
                                    mask_state=ma.state(dev), p=0.3)
    t0 = rb.randint((B,), dev, n_steps, advance=False)
    e0 = rb.normal((B, L), dev, offset=B, advance=False)
    assert torch.equal(t, t0) and torch.equal(noise, e0)
    x0 = ops.q_sample(z, e0, t0, ab)
    assert (xt - x0).abs().max().item() <= 1e-6 * max(1.0, x0.abs().max().item())
    if nm:
        m0 = mb.dropout((nm,), dev, 0.3, advance=False)
        assert torch.equal(masks, m0)
        assert 0.25 < (masks == 0).float().mean().item() < 0.35
    assert int(ra.state(dev).item()) == int(rb.state(dev).item())   # nothing advanced here: the caller's post hook does


def test_trainer_fused_chain_equals_separate_launches(dev):
    """DDPMTrainer.step with Philox draws: the fused heads / draws path and the FL_FUSED_HEADS=0 chain consume the same counters
    and give the same loss and update (exact-fp32 heads vs bf16x3 split heads: 1e-5)."""
    from oracle import procedural_init as pinit
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, steps

    B = 8
    g = torch.Generator().manual_seed(3)
    data, labels = torch.rand(B, 3, 64, 64, generator=g).to(dev), torch.randint(0, 102, (B,), generator=g).to(dev)
    res = []
    for fused in (True, False):
        engine.FUSED_HEADS = fused
        try:
            ae, un = M.SimpleAutoencoder(), M.ConditionalUNet()
            ae.load_state_dict(pinit.procedural_state_dict(ae.state_dict(), 42))
            un.load_state_dict(pinit.procedural_state_dict(un.state_dict(), 44))
            ae, un = ae.to(dev).eval(), un.to(dev).train()
            for r in (ae.rng, un.rng):
                r.manual_seed(5)
            tr = steps.DDPMTrainer(ae, un, lr=1e-3)
            tr.diffusion.rng.manual_seed(6)
            losses = [float(tr.step(data, labels)) for _ in range(2)]
            tr.check_programs()
            res.append((losses, un.state_dict()["layers.1.2.weight"].clone(),
                        [int(r.state(dev).item()) for r in (ae.rng, un.rng, tr.diffusion.rng)]))
        finally:
            engine.FUSED_HEADS = True
    (la, wa, ca), (lb, wb, cb) = res
    assert ca == cb and ca[0] > 0 and ca[1] > 0 and ca[2] > 0
    for x, y in zip(la, lb):
        assert abs(x - y) <= 1e-5 * max(1.0, abs(y)), (la, lb)
    assert (wa - wb).abs().max().item() <= 2e-4   # Adam: ~lr per step; the two paths agree to a fraction of one update
