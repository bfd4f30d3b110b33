"""Data-parallel path on CPU (gloo, world_size 2): the flat-buffer all-reduce of shard gradients reproduces the
global-batch gradient (the N>1 path of bench.py / steps.DDPMTrainer; RCCL replaces gloo on the GPU box)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import procedural_init as pinit
    from oracle import v2_oracle as O
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M
    from oxford_102_flower_gan_vae_latent_diffusion_amd import steps

    un = M.ConditionalUNet()
    un.load_state_dict(pinit.procedural_state_dict(un.state_dict(), 44))
    sd = {k: v.clone() for k, v in un.state_dict().items()}
    fp = steps.FlatParams(un, ("time_projections.4.", "attention_layers.4."))
    # state_dict still round-trips after re-homing the parameters into the flat buffer
    for k, v in un.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert fp.n_active < fp.n_total
    B = 8
    g = torch.Generator().manual_seed(123)
    z, eps = torch.randn(B, 256, generator=g), torch.randn(B, 256, generator=g)
    t, c = torch.randint(0, 1000, (B,), generator=g), torch.randint(0, 102, (B,), generator=g)
    sched = O.schedule()

    def grads(sl):
        sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        loss, _ = O.ddpm_loss(sdg, sched, z[sl], c[sl], t[sl], eps[sl])
        loss.backward()
        return {k: v.grad for k, v in sdg.items() if v.is_floating_point()}

    per = B // world
    mine = grads(slice(rank * per, (rank + 1) * per))
    fp.grad.zero_()
    for n, gr in mine.items():
        if gr is not None:
            fp.gviews[n].copy_(gr)
    steps.allreduce_grads(fp, world)
    fp.grad[:fp.n_active].mul_(1.0 / world)
    full = grads(slice(0, B))
    worst = 0.0
    for n, gr in full.items():
        if gr is None:
            assert float(fp.gviews[n].abs().max()) == 0.0, n
            continue
        worst = max(worst, float((fp.gviews[n] - gr).abs().max() / (gr.abs().max() + 1e-12)))
    # inactive tail untouched by the collective
    assert float(fp.grad[fp.n_active:].abs().max()) == 0.0
    assert steps.shard_seed(1000, rank) != steps.shard_seed(1000, (rank + 1) % world)
    ret[rank] = worst
    dist.destroy_process_group()


def test_flat_allreduce_equals_global_batch_gradient():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    for r, w in ret.items():
        assert w < 1e-4, (r, w)


def test_kl_term_data_parallel_rule():
    """The KL term of v2:380-386 is mean(clamped kl rows) + 1e-4 * SUM(mu^2): in a data-parallel VAE-GAN step the mean part
    is averaged over ranks, the sum part is added, and its shard gradient is scaled by the world size before the
    gradients are averaged (steps.VAEGANTrainer, fl_kl_bwd reg_scale). Checked here with the oracle on CPU."""
    from oracle import v2_oracle as O
    g = torch.Generator().manual_seed(5)
    B, W = 8, 2
    mu = (torch.randn(B, 16, generator=g) * 3).double().requires_grad_(True)
    lv = torch.randn(B, 16, generator=g).double().requires_grad_(True)
    full = O.kl_divergence(mu, lv)
    full.backward()
    g_mu, g_lv = mu.grad.clone(), lv.grad.clone()
    per = B // W
    mean_parts, reg_parts, gm, gl = [], [], torch.zeros_like(g_mu), torch.zeros_like(g_lv)
    for r in range(W):
        m = mu.detach()[r * per:(r + 1) * per].clone().requires_grad_(True)
        l = lv.detach()[r * per:(r + 1) * per].clone().requires_grad_(True)
        reg = 1e-4 * torch.clamp(m, -10, 10).pow(2).sum()
        mean_part = O.kl_divergence(m, l) - reg
        (mean_part + W * reg).backward()            # shard loss whose AVERAGED gradient is the global one
        mean_parts.append(mean_part.detach())
        reg_parts.append(reg.detach())
        gm[r * per:(r + 1) * per] = m.grad / W
        gl[r * per:(r + 1) * per] = l.grad / W
    assert abs(float(sum(mean_parts) / W + sum(reg_parts) - full.detach())) < 1e-12
    assert float((gm - g_mu).abs().max()) < 1e-12 and float((gl - g_lv).abs().max()) < 1e-12
