"""BASELINE config 4 on the HIP path: DDPMTrainer(process_group, world_size=2) as two ranks (gloo, both on cuda:0 - a
one-GPU box) against the oracle's single-process step on the GLOBAL batch: averaged gradient, pre-clip norm and post-step
weights; both ranks bit-identical. Plain eager steps AND the pipelined hipGraph path (encoder graph on stream E, denoiser
graph + eager all-reduce + update graph on stream U)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
KEYS = ("latent_proj.weight", "layers.0.0.0.weight", "layers.2.2.weight", "attention_layers.1.in_proj_weight",
        "attention_layers.3.out_proj.weight", "time_emb.lin1.weight", "class_emb.embedding.weight", "final.weight", "final.bias",
        "residual_weight")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(nb=2, B=8):
    g = torch.Generator().manual_seed(23)
    out = []
    for _ in range(nb):
        out.append(dict(data=torch.rand(B, 3, 64, 64, generator=g), labels=torch.randint(0, 102, (B,), generator=g),
                        re=torch.randn(B, 256, generator=g), t=torch.randint(0, 1000, (B,), generator=g),
                        eps=torch.randn(B, 256, generator=g)))
    return out


def _worker(rank, world, port, ret, pipelined):
    import faulthandler
    faulthandler.enable()   # a crash of a spawned rank (SIGSEGV in native code) leaves a Python stack in the test log
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import procedural_init as pinit
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, steps
    dev = torch.device("cuda:0")
    ae = M.SimpleAutoencoder()
    ae.load_state_dict(pinit.procedural_state_dict(ae.state_dict(), 42))
    un = M.ConditionalUNet()
    un.load_state_dict(pinit.procedural_state_dict(un.state_dict(), 44))
    ae, un = ae.to(dev).eval(), un.to(dev).eval()  # eval: dropout off, explicit noise -> comparable with the oracle
    # both ranks share ONE GPU here: the persistent programs need a whole device each, so this rehearsal runs the multi-kernel path
    tr = steps.DDPMTrainer(ae, un, lr=1e-3, process_group=dist.group.WORLD, world_size=world, persistent=False)
    batches = _inputs()
    per = batches[0]["data"].shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    res = {}
    if pipelined == "overlapped":
        # the N > 1 schedule of bench.py (capture_overlapped / prime_overlapped / replay_overlapped: all-reduce of step k
        # issued async while the encoder graph of batch k+1 replays, work.wait(), optimizer graph) against plain step() on
        # the same draws: same seeds per rank for both runs, weights / optimizer / RNG state restored in between
        def reseed():
            for r, s_ in ((ae.rng, 100 + rank), (un.rng, 300 + rank), (tr.diffusion.rng, 200 + rank)):
                r.manual_seed(s_)
        un.train()
        n = 3
        data = [batches[i % 2]["data"][sl].to(dev) for i in range(n + 1)]
        labs = [batches[i % 2]["labels"][sl].to(dev) for i in range(n + 1)]
        sd0 = {k: v.clone() for k, v in un.state_dict().items()}
        reseed()
        res["plain_loss"] = [float(tr.step(data[i], labs[i])) for i in range(n)]
        res["plain_flat"] = tr.fp.flat.detach().cpu().clone()
        un.load_state_dict(sd0)
        tr.opt.m.zero_(); tr.opt.v.zero_(); tr.opt.step_count.zero_()
        tr.capture_overlapped(data[0].clone(), labs[0].clone())
        un.load_state_dict(sd0)                  # capture ran warm-up steps
        tr.opt.m.zero_(); tr.opt.v.zero_(); tr.opt.step_count.zero_()
        reseed()
        tr.prime_overlapped(data[0], labs[0])
        res["over_loss"] = []
        for i in range(n):
            l = tr.replay_overlapped(data[i + 1], labs[i + 1])
            torch.cuda.synchronize()
            res["over_loss"].append(float(l))
        res["over_flat"] = tr.fp.flat.detach().cpu().clone()
        res["comm"] = tr.measure_comm(data, labs, steps_n=2, overlapped=True)
    elif not pipelined:
        for i, b in enumerate(batches):
            tr.step(b["data"][sl].to(dev), b["labels"][sl].to(dev), b["re"][sl].to(dev), b["t"][sl].to(dev), b["eps"][sl].to(dev))
            res["gn%d" % i] = float(tr.opt.grad_norm)
            # the clip path scales the summed gradient by 1/world in place: the buffer now holds the global-batch gradient
            res["g%d" % i] = {k: tr.fp.gviews[k].detach().cpu().reshape(-1)[:2048].clone() for k in KEYS}
    else:
        # graphs draw their own noise (Philox): seed the ranks differently, compare the ranks with each other only
        ae.rng.manual_seed(100 + rank)
        tr.diffusion.rng.manual_seed(200 + rank)
        d0, l0 = batches[0]["data"][sl].to(dev), batches[0]["labels"][sl].to(dev)
        tr.capture_pipelined(d0.clone(), l0.clone())
        tr.prime_pipeline(d0, l0)
        for i in range(3):
            b = batches[(i + 1) % 2]
            tr.replay_pipelined(b["data"][sl].to(dev), b["labels"][sl].to(dev))
        tr.sync_pipeline()
    torch.cuda.synchronize()
    res["params"] = {k: un.state_dict()[k].detach().cpu().reshape(-1)[:2048].clone() for k in KEYS}
    res["loss"] = float(tr.loss)
    ret[rank] = res
    dist.destroy_process_group()


def test_dp_ddpm_step_equals_global_batch_step(dev):
    from oracle import procedural_init as pinit, v2_oracle as O
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, False), nprocs=world, join=True)
    assert len(ret) == world
    ae_sd = pinit.procedural_state_dict(M.SimpleAutoencoder().state_dict(), 42)
    sd = pinit.procedural_state_dict(M.ConditionalUNet().state_dict(), 44)
    opt_state = None

    def close(a, b, tol, what):
        a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
        err, scale = (a - b).abs().max().item(), max(1e-6, b.abs().max().item())
        assert err <= tol * scale, "%s: err %.3e (ref %.3e)" % (what, err, scale)

    for i, b in enumerate(_inputs()):
        ref = O.ddpm_step(ae_sd, sd, b["data"], b["labels"], b["re"], b["t"], b["eps"], lr=1e-3, opt_state=opt_state)
        for r in range(world):
            close(ret[r]["gn%d" % i], ref["grad_norm"], 2e-4, "rank %d step %d grad norm" % (r, i))
            for k in KEYS:
                close(ret[r]["g%d" % i][k], ref["grads"][k].reshape(-1)[:2048], 5e-4, "rank %d step %d grad %s" % (r, i, k))
        sd, opt_state = ref["new_unet_sd"], ref["opt"].state_dict()
    for k in KEYS:
        assert torch.equal(ret[0]["params"][k], ret[1]["params"][k]), "ranks diverged: " + k
        a, bb = ret[0]["params"][k].double(), sd[k].reshape(-1)[:2048].double()
        assert (a - bb).abs().max().item() <= 4.2e-3 + 1e-7, k  # two Adam steps at lr 1e-3: sign-flip bound (gradients pinned above)


def test_dp_ddpm_pipelined_graphs_keep_ranks_identical(dev):
    """the pipelined path (graphs on two streams around the eager all-reduce) on two ranks: finite loss, and the replicated
    weights stay bit-identical across ranks after 3 steps although each rank drew its own noise."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, True), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert ret[r]["loss"] == ret[r]["loss"] and ret[r]["loss"] > 0
    assert ret[0]["loss"] != ret[1]["loss"]  # per-rank noise
    for k in KEYS:
        assert torch.equal(ret[0]["params"][k], ret[1]["params"][k]), "ranks diverged: " + k


def test_dp_ddpm_overlapped_schedule_two_ranks(dev):
    """bench.py's N > 1 schedule with world_size 2 (gloo, both ranks on cuda:0, multi-kernel denoiser: two persistent programs
    cannot share one GPU): the z / label double buffers, the async all-reduce under the next batch's encoder graph and the
    work.wait() ordering reproduce plain step() on the same draws bit for bit, on both ranks, and the replicated weights of the
    two ranks are identical although each rank drew its own noise."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, "overlapped"), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert ret[r]["over_loss"] == ret[r]["plain_loss"], "rank %d: %s vs %s" % (r, ret[r]["over_loss"], ret[r]["plain_loss"])
        assert torch.equal(ret[r]["over_flat"], ret[r]["plain_flat"]), "rank %d: overlapped schedule != plain steps" % r
        c = ret[r]["comm"]
        assert c["world"] == 2 and c["allreduce_ms"] > 0 and c["allreduce_exposed_ms"] >= 0 and c["allreduce_bytes"] > 4e7
    assert torch.equal(ret[0]["over_flat"], ret[1]["over_flat"]), "ranks diverged"
    assert ret[0]["over_loss"] != ret[1]["over_loss"]   # per-rank noise


def _rccl_worker(rank, world, port, ret):
    import faulthandler
    faulthandler.enable()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from oracle import procedural_init as pinit
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, steps
    dev = torch.device("cuda:0")
    B, n = 32, 4
    g = torch.Generator().manual_seed(7)
    data = [(torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).to(dev) for _ in range(n + 1)]
    labels = [torch.randint(0, 102, (B,), generator=g).to(dev) for _ in range(n + 1)]

    def make(dp):
        ae, un = M.SimpleAutoencoder(), M.ConditionalUNet()
        ae.load_state_dict(pinit.procedural_state_dict(ae.state_dict(), 42))
        un.load_state_dict(pinit.procedural_state_dict(un.state_dict(), 44))
        ae, un = ae.to(dev).eval(), un.to(dev).train()
        kw = dict(process_group=dist.group.WORLD, world_size=world, force_collectives=True) if dp else {}
        tr = steps.DDPMTrainer(ae, un, lr=1e-3, persistent=True, **kw)
        reseed(ae, un, tr)
        return ae, un, tr

    def reseed(ae, un, tr):
        for r, s_ in ((ae.rng, 1), (un.rng, 2), (tr.diffusion.rng, 3)):
            r.manual_seed(s_)

    ae, un, tr = make(False)
    la = [float(tr.step(data[i], labels[i])) for i in range(n)]
    pa = tr.fp.flat.clone()
    ae, un, tr = make(True)
    sd0 = {k: v.clone() for k, v in un.state_dict().items()}
    tr.capture_overlapped(data[0].clone(), labels[0].clone())
    un.load_state_dict(sd0)
    tr.opt.m.zero_(); tr.opt.v.zero_(); tr.opt.step_count.zero_()
    reseed(ae, un, tr)
    tr.prime_overlapped(data[0], labels[0])
    lb = []
    for i in range(n):
        l = tr.replay_overlapped(data[i + 1], labels[i + 1])
        torch.cuda.synchronize()
        lb.append(float(l))
    tr.check_programs()
    same = bool(torch.equal(tr.fp.flat, pa))
    comm = tr.measure_comm(data, labels, steps_n=3)   # (runs further steps)
    ret[rank] = dict(la=la, lb=lb, same=same, comm=comm)
    dist.destroy_process_group()


def test_ddpm_overlapped_schedule_on_rccl_one_rank(dev):
    """What `bench.py --gpus N` runs per rank - persistent denoiser programs, the flat-gradient all-reduce issued asynchronously on
    RCCL's stream under the next batch's encoder graph, work.wait(), optimizer program - on REAL RCCL. A test box has one GPU (and RCCL
    refuses two ranks on one device), so the group has one rank: the collective is trivial, the stream / event / graph choreography
    around it is the bench's. Bit-identical to plain step() on the same batches."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
    r = ret[0]
    assert r["lb"] == r["la"] and r["same"]
    assert r["comm"]["allreduce_ms"] > 0 and r["comm"]["allreduce_bytes"] > 4e7
