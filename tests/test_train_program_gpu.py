"""Training job kinds of the persistent executor (dgrad / wgrad GEMM operand layouts, bias column sums, activation
backward, LayerNorm backward rows + columns, loss rows, embedding scatter, clip + AdamW) against torch fp64 on the CPU,
and the two-launch denoiser training step (program_train.py) against the multi-kernel trainer and the oracle.
fp32 on the device; tolerances = fp32 re-association of the contractions, stated per test."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, ops, program as P, steps  # noqa: E402


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, tol=2e-5, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    ref = max(1.0, b.abs().max().item())
    assert err <= tol * ref, "%s max err %.3e (ref max %.3e)" % (what, err, ref)


def swish(v):
    return v * torch.sigmoid(v)


def swish_grad(v):
    s = torch.sigmoid(v)
    return s * (1 + v * (1 - s))


@pytest.mark.parametrize("B", [256, 64, 300, 5, 513])
def test_dgrad_wgrad_jobs(dev, B):
    """dx = (dy W) * head mask + R (W read as [k][n]); dW = dy^T x with the bias gradient (alpha, gate) riding on it;
    an activation-backward epilogue and a pre-activation copy. B = 300 / 5 / 64: ragged and short contraction chunks."""
    N, K = 512, 256
    dy, x, W = rnd(B, N, seed=1), rnd(B, K, seed=2), rnd(N, K, seed=3, scale=0.1)
    R, pre = rnd(B, K, seed=4), rnd(B, K, seed=5)
    hm = (torch.rand(B, 8, generator=torch.Generator().manual_seed(6)) > 0.3).float() / 0.7
    gate = torch.tensor([0.3])
    d = lambda t: t.to(dev)
    ddy, dx_, dW_, dR, dpre, dhm, dgate = map(d, (dy, x, W, R, pre, hm, gate))
    out_dx, out_dx2 = torch.empty(B, K, device=dev), torch.empty(B, K, device=dev)
    out_dW, out_db = torch.full((N, K), 7.0, device=dev), torch.full((N,), 7.0, device=dev)
    out_dW2, out_db2 = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    fwd, fwd_pre = torch.empty(B, N, device=dev), torch.empty(B, N, device=dev)
    bias = rnd(N, seed=7)
    pb = P.ProgramBuilder()
    pb.gemm(B, K, [(ddy, N, N, dW_, K)], out_dx, K, tw=True, mask=dhm, mask_group=K // 8, R=dR, ldr=K)
    pb.gemm(B, K, [(ddy, N, N, dW_, K)], out_dx2, K, tw=True, act=ops.ACT_SWISH, actbwd=dpre, ld_actbwd=K)
    pb.gemm(N, K, [(ddy, N, B, dx_, K)], out_dW, K, ta=True, tw=True, colsum=out_db, colsum_alpha=2.0)
    pb.gemm(N, K, [(ddy, N, B, dx_, K)], out_dW2, K, ta=True, tw=True, colsum=out_db2, gate=dgate)
    out_dW3, out_db3 = torch.full((N, K), 7.0, device=dev), torch.full((N,), 7.0, device=dev)
    pb.gemm(N, K, [(ddy, N, B, dx_, K)], out_dW3, K, ta=True, tw=True, colsum=out_db3, colsum_alpha=2.0, gate=dgate, t64=True)
    pb.gemm(B, N, [(dx_, K, K, dW_, K)], fwd, N, [(d(bias), 1.0)], act=ops.ACT_SWISH, pre_out=fwd_pre)
    prog = pb.finalize(dev)
    for _ in range(2):
        prog.run([])
    prog.check()
    ref = dy.double() @ W.double()
    close(out_dx, ref * hm.double().repeat_interleave(K // 8, 1) + R.double(), 2e-5, "dgrad")
    close(out_dx2, ref * swish_grad(pre.double()), 2e-5, "dgrad * act'")
    refW = dy.double().t() @ x.double()
    close(out_dW, refW, 2e-5, "wgrad")
    close(out_dW2, refW, 2e-5, "wgrad 2")
    close(out_db, 2.0 * dy.double().sum(0), 2e-5, "bias grad (alpha)")
    close(out_db2, (1 + torch.sigmoid(gate.double())) * dy.double().sum(0), 2e-5, "bias grad (gate)")
    close(out_dW3, refW, 2e-5, "wgrad, 64x64 tiles")
    close(out_db3, 2.0 * (1 + torch.sigmoid(gate.double())) * dy.double().sum(0), 2e-5, "bias grad, 64x64 tiles")
    p = x.double() @ W.double().t() + bias.double()
    close(fwd_pre, p, 2e-5, "pre-activation copy")
    close(fwd, swish(p), 2e-5, "activated")


def test_transposed_operand_shape_rules(dev):
    z = lambda *s: torch.zeros(*s, device=dev)
    pb = P.ProgramBuilder()
    pb.gemm(48, 64, [(z(10, 48), 48, 10, z(10, 64), 64)], z(48, 64), 64, ta=True, tw=True)   # M % 32 != 0 with TA
    with pytest.raises(ValueError):
        pb.finalize(dev)
    pb = P.ProgramBuilder()
    pb.gemm(64, 64, [(z(10, 64), 64, 10, z(10, 64), 64)], z(64, 64), 64, ta=True)            # TA without TW
    with pytest.raises(ValueError):
        pb.finalize(dev)


@pytest.mark.parametrize("D,act,masked", [(256, ops.ACT_SWISH, True), (1024, ops.ACT_NONE, False), (512, ops.ACT_SWISH, False)])
def test_layernorm_backward_jobs(dev, D, act, masked):
    """LNBWD rows (+ residual) and LNCOL columns vs torch autograd in fp64 on y = act(mask * LN(x)) (v2:514-516)."""
    B = 100
    x, dy, R = rnd(B, D, seed=1, scale=2.0), rnd(B, D, seed=2), rnd(B, D, seed=3)
    w, b = 1 + 0.1 * rnd(D, seed=4), 0.1 * rnd(D, seed=5)
    mask = ((torch.rand(B, D, generator=torch.Generator().manual_seed(6)) > 0.3).float() / 0.7) if masked else None
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xd, (D,), wd, bd, 1e-5)
    if masked:
        y = y * mask.double()
    if act == ops.ACT_SWISH:
        y = swish(y)
    y.backward(dy.double())
    d = lambda t: t.to(dev) if t is not None else None
    dx, gl, st = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev), torch.empty(B, 2, device=dev)
    dw, db = torch.empty(D, device=dev), torch.empty(D, device=dev)
    ddy, dxx = d(dy), d(x)
    need_gl = masked or act != ops.ACT_NONE
    pb = P.ProgramBuilder()
    pb.lnbwd(B, D, dxx, D, ddy, D, d(w), d(b), 1e-5, dx, D, act=act, mask=d(mask), R=d(R), ldr=D, gl=gl if need_gl else None,
             stats=st)
    pb.barrier()
    pb.lncol(B, D, gl if need_gl else ddy, D, dxx, D, st, dw, db)
    prog = pb.finalize(dev)
    prog.run([])
    prog.check()
    close(dx, xd.grad + R.double(), 3e-5, "dx")
    close(dw, wd.grad, 3e-5, "dweight")
    close(db, bd.grad, 3e-5, "dbias")
    close(st[:, 0], x.double().mean(1), 1e-5, "mean")


@pytest.mark.parametrize("D,B", [(512, 70), (1024, 256)])
def test_two_stage_layernorm_backward_job(dev, D, B):
    """The layer's LayerNorm pair (v2:514-516 then v2:549) backward as ONE row job vs torch autograd in fp64."""
    g_, dy, R = rnd(B, D, seed=1, scale=2.0), rnd(B, D, seed=2), rnd(B, D, seed=3)
    w, b, w2, b2 = 1 + 0.1 * rnd(D, seed=4), 0.1 * rnd(D, seed=5), 1 + 0.1 * rnd(D, seed=6), 0.1 * rnd(D, seed=7)
    mask = (torch.rand(B, D, generator=torch.Generator().manual_seed(8)) > 0.3).float() / 0.7
    gd = g_.double().requires_grad_(True)
    wd, bd, w2d, b2d = (t.double().requires_grad_(True) for t in (w, b, w2, b2))
    h2 = swish(torch.nn.functional.layer_norm(gd, (D,), wd, bd, 1e-5) * mask.double())
    h2.retain_grad()
    hn = torch.nn.functional.layer_norm(h2, (D,), w2d, b2d, 1e-5)
    ((hn * dy.double()).sum() + (h2 * R.double()).sum()).backward()
    d = lambda t: t.to(dev)
    dg, mid, gl = (torch.empty(B, D, device=dev) for _ in range(3))
    st_o, st_i = torch.empty(B, 2, device=dev), torch.empty(B, 2, device=dev)
    dw, db, dw2, db2 = (torch.empty(D, device=dev) for _ in range(4))
    dh2x, ddy, dgx = d(h2.detach().float()), d(dy), d(g_)
    pb = P.ProgramBuilder()
    pb.lnbwd2(B, D, dh2x, ddy, d(w2), d(b2), 1e-5, d(R), mid, st_o, dgx, d(w), d(b), 1e-5, dg, act=ops.ACT_SWISH, mask=d(mask),
              gl=gl, stats=st_i)
    pb.barrier()
    pb.lncol(B, D, ddy, D, dh2x, D, st_o, dw2, db2)
    pb.lncol(B, D, gl, D, dgx, D, st_i, dw, db)
    prog = pb.finalize(dev)
    prog.run([])
    prog.check()
    close(mid, h2.grad, 3e-5, "d h2")
    close(dg, gd.grad, 3e-5, "d g")
    close(dw2, w2d.grad, 3e-5, "outer weight")
    close(db2, b2d.grad, 3e-5, "outer bias")
    close(dw, wd.grad, 3e-5, "inner weight")
    close(db, bd.grad, 3e-5, "inner bias")


def test_loss_fin_ewadd_embedding_jobs(dev):
    B, D, rows = 70, 256, 11
    pred, tgt, r = rnd(B, D, seed=1), rnd(B, D, seed=2), rnd(B, D, seed=3)
    idx = torch.randint(0, rows, (B,), generator=torch.Generator().manual_seed(4))
    gate = torch.tensor([-0.2])
    d = lambda t: t.to(dev)
    dpred, dtgt, dr, didx, dgate = map(d, (pred, tgt, r, idx, gate))
    dout, norms, rowdot = torch.empty(B, D, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)
    loss, gg = torch.empty(1, device=dev), torch.empty(1, device=dev)
    s3, tab = torch.empty(B, D, device=dev), torch.full((rows, D), 3.0, device=dev)
    pb = P.ProgramBuilder()
    pb.loss(B, D, dpred, dtgt, dout, norms, 1.0 / B, r=dr, rowdot=rowdot)
    pb.ewadd(B, D, dpred, dtgt, s3, c=dr)
    pb.barrier()
    pb.fin(B, norms, loss, rowdot=rowdot, gate=dgate, ggate=gg)
    pb.embbwd(B, D, rows, dout, didx, tab)
    prog = pb.finalize(dev)
    prog.run([])
    prog.check()
    pd = pred.double().requires_grad_(True)
    nrm = torch.sqrt(((pd - tgt.double()) ** 2).sum(1) + 1e-8)   # v2:293-302
    nrm.mean().backward()
    close(loss, nrm.mean().detach(), 1e-6, "loss")
    close(dout, pd.grad, 1e-6, "dloss")
    sg = torch.sigmoid(gate.double())
    close(gg, sg * (1 - sg) * (pd.grad * r.double()).sum(), 2e-5, "gate grad")
    close(s3, pred.double() + tgt.double() + r.double(), 1e-6, "ewadd")
    ref = torch.zeros(rows, D, dtype=torch.float64).index_add_(0, idx, pd.grad)
    close(tab, ref, 1e-6, "embedding grad")


@pytest.mark.parametrize("n,max_norm,decoupled", [(1000003 // 4 * 4, 1.0, True), (4096, None, False), (52, 0.05, True)])
def test_optimizer_program_matches_torch(dev, n, max_norm, decoupled):
    """SQNORM + ADAM phases vs torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW/Adam, three steps."""
    p0, g = rnd(n, seed=1), [rnd(n, seed=10 + i, scale=0.01 * (i + 1)) for i in range(3)]
    mod = torch.nn.ParameterList([torch.nn.Parameter(p0.clone().to(dev))])
    fp = steps.FlatParams(mod)
    opt = steps.FusedAdam(fp, 1e-3, weight_decay=1e-2, decoupled=decoupled, max_norm=max_norm)
    from oxford_102_flower_gan_vae_latent_diffusion_amd.program_train import OptimizerProgram
    op = OptimizerProgram(opt, fp.n_active, gscale=0.5)
    ref_p = torch.nn.Parameter(p0.clone().double())
    ropt = (torch.optim.AdamW if decoupled else torch.optim.Adam)([ref_p], lr=1e-3, weight_decay=1e-2)
    for i in range(3):
        fp.grad[:n].copy_(g[i].to(dev))
        op.run()
        ref_p.grad = 0.5 * g[i].double()
        gn = ref_p.grad.norm()
        if max_norm is not None:
            torch.nn.utils.clip_grad_norm_([ref_p], max_norm)
            close(opt.grad_norm, gn, 1e-5, "grad norm")
        ropt.step()
    op.prog.check()
    assert int(opt.step_count) == 3
    close(fp.flat[:n], ref_p.detach(), 2e-6, "parameters after 3 steps")


def _seeded(m, seed):
    from oracle import procedural_init as pinit
    m.load_state_dict(pinit.procedural_state_dict(m.state_dict(), seed))
    return m


@pytest.mark.parametrize("B,train", [(64, True), (256, False), (40, True)])
def test_program_step_equals_multikernel_step(dev, B, train):
    """DDPMTrainer(persistent=True) vs (persistent=False): same draws -> same loss, gradients, clip norm and parameters
    after three steps (fp32 re-association only: 2e-5 of each tensor's max, 1e-4 for the tiny gate gradient)."""
    g = torch.Generator().manual_seed(5)
    data = [torch.rand(B, 3, 64, 64, generator=g) * 2 - 1 for _ in range(3)]
    labels = [torch.randint(0, 102, (B,), generator=g) for _ in range(3)]
    res = []
    for persistent in (False, True):
        ae = _seeded(M.SimpleAutoencoder(), 42).to(dev).eval()
        un = _seeded(M.ConditionalUNet(), 44).to(dev)
        un.train(train)
        tr = steps.DDPMTrainer(ae, un, lr=1e-3, persistent=persistent)
        out = {"loss": [], "gn": []}
        for i in range(3):
            l = tr.step(data[i].to(dev), labels[i].to(dev))
            out["loss"].append(float(l))
            out["gn"].append(float(tr.opt.grad_norm))
            if i == 0:
                out["g"] = {k: v.detach().cpu().clone() for k, v in tr.fp.gviews.items()}
                out["eps_theta"] = tr.eps_theta.detach().cpu().clone()
        out["p"] = {k: v.detach().cpu().clone() for k, v in un.state_dict().items()}
        if persistent:
            assert tr._tprog and tr._oprog is not None, "the persistent path did not run"
            for pr in tr._tprog.values():
                for q in pr.programs:
                    q.check()
            tr._oprog.prog.check()
        res.append(out)
    a, b = res
    np.testing.assert_allclose(b["loss"], a["loss"], rtol=2e-6)
    np.testing.assert_allclose(b["gn"], a["gn"], rtol=2e-5)
    close(b["eps_theta"], a["eps_theta"], 2e-5, "eps_theta")
    for k in a["g"]:
        tol = 1e-4 if k == "residual_weight" else 2e-5
        err = (a["g"][k].double() - b["g"][k].double()).abs().max().item()
        assert err <= tol * max(a["g"][k].abs().max().item(), 1e-3), "grad %s: %.3e vs max %.3e" % (k, err, a["g"][k].abs().max().item())
    # three Adam steps of lr 1e-3: an element whose gradient is ~0 may take a step of opposite sign (|diff| up to 2 lr per
    # step); everything else agrees to rounding
    bad = tot = 0
    for k in a["p"]:
        if not a["p"][k].dtype.is_floating_point:
            continue
        diff = (a["p"][k].double() - b["p"][k].double()).abs()
        bad += int((diff > 2.5e-4).sum())
        tot += diff.numel()
        assert diff.max().item() <= 6.1e-3, "param %s moved by %.3e" % (k, diff.max().item())
    assert bad <= 1e-3 * tot, "%d of %d parameters differ by more than lr/4" % (bad, tot)


def test_program_step_graph_capture_and_replay(dev):
    """capture() with the persistent programs: N replays == N eager steps (same Philox streams), bit-exact."""
    B = 64
    g = torch.Generator().manual_seed(6)
    data, labels = torch.rand(B, 3, 64, 64, generator=g).to(dev) * 2 - 1, torch.randint(0, 102, (B,), generator=g).to(dev)
    outs = []
    for captured in (False, True):
        ae = _seeded(M.SimpleAutoencoder(), 42).to(dev).eval()
        un = _seeded(M.ConditionalUNet(), 44).to(dev).train()
        tr = steps.DDPMTrainer(ae, un, lr=1e-3, persistent=True)
        if captured:
            tr.capture(data, labels)          # two warm-up steps inside
            for _ in range(3):
                tr.replay()
        else:
            for _ in range(5):
                tr.step(data, labels)
        torch.cuda.synchronize()
        outs.append((float(tr.loss), {k: v.detach().cpu().clone() for k, v in un.state_dict().items()}))
    assert outs[0][0] == outs[1][0]
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k
    # the eval-mode (folded) forward must see the weights the replayed graphs left behind
    from oxford_102_flower_gan_vae_latent_diffusion_amd import engine
    un.eval()
    x = torch.randn(B, 256, device=dev)
    t = torch.randint(0, 1000, (B,), device=dev)
    with torch.no_grad():
        o1 = un(x, t, labels)
        tr.replay()
        o2 = un(x, t, labels)
        ref, _ = engine.unet_forward(un, x, t, labels)
    assert (o2 - o1).abs().max().item() > 1e-6 and un._folded
    close(o2, ref, 3e-5, "folded forward after a replayed update")


def test_overlapped_dp_schedule_equals_plain_steps(dev):
    """capture_overlapped / replay_overlapped (the N > 1 bench schedule: encoder of batch k+1 between the denoiser program and
    the optimizer program of batch k) on one rank == plain step() on the same batches, bit for bit."""
    B, n = 32, 4
    g = torch.Generator().manual_seed(7)
    data = [(torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).to(dev) for _ in range(n + 1)]
    labels = [torch.randint(0, 102, (B,), generator=g).to(dev) for _ in range(n + 1)]

    def make():
        ae = _seeded(M.SimpleAutoencoder(), 42).to(dev).eval()
        un = _seeded(M.ConditionalUNet(), 44).to(dev).train()
        tr = steps.DDPMTrainer(ae, un, lr=1e-3, persistent=True)
        reseed(ae, un, tr)       # before any capture: the Philox seed is a kernel argument baked into the graphs
        return ae, un, tr

    def reseed(ae, un, tr):
        for r, s_ in ((ae.rng, 1), (un.rng, 2), (tr.diffusion.rng, 3)):
            r.manual_seed(s_)

    ae, un, tr = make()
    reseed(ae, un, tr)
    la = [float(tr.step(data[i], labels[i])) for i in range(n)]
    pa = tr.fp.flat.clone()
    ae, un, tr = make()
    sd0 = {k: v.clone() for k, v in un.state_dict().items()}
    tr.capture_overlapped(data[0].clone(), labels[0].clone())
    un.load_state_dict(sd0)                      # capture ran warm-up steps: restore weights, optimizer and RNG state
    tr.opt.m.zero_(); tr.opt.v.zero_(); tr.opt.step_count.zero_()
    reseed(ae, un, tr)
    # plain steps draw reparam noise (ae.rng) then t / eps / masks per step, in that order per batch: the overlapped schedule
    # encodes batch k+1 before step k's draws only on ae.rng (its own stream), so every stream sees the same sequence
    tr.prime_overlapped(data[0], labels[0])
    lb = []
    for i in range(n):
        l = tr.replay_overlapped(data[i + 1], labels[i + 1])
        torch.cuda.synchronize()
        lb.append(float(l))
    assert lb == la
    assert torch.equal(tr.fp.flat, pa)


def test_beta1_schedule_reaches_captured_and_resident_optimizers(dev):
    """torch's OneCycleLR cycles AdamW's beta1 every batch (v2:1031-1037). beta1 lives on the device like lr: a captured
    multi-kernel step and the resident optimizer program both follow set_beta1 without being rebuilt, and equal eager steps with
    the same schedule (it used to be baked into graphs and to rebuild the optimizer program on every change)."""
    n = 4096 * 3
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) for _ in range(4)]
    b1s = [0.95, 0.9, 0.85, 0.93]

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(p0.clone())

    def run(mode):
        m = Holder().to(dev)
        fp = steps.FlatParams(m)
        opt = steps.FusedAdam(fp, lr=1e-2, betas=(0.95, 0.999), weight_decay=1e-2, decoupled=True, max_norm=1.0)
        gbuf = fp.grad
        if mode == "graph":
            gbuf.copy_(grads[0].to(dev))
            opt.step()                       # warm-up (sizes workspaces), then restore
            fp.flat.copy_(p0.to(dev)); opt.m.zero_(); opt.v.zero_(); opt.step_count.zero_()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                opt.step()
        elif mode == "program":
            from oxford_102_flower_gan_vae_latent_diffusion_amd.program_train import OptimizerProgram
            op = OptimizerProgram(opt, fp.n_active)
        for gi, b1 in zip(grads, b1s):
            opt.set_beta1(b1)
            gbuf.copy_(gi.to(dev))
            if mode == "eager":
                opt.step()
            elif mode == "graph":
                gr.replay()
            else:
                op.run()
        torch.cuda.synchronize()
        if mode == "program":
            op.prog.check()
        return fp.flat.detach().cpu().clone()

    eager = run("eager")
    # torch reference with the same per-step beta1
    pt = torch.nn.Parameter(p0.clone().double())
    o = torch.optim.AdamW([pt], lr=1e-2, betas=(0.95, 0.999), weight_decay=1e-2)
    for gi, b1 in zip(grads, b1s):
        o.param_groups[0]["betas"] = (b1, 0.999)
        pt.grad = gi.double().clone()
        torch.nn.utils.clip_grad_norm_([pt], 1.0)
        o.step()
    close(eager, pt.detach(), 2e-6, "eager steps vs torch AdamW with a cycled beta1")
    assert torch.equal(run("graph"), eager), "a captured step ignores set_beta1"
    close(run("program"), eager, 1e-6, "optimizer program vs eager steps with a cycled beta1")


def test_optimizer_program_does_not_apply_a_void_gradient(dev):
    """The optimizer program watches the training program's abort word: after a voided forward / backward run (a label outside the
    embedding table) it does nothing - weights, moments and the step count stay put - until the host has seen and cleared it."""
    B = 32
    ae = _seeded(M.SimpleAutoencoder(), 42).to(dev).eval()
    un = _seeded(M.ConditionalUNet(), 44).to(dev).train()
    tr = steps.DDPMTrainer(ae, un, lr=1e-3, persistent=True)
    g = torch.Generator().manual_seed(1)
    data = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).to(dev)
    labels = torch.randint(0, 102, (B,), generator=g).to(dev)
    tr.step(data, labels)
    torch.cuda.synchronize()
    tr.check_programs()
    before, steps_before = tr.fp.flat.clone(), int(tr.opt.step_count)
    bad = labels.clone()
    bad[3] = 102                                   # outside the 102-row class table: the training program voids its run
    z = tr.encode(data)
    tr.fwd_bwd(z, bad)
    torch.cuda.synchronize()
    tr._oprog.prog._abort_host.zero_()              # (keep the eager host-side shortcut out of the way: the DEVICE-side guard is under test)
    tr._tprog[(B, True)].prog._abort_host.zero_()
    tr._oprog.run()
    torch.cuda.synchronize()
    assert torch.equal(tr.fp.flat, before) and int(tr.opt.step_count) == steps_before
    with pytest.raises(RuntimeError, match="outside its table"):
        tr.check_programs()
    tr.step(data, labels)                           # cleared: training goes on
    torch.cuda.synchronize()
    tr.check_programs()
    assert int(tr.opt.step_count) == steps_before + 1 and not torch.equal(tr.fp.flat, before)
