"""The latent-DDPM training step's denoiser half (v2:1260-1264) as TWO persistent launches (csrc/program.hip):

  DenoiserTrainProgram   ConditionalUNet forward with saved activations (v2:534-561), euclidean_distance_loss and its
                         gradient (v2:293-302), the whole manual backward of engine.unet_backward - every weight gradient
                         written straight into the trainer's flat gradient buffer, bias gradients riding on the weight-
                         gradient tiles, LayerNorm weight/bias gradients as column jobs - 60 phases, ~108 jobs;
  OptimizerProgram       clip_grad_norm_(1.0) + AdamW on the flat buffers (v2:1262-1263): 2 phases.

The data-parallel all-reduce sits between the two, exactly where DDPMTrainer.step has it. The multi-kernel
path (engine.unet_forward/unet_backward + FusedAdam.step, ~240 launches) stays as the reference implementation of the
same arithmetic; tests compare the two and both with the oracle.

A program needs every block resident (device-wide barriers): it must not run next to another stream's kernels, so the
trainer runs encoder graph -> programs on ONE stream.
"""
import torch

from . import ops
from .program import IO, ProgramBuilder

XT, T, CLS, EPS, BM0, HM0 = 0, 1, 2, 3, 4, 8  # IO slots


class _GlobalJobs:
    """Jobs that contract over the BATCH (weight gradients with their bias column sums, LayerNorm weight / bias gradients, the
    embedding gradient, the loss mean) need every row of the batch: they run device-wide. Nothing reads their outputs before
    the optimizer, so they are queued here while the backward's row-local chain (input gradients, LayerNorm backward rows) is
    laid out, and emitted as the program's LAST phase(s): the forward and the whole input-gradient chain then run XCD-local
    (32 rows per XCD, data handed over through that XCD's L2), followed by ONE L2 write-back + device-wide barrier and one big
    phase of ~2.8 k independent tiles. (Round 2 placed the weight gradients next to the input gradients of the same layer and
    cut the largest into filler pieces for idle blocks: every one of its 54 phases was device-wide.)"""

    def __init__(self, pb):
        self.pb, self.q = pb, []

    def add(self, emit):
        self.q.append(emit)

    def flush(self, per_phase=64):
        for i, emit in enumerate(self.q):
            if i and i % per_phase == 0:
                self.pb.barrier()
            emit()
        self.q = []


class DenoiserTrainProgram:
    def __init__(self, un, gviews, B, device, masks=True, loss_out=None):
        """un: ConditionalUNet; gviews: {parameter name: gradient tensor} (FlatParams.gviews); B: batch rows;
        masks: train-mode dropout scale tensors are bound at run time (IO slots BM0.., HM0..)."""
        d = lambda p: p.detach()
        SW = ops.ACT_SWISH
        f32 = dict(dtype=torch.float32, device=device)
        self.un, self.B, self.masks = un, B, masks
        self.key = tuple(p.data_ptr() for p in un.parameters()) + tuple(g.data_ptr() for g in gviews.values())
        E, L = un.time_emb_dim, un.latent_dim
        te_m, ce_m = un.time_emb, un.class_emb
        H1, C1 = te_m.lin1.weight.shape[0], ce_m.lin1.weight.shape[0]
        n_cls = ce_m.embedding.weight.shape[0]
        G = lambda name: gviews[name]
        new = lambda *shape: torch.empty(*shape, **f32)
        pb = ProgramBuilder()
        ws = self.ws = {}

        gq = _GlobalJobs(pb)

        def wgrad(dy, n_out, x, k_in, wname, bname=None, alpha=1.0, gate=None, w_off=0, b_off=0):
            """dW[n_out, k_in] = dy^T x (+ bias gradient = alpha * column sums of dy) into the flat gradient buffer (queued)."""
            t64 = n_out % 64 == 0 and k_in % 64 == 0
            gq.add(lambda: pb.gemm(n_out, k_in, [(dy, n_out, B, x, k_in)], (G(wname), w_off) if w_off else G(wname), k_in, ta=True,
                                   tw=True, colsum=((G(bname), b_off) if b_off else G(bname)) if bname else None,
                                   colsum_alpha=alpha, gate=gate, t64=t64))

        def dgrad(dy, n_out, W, k_in, dx, **kw):
            """dx[B, k_in] = dy[B, n_out] W[n_out, k_in]"""
            pb.gemm(B, k_in, [(dy, n_out, n_out, W, k_in)], dx, k_in, tw=True, **kw)

        # ------------------------------------------------------------------ forward (as UNetProgram, plus what backward needs)
        feat, emb = new(B, E), new(B, E)
        te1, te1a, ce1, ce1a = new(B, H1), new(B, H1), new(B, C1), new(B, C1)
        te, ce, tc = new(B, E), new(B, E), new(B, E)
        rfin = new(B, L)
        fin = un.final
        rw = d(un.residual_weight)
        pb.sinusoid(B, E, IO(T), 1, un._freqs(device), feat, E)
        pb.gather(B, E, d(ce_m.embedding.weight), E, IO(CLS), emb, E, rows=n_cls)   # a label outside the table voids the run
        pb.gemm(B, L, [(IO(XT), L, L, d(fin.weight), L)], rfin, L, [(d(fin.bias), 1.0)])        # final(x) of v2:561
        pb.barrier()
        pb.gemm(B, H1, [(feat, E, E, d(te_m.lin1.weight), E)], te1a, H1, [(d(te_m.lin1.bias), 1.0)], act=SW, pre_out=te1)
        pb.gemm(B, C1, [(emb, E, E, d(ce_m.lin1.weight), E)], ce1a, C1, [(d(ce_m.lin1.bias), 1.0)], act=SW, pre_out=ce1)
        pb.barrier()
        pb.gemm(B, E, [(te1a, H1, H1, d(te_m.lin2.weight), H1)], te, E, [(d(te_m.lin2.bias), 1.0)])
        pb.gemm(B, E, [(ce1a, C1, C1, d(ce_m.lin2.weight), C1)], ce, E, [(d(ce_m.lin2.bias), 1.0)])
        pb.gemm(B, E, [(te1a, H1, H1, d(te_m.lin2.weight), H1), (ce1a, C1, C1, d(ce_m.lin2.weight), C1)], tc, E,
                [(d(te_m.lin2.bias), 1.0), (d(ce_m.lin2.bias), 1.0)])
        pb.barrier()
        prev, prev_ld, prev_w, prev_b, Kprev = IO(XT), L, d(un.latent_proj.weight), d(un.latent_proj.bias), L
        saved = []
        for i, (block, norm, down) in enumerate(un.layers):
            lin, ln = block[0], block[1]
            D = lin.weight.shape[0]
            tp, att = un.time_projections[i], un.attention_layers[i]
            h1, g, h2, hn, vm, h3 = (new(B, D) for _ in range(6))
            saved.append((h1, g, h2, hn, vm, h3))
            pb.gemm(B, D, [(prev, prev_ld, Kprev, prev_w, Kprev), (tc, E, E, d(tp.weight), E)], h1, D,
                    [(prev_b, 1.0), (d(tp.bias), 2.0)])
            pb.barrier()
            pb.gemm(B, D, [(h1, D, D, d(lin.weight), D)], g, D, [(d(lin.bias), 1.0)])
            pb.barrier()
            pb.rownorm(B, D, g, D, d(ln.weight), d(ln.bias), ln.eps, h2, D, act=SW, mask=IO(BM0 + i) if masks else None,
                       R=h1, ldr=D, w2=d(norm.weight), b2=d(norm.bias), eps2=norm.eps, C2=hn)
            pb.barrier()
            pb.gemm(B, D, [(hn, D, D, (d(att.in_proj_weight), 2 * D * D), D)], vm, D, [((d(att.in_proj_bias), 2 * D), 1.0)],
                    mask=IO(HM0 + i) if masks else None, mask_group=D // att.num_heads)
            pb.barrier()
            pb.gemm(B, D, [(vm, D, D, d(att.out_proj.weight), D)], h3, D, [(d(att.out_proj.bias), 1.0)], R=h2, ldr=D)
            pb.barrier()
            prev, prev_ld, prev_w, prev_b, Kprev = h3, D, d(down.weight), d(down.bias), D
        ft, fcp, fn = un.final_time_proj, un.final_class_proj, un.final_norm
        Dl = ft.weight.shape[0]
        hpre, u, out = new(B, Dl), new(B, Dl), new(B, L)
        pb.gemm(B, Dl, [(prev, prev_ld, Kprev, prev_w, Kprev), (te, E, E, d(ft.weight), E), (ce, E, E, d(fcp.weight), E)], hpre, Dl,
                [(prev_b, 1.0), (d(ft.bias), 1.0), (d(fcp.bias), 1.0)])
        pb.barrier()
        # u = LN(h) + sigmoid(rw)*x ; out = final(u) + sigmoid(rw)*final.bias == final(LN(h)) + sigmoid(rw)*final(x)  (v2:561)
        pb.rownorm(B, Dl, hpre, Dl, d(fn.weight), d(fn.bias), fn.eps, u, Dl, gate=rw, gx=IO(XT), ldgx=L)
        pb.barrier()
        pb.gemm(B, L, [(u, Dl, Dl, d(fin.weight), Dl)], out, L, [(d(fin.bias), 1.0)], gate=rw)
        pb.barrier()
        self.n_fwd_phases = len([p for p in pb._phases if p])

        # ------------------------------------------------------------------ loss (v2:293-302) and its gradient
        dout, norms, rowdot = new(B, L), new(B), new(B)
        self.loss = loss_out if loss_out is not None else torch.zeros(1, **f32)
        pb.loss(B, L, out, IO(EPS), dout, norms, 1.0 / B, r=rfin, rowdot=rowdot)
        pb.barrier()

        # ------------------------------------------------------------------ backward (engine.unet_backward)
        dhf = new(B, Dl)
        gq.add(lambda: pb.fin(B, norms, self.loss, rowdot=rowdot, gate=rw, ggate=G("residual_weight")))
        wgrad(dout, L, u, Dl, "final.weight", "final.bias", gate=rw)     # one weight, two uses: (1+sig) * column sums
        dgrad(dout, L, d(fin.weight), Dl, dhf)
        pb.barrier()
        dh, st_f = new(B, Dl), new(B, 2)
        pb.lnbwd(B, Dl, hpre, Dl, dhf, Dl, d(fn.weight), d(fn.bias), fn.eps, dh, Dl, stats=st_f)
        pb.barrier()
        gq.add(lambda: pb.lncol(B, Dl, dhf, Dl, hpre, Dl, st_f, G("final_norm.weight"), G("final_norm.bias")))
        dte0, dce0 = new(B, E), new(B, E)
        wgrad(dh, Dl, te, E, "final_time_proj.weight", "final_time_proj.bias")
        wgrad(dh, Dl, ce, E, "final_class_proj.weight", "final_class_proj.bias")
        dgrad(dh, Dl, d(ft.weight), E, dte0)
        dgrad(dh, Dl, d(fcp.weight), E, dce0)
        dtc = None
        n_layers = len(un.layers)
        for i in reversed(range(n_layers)):
            block, norm, down = un.layers[i]
            lin, ln = block[0], block[1]
            h1, g, h2, hn, vm, h3 = saved[i]
            D, Dn = lin.weight.shape[0], down.weight.shape[0]
            att, tp = un.attention_layers[i], un.time_projections[i]
            pre, ap, tpp = "layers.%d." % i, "attention_layers.%d." % i, "time_projections.%d." % i
            # (same phase as the previous layer's time-projection jobs / the final projections above)
            dh3 = new(B, D)
            wgrad(dh, Dn, h3, D, pre + "2.weight", pre + "2.bias")
            dgrad(dh, Dn, d(down.weight), D, dh3)
            pb.barrier()
            dv = new(B, D)
            wgrad(dh3, D, vm, D, ap + "out_proj.weight", ap + "out_proj.bias")
            dgrad(dh3, D, d(att.out_proj.weight), D, dv, mask=IO(HM0 + i) if masks else None, mask_group=D // att.num_heads)
            pb.barrier()
            dhn = new(B, D)
            # q/k rows of in_proj: exactly zero gradient (softmax over one key): never written, the buffer is zero there
            wgrad(dv, D, hn, D, ap + "in_proj_weight", ap + "in_proj_bias", w_off=2 * D * D, b_off=2 * D)
            dgrad(dv, D, (d(att.in_proj_weight), 2 * D * D), D, dhn)
            pb.barrier()
            # both LayerNorm backwards of the layer (v2:549 then v2:514-516) are row-local: one row job
            dh2, st_n, dg, gl, st_g = new(B, D), new(B, 2), new(B, D), new(B, D), new(B, 2)
            pb.lnbwd2(B, D, h2, dhn, d(norm.weight), d(norm.bias), norm.eps, dh3, dh2, st_n, g, d(ln.weight), d(ln.bias), ln.eps,
                      dg, act=SW, mask=IO(BM0 + i) if masks else None, gl=gl, stats=st_g)
            pb.barrier()
            gq.add(lambda dhn=dhn, h2=h2, st_n=st_n, pre=pre, D=D: pb.lncol(B, D, dhn, D, h2, D, st_n, G(pre + "1.weight"), G(pre + "1.bias")))
            dh1 = new(B, D)
            wgrad(dg, D, h1, D, pre + "0.0.weight", pre + "0.0.bias")
            dgrad(dg, D, d(lin.weight), D, dh1, R=dh2, ldr=D)
            gq.add(lambda gl=gl, g=g, st_g=st_g, pre=pre, D=D: pb.lncol(B, D, gl, D, g, D, st_g, G(pre + "0.1.weight"), G(pre + "0.1.bias")))
            pb.barrier()
            dtc_i = new(B, E)
            wgrad(dh1, D, tc, E, tpp + "weight", tpp + "bias", alpha=2.0)   # the bias enters twice (W te + b + W ce + b)
            dgrad(dh1, D, d(tp.weight), E, dtc_i, R=dtc, ldr=E)
            dtc = dtc_i
            dh = dh1
        wgrad(dh, un.layers[0][0][0].weight.shape[0], IO(XT), L, "latent_proj.weight", "latent_proj.bias")
        pb.barrier()
        dte, dce = new(B, E), new(B, E)
        pb.ewadd(B, E, dte0, dtc, dte)
        pb.ewadd(B, E, dce0, dtc, dce)
        pb.barrier()
        d1t, d1c = new(B, H1), new(B, C1)
        wgrad(dte, E, te1a, H1, "time_emb.lin2.weight", "time_emb.lin2.bias")
        dgrad(dte, E, d(te_m.lin2.weight), H1, d1t, act=SW, actbwd=te1, ld_actbwd=H1)
        wgrad(dce, E, ce1a, C1, "class_emb.lin2.weight", "class_emb.lin2.bias")
        dgrad(dce, E, d(ce_m.lin2.weight), C1, d1c, act=SW, actbwd=ce1, ld_actbwd=C1)
        pb.barrier()
        demb = new(B, E)
        wgrad(d1t, H1, feat, E, "time_emb.lin1.weight", "time_emb.lin1.bias")
        wgrad(d1c, C1, emb, E, "class_emb.lin1.weight", "class_emb.lin1.bias")
        dgrad(d1c, C1, d(ce_m.lin1.weight), E, demb)
        pb.barrier()
        pb.barrier()
        gq.add(lambda: pb.embbwd(B, E, n_cls, demb, IO(CLS), G("class_emb.embedding.weight")))
        gq.flush()
        self.prog = pb.finalize(device)
        self.out = out
        self._dummy = torch.zeros(4, **f32)
        ws.update(dict(out=out, dout=dout, norms=norms, u=u, hpre=hpre, tc=tc, te=te, ce=ce, saved=saved))

    @property
    def programs(self):
        return (self.prog,)

    def run(self, xt, t, labels, eps, block_masks=None, head_masks=None, post=None):
        """post: [(int64 device counter, delta), ...] (at most two) bumped once after the last phase (Philox offsets)"""
        B = self.B
        assert xt.shape[0] == B and t.numel() == B and labels.numel() == B and eps.shape == xt.shape
        ops._chk(xt, t, labels, eps)
        io = [self._dummy] * 12
        io[XT], io[T], io[CLS], io[EPS] = xt, t, labels, eps
        if self.masks:
            for i in range(len(self.un.layers)):
                ops._chk(block_masks[i], head_masks[i])
                io[BM0 + i], io[HM0 + i] = block_masks[i], head_masks[i]
        self.prog.run(io, post=post or None)
        return self.out


class OptimizerProgram:
    """clip_grad_norm_ + Adam/AdamW of a steps.FusedAdam as one persistent launch (sum-of-squares slices, then the update
    with the clip coefficient derived per tile); bumps the device-resident step count after the last phase."""

    def __init__(self, opt, n, gscale=1.0):
        fp = opt.fp
        dev = fp.flat.device
        self.opt, self.n, self.gscale = opt, n, gscale
        self.partials = torch.zeros(256, dtype=torch.float32, device=dev)
        self.key = (n, gscale, opt.eps, opt.wd, opt.decoupled, opt.max_norm, fp.flat.data_ptr(), fp.grad.data_ptr())
        pb = ProgramBuilder()
        pb.sqnorm(n, fp.grad, self.partials)
        pb.barrier()
        pb.adam(n, fp.flat, fp.grad, opt.m, opt.v, self.partials, opt.lr, opt.step_count, opt.betas[0], opt.betas[1], opt.eps,
                opt.wd, opt.decoupled, gscale=gscale, max_norm=opt.max_norm, coef=opt.coef, gnorm=opt.grad_norm,
                betas_dev=opt.betas_dev)
        self.prog = pb.finalize(dev)

    def run(self):
        self.prog.run([], post=[(self.opt.step_count, 1)])
        self.opt.fp.bump()
