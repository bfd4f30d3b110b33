"""Training-step harnesses (the build's counterpart of the reference's loop bodies).

ddpm step    = v2:1254-1264: frozen-encoder forward -> reparameterize -> q_sample -> denoiser fwd/bwd ->
               global-norm clip (1.0) -> AdamW(lr, wd 1e-5).
vae-gan step = v2:1060-1134: AE forward, 6 loss terms, D update (before the generator's adversarial term),
               device-side loss balancing, AE backward, clip, AdamW, centre EMA.
Every host sync of the reference (.item(), clip_grad_norm_) is removed: losses, norms, clip coefficient, balancing
scales, step count and lr stay on the device. Data-parallel runs add ONE RCCL all-reduce of the flat gradient
buffer per step; the ddpm step can be replayed as hipGraphs.
"""
import os

import torch

from . import engine, engine_bwd, ops

# the discriminator update's real and fake passes as one batch-stacked pass (engine_bwd.disc_fwd_pair); FL_PAIRED_D=0: two passes
PAIRED_D_PASSES = os.environ.get("FL_PAIRED_D", "1") != "0"
# the perceptual loss's two VGG passes (reconstruction, target) as one batch-stacked pass; FL_STACKED_VGG=0: two passes
STACKED_VGG = os.environ.get("FL_STACKED_VGG", "1") != "0"


class FlatParams:
    """Re-homes a module's parameters into ONE flat fp32 buffer (grads into another), 16-byte aligned slots.
    Parameters stay real nn.Parameters (views), so state_dict()/load_state_dict()/torch.optim keep working.
    Parameters whose name starts with one of `tail_prefixes` are placed last, in the given prefix order, so that an
    optimizer can exclude them (parameters the forward never touches get grad None in the reference and are skipped
    by its optimizer: `n_upto(prefix)` gives the length of the buffer before that group)."""

    def __init__(self, module, tail_prefixes=()):
        named = list(module.named_parameters())

        def group(n):
            for gi, pfx in enumerate(tail_prefixes):
                if n.startswith(pfx):
                    return gi + 1
            return 0

        order = sorted(range(len(named)), key=lambda i: (group(named[i][0]), i))
        dev = named[0][1].device
        self.slots, self._starts = {}, {}
        off = 0
        for i in order:
            n, p = named[i]
            gi = group(n)
            if gi and tail_prefixes[gi - 1] not in self._starts:
                self._starts[tail_prefixes[gi - 1]] = off
            self.slots[n] = (off, p.numel(), tuple(p.shape))
            off += (p.numel() + 3) // 4 * 4
        self.n_total = off
        first_tail = min(self._starts.values()) if self._starts else off
        self.n_active = first_tail
        self.flat = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.gviews = {}
        with torch.no_grad():
            for n, p in named:
                o, k, shp = self.slots[n]
                v = self.flat[o:o + k].view(shp)
                v.copy_(p.data)
                p.data = v
                gv = self.grad[o:o + k].view(shp)
                self.gviews[n] = gv
                p.grad = gv
        self.module = module

    def n_upto(self, prefix):
        return self._starts.get(prefix, self.n_total)

    def set_grads(self, G, prefix=""):
        """copy a {name: grad} dict into the flat gradient buffer."""
        ops.multi_axpby([(g.contiguous(), self.gviews[prefix + n]) for n, g in G.items() if g is not None])

    def add_grads(self, G, prefix=""):
        ops.multi_axpby([(g.contiguous(), self.gviews[prefix + n]) for n, g in G.items() if g is not None], 1.0, 1.0)

    def bump(self):
        """Parameters were updated through raw pointers: invalidate derived (packed) layouts."""
        for m in self.module.modules():
            pk = getattr(m, "_pk", None)
            if pk is not None:
                pk.epoch += 1
        self.module._raw_epoch = getattr(self.module, "_raw_epoch", 0) + 1  # folded inference programs key on it


class FusedAdam:
    """Adam/AdamW (torch semantics) + clip_grad_norm_ on a FlatParams, all scalars device-resident."""

    def __init__(self, fp, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=True, max_norm=None):
        dev = fp.flat.device
        self.fp, self.betas, self.eps, self.wd, self.decoupled, self.max_norm = fp, betas, eps, weight_decay, decoupled, max_norm
        self.m = torch.zeros_like(fp.flat)
        self.v = torch.zeros_like(fp.flat)
        self.lr = torch.tensor([lr], dtype=torch.float32, device=dev)
        # (beta1, beta2) live on the device like lr: torch's OneCycleLR cycles beta1 every batch (v2:1031-1037), and a captured
        # step or a resident optimizer program must follow it without being rebuilt
        self.betas_dev = torch.tensor([betas[0], betas[1]], dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.coef = torch.ones(1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)

    def set_lr(self, lr):
        self.lr.fill_(lr)

    def set_beta1(self, b1):
        """torch's OneCycleLR cycles AdamW's beta1 (schedules.OneCycle). Device-resident like lr: eager steps, captured
        graphs and the resident optimizer program all read it at run time."""
        self.betas = (float(b1), self.betas[1])
        self.betas_dev[0:1].fill_(float(b1))

    def state_dict(self):
        return {"m": self.m.detach().cpu(), "v": self.v.detach().cpu(), "step": self.step_count.detach().cpu(),
                "lr": self.lr.detach().cpu(), "betas": tuple(self.betas)}

    def load_state_dict(self, st):
        self.m.copy_(st["m"])
        self.v.copy_(st["v"])
        self.step_count.copy_(st["step"])
        self.lr.copy_(st["lr"])
        if "betas" in st:
            self.betas = tuple(st["betas"])
            self.betas_dev.copy_(torch.tensor(self.betas, dtype=torch.float32))

    def step(self, gscale=1.0, n=None):
        fp = self.fp
        n = fp.n_active if n is None else n
        g = fp.grad[:n]
        coef = None
        if self.max_norm is not None:
            if gscale != 1.0:
                ops.axpby(gscale, g, out=g)
                gscale = 1.0
            ops.sqnorm(g, self.sq)
            ops.clip_coef(self.sq, self.max_norm, self.coef, self.grad_norm)
            coef = self.coef
        ops.step_counter_add(self.step_count, None, 1)
        ops.adam_step(fp.flat[:n], g, self.m[:n], self.v[:n], self.lr, self.step_count, self.betas[0], self.betas[1],
                      self.eps, self.wd, coef, gscale, self.decoupled, betas_dev=self.betas_dev)
        fp.bump()


def allreduce_grads(fp, world_size, group=None, force=False):
    """The ONE collective of a data-parallel step: sum the flat active-gradient buffer over ranks (RCCL over xGMI
    when the buffer lives on a GPU, gloo in the CPU tests). The 1/world factor is applied by the caller (folded into
    the clip/Adam kernel), so that mean-of-shard-gradients == global-batch gradient (the loss is a batch mean,
    v2:298)."""
    if world_size > 1 or force:   # force: a one-rank group rehearsing the data-parallel path (tests)
        import torch.distributed as dist
        dist.all_reduce(fp.grad[:fp.n_active], group=group)
    return fp.grad[:fp.n_active]


def shard_seed(base, rank):
    """Per-rank RNG seed offset (t / eps / dropout draws are independent per rank)."""
    return int(base) + 97 * int(rank)


class DDPMTrainer:
    """Latent-DDPM training step (configs 3/4 of BASELINE.json)."""

    def __init__(self, autoencoder, unet, lr=1e-3, n_steps=1000, process_group=None, world_size=1, persistent=None,
                 force_collectives=False):
        """persistent: run the denoiser forward/backward and clip+AdamW as persistent phase programs (program_train.py:
        2 launches instead of ~240; they need the whole device, so nothing else may run next to them). None = the
        FL_TRAIN_PROGRAM environment variable (default on)."""
        from .modules import ConditionalDenoiseDiffusion
        import os
        self.persistent = (os.environ.get("FL_TRAIN_PROGRAM", "1") == "1") if persistent is None else bool(persistent)
        self._tprog, self._oprog = {}, None
        self.ae, self.unet = autoencoder, unet
        self.ae.eval()
        dev = next(unet.parameters()).device
        self.dev = dev
        self.diffusion = ConditionalDenoiseDiffusion(unet, n_steps=n_steps, device=dev)
        n_layers = len(unet.layers)
        # never touched by the forward (v2:540-553 loops over len(layers)): grad None in the reference
        inactive = ("time_projections.%d." % n_layers, "attention_layers.%d." % n_layers)
        self.fp = FlatParams(unet, inactive)
        self.opt = FusedAdam(self.fp, lr, weight_decay=1e-5, decoupled=True, max_norm=1.0)
        self.pg, self.world = process_group, world_size
        # force_collectives: issue the gradient all-reduce on a ONE-rank group too - the rehearsal of the N > 1 schedule on real RCCL
        # that a one-GPU box allows (tests/test_ddpm_dp_gpu.py)
        self.dp = world_size > 1 or bool(force_collectives)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.eps_theta = None
        self.graph = None

    @torch.no_grad()
    def encode(self, data, reparam_eps=None):
        """v2:1257-1259 (frozen encoder, eval)."""
        if reparam_eps is None:   # draws from the autoencoder's Philox stream: made inside the heads' tail kernel where that runs
            return engine.encoder_latent(self.ae.encoder, data, self.ae.rng)
        mu, logvar, _ = engine.encoder_infer(self.ae.encoder, data, want_feats=False)
        z, _ = ops.reparameterize(mu, logvar, reparam_eps)
        return z

    @torch.no_grad()
    def fwd_bwd(self, z, labels, t=None, eps=None, block_masks=None, head_masks=None):
        d = self.diffusion
        B = z.shape[0]
        use_prog = self.persistent and labels is not None and (t is None or t.numel() == B) and self._program_shapes_ok()
        # on the program path the Philox counters of this step's draws are bumped by the program's post hook (the last block
        # to leave) instead of one tiny launch per draw: same numbers (offset = counters consumed so far)
        d_adv = m_adv = 0
        xt = None
        want_masks = self.unet.training and block_masks is None and self.unet.dropout_rate > 0
        if use_prog and t is None and eps is None and engine.FUSED_HEADS and z.shape[1] % 4 == 0:
            # t, noise, x_t and the dropout scales in ONE launch (same numbers as the draws below, counter for counter)
            spec = self.unet.dropout_masks(B, z.device, defer=True, draw=False) if want_masks else None
            flat = spec[0] if spec is not None and spec[0] is not None else None
            t, eps, xt = ops.diffuse_prep(z, d.alpha_bar, d.n_steps, d.rng.seed, d.rng.state(z.device), masks=flat,
                                          mask_seed=self.unet.rng.seed, mask_state=self.unet.rng.state(z.device),
                                          p=spec[1] if flat is not None else 0.0)
            d_adv = B + d.rng.fill_count(z.numel())
            if flat is not None:
                block_masks, head_masks, m_adv = spec[2], spec[3], spec[4]
        if t is None:
            t = d.rng.randint((B,), z.device, d.n_steps, advance=not use_prog)
            d_adv += B
        if eps is None:
            eps = d.rng.normal(z.shape, z.device, offset=d_adv if use_prog else 0, advance=not use_prog)
            d_adv += d.rng.fill_count(z.numel())
        if self.unet.training and block_masks is None and self.unet.dropout_rate > 0:
            if use_prog:
                block_masks, head_masks, m_adv = self.unet.dropout_masks(B, z.device, defer=True)
            else:
                block_masks, head_masks = self.unet.dropout_masks(B, z.device)
        if xt is None:
            xt = ops.q_sample(z, eps, t, d.alpha_bar)
        if use_prog:
            prog = self._train_program(B, block_masks is not None)
            self._last_tp = prog.programs[-1]   # the optimizer program watches this run's abort word
            post = [(r.state(z.device), n) for r, n in ((d.rng, d_adv), (self.unet.rng, m_adv)) if n]
            self.eps_theta = prog.run(xt, t, labels, eps, block_masks, head_masks, post=post)   # loss and every gradient included
            return
        eps_theta, S = engine.unet_forward(self.unet, xt, t, labels, block_masks, head_masks, save=True)
        loss, norms = ops.euclid_loss_fwd(eps_theta, eps)
        dout = ops.euclid_loss_bwd(eps_theta, eps, norms)
        engine.unet_backward(self.unet, S, dout, gout=self.fp.gviews)
        ops.axpby(1.0, loss, out=self.loss)
        self.eps_theta = eps_theta

    def _program_shapes_ok(self):
        un = self.unet
        dims = [blk[0].weight.shape[0] for blk, _, _ in un.layers] + [un.final_time_proj.weight.shape[0]]
        return all(dd % 256 == 0 and dd <= 1024 for dd in dims) and un.latent_dim % 256 == 0 and un.latent_dim <= 1024 and \
            un.time_emb_dim % 32 == 0 and un.time_emb.lin1.weight.shape[0] % 32 == 0 and \
            un.class_emb.lin1.weight.shape[0] % 32 == 0

    def _train_program(self, B, masks):
        from .program_train import DenoiserTrainProgram
        prog = self._tprog.get((B, masks))
        if prog is not None and prog.key != tuple(p.data_ptr() for p in self.unet.parameters()) + tuple(
                g.data_ptr() for g in self.fp.gviews.values()):
            prog = None  # a parameter was re-allocated: rebuild
        if prog is None:
            prog = DenoiserTrainProgram(self.unet, self.fp.gviews, B, self.dev, masks=masks, loss_out=self.loss)
            self._tprog[(B, masks)] = prog
        return prog

    def allreduce(self):
        allreduce_grads(self.fp, self.world, self.pg, force=self.dp)

    @torch.no_grad()
    def update(self):
        if self.persistent:
            from .program_train import OptimizerProgram
            o, n, gs = self.opt, self.fp.n_active, 1.0 / self.world
            key = (n, gs, o.eps, o.wd, o.decoupled, o.max_norm, self.fp.flat.data_ptr(), self.fp.grad.data_ptr())
            if self._oprog is None or self._oprog.key != key:
                self._oprog = OptimizerProgram(o, n, gs)
            # the optimizer must not apply a gradient whose forward / backward run was void (barrier time-out, label outside
            # the table): it watches the training program's abort word
            self._oprog.prog.guard = getattr(self, "_last_tp", None)
            self._oprog.run()
            return
        self.opt.step(gscale=1.0 / self.world)

    @torch.no_grad()
    def step(self, data, labels, reparam_eps=None, t=None, eps=None, block_masks=None, head_masks=None):
        """Returns the loss as a device tensor (no host sync)."""
        z = self.encode(data, reparam_eps)
        self.fwd_bwd(z, labels, t, eps, block_masks, head_masks)
        self.allreduce()
        self.update()
        return self.loss

    # hipGraph: capture [encode + fwd/bwd] and [clip + AdamW] around the (eager) all-reduce
    def capture(self, data, labels):
        """data/labels become static input buffers; call replay() after copying new batches into them."""
        self.static_data, self.static_labels = data, labels
        for _ in range(2):  # warm-up: sizes the workspace, fills pack caches
            self.step(data, labels)
        torch.cuda.synchronize(self.dev)
        self.g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g1):
            z = self.encode(self.static_data)
            self.fwd_bwd(z, self.static_labels)
            if not self.dp:   # nothing sits between the gradients and the update: one graph launch per step instead of two
                self.update()
        self.g2 = None
        if self.dp:
            self.g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g2, pool=self.g1.pool()):
                self.update()
        self.graph = True

    def check_programs(self):
        """Synchronising health check of the persistent programs (a barrier that timed out voids the run: raises)."""
        for prog in self._tprog.values():
            for q in prog.programs:
                q.check()
        if self._oprog is not None:
            self._oprog.prog.check()

    def _replayed(self):
        # a replayed optimizer graph updates the weights through raw pointers without running FlatParams.bump(): tell the
        # weight-version consumers (program.FoldedUNetProgram) that the denoiser's weights moved
        self.unet._raw_epoch = getattr(self.unet, "_raw_epoch", 0) + 1
        # graph replays skip Program.run's own (eager-only) health check: look every 256 steps (one sync)
        self._replays = getattr(self, "_replays", 0) + 1
        if self.persistent and self._replays % 256 == 0:
            self.check_programs()

    def replay(self):
        self.g1.replay()
        if self.g2 is not None:
            self.allreduce()
            self.g2.replay()
        self._replayed()
        return self.loss

    # ---- data-parallel step with the gradient all-reduce hidden behind the NEXT batch's encoder ---------------
    # The persistent programs need the device to themselves, the frozen encoder does not, and it does not depend on
    # the update either (v2:1257-1259): [denoiser fwd/bwd program (k)] -> all-reduce (k) on RCCL's stream, WHILE the
    # encoder graph of batch k+1 runs on the compute stream -> [optimizer program (k)]. Same arithmetic and update order
    # as step(); one collective per step, off the critical path instead of serialised between two launches.
    def capture_overlapped(self, data, labels):
        dev = self.dev
        self.static_data, self.static_labels = data, labels
        for _ in range(2):
            self.step(data, labels)
        torch.cuda.synchronize(dev)
        B, L = data.shape[0], self.ae.latent_dim
        self.z_next = torch.zeros(B, L, device=dev)
        self.z_in = torch.zeros(B, L, device=dev)
        self.lab_next, self.lab_in = labels.clone(), labels.clone()
        self.gE = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gE):
            self.z_next.copy_(self.encode(self.static_data))
        self.gU = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gU, pool=self.gE.pool()):
            self.fwd_bwd(self.z_in, self.lab_in)
        self.gO = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gO, pool=self.gE.pool()):
            self.update()

    def prime_overlapped(self, data, labels):
        """encode the first batch (before the timed region)."""
        self.static_data.copy_(data)
        self.gE.replay()
        self.lab_next.copy_(labels)

    def replay_overlapped(self, next_data, next_labels):
        """one step on the latents encoded by the previous call; `next_data` is encoded under the all-reduce."""
        self.z_in.copy_(self.z_next)
        self.lab_in.copy_(self.lab_next)
        self.gU.replay()
        work = None
        if self.dp:
            import torch.distributed as dist
            work = dist.all_reduce(self.fp.grad[:self.fp.n_active], group=self.pg, async_op=True)
        self.static_data.copy_(next_data)
        self.gE.replay()
        self.lab_next.copy_(next_labels)
        if work is not None:
            work.wait()      # the compute stream waits for the collective; the host does not block
        self.gO.replay()
        self._replayed()
        return self.loss

    def measure_comm(self, batches, labels, steps_n=10, overlapped=True):
        """Timing of the data-parallel collective (N > 1; run AFTER a timed region, it synchronises): `allreduce_ms` = the
        flat-gradient all-reduce alone (HIP events around `steps_n` back-to-back collectives, median), `allreduce_exposed_ms` =
        what a step waits for it (overlapped schedule: from the end of the next batch's encoder graph to the collective's
        completion on the compute stream; plain schedule: its whole duration), plus the step pieces it hides under."""
        import statistics
        dev = self.dev
        ev = lambda: torch.cuda.Event(enable_timing=True)
        n = self.fp.n_active
        alone = []
        for _ in range(steps_n):
            e0, e1 = ev(), ev()
            torch.cuda.synchronize(dev)
            e0.record()
            if self.dp:
                import torch.distributed as dist
                dist.all_reduce(self.fp.grad[:n], group=self.pg)
            e1.record()
            torch.cuda.synchronize(dev)
            alone.append(e0.elapsed_time(e1))
        res = {"allreduce_ms": statistics.median(alone), "allreduce_bytes": 4 * n, "world": self.world,
               "schedule": "all-reduce of step k under the encoder of batch k+1" if overlapped else "all-reduce between the two graphs"}
        if overlapped and hasattr(self, "gE"):
            exposed, enc, den = [], [], []
            for i in range(steps_n):
                e0, e1, e2, e3 = ev(), ev(), ev(), ev()
                self.z_in.copy_(self.z_next)
                self.lab_in.copy_(self.lab_next)
                e0.record()
                self.gU.replay()
                e1.record()
                work = None
                if self.dp:
                    import torch.distributed as dist
                    work = dist.all_reduce(self.fp.grad[:n], group=self.pg, async_op=True)
                self.static_data.copy_(batches[i % len(batches)])
                self.gE.replay()
                self.lab_next.copy_(labels[i % len(labels)])
                e2.record()
                if work is not None:
                    work.wait()
                e3.record()
                self.gO.replay()
                self._replayed()
                torch.cuda.synchronize(dev)
                den.append(e0.elapsed_time(e1))
                enc.append(e1.elapsed_time(e2))
                exposed.append(e2.elapsed_time(e3))
            res.update({"allreduce_exposed_ms": statistics.median(exposed), "encoder_graph_ms": statistics.median(enc),
                        "denoiser_program_ms": statistics.median(den)})
        else:
            res["allreduce_exposed_ms"] = res["allreduce_ms"]
        return res

    # ---- cross-step software pipeline -------------------------------------------------------------------
    # The frozen encoder of batch k+1 does not depend on the denoiser update of batch k (v2:1257-1259 runs under
    # no_grad on a module that is never trained here), so the two phases run concurrently: stream E replays the
    # encoder graph for the NEXT batch (large MFMA-bound conv kernels) while stream U replays the denoiser
    # fwd/bwd + all-reduce + clip/AdamW for the CURRENT latents (hundreds of small latency-bound launches that
    # leave most CUs idle). Same arithmetic, same order of parameter updates; z/labels are double-buffered.
    def capture_pipelined(self, data, labels):
        dev = self.dev
        self.static_data, self.static_labels = data, labels
        self.persistent = False  # two streams share the device here: a persistent program cannot run next to stream E
        for _ in range(2):
            self.step(data, labels)
        # stream priorities (experiment knob, default none): FL_PIPE_PRIO="e,u", -1 = high
        import os
        pe, pu = (int(v) for v in os.environ.get("FL_PIPE_PRIO", "0,0").split(","))
        self.sE, self.sU = torch.cuda.Stream(dev, priority=pe), torch.cuda.Stream(dev, priority=pu)
        torch.cuda.synchronize(dev)
        B, L = data.shape[0], self.ae.latent_dim
        self.z_next = torch.zeros(B, L, device=dev)
        self.z_in = torch.zeros(B, L, device=dev)
        self.lab_next, self.lab_in = labels.clone(), labels.clone()
        old = ops.set_workspace_slot(1)           # encoder graph: its own scratch (runs concurrently with U)
        self.encode(data)                         # size slot-1 workspace outside capture
        torch.cuda.synchronize(dev)
        # the encoder graph runs NEXT to the denoiser graph: capture its 3x3 convs in the variant that leaves registers and
        # LDS for the other stream's waves (FL_PIPE_HALO_MODE=0 keeps the stand-alone-fastest variant)
        halo_mode = int(os.environ.get("FL_PIPE_HALO_MODE", "1"))
        old_mode = ops.set_conv_halo_mode(halo_mode)
        self.gE = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gE):
            self.z_out = self.encode(self.static_data)
        ops.set_conv_halo_mode(old_mode)
        ops.set_workspace_slot(old)
        # the denoiser shares the chip with the encoder stream: optionally capture it with a lower split-K threshold
        # (fewer dependent launches; FL_PIPE_SPLITK_TILES, experiment knob)
        import os as _os
        skt = _os.environ.get("FL_PIPE_SPLITK_TILES")
        old_skt = ops.set_gemm_splitk_tiles(int(skt)) if skt is not None else None
        self.gU = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gU):
            self.fwd_bwd(self.z_in, self.lab_in)
        if old_skt is not None:
            ops.set_gemm_splitk_tiles(old_skt)
        self.gO = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gO):
            self.update()
        self.ev_enc, self.ev_used = torch.cuda.Event(), torch.cuda.Event()
        self._primed = False

    def _launch_encoder(self, data, labels):
        cur = torch.cuda.current_stream(self.dev)
        self.sE.wait_stream(cur)
        with torch.cuda.stream(self.sE):
            self.static_data.copy_(data, non_blocking=True)
            self.gE.replay()
            if self._primed:
                self.sE.wait_event(self.ev_used)   # z_next / lab_next of the previous batch have been consumed
            self.z_next.copy_(self.z_out, non_blocking=True)
            self.lab_next.copy_(labels, non_blocking=True)
            self.ev_enc.record(self.sE)

    def prime_pipeline(self, data, labels):
        """encode the first batch (before the timed region)."""
        self._launch_encoder(data, labels)
        self._primed = True

    def replay_pipelined(self, next_data, next_labels):
        """one step: denoiser update on the latents encoded by the previous call, while `next_data` is encoded."""
        with torch.cuda.stream(self.sU):
            self.sU.wait_event(self.ev_enc)
            self.z_in.copy_(self.z_next, non_blocking=True)
            self.lab_in.copy_(self.lab_next, non_blocking=True)
            self.ev_used.record(self.sU)
        self._launch_encoder(next_data, next_labels)
        with torch.cuda.stream(self.sU):
            self.gU.replay()
            self.allreduce()
            self.gO.replay()
        self.unet._raw_epoch = getattr(self.unet, "_raw_epoch", 0) + 1
        return self.loss

    def sync_pipeline(self):
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_stream(self.sE)
        cur.wait_stream(self.sU)


def loss_phase(epoch):
    """Loss gating of v2:1072-1087 -> (kl_factor, cls_factor, center_factor)."""
    if epoch < 40:
        return 0.0, 0.0, 0.0
    if epoch < 80:
        return min(1.0, (epoch - 20) / 20), 0.0, 0.0
    if epoch < 160:
        return 1.0, min(0.2, (epoch - 40) / 20), 0.0
    return 1.0, 1.0, min(1.0, (epoch - 60) / 20)


def kl_weight_at(epoch, num_epochs, start=0.001, end=0.05):
    """v2:1055."""
    return min(end, start + (epoch / (num_epochs * 0.6)) * (end - start))


class VAEGANTrainer:
    """VAE-GAN training step (configs 1/2 of BASELINE.json), v2:1060-1134.

    Data parallel (world_size > 1; one process per GPU, each with a shard of the batch): the step is the
    GLOBAL-batch step of the reference -
      * the gradients of the autoencoder and of the discriminator are all-reduced (flat buffers, two collectives) and
        averaged inside the Adam kernels;
      * the six loss terms that drive the balancing coefficients (v2:1107-1125) are all-reduced first, so every rank uses
        the coefficients of the global batch; the KL term is the mean over the batch PLUS a batch sum 1e-4*sum(mu^2)
        (v2:385): the mean part is averaged, the sum part added over ranks, and its gradient scaled by world_size
        (shard gradients are averaged afterwards);
      * the class-centre EMA (v2:369-378) runs on the all-gathered latents and labels, identically on every rank;
      * Discriminator64's BatchNorm layers are synchronised (sync_bn=True, the default): the per-channel sums of the
        forward and of the backward are all-reduced (2C doubles per layer per pass, fl_batchnorm_sums ->
        fl_batchnorm_*_from_sums), so the statistics, the running buffers and dx are those of the global batch.
        sync_bn=False keeps the LOCAL shard statistics (what torch DDP does without SyncBatchNorm); the adversarial
        gradients then differ from a single-process run on the global batch.
    The collectives are torch.distributed calls on the step's stream: with RCCL (backend "nccl") capture() records them into the
    step's hipGraph like any kernel (round 3; a gloo group cannot be captured: step() only). force_collectives=True runs the
    data-parallel code path on a ONE-rank group - the rehearsal the one-GPU test box allows (tests/test_vaegan_dp_gpu.py)."""

    def __init__(self, autoencoder, discriminator, vgg, lr=1e-4, lambda_cls=0.1, lambda_center=0.05, lambda_vgg=0.4,
                 lambda_gan=0.2, process_group=None, world_size=1, sync_bn=True, force_collectives=False):
        self.ae, self.disc, self.vgg = autoencoder, discriminator, vgg
        self.pg, self.world = process_group, int(world_size)
        self.dp = self.world > 1 or bool(force_collectives)
        self.bn_sync = None
        if self.dp and sync_bn:
            import torch.distributed as dist
            self.bn_sync = lambda t: dist.all_reduce(t, group=self.pg)
        dev = next(autoencoder.parameters()).device
        self.dev = dev
        self.fp = FlatParams(autoencoder, ("classifier.",))
        self.opt = FusedAdam(self.fp, lr, weight_decay=1e-5, decoupled=True, max_norm=1.0)       # v2:1026, v2:1128
        self.dfp = FlatParams(discriminator)
        self.d_opt = FusedAdam(self.dfp, 1e-4, betas=(0.5, 0.999), weight_decay=0.0, decoupled=False)  # v2:1028
        self.lam = (lambda_cls, lambda_center, lambda_vgg, lambda_gan)
        self.terms = torch.zeros(6, dtype=torch.float32, device=dev)
        self._wcache = {}
        self.out = {}
        self.graph = None

    def capture(self, data, labels, epoch=0, num_epochs=300):
        """Capture the whole step (one loss phase = one graph) as a hipGraph; `data`/`labels` become the static input
        buffers. replay() after copying a new batch into them."""
        if self.dp:
            import torch.distributed as dist
            if dist.get_backend(self.pg) != "nccl":
                raise RuntimeError("only RCCL collectives (backend 'nccl') can be captured into the step's hipGraph: with %r call step(), "
                                   "not capture()" % dist.get_backend(self.pg))
        self.static_data, self.static_labels = data, labels
        for _ in range(2):
            self.step(data, labels, epoch=epoch, num_epochs=num_epochs)
        torch.cuda.synchronize(self.dev)
        # the graph updates the weights it packs: every derived layout (PackCache) must be rebuilt INSIDE the graph, from
        # the first use on - a pack made by the warm-up steps would be frozen into the replays (and freed under them)
        self.fp.bump()
        self.dfp.bump()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.step(self.static_data, self.static_labels, epoch=epoch, num_epochs=num_epochs)

    def replay(self):
        self.graph.replay()
        # the graph's own packs were made from the weights BEFORE its update: eager code that runs after a replay (evaluation,
        # sampling, a checkpointed decode) must re-derive its packed layouts from the weights the graph left behind
        self.fp.bump()
        self.dfp.bump()
        return self.out

    @torch.no_grad()
    def step(self, data, labels, epoch=0, num_epochs=300, lr=None, reparam_eps=None, cls_masks=(None, None)):
        ae, D, vgg = self.ae, self.disc, self.vgg
        l_cls, l_cen, l_vgg, l_gan = self.lam
        klw = kl_weight_at(epoch, num_epochs)
        klf, clsf, cenf = loss_phase(epoch)
        if lr is not None:
            self.opt.set_lr(lr)
        B = data.shape[0]
        data = data.contiguous()
        # ---- forward (v2:1069) ----
        mu, logvar, feats, S_enc = engine_bwd.encoder_fwd(ae.encoder, data)
        ae.encoder.skip_features = [f.permute(0, 3, 1, 2) for f in feats]
        if reparam_eps is None:
            reparam_eps = ae.rng.normal(mu.shape, mu.device)
        z, _ = ops.reparameterize(mu, logvar, reparam_eps)
        recon, S_dec = engine_bwd.decoder_fwd(ae.decoder, z)
        # ---- losses (v2:1090-1094) ----
        recon_loss, rn = ops.euclid_loss_fwd(recon, data)
        if STACKED_VGG:
            fx, fy, tape = engine_bwd.vgg_features_pair(vgg, recon, data)
        else:
            fx, tape = engine_bwd.vgg_features(vgg, recon, True)
            fy, _ = engine_bwd.vgg_features(vgg, data, False)
        percep, pn = ops.euclid_loss_fwd(fx, fy)
        zero = torch.zeros(1, dtype=torch.float32, device=self.dev)
        W = self.world
        musq = None
        if klf > 0:
            kl, kl_row, musq = ops.kl_fwd(mu, logvar, want_musq=True)
        else:
            kl, kl_row = zero, None
        if clsf > 0:
            m1, m2 = cls_masks
            if ae.training and m1 is None:
                c = ae.classifier
                m1 = ae.rng.dropout((B, 512), z.device, c[3].p)
                m2 = ae.rng.dropout((B, 256), z.device, c[7].p)
            logits, S_cls = engine_bwd.classifier_fwd(ae, z, m1, m2)
            ce, probs = ops.cross_entropy_fwd(logits, labels)
        else:
            ce = zero
        if cenf > 0:
            cb = ops.gather_rows(ae.class_centers, labels)
            cen, cn = ops.euclid_loss_fwd(z, cb)
        else:
            cen = zero
        # ---- discriminator update (v2:1097-1102) ----
        bns = dict(sync=self.bn_sync, world=self.world)
        if PAIRED_D_PASSES and D.training:  # (an eval-mode discriminator normalises with its running statistics: two plain passes)
            # the real and the fake pass as one batch-stacked pass with per-pass BatchNorm statistics: half the launches, weight-
            # gradient GEMMs over both batches, and one SyncBatchNorm exchange per layer and direction for both (12 per step, was 18)
            (p_real, p_fake), S_d = engine_bwd.disc_fwd_pair(D, data, recon, **bns)  # recon_x.detach(): no gradient reaches the AE
            l_real, l_fake = ops.bce_fwd(p_real, 1.0), ops.bce_fwd(p_fake, 0.0)
            d_loss = ops.axpby(0.5, l_real, 0.5, l_fake)
            self.dfp.set_grads(engine_bwd.disc_bwd_pair(D, S_d, ops.bce_bwd(p_real, 1.0, 0.5), ops.bce_bwd(p_fake, 0.0, 0.5)))
        else:
            p_real, S_r = engine_bwd.disc_fwd(D, data, **bns)
            p_fake, S_f = engine_bwd.disc_fwd(D, recon, **bns)
            l_real, l_fake = ops.bce_fwd(p_real, 1.0), ops.bce_fwd(p_fake, 0.0)
            d_loss = ops.axpby(0.5, l_real, 0.5, l_fake)
            Gr, _ = engine_bwd.disc_bwd(D, S_r, ops.bce_bwd(p_real, 1.0, 0.5), need_wgrad=True, need_dx=False)
            Gf, _ = engine_bwd.disc_bwd(D, S_f, ops.bce_bwd(p_fake, 0.0, 0.5), need_wgrad=True, need_dx=False)
            self.dfp.set_grads(Gr)
            self.dfp.add_grads(Gf)
        allreduce_grads(self.dfp, W, self.pg, force=self.dp)
        self.d_opt.step(gscale=1.0 / W)
        # ---- generator adversarial term with the UPDATED discriminator (v2:1105) ----
        p_adv, S_a = engine_bwd.disc_fwd(D, recon, **bns)
        adv = ops.bce_fwd(p_adv, 1.0)
        # ---- device-side loss balancing (v2:1107-1125) ----
        if self.dp:
            import torch.distributed as dist
            # [recon, percep, kl mean part, adv, ce, centre | kl batch-sum part | d_loss]: means are averaged, the sum is summed
            tv = torch.zeros(8, dtype=torch.float32, device=self.dev)
            reg = ops.colsum(musq.view(B, 1), alpha=1e-4) if musq is not None else zero
            klmean = ops.axpby(1.0, kl, -1.0, reg) if musq is not None else zero
            for i, tsr in enumerate((recon_loss, percep, klmean, adv, ce, cen)):
                ops.axpby(1.0 / W, tsr, out=tv[i:i + 1])
            ops.axpby(1.0, reg, out=tv[6:7])
            ops.axpby(1.0 / W, d_loss, out=tv[7:8])
            dist.all_reduce(tv, group=self.pg)
            recon_loss, percep, adv, ce, cen, d_loss = tv[0:1], tv[1:2], tv[3:4], tv[4:5], tv[5:6], tv[7:8]
            kl = ops.axpby(1.0, tv[2:3], 1.0, tv[6:7])
        for i, tsr in enumerate((recon_loss, percep, kl, adv, ce, cen)):
            ops.axpby(1.0, tsr, out=self.terms[i:i + 1])
        wkey = (1.0, l_vgg, klw * klf, l_gan, l_cls * clsf, l_cen * cenf)
        w = self._wcache.get(wkey)
        if w is None:  # one H2D copy per distinct (epoch-dependent) weight vector, never inside the step proper
            w = torch.tensor(wkey, dtype=torch.float32, device=self.dev)
            self._wcache[wkey] = w
        coeff, total = ops.balance_losses(self.terms, w)
        # ---- backward of total_loss w.r.t. the autoencoder (v2:1127) ----
        drecon = ops.euclid_loss_bwd(recon, data, rn, 1.0, coeff[0:1])
        dfx = ops.euclid_loss_bwd(fx, fy, pn, 1.0, coeff[1:2])
        drecon = engine_bwd._add(drecon, engine_bwd.vgg_features_bwd(vgg, tape, dfx))
        _, dr_adv = engine_bwd.disc_bwd(D, S_a, ops.bce_bwd(p_adv, 1.0, 1.0, coeff[3:4]), need_wgrad=False, need_dx=True)
        drecon = engine_bwd._add(drecon, dr_adv)
        G_dec, dz = engine_bwd.decoder_bwd(ae.decoder, S_dec, drecon)
        n_act = self.fp.n_upto("classifier.")
        if clsf > 0:
            G_cls, dz_cls = engine_bwd.classifier_bwd(ae, S_cls, ops.cross_entropy_bwd(probs, labels, 1.0, coeff[4:5]))
            dz = engine_bwd._add(dz, dz_cls)
            self.fp.set_grads(G_cls, "classifier.")
            n_act = self.fp.n_total
        if cenf > 0:
            dz = engine_bwd._add(dz, ops.euclid_loss_bwd(z, cb, cn, 1.0, coeff[5:6]))
        dmu, dlv = dz, ops.reparam_bwd(dz, logvar, reparam_eps)
        if klf > 0:
            dmu_k, dlv_k = ops.kl_bwd(mu, logvar, kl_row, 1.0, coeff[2:3], reg_scale=float(W))
            dmu, dlv = engine_bwd._add(dmu, dmu_k), engine_bwd._add(dlv, dlv_k)
        G_enc, _ = engine_bwd.encoder_bwd(ae.encoder, S_enc, dmu, dlv)
        self.fp.set_grads(G_enc, "encoder.")
        self.fp.set_grads(G_dec, "decoder.")
        if self.dp:
            import torch.distributed as dist
            dist.all_reduce(self.fp.grad[:n_act], group=self.pg)
        self.opt.step(gscale=1.0 / W, n=n_act)                   # clip 1.0 + AdamW (v2:1128-1129)
        if epoch >= 60 and cenf > 0:                             # v2:1132-1134
            zc, lc = z, labels
            if self.dp:  # every rank applies the EMA of the global batch: gather (sum of disjoint slices: any backend)
                import torch.distributed as dist
                r = dist.get_rank(self.pg)
                zc = torch.zeros(W * B, z.shape[1], dtype=torch.float32, device=self.dev)
                lc = torch.zeros(W * B, dtype=torch.int64, device=self.dev)
                zc[r * B:(r + 1) * B].copy_(z)
                lc[r * B:(r + 1) * B].copy_(labels)
                dist.all_reduce(zc, group=self.pg)
                dist.all_reduce(lc, group=self.pg)
            ops.update_centers(ae.class_centers, zc, lc, 0.9)
        self.out = dict(total=total, recon=recon_loss, perceptual=percep, kl=kl, cls=ce, center=cen, adv=adv,
                        d_loss=d_loss, recon_x=recon, mu=mu, logvar=logvar, z=z, grad_norm=self.opt.grad_norm)
        return self.out
