"""Training-step harnesses (the build's counterpart of the reference's loop bodies).

ddpm step  = v2:1254-1264: frozen-encoder forward -> reparameterize -> q_sample -> denoiser fwd/bwd ->
             global-norm clip (1.0) -> AdamW(lr, wd 1e-5), with every host sync of the reference removed
             (loss, norm, clip coefficient, step count and lr stay on the device), one RCCL all-reduce of the
             flat gradient buffer per step when data-parallel, and optional hipGraph capture of the whole step.
"""
import torch

from . import engine, ops


class FlatParams:
    """Re-homes a module's parameters into ONE flat fp32 buffer (grads into another), 16-byte aligned slots.
    Parameters stay real nn.Parameters (views), so state_dict()/load_state_dict()/torch.optim keep working.
    Parameters named in `inactive` (never touched by the forward: grad None in the reference, skipped by its
    optimizer) are placed last and excluded from clip / AdamW / all-reduce."""

    def __init__(self, module, inactive_prefixes=()):
        named = list(module.named_parameters())
        act = [(n, p) for n, p in named if not any(n.startswith(x) for x in inactive_prefixes)]
        ina = [(n, p) for n, p in named if any(n.startswith(x) for x in inactive_prefixes)]
        dev = named[0][1].device
        self.slots = {}
        off = 0
        for n, p in act + ina:
            self.slots[n] = (off, p.numel(), tuple(p.shape))
            off += (p.numel() + 3) // 4 * 4
            if n == act[-1][0]:
                self.n_active = off
        self.n_total = off
        self.flat = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.gviews = {}
        with torch.no_grad():
            for n, p in act + ina:
                o, k, shp = self.slots[n]
                v = self.flat[o:o + k].view(shp)
                v.copy_(p.data)
                p.data = v
                gv = self.grad[o:o + k].view(shp)
                self.gviews[n] = gv
                p.grad = gv
        self.module = module

    def bump(self):
        """Parameters were updated through raw pointers: invalidate derived (packed) layouts."""
        for m in self.module.modules():
            pk = getattr(m, "_pk", None)
            if pk is not None:
                pk.epoch += 1


class FusedAdam:
    """Adam/AdamW (torch semantics) + clip_grad_norm_ on a FlatParams, all scalars device-resident."""

    def __init__(self, fp, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=True, max_norm=None):
        dev = fp.flat.device
        self.fp, self.betas, self.eps, self.wd, self.decoupled, self.max_norm = fp, betas, eps, weight_decay, decoupled, max_norm
        self.m = torch.zeros_like(fp.flat)
        self.v = torch.zeros_like(fp.flat)
        self.lr = torch.tensor([lr], dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.coef = torch.ones(1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)

    def set_lr(self, lr):
        self.lr.fill_(lr)

    def step(self, gscale=1.0):
        fp = self.fp
        n = fp.n_active
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
                      self.eps, self.wd, coef, gscale, self.decoupled)
        fp.bump()


UNET_INACTIVE = ("time_projections.4.", "attention_layers.4.")


class DDPMTrainer:
    """Latent-DDPM training step (configs 3/4 of BASELINE.json)."""

    def __init__(self, autoencoder, unet, lr=1e-3, n_steps=1000, process_group=None, world_size=1):
        from .modules import ConditionalDenoiseDiffusion
        self.ae, self.unet = autoencoder, unet
        self.ae.eval()
        dev = next(unet.parameters()).device
        self.dev = dev
        self.diffusion = ConditionalDenoiseDiffusion(unet, n_steps=n_steps, device=dev)
        n_layers = len(unet.layers)
        inactive = tuple(p for p in ("time_projections.%d." % n_layers, "attention_layers.%d." % n_layers))
        self.fp = FlatParams(unet, inactive)
        self.opt = FusedAdam(self.fp, lr, weight_decay=1e-5, decoupled=True, max_norm=1.0)
        self.pg, self.world = process_group, world_size
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.eps_theta = None
        self.graph = None

    # -- pieces -------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, data, reparam_eps=None):
        """v2:1257-1259 (frozen encoder, eval)."""
        mu, logvar, _ = engine.encoder_infer(self.ae.encoder, data)
        if reparam_eps is None:
            reparam_eps = self.ae.rng.normal(mu.shape, mu.device)
        z, _ = ops.reparameterize(mu, logvar, reparam_eps)
        return z

    @torch.no_grad()
    def fwd_bwd(self, z, labels, t=None, eps=None, block_masks=None, head_masks=None):
        d = self.diffusion
        B = z.shape[0]
        if t is None:
            t = d.rng.randint((B,), z.device, d.n_steps)
        if eps is None:
            eps = d.rng.normal(z.shape, z.device)
        if self.unet.training and block_masks is None and self.unet.dropout_rate > 0:
            block_masks, head_masks = self.unet.dropout_masks(B, z.device)
        xt = ops.q_sample(z, eps, t, d.alpha_bar)
        eps_theta, S = engine.unet_forward(self.unet, xt, t, labels, block_masks, head_masks, save=True)
        loss, norms = ops.euclid_loss_fwd(eps_theta, eps)
        dout = ops.euclid_loss_bwd(eps_theta, eps, norms)
        engine.unet_backward(self.unet, S, dout, gout=self.fp.gviews)
        ops.axpby(1.0, loss, out=self.loss)
        self.eps_theta = eps_theta

    def allreduce(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.fp.grad[:self.fp.n_active], group=self.pg)

    @torch.no_grad()
    def update(self):
        self.opt.step(gscale=1.0 / self.world)

    # -- whole step ---------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, data, labels, reparam_eps=None, t=None, eps=None, block_masks=None, head_masks=None):
        """Returns the loss as a device tensor (no host sync)."""
        z = self.encode(data, reparam_eps)
        self.fwd_bwd(z, labels, t, eps, block_masks, head_masks)
        self.allreduce()
        self.update()
        return self.loss

    # -- hipGraph: capture [encode + fwd/bwd] and [clip + AdamW] around the (eager) all-reduce ----
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
        self.g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g2, pool=self.g1.pool()):
            self.update()
        self.graph = True

    def replay(self):
        self.g1.replay()
        self.allreduce()
        self.g2.replay()
        return self.loss
