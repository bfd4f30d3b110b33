"""DDPM reverse process (v2:594-598) as a replayed hipGraph.

The reference's loop makes a tensor->bool host sync every step (`if t[0] > 0`, v2:588) and launches ~100
eager kernels per step. Here one reverse step = denoiser forward + Philox noise + fused update + device-side
step counter is captured once and replayed; `t` lives in device memory (int32 for the update kernel, int64 for
the time embedding), so there is no host round trip inside the 1000-step loop."""
import os

import torch

from . import engine, ops

PERSISTENT = os.environ.get("FL_PERSISTENT", "1") != "0"


class GraphSampler:
    """persistent=True (default): one reverse step is ONE launch of the persistent phase program (program.SamplerProgram:
    denoiser with the x-independent paths hoisted and the eval-mode linear chains folded, the update, inline Philox
    noise and the step counters); False: the multi-kernel engine forward + three small kernels per step."""

    def __init__(self, diffusion, shape, device, c=None, use_graph=True, persistent=None):
        self.d = diffusion
        self.shape = tuple(shape)
        self.device = torch.device(device)
        self.c = c.long().contiguous() if c is not None else None
        self.use_graph = use_graph
        self.graph = None
        self.persistent = PERSISTENT if persistent is None else persistent
        self.sprog = None

    def _step(self):
        d = self.d
        if self.persistent:
            if self.sprog is None:
                from .program import SamplerProgram
                self.sprog = SamplerProgram(d, self.shape[0], self.device, self.x, self.t64, self.t32, c=self.c)
            self.sprog.step()
            return
        eps, _ = engine.unet_forward(d.eps_model, self.x, self.t64, self.c, None, None, save=False)
        d.rng.normal(self.shape, self.device, out=self.noise)
        ops.p_sample_update(self.x, eps, self.noise, self.t32, d.alpha, d.alpha_bar, d.beta, out=self.x)
        ops.step_counter_add(self.t32, self.t64, -1)

    @torch.no_grad()
    def run(self, x_T=None, n_steps=None, t_start=None):
        d = self.d
        was_training = getattr(d.eps_model, "training", False)
        if was_training:
            d.eps_model.eval()
        n_steps = d.n_steps if n_steps is None else n_steps
        t0 = (d.n_steps - 1) if t_start is None else t_start
        self.sprog = self.graph = None  # both bake in the x / t buffers allocated below: rebuilt per run
        self.x = d.rng.normal(self.shape, self.device) if x_T is None else x_T.to(self.device, torch.float32).clone()
        self.noise = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        self.t32 = torch.tensor([t0], dtype=torch.int32, device=self.device)
        self.t64 = torch.tensor([t0], dtype=torch.int64, device=self.device)
        done = 0
        if n_steps > 0:
            self._step()  # eager first step: sizes the workspace, warms every kernel
            done = 1
        if self.use_graph and n_steps - done > 1:
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step()
            self.graph = g
            for _ in range(n_steps - done):
                g.replay()
        else:
            for _ in range(n_steps - done):
                self._step()
        if self.sprog is not None:
            self.sprog.prog.check()  # a barrier time-out voids the run (raises)
        if was_training:
            d.eps_model.train()
        return self.x


# ---- visualisation-facing helpers: the access patterns of visualize_denoising_steps (v2:687-691) and
# create_diffusion_animation (v2:928-935) as batched device-side operations ------------------------------------------
@torch.no_grad()
def denoise_from(diffusion, x_t, t_start, c=None, use_graph=True):
    """Run the reverse process from step `t_start` down to 0 starting at x_t (v2:687-691 restarts from the same x_T for
    several t values; pass x_T and t_start = n_steps-1 for a full trajectory)."""
    g = GraphSampler(diffusion, tuple(x_t.shape), x_t.device, c, use_graph=use_graph)
    return g.run(x_T=x_t, n_steps=int(t_start) + 1, t_start=int(t_start))


@torch.no_grad()
def renoise(diffusion, x0, t, eps=None):
    """x_t ~ q(x_t | x_0) for an integer t shared by the batch (v2:928-935 uses alpha_bar[t] directly)."""
    tt = torch.full((x0.shape[0],), int(t), dtype=torch.int64, device=x0.device)
    return diffusion.q_sample(x0, tt, eps)
