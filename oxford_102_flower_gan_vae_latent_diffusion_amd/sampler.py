"""DDPM reverse process (v2:594-598) as a replayed hipGraph.

The reference's loop makes a tensor->bool host sync every step (`if t[0] > 0`, v2:588) and launches ~100
eager kernels per step. Here one reverse step = denoiser forward + Philox noise + fused update + device-side
step counter is captured once and replayed; `t` lives in device memory (int32 for the update kernel, int64 for
the time embedding), so there is no host round trip inside the 1000-step loop."""
import os

import torch

from . import engine, ops

PERSISTENT = os.environ.get("FL_PERSISTENT", "1") != "0"


STEPS_PER_GRAPH = int(os.environ.get("FL_SAMPLER_STEPS_PER_GRAPH", "10"))


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
            left = n_steps - done
            # every graph launch costs ~9 us of idle device between its last kernel and the next launch's first (profiles/
            # r03_step_timeline.txt): a reverse step is 0.17 ms, so long runs replay STEPS_PER_GRAPH steps per launch (the step's
            # timestep, counters and draws are device-resident: the captured step is the same launch every time)
            if left >= 2 * STEPS_PER_GRAPH:
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, pool=g.pool()):
                    for _ in range(STEPS_PER_GRAPH):
                        self._step()
                self.graph_k = gk
                for _ in range(left // STEPS_PER_GRAPH):
                    gk.replay()
                left -= (left // STEPS_PER_GRAPH) * STEPS_PER_GRAPH
            for _ in range(left):
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


@torch.no_grad()
def denoise_from_many(diffusion, x_T, t_starts, c=None, use_graph=True, persistent=None):
    """visualize_denoising_steps (v2:687-691) as ONE batched run: the reference restarts from the same x_T for each t in
    `t_starts` and walks t..0 one p_sample at a time (sum(t+1) sequential steps). Here the restarts are the row groups of one
    batch (group i = the x_T rows started at the i-th largest t): at global step s exactly the groups with t >= s are alive,
    so the run is len(t_starts) phases of (t_i - t_{i+1}) steps on a growing row prefix - max(t)+1 sequential steps in all.
    Returns (len(t_starts) * n, latent) in the ORDER of `t_starts` (fresh, independent noise per trajectory, like the
    reference's per-call randn_like)."""
    n = x_T.shape[0]
    order = sorted(range(len(t_starts)), key=lambda i: -int(t_starts[i]))
    ts = [int(t_starts[i]) for i in order]
    G = len(ts)
    x = x_T.to(torch.float32).repeat(G, 1).contiguous()
    cc = c.long().repeat(G).contiguous() if c is not None else None
    for i, t in enumerate(ts):
        rows = (i + 1) * n
        nxt = ts[i + 1] if i + 1 < G else -1
        if t - nxt <= 0:
            continue
        g = GraphSampler(diffusion, (rows, x.shape[1]), x.device, cc[:rows] if cc is not None else None, use_graph=use_graph,
                         persistent=persistent)
        x[:rows].copy_(g.run(x_T=x[:rows], n_steps=t - nxt, t_start=t))
    out = torch.empty_like(x)
    for pos, i in enumerate(order):  # back to the caller's order
        out[i * n:(i + 1) * n].copy_(x[pos * n:(pos + 1) * n])
    return out


@torch.no_grad()
def denoising_steps_decoded(diffusion, autoencoder, x_T, t_starts, c=None):
    """v2:687-693: the partial trajectories above, decoded in one decoder batch -> (len(t_starts), n, 3, 64, 64)."""
    z = denoise_from_many(diffusion, x_T, t_starts, c)
    img = autoencoder.decode(z)
    return img.view(len(t_starts), x_T.shape[0], *img.shape[1:]), z


@torch.no_grad()
def renoise_decode(diffusion, autoencoder, x0, t_list, eps):
    """create_diffusion_animation's frames (v2:928-935) as two batched ops: every frame re-noises the SAME clean latent with
    the SAME eps (the reference re-seeds before each randn_like) at its own t - one fused q_sample over (frames x n) rows with
    per-row t (t = 0 keeps x0 untouched, v2:930) - and all frames go through the decoder together.
    Returns (len(t_list), n, 3, 64, 64)."""
    n, F = x0.shape[0], len(t_list)
    x0r = x0.to(torch.float32).repeat(F, 1).contiguous()
    er = eps.to(torch.float32).repeat(F, 1).contiguous()
    tt = torch.tensor([int(t) for t in t_list for _ in range(n)], dtype=torch.int64, device=x0.device)
    xt = diffusion.q_sample(x0r, tt, er)
    for f, t in enumerate(t_list):
        if int(t) <= 0:
            xt[f * n:(f + 1) * n].copy_(x0r[f * n:(f + 1) * n])
    img = autoencoder.decode(xt)
    return img.view(F, n, *img.shape[1:]), xt
