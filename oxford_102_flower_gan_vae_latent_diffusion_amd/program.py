"""Host side of the persistent phase-program executor (include/flowers_hip.h, fl_program_run; csrc/program.hip).

`ProgramBuilder` packs jobs/phases into the C structs of the header; `build_unet_program` lays the ConditionalUNet
forward (v2:534-561) out as 26 phases: the per-layer GEMMs are fused by concatenating K (h + time_proj(tc) is one
contraction, the `down` projection of layer i is folded into the first GEMM of layer i+1), both LayerNorms of a layer
share one row phase, the attention head-dropout scale is an epilogue of the V projection. One launch replaces the 71
kernels of engine.unet_forward.
"""
import ctypes

import numpy as np
import torch

from . import lib, ops

import os

GEMM, ROWNORM, SINUSOID, GATHER, PSAMPLE, LNBWD, LNCOL, LOSS, FIN, EWADD, EMBBWD, SQNORM, ADAM = range(13)
PG_TA, PG_TW, PG_COLSUM, PG_ACTBWD, PG_DECOUPLED, PG_T64, PG_DMA, PG_WCONST, PG_WPACK = 1, 2, 4, 8, 16, 32, 64, 128, 256
# XCD-local execution of the row-local phases (program.hip): on by default, FL_PROG_LOCAL=0 runs every phase device-wide
LOCAL_DEFAULT = os.environ.get("FL_PROG_LOCAL", "1") != "0"
ROW_KINDS = (GEMM, ROWNORM, SINUSOID, GATHER, PSAMPLE, LNBWD, LOSS, EWADD)  # row tile rt of the output needs row tile rt of the inputs
MAX_IO = 16

JOB_DTYPE = np.dtype([
    ("A", "<u8", 3), ("W", "<u8", 3), ("bias", "<u8", 3), ("R", "<u8"), ("C", "<u8"), ("mask", "<u8"), ("gate", "<u8"),
    ("aux", "<u8", 4), ("kind", "<i4"), ("M", "<i4"), ("N", "<i4"), ("nseg", "<i4"), ("K", "<i4", 3), ("lda", "<i4", 3),
    ("ldw", "<i4", 3), ("ldr", "<i4"), ("ldc", "<i4"), ("act", "<i4"), ("mask_group", "<i4"), ("tile_begin", "<i4"),
    ("ntiles", "<i4"), ("bias_scale", "<f4", 3), ("eps", "<f4"), ("eps2", "<f4"), ("bias_row_stride", "<i4"), ("flags", "<i4"),
    ("alpha", "<f4"), ("ld2", "<i4"), ("nrt", "<i4"), ("per", "<i4")])
POST_DTYPE = np.dtype([("ptr64", "<u8", 2), ("delta64", "<i8", 2), ("ptr32", "<u8"), ("delta32", "<i4"), ("pad_", "<i4")])
PHASE_DTYPE = np.dtype([("job_begin", "<i4"), ("job_end", "<i4"), ("ntiles", "<i4"), ("flags", "<i4")])
PHASE_ACQUIRE, PHASE_LOCAL, PHASE_RINGSAFE = 1, 2, 4
RINGSAFE_KINDS = (ROWNORM, SINUSOID, GATHER, PSAMPLE, LNBWD, LOSS, EWADD)  # tiles that leave the GEMM tiles' LDS rings alone
SYNC_ABORT_WORD = 8 * 1024 + 16


class IO:
    """Names slot `index` of the io array given to Program.run (a tensor that changes from call to call)."""

    def __init__(self, index):
        assert 0 <= index < MAX_IO
        self.index = index


def _ptr(x, offset_elems=0):
    if x is None:
        return 0
    if isinstance(x, IO):
        assert offset_elems == 0
        return x.index + 1
    if isinstance(x, tuple):  # (tensor, element offset)
        return _ptr(x[0], x[1])
    return x.data_ptr() + 4 * offset_elems


class ProgramBuilder:
    def __init__(self, frozen_weights=False):
        """frozen_weights: the weights do not change for the life of the program (inference): every eligible GEMM gets the
        tile-contiguous packed copy of its weights (PG_WPACK)."""
        self._frozen = frozen_weights
        self._phases = [[]]
        self._keep = []  # tensors whose storage the program points into
        self._wkeys = {}  # id(job array) -> buffers behind a GEMM job's W operands

    def _hold(self, *xs):
        for x in xs:
            if isinstance(x, tuple):
                x = x[0]
            if isinstance(x, torch.Tensor):
                self._keep.append(x)

    def barrier(self):
        """Everything added afterwards may read what was added before."""
        if self._phases[-1]:
            self._phases.append([])

    def tiles_in_phase(self):
        """tiles of the jobs added to the current phase so far (as the library counts them)"""
        L = lib.load()
        return sum(max(0, L.fl_program_job_tiles(ctypes.c_char_p(j.tobytes()))) for j, _, _ in self._phases[-1])

    @staticmethod
    def _key(x):
        """Identity of the buffer behind a pointer argument (for the stale-cache analysis in finalize)."""
        if x is None:
            return None
        if isinstance(x, IO):
            return ("io", x.index)
        if isinstance(x, tuple):
            x = x[0]
        return ("mem", x.untyped_storage().data_ptr())

    def _add(self, job, reads, writes, weights=()):
        self._phases[-1].append((job, [self._key(r) for r in reads if r is not None],
                                 [self._key(w) for w in writes if w is not None]))
        self._wkeys[id(job)] = [self._key(w) for w in weights if w is not None]

    def gemm(self, M, N, segs, C, ldc, biases=(), act=ops.ACT_NONE, R=None, ldr=0, mask=None, mask_group=0, gate=None,
             bias_row_index=None, bias_row_stride=0, ta=False, tw=False, colsum=None, colsum_alpha=1.0, actbwd=None,
             ld_actbwd=0, pre_out=None, t64=False, wpack=None):
        """C[M,N] = epi(sum_s A_s W_s^T). segs: [(A, lda, K, W, ldw)]; biases: [(bias, scale)]; with bias_row_index (a
        device int64) biases[0] is row *index of a table with row stride bias_row_stride.
        ta / tw: operand stored with the contraction index as its row (A[k][m], W[k][n]); colsum (needs ta): vector that
        receives colsum_alpha * (1 + sigmoid(gate)) * sum_k A[k][m]; actbwd: tensor P, the result is multiplied by
        act'(P[row][col]) instead of being activated; pre_out: also store the pre-activation value there; t64 (ta and tw,
        M, N % 64 == 0, no epilogue): 64x64 tiles from one LDS stage (weight gradients). wpack: the weights are FROZEN
        for the life of the program: every W_s is copied into the tile-contiguous image the LDS-DMA tile streams fastest
        (N % 32 == 0, K_s % 128 == 0, not ta / tw)."""
        wpack = self._frozen if wpack is None else wpack
        if wpack and not (ta or tw) and N % 32 == 0 and all(K % 128 == 0 for _, _, K, _, _ in segs):
            packed = []
            for (A, lda, K, W, ldw) in segs:
                Wt, off = (W if isinstance(W, tuple) else (W, 0))
                packed.append((A, lda, K, ops.program_pack_weight(Wt, N, K, ldw, off), K))
            segs = packed
        else:
            wpack = False
        j = np.zeros((), JOB_DTYPE)
        j["flags"] = (PG_TA if ta else 0) | (PG_TW if tw else 0) | (PG_COLSUM if colsum is not None else 0) | \
            (PG_ACTBWD if actbwd is not None else 0) | (PG_T64 if t64 else 0) | (PG_WPACK if wpack else 0)
        j["aux"][0], j["alpha"], j["aux"][1], j["ld2"], j["aux"][3] = _ptr(colsum), colsum_alpha, _ptr(actbwd), ld_actbwd, _ptr(pre_out)
        self._hold(colsum, actbwd, pre_out)
        j["kind"], j["M"], j["N"], j["nseg"] = GEMM, M, N, len(segs)
        for s, (A, lda, K, W, ldw) in enumerate(segs):
            j["A"][s], j["lda"][s], j["K"][s], j["W"][s], j["ldw"][s] = _ptr(A), lda, K, _ptr(W), ldw
            self._hold(A, W)
        for s, (b, sc) in enumerate(biases):
            j["bias"][s], j["bias_scale"][s] = _ptr(b), sc
            self._hold(b)
        j["C"], j["ldc"], j["act"] = _ptr(C), ldc, act
        j["R"], j["ldr"] = _ptr(R), ldr
        j["mask"], j["mask_group"], j["gate"] = _ptr(mask), mask_group, _ptr(gate)
        j["aux"][2], j["bias_row_stride"] = _ptr(bias_row_index), bias_row_stride
        self._hold(C, R, mask, gate, bias_row_index)
        self._add(j, [a[0] for a in segs] + [a[3] for a in segs if ta or tw] + [R, mask, bias_row_index, actbwd],
                  [C, colsum, pre_out], weights=[a[3] for a in segs])

    def lnbwd(self, M, D, x, ldx, dy, lddy, w, b, eps, dx, ld, act=ops.ACT_NONE, mask=None, R=None, ldr=0, gl=None, stats=None):
        """row LayerNorm backward: dx = LN'(dy * act'(u) * mask) + R; gl (optional) receives the gradient w.r.t. the LN
        output, stats [M,2] the recomputed (mean, rstd)."""
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"] = LNBWD, M, D
        j["A"][0], j["lda"][0], j["A"][1], j["lda"][1] = _ptr(x), ldx, _ptr(dy), lddy
        j["W"][0], j["bias"][0], j["eps"], j["act"], j["mask"] = _ptr(w), _ptr(b), eps, act, _ptr(mask)
        j["R"], j["ldr"], j["C"], j["ldc"], j["aux"][0], j["aux"][1] = _ptr(R), ldr, _ptr(dx), ld, _ptr(gl), _ptr(stats)
        self._hold(x, dy, w, b, mask, R, dx, gl, stats)
        self._add(j, [x, dy, mask, R], [dx, gl, stats])

    def lnbwd2(self, M, D, x_outer, dy, w_outer, b_outer, eps_outer, R, mid, stats_outer, x_inner, w, b, eps, dx, act=ops.ACT_NONE,
               mask=None, gl=None, stats=None):
        """backward of ROWNORM's LN pair in one row job: mid = LN_outer'(dy; x_outer) + R, dx = LN_inner'(mid * act' * mask;
        x_inner); all tensors [M, D] contiguous."""
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"] = LNBWD, M, D
        j["A"][0], j["A"][1], j["A"][2] = _ptr(x_outer), _ptr(dy), _ptr(x_inner)
        j["lda"][0] = j["lda"][1] = j["lda"][2] = j["ldr"] = j["ldc"] = D
        j["W"][0], j["bias"][0], j["eps"], j["W"][1], j["bias"][1], j["eps2"] = _ptr(w), _ptr(b), eps, _ptr(w_outer), _ptr(b_outer), eps_outer
        j["act"], j["mask"], j["R"], j["C"] = act, _ptr(mask), _ptr(R), _ptr(dx)
        j["aux"][0], j["aux"][1], j["aux"][2], j["aux"][3] = _ptr(gl), _ptr(stats), _ptr(mid), _ptr(stats_outer)
        self._hold(x_outer, dy, x_inner, w, b, w_outer, b_outer, mask, R, dx, gl, stats, mid, stats_outer)
        self._add(j, [x_outer, dy, x_inner, mask, R], [dx, gl, stats, mid, stats_outer])

    def lncol(self, M, D, gl, ldg, x, ldx, stats, dw, db):
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"] = LNCOL, M, D
        j["A"][0], j["lda"][0], j["A"][1], j["lda"][1], j["aux"][1] = _ptr(gl), ldg, _ptr(x), ldx, _ptr(stats)
        j["C"], j["aux"][0] = _ptr(dw), _ptr(db)
        self._hold(gl, x, stats, dw, db)
        self._add(j, [gl, x, stats], [dw, db])

    def loss(self, M, D, pred, target, dout, norms, alpha, r=None, rowdot=None):
        """euclidean_distance_loss rows (v2:293-302): norms[m], dout = alpha * d/norm, rowdot[m] = <dout[m], r[m]>."""
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"], j["alpha"] = LOSS, M, D, alpha
        j["A"][0], j["A"][1], j["A"][2] = _ptr(pred), _ptr(target), _ptr(r)
        j["lda"][0] = j["lda"][1] = j["lda"][2] = D
        j["C"], j["ldc"], j["aux"][0], j["aux"][1] = _ptr(dout), D, _ptr(norms), _ptr(rowdot)
        self._hold(pred, target, r, dout, norms, rowdot)
        self._add(j, [pred, target, r], [dout, norms, rowdot])

    def fin(self, M, norms, loss, rowdot=None, gate=None, ggate=None):
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"] = FIN, M, 1
        j["A"][0], j["A"][1], j["C"], j["aux"][0], j["gate"] = _ptr(norms), _ptr(rowdot), _ptr(loss), _ptr(ggate), _ptr(gate)
        self._hold(norms, rowdot, loss, gate, ggate)
        self._add(j, [norms, rowdot], [loss, ggate])

    def ewadd(self, M, N, a, b, C, c=None):
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"] = EWADD, M, N
        j["A"][0], j["A"][1], j["A"][2], j["C"], j["ldc"] = _ptr(a), _ptr(b), _ptr(c), _ptr(C), N
        j["lda"][0] = j["lda"][1] = j["lda"][2] = N
        self._hold(a, b, c, C)
        self._add(j, [a, b, c], [C])

    def embbwd(self, B, D, rows, dout, idx, dtable):
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"] = EMBBWD, B, D
        j["K"][0], j["A"][0], j["lda"][0], j["W"][0], j["C"], j["ldc"] = rows, _ptr(dout), D, _ptr(idx), _ptr(dtable), D
        self._hold(dout, idx, dtable)
        self._add(j, [dout, idx], [dtable])

    def sqnorm(self, n, g, partials):
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"] = SQNORM, 1, 1
        j["K"][0], j["A"][0], j["C"] = n, _ptr(g), _ptr(partials)
        self._hold(g, partials)
        self._add(j, [g], [partials])

    def adam(self, n, p, g, m, v, partials, lr, step, b1, b2, eps, wd, decoupled, gscale=1.0, max_norm=None, coef=None,
             gnorm=None, betas_dev=None):
        """betas_dev: device tensor (beta1, beta2) read at run time instead of b1, b2."""
        j = np.zeros((), JOB_DTYPE)
        j["bias"][0] = _ptr(betas_dev)
        self._hold(betas_dev)
        j["kind"], j["M"], j["N"] = ADAM, 1, 1
        j["K"][0], j["A"][0], j["C"], j["aux"][0], j["aux"][1] = n, _ptr(g), _ptr(p), _ptr(m), _ptr(v)
        j["W"][0], j["W"][1], j["W"][2], j["aux"][2], j["aux"][3] = _ptr(partials), _ptr(lr), _ptr(step), _ptr(coef), _ptr(gnorm)
        j["bias_scale"][0], j["bias_scale"][1], j["bias_scale"][2] = b1, b2, (max_norm if max_norm is not None else 0.0)
        j["eps"], j["eps2"], j["alpha"], j["flags"] = eps, wd, gscale, (PG_DECOUPLED if decoupled else 0)
        self._hold(p, g, m, v, partials, lr, step, coef, gnorm)
        self._add(j, [g, p, m, v, partials, lr, step], [p, m, v, coef, gnorm])

    def rownorm(self, M, D, x, ldx, w, b, eps, C, ldc, act=ops.ACT_NONE, mask=None, R=None, ldr=0, w2=None, b2=None,
                eps2=1e-5, C2=None, gate=None, gx=None, ldgx=0):
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"] = ROWNORM, M, D
        j["A"][0], j["lda"][0], j["W"][0], j["bias"][0], j["eps"] = _ptr(x), ldx, _ptr(w), _ptr(b), eps
        j["C"], j["ldc"], j["act"], j["mask"], j["R"], j["ldr"] = _ptr(C), ldc, act, _ptr(mask), _ptr(R), ldr
        j["W"][1], j["bias"][1], j["eps2"], j["aux"][0] = _ptr(w2), _ptr(b2), eps2, _ptr(C2)
        j["gate"], j["aux"][1], j["lda"][1] = _ptr(gate), _ptr(gx), ldgx
        self._hold(x, w, b, C, mask, R, w2, b2, C2, gate, gx)
        self._add(j, [x, mask, R, gx], [C, C2])

    def sinusoid(self, M, N, t, t_stride, freqs, C, ldc):
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"] = SINUSOID, M, N
        j["A"][0], j["lda"][0], j["W"][0], j["C"], j["ldc"] = _ptr(t), t_stride, _ptr(freqs), _ptr(C), ldc
        self._hold(t, freqs, C)
        self._add(j, [t], [C])

    def gather(self, M, N, table, ld_table, idx, C, ldc, rows=0):
        """rows > 0: the table's row count - an index outside [0, rows) voids the run instead of reading outside the table"""
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"] = GATHER, M, N
        j["K"][0] = rows
        j["A"][0], j["lda"][0], j["W"][0], j["C"], j["ldc"] = _ptr(table), ld_table, _ptr(idx), _ptr(C), ldc
        self._hold(table, idx, C)
        self._add(j, [idx], [C])

    def psample(self, M, N, x, eps, t64, alpha, alpha_bar, beta, noise=None, philox_offset=None, seed=0):
        """x <- DDPM reverse update (in place); noise tensor or inline Philox (seed, device offset)."""
        j = np.zeros((), JOB_DTYPE)
        j["kind"], j["M"], j["N"] = PSAMPLE, M, N
        j["A"][0], j["A"][1], j["A"][2], j["C"], j["ldc"] = _ptr(x), _ptr(eps), _ptr(noise), _ptr(x), N
        j["W"][0], j["W"][1], j["W"][2] = _ptr(alpha), _ptr(alpha_bar), _ptr(beta)
        j["aux"][0], j["aux"][2] = _ptr(philox_offset), _ptr(t64)
        j["K"][0], j["K"][1] = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32).view(np.int32)
        self._hold(x, eps, t64, alpha, alpha_bar, beta, noise, philox_offset)
        self._add(j, [x, eps, t64, noise, philox_offset], [x])

    def finalize(self, device):
        L = lib.load()
        assert JOB_DTYPE.itemsize == L.fl_program_job_bytes() and PHASE_DTYPE.itemsize == L.fl_program_phase_bytes(), \
            "program struct packing differs from the library's"
        phases = [p for p in self._phases if p]
        jobs = []
        ph = np.zeros(len(phases), PHASE_DTYPE)
        max_tiles = 0
        written = set(k for p in phases for _, _, w in p for k in w)
        # XCD-local execution of a phase (its row tiles never leave their XCD) is legal when all of its jobs are row-local
        # and work on the same batch rows as every other row-local job of the program
        row_ms = set(int(j["M"]) for p in phases for j, _, _ in p
                     if int(j["kind"]) in ROW_KINDS and not (int(j["kind"]) == GEMM and int(j["flags"]) & PG_TA))
        local_ok = len(row_ms) == 1
        m_ref = next(iter(row_ms)) if local_ok else -1
        # A run starts with clean caches and the barrier only writes dirty lines back. A phase needs the invalidating
        # acquire iff it reads a buffer that was overwritten after some block may have cached an older version of it.
        cached, stale = set(), set()
        for pi, p in enumerate(phases):
            reads = set(k for _, r, _ in p for k in r)
            if reads & stale:
                ph[pi]["flags"] |= PHASE_ACQUIRE
                cached, stale = set(), set()
            for _, _, w in p:
                stale |= set(w) & cached
            cached |= reads
            ph[pi]["job_begin"] = len(jobs)
            if len(p) > 64:
                raise ValueError("a phase holds %d jobs: the executor deals a phase from a 64-lane window of its jobs" % len(p))
            t0 = 0
            all_rows = local_ok
            for j, _, _ in p:
                kind, flags = int(j["kind"]), int(j["flags"])
                if kind == GEMM and not (set(self._wkeys.get(id(j), ())) & written):
                    j["flags"] = flags | PG_WCONST   # its W operands are weights: never written by the program
                buf = ctypes.create_string_buffer(j.tobytes(), JOB_DTYPE.itemsize)
                n = L.fl_program_job_layout(buf)
                if n < 0:
                    raise ValueError("program job rejected: %s" % lib.last_error())
                laid = np.frombuffer(buf.raw, dtype=JOB_DTYPE)[0]
                j["nrt"], j["per"], j["flags"] = laid["nrt"], laid["per"], laid["flags"]
                j["tile_begin"], j["ntiles"] = t0, n
                t0 += n
                jobs.append(j)
                all_rows = all_rows and kind in ROW_KINDS and not (kind == GEMM and flags & PG_TA) and int(j["M"]) == m_ref
            if all_rows:
                ph[pi]["flags"] |= PHASE_LOCAL
            if all(int(j["kind"]) in RINGSAFE_KINDS for j, _, _ in p):
                ph[pi]["flags"] |= PHASE_RINGSAFE
            ph[pi]["job_end"], ph[pi]["ntiles"] = len(jobs), t0
            max_tiles = max(max_tiles, t0)
        jobs_np = np.stack(jobs)
        jobs_t = torch.from_numpy(np.frombuffer(jobs_np.tobytes(), dtype=np.uint8).copy()).to(device)
        ph_t = torch.from_numpy(np.frombuffer(ph.tobytes(), dtype=np.uint8).copy()).to(device)
        prog = Program(ph_t, jobs_t, len(phases), len(jobs), max_tiles, self._keep, device)
        prog.local_ok = local_ok
        prog.local = local_ok and LOCAL_DEFAULT
        prog.n_local_phases = int(((ph["flags"] & PHASE_LOCAL) != 0).sum())
        return prog


class Program:
    def __init__(self, phases, jobs, n_phases, n_jobs, max_tiles, keep, device):
        self.phases, self.jobs, self.n_phases, self.n_jobs, self.max_tiles = phases, jobs, n_phases, n_jobs, max_tiles
        self._keep = keep
        self.sync = torch.zeros(lib.load().fl_program_sync_bytes() // 4, dtype=torch.int32, device=device)
        self._abort_host = torch.zeros(1, dtype=torch.int32).pin_memory() if str(device).startswith("cuda") else None
        self.max_blocks = 0  # 0 = one block per CU
        self.local_ok = False  # the row-local jobs all work on the same batch rows: phases flagged PHASE_LOCAL may run XCD-local
        self.local = False     # ask for FL_PROG_XCD_LOCAL (set by the builder when local_ok; FL_PROG_LOCAL=0 turns it off)
        self.n_local_phases = 0
        self.guard = None      # another Program whose outputs this one consumes: while THAT one's run is void, this one does nothing

    def run(self, io, post=None):
        """io: list of tensors bound to the IO slots. Enqueues one kernel on the current stream (graph-capturable).
        post: [(int64 tensor, delta), ...] (at most 2) and/or an (int32 tensor, delta): counters bumped once per run,
        after the last phase."""
        arr = (ctypes.c_void_p * MAX_IO)(*[t.data_ptr() for t in io])
        pbuf = None
        if post:
            pp = np.zeros((), POST_DTYPE)
            k = 0
            for t, d in post:
                if t.dtype == torch.int64:
                    pp["ptr64"][k], pp["delta64"][k] = t.data_ptr(), d
                    k += 1
                else:
                    assert t.dtype == torch.int32
                    pp["ptr32"], pp["delta32"] = t.data_ptr(), d
            pbuf = ctypes.c_char_p(pp.tobytes())
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing and self._abort_host is not None and int(self._abort_host[0]) != 0:
            self.check()  # an earlier run timed out (seen through the pinned mirror, no sync): re-arm and raise
        if not capturing and self.guard is not None and self.guard._abort_host is not None and int(self.guard._abort_host[0]) != 0:
            self.guard.check()  # the producer's run was void: raise instead of consuming it
        lib.call("fl_program_run", ctypes.c_void_p(self.phases.data_ptr()), ctypes.c_void_p(self.jobs.data_ptr()),
                 self.n_phases, self.n_jobs, arr, len(io), ctypes.c_void_p(self.sync.data_ptr()), self.max_blocks,
                 1 if self.local else 0, pbuf, ctypes.c_void_p(self.guard.sync.data_ptr() if self.guard is not None else 0), ops._stream())
        if not capturing and self._abort_host is not None:  # cheap health check: async mirror of the abort word, read next call
            self._abort_host.copy_(self.sync[SYNC_ABORT_WORD:SYNC_ABORT_WORD + 1], non_blocking=True)

    def check(self):
        """Synchronising health check: raises if a barrier timed out (and re-arms the counters)."""
        word = int(self.sync[SYNC_ABORT_WORD])
        if word != 0:
            self.sync.zero_()
            if self._abort_host is not None:
                self._abort_host.zero_()
            if word == 2:
                raise RuntimeError("persistent program aborted: a gather index (timestep / class label) lies outside its table")
            if word == 3:
                raise RuntimeError("persistent program aborted: the grid's blocks were not spread evenly over the XCDs "
                                   "(XCD-local phases need that; FL_PROG_LOCAL=0 runs every phase device-wide)")
            raise RuntimeError("persistent program aborted: a device-wide barrier timed out (is another persistent "
                               "program running on this device?)")


# ------------------------------------------------------------------------------------------------ ConditionalUNet
X, T, CLS, OUT, BM0, HM0, CE = 0, 1, 2, 3, 4, 8, 12  # IO slots


class UNetProgram:
    """The ConditionalUNet forward as one persistent launch. Built for a fixed (B, shared-t?, class?, masks?, ce given?)
    signature; weights are referenced in place (rebuild if a parameter is re-allocated)."""

    def __init__(self, un, B, device, shared_t=False, has_class=True, masks=False, ce_given=False):
        self.un, self.B, self.key = un, B, self.weights_key(un)
        self.shared_t, self.has_class, self.masks, self.ce_given = shared_t, has_class, masks, ce_given
        d = lambda p: p.detach()
        SW = ops.ACT_SWISH
        f32 = dict(dtype=torch.float32, device=device)
        E = un.time_emb_dim
        Bt = 1 if shared_t else B
        te_m, ce_m = un.time_emb, un.class_emb
        H1 = te_m.lin1.weight.shape[0]
        pb = ProgramBuilder()
        feat = torch.empty(Bt, E, **f32)
        te1a = torch.empty(Bt, H1, **f32)
        te = torch.empty(Bt, E, **f32)
        ldt = 0 if shared_t else 1      # row stride multiplier of the time tensors (one shared row: stride 0)
        pb.sinusoid(Bt, E, IO(T), 1, un._freqs(device), feat, E)
        if has_class and not ce_given:
            emb = torch.empty(B, E, **f32)
            ce1a = torch.empty(B, ce_m.lin1.weight.shape[0], **f32)
            ce = torch.empty(B, E, **f32)
            pb.gather(B, E, d(ce_m.embedding.weight), E, IO(CLS), emb, E, rows=ce_m.embedding.weight.shape[0])
        pb.barrier()
        pb.gemm(Bt, H1, [(feat, E, E, d(te_m.lin1.weight), E)], te1a, H1, [(d(te_m.lin1.bias), 1.0)], act=SW)
        if has_class and not ce_given:
            C1 = ce1a.shape[1]
            pb.gemm(B, C1, [(emb, E, E, d(ce_m.lin1.weight), E)], ce1a, C1, [(d(ce_m.lin1.bias), 1.0)], act=SW)
        pb.barrier()
        pb.gemm(Bt, E, [(te1a, H1, H1, d(te_m.lin2.weight), H1)], te, E, [(d(te_m.lin2.bias), 1.0)])
        if has_class:
            tc = torch.empty(B, E, **f32)
            if ce_given:
                ce = IO(CE)
                # tc = te (one row or per row) + ce
                pb.gemm(B, E, [(te1a, H1 * ldt, H1, d(te_m.lin2.weight), H1)], tc, E, [(d(te_m.lin2.bias), 1.0)], R=ce, ldr=E)
            else:
                pb.gemm(B, E, [(ce1a, C1, C1, d(ce_m.lin2.weight), C1)], ce, E, [(d(ce_m.lin2.bias), 1.0)])
                # te + ce as one contraction over [te1a | ce1a]  (W(te)+W(ce)+2b of v2:543-544 needs only the sum)
                pb.gemm(B, E, [(te1a, H1 * ldt, H1, d(te_m.lin2.weight), H1), (ce1a, C1, C1, d(ce_m.lin2.weight), C1)], tc, E,
                        [(d(te_m.lin2.bias), 1.0), (d(ce_m.lin2.bias), 1.0)])
            tc_ld, tb = E, 2.0
        else:
            tc, tc_ld, tb = te, E * ldt, 1.0
        pb.barrier()
        prev, prev_ld, prev_w, prev_b = IO(X), un.latent_dim, d(un.latent_proj.weight), d(un.latent_proj.bias)
        Kprev = un.latent_dim
        self.ws = [feat, te1a, te, tc]
        for i, (block, norm, down) in enumerate(un.layers):
            lin, ln = block[0], block[1]
            D = lin.weight.shape[0]
            tp, att = un.time_projections[i], un.attention_layers[i]
            h1, g, h2, hn, vm, h3 = (torch.empty(B, D, **f32) for _ in range(6))
            self.ws += [h1, g, h2, hn, vm, h3]
            pb.gemm(B, D, [(prev, prev_ld, Kprev, prev_w, Kprev), (tc, tc_ld, E, d(tp.weight), E)], h1, D,
                    [(prev_b, 1.0), (d(tp.bias), tb)])
            pb.barrier()
            pb.gemm(B, D, [(h1, D, D, d(lin.weight), D)], g, D, [(d(lin.bias), 1.0)])
            pb.barrier()
            pb.rownorm(B, D, g, D, d(ln.weight), d(ln.bias), ln.eps, h2, D, act=SW, mask=IO(BM0 + i) if masks else None,
                       R=h1, ldr=D, w2=d(norm.weight), b2=d(norm.bias), eps2=norm.eps, C2=hn)
            pb.barrier()
            wv = (d(att.in_proj_weight), 2 * D * D)
            bv = (d(att.in_proj_bias), 2 * D)
            pb.gemm(B, D, [(hn, D, D, wv, D)], vm, D, [(bv, 1.0)], mask=IO(HM0 + i) if masks else None,
                    mask_group=D // att.num_heads)
            pb.barrier()
            pb.gemm(B, D, [(vm, D, D, d(att.out_proj.weight), D)], h3, D, [(d(att.out_proj.bias), 1.0)], R=h2, ldr=D)
            pb.barrier()
            prev, prev_ld, prev_w, prev_b, Kprev = h3, D, d(down.weight), d(down.bias), D
        ft, fcp, fn, fin = un.final_time_proj, un.final_class_proj, un.final_norm, un.final
        Dl = ft.weight.shape[0]
        hpre, u = torch.empty(B, Dl, **f32), torch.empty(B, Dl, **f32)
        self.ws += [hpre, u]
        segs = [(prev, prev_ld, Kprev, prev_w, Kprev), (te, E * ldt, E, d(ft.weight), E)]
        biases = [(prev_b, 1.0), (d(ft.bias), 1.0)]
        if has_class:
            segs.append((ce, E, E, d(fcp.weight), E))
            biases.append((d(fcp.bias), 1.0))
        pb.gemm(B, Dl, segs, hpre, Dl, biases)
        pb.barrier()
        rw = d(un.residual_weight)
        # u = LN(h) + sigmoid(rw) * x ; out = final(u) + sigmoid(rw)*final.bias  ==  final(LN(h)) + sigmoid(rw)*final(x)  (v2:561)
        pb.rownorm(B, Dl, hpre, Dl, d(fn.weight), d(fn.bias), fn.eps, u, Dl, gate=rw, gx=IO(X), ldgx=un.latent_dim)
        pb.barrier()
        pb.gemm(B, un.latent_dim, [(u, Dl, Dl, d(fin.weight), Dl)], IO(OUT), un.latent_dim, [(d(fin.bias), 1.0)], gate=rw)
        self.prog = pb.finalize(device)
        self._dummy = torch.zeros(4, **f32)

    @staticmethod
    def weights_key(un):
        return tuple(p.data_ptr() for p in un.parameters())

    def __call__(self, x, t, c=None, block_masks=None, head_masks=None, ce=None, out=None):
        B = self.B
        assert x.shape[0] == B and (t.numel() == (1 if self.shared_t else B))
        if out is None:
            out = torch.empty(B, self.un.latent_dim, dtype=torch.float32, device=x.device)
        ops._chk(x, t, c, out, ce)
        z = self._dummy
        io = [z] * 13
        io[X], io[T], io[OUT] = x, t, out
        if self.has_class and not self.ce_given:
            io[CLS] = c
        if self.ce_given:
            io[CE] = ce
        if self.masks:
            for i in range(4):
                ops._chk(block_masks[i], head_masks[i])
                io[BM0 + i], io[HM0 + i] = block_masks[i], head_masks[i]
        self.prog.run(io)
        return out


class FoldedUNetProgram:
    """Eval-mode ConditionalUNet forward (per-sample t, class-conditional, no dropout) with the linear chains between the
    nonlinearities folded, 12 phases instead of 26 (a phase costs ~4.4 us of barrier + hand-over whatever it computes, §3b):
      * attention with one token is out_proj(v_proj(.)) and nothing nonlinear follows before the next layer, so
        down(out_proj(v_proj(hn)) + h2) = (W_d W_o W_v) hn + W_d h2 + const is one contraction over [hn | h2]  (as in
        SamplerProgram);
      * the Linear that follows h1 = h + time_proj(tc) is linear in the same inputs, so g = W_lin h1 + b is computed NEXT TO
        h1 in the same phase with pre-multiplied weights (W_lin W_cat, W_lin W_tp) instead of one phase later;
      * layer 0 and the final projections take te1a / ce1a through pre-multiplied weights (W_tp0 W_te2, ...), so nothing on
        the chain waits for te / ce / tc;
      * te1a = swish(lin1(sinusoid(t))) and ce1a = swish(lin1(embedding(c))) depend on the timestep / the class alone:
        tables over all n_steps timesteps and all classes, ONE checked gather phase (an index outside its table voids the
        run loudly) instead of sinusoid + embedding + two GEMM phases.
    The folded weights are products of the module's weights computed once per weight version (`version_key`): this program
    is for frozen weights (inference). Matches v2:534-561 to fp32 re-association (tests/test_program_gpu.py)."""

    def __init__(self, un, B, device, n_steps=1000):
        """n_steps: timesteps the per-timestep table covers (t >= n_steps voids the run: Program.check raises)"""
        d = lambda p: p.detach()
        SW = ops.ACT_SWISH
        f32 = dict(dtype=torch.float32, device=device)
        self.un, self.B, self.key, self.n_steps = un, B, self.version_key(un), n_steps
        E, L = un.time_emb_dim, un.latent_dim
        te_m, ce_m = un.time_emb, un.class_emb
        H1, C1 = te_m.lin1.weight.shape[0], ce_m.lin1.weight.shape[0]
        new = lambda *shape: torch.empty(*shape, **f32)
        mm = lambda A, Bm: ops.gemm(A.contiguous(), Bm.contiguous(), A.shape[0], Bm.shape[1], A.shape[1], b_layout=ops.KS,
                                    ldb=Bm.shape[1])                                    # A @ Bm
        mv = lambda W, v, b=None: ops.linear(v.reshape(1, -1).contiguous(), W, b).reshape(-1)   # W v (+ b)
        keep = self.keep = []
        pb = ProgramBuilder(frozen_weights=True)
        with torch.no_grad():
            Wte2, bte2, Wce2, bce2 = d(te_m.lin2.weight), d(te_m.lin2.bias), d(ce_m.lin2.weight), d(ce_m.lin2.bias)
            bsum = (bte2 + bce2).contiguous()
            te1a, ce1a, tc = new(B, H1), new(B, C1), new(B, E)
            # x-independent and (frozen weights) a function of the timestep / the class alone: tables over every timestep
            # and every class, one gather phase instead of sinusoid + embedding + two GEMM phases
            steps_all = torch.arange(n_steps, dtype=torch.int64, device=device)
            te_tab = ops.linear(ops.time_sinusoid(steps_all, un._freqs(device)), d(te_m.lin1.weight), d(te_m.lin1.bias), act=SW)
            ce_tab = ops.linear(d(ce_m.embedding.weight).contiguous(), d(ce_m.lin1.weight), d(ce_m.lin1.bias), act=SW)
            n_cls = ce_tab.shape[0]
            keep += [te_tab, ce_tab]
            pb.gather(B, H1, te_tab, H1, IO(T), te1a, H1, rows=n_steps)
            pb.gather(B, C1, ce_tab, C1, IO(CLS), ce1a, C1, rows=n_cls)
            pb.barrier()
            pb.gemm(B, E, [(te1a, H1, H1, Wte2, H1), (ce1a, C1, C1, Wce2, C1)], tc, E, [(bte2, 1.0), (bce2, 1.0)])
            # final_time_proj(te) + final_class_proj(ce) (+ the folded constants of the last layer, added below)
            ft, fcp, fn, fin = un.final_time_proj, un.final_class_proj, un.final_norm, un.final
            Dl = ft.weight.shape[0]
            extra = new(B, Dl)
            Wft2, Wfc2 = mm(d(ft.weight), Wte2), mm(d(fcp.weight), Wce2)
            extra_const = mv(d(ft.weight), bte2, d(ft.bias)) + mv(d(fcp.weight), bce2, d(fcp.bias))
            prev = None            # (hn, h2, D, Wcat, const): the folded input of the next layer
            for i, (block, norm, down) in enumerate(un.layers):
                lin, ln = block[0], block[1]
                D = lin.weight.shape[0]
                tp, att = un.time_projections[i], un.attention_layers[i]
                Wlin, blin, Wtp, btp = d(lin.weight), d(lin.bias), d(tp.weight), d(tp.bias)
                h1, g, h2, hn = (new(B, D) for _ in range(4))
                if prev is None:
                    Wlp, blp = d(un.latent_proj.weight), d(un.latent_proj.bias)
                    T0, C0 = mm(Wtp, Wte2), mm(Wtp, Wce2)
                    c_h1 = (blp + 2.0 * btp + mv(Wtp, bsum)).contiguous()
                    segs_h1 = [(IO(X), L, L, Wlp, L), (te1a, H1, H1, T0, H1), (ce1a, C1, C1, C0, C1)]
                    Wg0, Tg, Cg = mm(Wlin, Wlp), mm(Wlin, T0), mm(Wlin, C0)
                    segs_g = [(IO(X), L, L, Wg0, L), (te1a, H1, H1, Tg, H1), (ce1a, C1, C1, Cg, C1)]
                    keep += [T0, C0, Wg0, Tg, Cg]
                else:
                    phn, ph2, Dp, Wcat, cprev = prev
                    c_h1 = (cprev + 2.0 * btp).contiguous()
                    segs_h1 = [(phn, Dp, Dp, Wcat, 2 * Dp), (ph2, Dp, Dp, (Wcat, Dp), 2 * Dp), (tc, E, E, Wtp, E)]
                    WG, WGt = mm(Wlin, Wcat), mm(Wlin, Wtp)
                    segs_g = [(phn, Dp, Dp, WG, 2 * Dp), (ph2, Dp, Dp, (WG, Dp), 2 * Dp), (tc, E, E, WGt, E)]
                    keep += [WG, WGt]
                c_g = mv(Wlin, c_h1, blin).contiguous()
                keep += [c_h1, c_g, h1, g, h2, hn]
                pb.gemm(B, D, segs_h1, h1, D, [(c_h1, 1.0)])
                pb.gemm(B, D, segs_g, g, D, [(c_g, 1.0)])
                if prev is None:
                    pb.gemm(B, Dl, [(te1a, H1, H1, Wft2, H1), (ce1a, C1, C1, Wfc2, C1)], extra, Dl)   # off the chain
                pb.barrier()
                pb.rownorm(B, D, g, D, d(ln.weight), d(ln.bias), ln.eps, h2, D, act=SW, R=h1, ldr=D, w2=d(norm.weight),
                           b2=d(norm.bias), eps2=norm.eps, C2=hn)
                pb.barrier()
                Wv, bv = d(att.in_proj_weight)[2 * D:], d(att.in_proj_bias)[2 * D:]
                Wo, bo, Wd, bd = d(att.out_proj.weight), d(att.out_proj.bias), d(down.weight), d(down.bias)
                Dn = Wd.shape[0]
                Wcat = new(Dn, 2 * D)
                Wcat[:, :D] = mm(Wd, mm(Wo, Wv))
                Wcat[:, D:] = Wd
                cnext = mv(Wd, mv(Wo, bv, bo), bd)
                keep.append(Wcat)
                prev = (hn, h2, D, Wcat, cnext)
            phn, ph2, Dp, Wcat, cprev = prev
            c_f = (cprev + extra_const).contiguous()
            hpre, u = new(B, Dl), new(B, Dl)
            pb.gemm(B, Dl, [(phn, Dp, Dp, Wcat, 2 * Dp), (ph2, Dp, Dp, (Wcat, Dp), 2 * Dp)], hpre, Dl, [(c_f, 1.0)], R=extra, ldr=Dl)
            pb.barrier()
            rw = d(un.residual_weight)
            pb.rownorm(B, Dl, hpre, Dl, d(fn.weight), d(fn.bias), fn.eps, u, Dl, gate=rw, gx=IO(X), ldgx=L)
            pb.barrier()
            pb.gemm(B, L, [(u, Dl, Dl, d(fin.weight), Dl)], IO(OUT), L, [(d(fin.bias), 1.0)], gate=rw)
            keep += [Wft2, Wfc2, c_f, extra, hpre, u, bsum, te1a, ce1a, tc]
            self.prog = pb.finalize(device)
        self._dummy = torch.zeros(4, **f32)

    @staticmethod
    def version_key(un):
        """changes whenever a weight does: torch's in-place version counters + the raw-pointer update epoch that
        steps.FlatParams.bump() advances + the storage addresses"""
        return (sum(p._version for p in un.parameters()), getattr(un, "_raw_epoch", 0)) + tuple(p.data_ptr() for p in un.parameters())

    def __call__(self, x, t, c, out=None):
        B = self.B
        assert x.shape[0] == B and t.numel() == B and c.numel() == B
        if out is None:
            out = torch.empty(B, self.un.latent_dim, dtype=torch.float32, device=x.device)
        ops._chk(x, t, c, out)
        io = [self._dummy] * 4
        io[X], io[T], io[CLS], io[OUT] = x, t, c, out
        self.prog.run(io)
        return out


# ------------------------------------------------------------------------------------------------ sampler step
class SamplerProgram:
    """One DDPM reverse step (v2:580-592) as ONE persistent launch of 12 phases: denoiser + update + Philox noise +
    step counter. Everything that does not depend on x is hoisted out of the 1000-step loop ("class path
    hoistable"), and with frozen eval-mode weights the linear chains are folded:
      * per-timestep tables (n_steps rows): TE = time_emb(t); TPt_i = W_tp_i TE; FT = W_ft TE  -> one table row per
        step is added as a bias picked by the device-resident step counter;
      * per-sample constants: ce = class_emb(c); TPc_i = W_tp_i ce + biases; FC = W_fc ce + biases -> residual inputs;
      * attention with one token is out_proj(v_proj(.)) and no nonlinearity follows before the next
        layer's Linear, so  down(out_proj(v_proj(hn)) + h2) = (W_d W_o W_v) hn + W_d h2 + const  is ONE contraction
        over [hn | h2] with pre-multiplied weights.
      * the block's Linear is linear in h1's own inputs, so g = W_lin h1 + b is computed next to h1 in the same phase with
        pre-multiplied weights and tables (as FoldedUNetProgram).
    Per layer: {h1, g} GEMMs, (LN, Swish, +h1, LN) row phase = 2 phases instead of 5."""

    def __init__(self, diffusion, B, device, x, t64, t32, c=None, noise=None, fold_lin=None):
        """fold_lin: compute g = W_lin h1 + b next to h1 with pre-multiplied weights (one phase fewer per layer, ~1.5x the
        contraction work of those phases). None = FL_SAMPLER_FOLD_LIN, default on."""
        if fold_lin is None:
            fold_lin = os.environ.get("FL_SAMPLER_FOLD_LIN", "1") != "0"
        d_ = lambda p: p.detach()
        un = diffusion.eps_model
        self.d, self.B, self.x, self.t64, self.t32 = diffusion, B, x, t64, t32
        f32 = dict(dtype=torch.float32, device=device)
        E, T = un.time_emb_dim, diffusion.n_steps
        SW = ops.ACT_SWISH
        with torch.no_grad():
            TE = un.time_emb(torch.arange(T, dtype=torch.int64, device=device))          # (T, E)
            ce = un.class_emb(c) if c is not None else None                              # (B, E)
            tb = 2.0 if c is not None else 1.0
            pb = ProgramBuilder(frozen_weights=True)
            self.keep = [TE, ce]
            mm = lambda A, Bm: ops.gemm(A.contiguous(), Bm.contiguous(), A.shape[0], Bm.shape[1], A.shape[1], b_layout=ops.KS,
                                        ldb=Bm.shape[1])                                    # A @ Bm
            prev_segs = [(x, un.latent_dim, un.latent_dim, d_(un.latent_proj.weight), un.latent_dim)]
            const = d_(un.latent_proj.bias).clone()
            for i, (block, norm, down) in enumerate(un.layers):
                lin, ln = block[0], block[1]
                D = lin.weight.shape[0]
                tp, att = un.time_projections[i], un.attention_layers[i]
                TPt = ops.linear(TE, d_(tp.weight))                                        # (T, D), no bias
                const = const + tb * d_(tp.bias)
                if ce is not None:
                    extra = ops.linear(ce, d_(tp.weight), const.contiguous())             # (B, D): W_tp ce + consts
                h1, g, h2, hn = (torch.empty(B, D, **f32) for _ in range(4))
                self.keep += [TPt, h1, g, h2, hn]
                # g = lin(h1) is linear in h1's own inputs: computed NEXT TO h1 with pre-multiplied weights / tables
                Wlin, blin = d_(lin.weight), d_(lin.bias)
                TGt = ops.linear(TPt, Wlin)                                                # (T, D): W_lin W_tp TE
                if len(prev_segs) == 1:
                    A0, lda0, K0, W0, ldw0 = prev_segs[0]
                    Wg = mm(Wlin, W0)
                    segs_g = [(A0, lda0, K0, Wg, K0)]
                else:
                    (A0, lda0, K0, Wc, ldw0), (A1, lda1, K1, _, _) = prev_segs
                    Wg = mm(Wlin, Wc)
                    segs_g = [(A0, lda0, K0, Wg, ldw0), (A1, lda1, K1, (Wg, K0), ldw0)]
                self.keep += [TGt, Wg]
                if ce is not None:
                    self.keep.append(extra)
                    extra_g = ops.linear(extra, Wlin, blin)
                    self.keep.append(extra_g)
                    pb.gemm(B, D, prev_segs, h1, D, [(TPt, 1.0)], R=extra, ldr=D, bias_row_index=t64, bias_row_stride=D)
                    if fold_lin:
                        pb.gemm(B, D, segs_g, g, D, [(TGt, 1.0)], R=extra_g, ldr=D, bias_row_index=t64, bias_row_stride=D)
                else:
                    cst = const.contiguous()
                    cst_g = ops.linear(cst.reshape(1, -1), Wlin, blin).reshape(-1)
                    self.keep += [cst, cst_g]
                    pb.gemm(B, D, prev_segs, h1, D, [(TPt, 1.0), (cst, 1.0)], bias_row_index=t64, bias_row_stride=D)
                    if fold_lin:
                        pb.gemm(B, D, segs_g, g, D, [(TGt, 1.0), (cst_g, 1.0)], bias_row_index=t64, bias_row_stride=D)
                pb.barrier()
                if not fold_lin:
                    pb.gemm(B, D, [(h1, D, D, Wlin, D)], g, D, [(blin, 1.0)])
                    pb.barrier()
                pb.rownorm(B, D, g, D, d_(ln.weight), d_(ln.bias), ln.eps, h2, D, act=SW, R=h1, ldr=D, w2=d_(norm.weight),
                           b2=d_(norm.bias), eps2=norm.eps, C2=hn)
                pb.barrier()
                # fold: next input = W_d (W_o (W_v hn + b_v) + b_o + h2) + b_d
                Wv, bv = d_(att.in_proj_weight)[2 * D:], d_(att.in_proj_bias)[2 * D:]
                Wo, bo = d_(att.out_proj.weight), d_(att.out_proj.bias)
                Wd, bd = d_(down.weight), d_(down.bias)
                Dn = Wd.shape[0]
                Wov = ops.gemm(Wo, Wv, D, D, D, b_layout=ops.KS, ldb=D)                    # W_o W_v
                bov = ops.linear(bv.reshape(1, D).contiguous(), Wo, bo)                    # W_o b_v + b_o  (1, D)
                Wcat = torch.empty(Dn, 2 * D, **f32)
                Wcat[:, :D] = ops.gemm(Wd, Wov, Dn, D, D, b_layout=ops.KS, ldb=D)          # W_d W_o W_v
                Wcat[:, D:] = Wd
                const = ops.linear(bov, Wd, bd).reshape(Dn)                                # W_d b_ov + b_d
                self.keep.append(Wcat)
                prev_segs = [(hn, D, D, Wcat, 2 * D), (h2, D, D, (Wcat, D), 2 * D)]
            ft, fcp, fn, fin = un.final_time_proj, un.final_class_proj, un.final_norm, un.final
            Dl = ft.weight.shape[0]
            FT = ops.linear(TE, d_(ft.weight))                                            # (T, Dl)
            const = const + d_(ft.bias)
            hpre, u, eps = torch.empty(B, Dl, **f32), torch.empty(B, Dl, **f32), torch.empty(B, un.latent_dim, **f32)
            self.keep += [FT, hpre, u, eps]
            if ce is not None:
                extra = ops.linear(ce, d_(fcp.weight), (const + d_(fcp.bias)).contiguous())
                self.keep.append(extra)
                pb.gemm(B, Dl, prev_segs, hpre, Dl, [(FT, 1.0)], R=extra, ldr=Dl, bias_row_index=t64, bias_row_stride=Dl)
            else:
                cst = const.contiguous()
                self.keep.append(cst)
                pb.gemm(B, Dl, prev_segs, hpre, Dl, [(FT, 1.0), (cst, 1.0)], bias_row_index=t64, bias_row_stride=Dl)
            pb.barrier()
            rw = d_(un.residual_weight)
            pb.rownorm(B, Dl, hpre, Dl, d_(fn.weight), d_(fn.bias), fn.eps, u, Dl, gate=rw, gx=x, ldgx=un.latent_dim)
            pb.barrier()
            pb.gemm(B, un.latent_dim, [(u, Dl, Dl, d_(fin.weight), Dl)], eps, un.latent_dim, [(d_(fin.bias), 1.0)], gate=rw)
            pb.barrier()
            st = diffusion.rng.state(device)
            self.n4 = (B * un.latent_dim + 3) // 4
            pb.psample(B, un.latent_dim, x, eps, t64, diffusion.alpha, diffusion.alpha_bar, diffusion.beta, noise=noise,
                       philox_offset=None if noise is not None else st, seed=diffusion.rng.seed)
            self.post = [(t64, -1), (t32, -1)] + ([] if noise is not None else [(st, self.n4)])
            self.eps = eps
            self.prog = pb.finalize(device)

    def step(self):
        self.prog.run([], post=self.post)
