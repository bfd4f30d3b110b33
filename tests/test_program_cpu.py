"""Host side of the persistent phase-program executor (no GPU): struct packing against the library, job shape rules,
tile counts, the stale-cache (acquire) analysis of the builder, and the io-slot encoding."""
import numpy as np
import pytest
import torch

from oxford_102_flower_gan_vae_latent_diffusion_amd import lib, program as P


def phases_of(prog):
    return np.frombuffer(prog.phases.numpy().tobytes(), dtype=P.PHASE_DTYPE)


def jobs_of(prog):
    return np.frombuffer(prog.jobs.numpy().tobytes(), dtype=P.JOB_DTYPE)


def test_struct_packing_matches_library():
    L = lib.load()
    assert P.JOB_DTYPE.itemsize == L.fl_program_job_bytes() == 256
    assert P.PHASE_DTYPE.itemsize == L.fl_program_phase_bytes() == 16
    assert L.fl_program_sync_bytes() >= 4 * (P.SYNC_ABORT_WORD + 1)
    assert P.POST_DTYPE.itemsize == 48


def test_tile_counts_and_phase_layout():
    pb = P.ProgramBuilder()
    a, w, c = torch.zeros(100, 64), torch.zeros(96, 64), torch.zeros(100, 96)
    x, g, b = torch.zeros(70, 512), torch.zeros(512), torch.zeros(512)
    pb.gemm(100, 96, [(a, 64, 64, w, 64)], c, 96)                      # ceil(100/32) * 96/32 = 12 tiles
    pb.rownorm(70, 512, x, 512, g, b, 1e-5, torch.zeros(70, 512), 512)  # 3 row tiles x 32 rows = 96 tiles
    pb.barrier()
    pb.sinusoid(3, 256, torch.zeros(3, dtype=torch.int64), 1, torch.zeros(128), torch.zeros(3, 256), 256)  # 1 row tile x 16
    pb.gather(40, 256, torch.zeros(10, 256), 256, torch.zeros(40, dtype=torch.int64), torch.zeros(40, 256), 256)  # 2 row tiles x 8
    prog = pb.finalize("cpu")
    ph, jb = phases_of(prog), jobs_of(prog)
    assert [tuple(p)[:3] for p in ph] == [(0, 2, 108), (2, 4, 32)]
    assert list(jb["ntiles"]) == [12, 96, 16, 16] and list(jb["tile_begin"]) == [0, 12, 0, 16]
    assert prog.max_tiles == 108 and prog.n_jobs == 4
    assert not prog.local_ok and not prog.local  # jobs with different row counts: device-wide barriers only


@pytest.mark.parametrize("bad", ["K", "lda", "N", "rownorm_D", "rownorm_second", "psample"])
def test_shape_rules_are_enforced(bad):
    pb = P.ProgramBuilder()
    z = torch.zeros
    if bad == "K":
        pb.gemm(8, 32, [(z(8, 42), 44, 42, z(32, 44), 44)], z(8, 32), 32)
    elif bad == "lda":
        pb.gemm(8, 32, [(z(8, 42), 42, 40, z(32, 40), 40)], z(8, 32), 32)
    elif bad == "N":
        pb.gemm(8, 40, [(z(8, 64), 64, 64, z(40, 64), 64)], z(8, 40), 40)
    elif bad == "rownorm_D":
        pb.rownorm(8, 320, z(8, 320), 320, z(320), z(320), 1e-5, z(8, 320), 320)
    elif bad == "rownorm_second":
        pb.rownorm(8, 256, z(8, 256), 256, z(256), z(256), 1e-5, z(8, 256), 256, w2=z(256))
    else:
        pb.psample(3, 6, z(3, 6), z(3, 6), z(1, dtype=torch.int64), z(10), z(10), z(10), philox_offset=z(1, dtype=torch.int64))
    with pytest.raises(ValueError):
        pb.finalize("cpu")


def test_acquire_only_where_a_cached_buffer_was_overwritten():
    a, b, c, w = torch.zeros(64, 64), torch.zeros(64, 64), torch.zeros(64, 64), torch.zeros(64, 64)
    pb = P.ProgramBuilder()
    pb.gemm(64, 64, [(a, 64, 64, w, 64)], b, 64)   # reads a
    pb.barrier()
    pb.gemm(64, 64, [(b, 64, 64, w, 64)], c, 64)   # reads b (fresh), writes c
    pb.barrier()
    pb.gemm(64, 64, [(c, 64, 64, w, 64)], a, 64)   # overwrites a, which phase 0 may have cached
    pb.barrier()
    pb.gemm(64, 64, [(b, 64, 64, w, 64)], c, 64)   # b is still valid: no acquire; overwrites c (cached by phase 2)
    pb.barrier()
    pb.gemm(64, 64, [(a, 64, 64, w, 64)], b, 64)   # re-reads a after the overwrite -> acquire
    pb.barrier()
    pb.gemm(64, 64, [(c, 64, 64, w, 64)], a, 64)   # the acquire dropped every cached line: c is read fresh -> none
    flags = phases_of(pb.finalize("cpu"))["flags"]
    assert list(flags & P.PHASE_ACQUIRE) == [0, 0, 0, 0, P.PHASE_ACQUIRE, 0]
    assert all(flags & P.PHASE_LOCAL)   # six row-local GEMM phases on the same 64 rows


def test_io_slots_are_encoded_as_small_integers():
    pb = P.ProgramBuilder()
    w = torch.zeros(32, 32)
    pb.gemm(4, 32, [(P.IO(0), 32, 32, w, 32)], P.IO(3), 32, R=P.IO(15), ldr=32)
    jb = jobs_of(pb.finalize("cpu"))[0]
    assert jb["A"][0] == 1 and jb["C"] == 4 and jb["R"] == 16 and jb["W"][0] == w.data_ptr()
    with pytest.raises(AssertionError):
        P.IO(16)


def test_training_job_kinds_tile_counts():
    """Tile counts of the training kinds as the library computes them (include/flowers_hip.h) and the flag encoding."""
    z = torch.zeros
    pb = P.ProgramBuilder()
    dy, x, W = z(256, 512), z(256, 128), z(512, 128)
    pb.gemm(256, 128, [(dy, 512, 512, W, 128)], z(256, 128), 128, tw=True)                          # dgrad: 8 x 4
    pb.gemm(512, 128, [(dy, 512, 256, x, 128)], z(512, 128), 128, ta=True, tw=True, colsum=z(512))  # wgrad 32x32: 16 x 4
    pb.gemm(512, 128, [(dy, 512, 256, x, 128)], z(512, 128), 128, ta=True, tw=True, t64=True)       # wgrad 64x64: 8 x 2
    pb.barrier()
    pb.lnbwd(70, 512, z(70, 512), 512, z(70, 512), 512, z(512), z(512), 1e-5, z(70, 512), 512)      # 3 row tiles x 8
    pb.lncol(70, 512, z(70, 512), 512, z(70, 512), 512, z(70, 2), z(512), z(512))                   # 512 / 32
    pb.loss(70, 256, z(70, 256), z(70, 256), z(70, 256), z(70), 1.0 / 70)                           # 3 x 8
    pb.fin(70, z(70), z(1))                                                                          # 1
    pb.ewadd(70, 256, z(70, 256), z(70, 256), z(70, 256))                                            # 3 x 8
    pb.embbwd(70, 256, 102, z(70, 256), z(70, dtype=torch.int64), z(102, 256))                      # one per table row
    pb.barrier()
    pb.sqnorm(1000, z(1000), z(256))
    pb.barrier()
    pb.adam(1000, z(1000), z(1000), z(1000), z(1000), z(256), z(1), z(1, dtype=torch.int32), 0.9, 0.999, 1e-8, 1e-5, True,
            max_norm=1.0)
    prog = pb.finalize("cpu")
    jb = jobs_of(prog)
    assert list(jb["ntiles"]) == [32, 64, 16, 24, 16, 24, 1, 24, 102, 256, 256]
    derived = P.PG_DMA | P.PG_WCONST   # set by fl_program_job_layout / the builder's weight analysis
    assert list(jb["flags"][:3] & ~derived) == [P.PG_TW, P.PG_TA | P.PG_TW | P.PG_COLSUM, P.PG_TA | P.PG_TW | P.PG_T64]
    assert jb["flags"][0] & P.PG_DMA and not jb["flags"][1] & P.PG_DMA and not jb["flags"][2] & P.PG_DMA   # never a TA operand
    assert jb["flags"][-1] == P.PG_DECOUPLED and prog.n_phases == 4
    assert list(jb["nrt"] * jb["per"]) == list(jb["ntiles"])
    # rows 70 vs 256: the row-local jobs do not share their rows -> no local phase
    assert not prog.local_ok and not any(phases_of(prog)["flags"] & P.PHASE_LOCAL)


def test_local_phases_dma_and_constant_weights():
    """Which phases may run XCD-local, which GEMMs take the LDS-DMA tile, which may have their weights prefetched."""
    z = torch.zeros
    B = 64
    pb = P.ProgramBuilder()
    x, w1, h, w2, y = z(B, 256), z(128, 256), z(B, 128), z(256, 128), z(B, 256)
    dy, dx, gw = z(B, 256), z(B, 128), z(256, 128)
    pb.gemm(B, 128, [(x, 256, 256, w1, 256)], h, 128)                         # row-local, K % 128 == 0: DMA, constant weights
    pb.barrier()
    pb.gemm(B, 256, [(h, 128, 128, w2, 128)], y, 256)
    pb.rownorm(B, 256, x, 256, z(256), z(256), 1e-5, z(B, 256), 256)
    pb.barrier()
    pb.gemm(B, 128, [(dy, 256, 256, w2, 128)], dx, 128, tw=True)              # dgrad: row-local, DMA
    pb.gemm(256, 128, [(dy, 256, B, h, 128)], gw, 128, ta=True, tw=True)      # wgrad: contraction over the batch -> device-wide phase
    pb.barrier()
    pb.gemm(B, 96, [(x, 256, 200, z(96, 200), 200)], z(B, 96), 96)            # K % 128 != 0: register-staged tile
    pb.gemm(B, 128, [(x, 256, 256, h, 256)], z(B, 128), 128)                  # "weights" that the program itself writes (h)
    prog = pb.finalize("cpu")
    ph, jb = phases_of(prog), jobs_of(prog)
    assert prog.local_ok and prog.local == P.LOCAL_DEFAULT and prog.n_local_phases == 3
    assert [bool(f & P.PHASE_LOCAL) for f in ph["flags"]] == [True, True, False, True]
    assert [bool(f & P.PG_DMA) for f in jb["flags"]] == [True, True, False, True, False, False, True]
    assert [bool(f & P.PG_WCONST) for f in jb["flags"][[0, 1, 3, 5, 6]]] == [True, True, True, True, False]


@pytest.mark.parametrize("bad", ["ta_alone", "ta_ragged", "t64_epilogue", "t64_shape", "lnbwd_D", "lnbwd_two_stage", "adam_n"])
def test_training_shape_rules_are_enforced(bad):
    z = torch.zeros
    pb = P.ProgramBuilder()
    if bad == "ta_alone":
        pb.gemm(64, 64, [(z(10, 64), 64, 10, z(10, 64), 64)], z(64, 64), 64, ta=True)
    elif bad == "ta_ragged":
        pb.gemm(48, 64, [(z(10, 48), 48, 10, z(10, 64), 64)], z(48, 64), 64, ta=True, tw=True)
    elif bad == "t64_epilogue":
        pb.gemm(64, 64, [(z(10, 64), 64, 10, z(10, 64), 64)], z(64, 64), 64, [(z(64), 1.0)], ta=True, tw=True, t64=True)
    elif bad == "t64_shape":
        pb.gemm(96, 64, [(z(10, 96), 96, 10, z(10, 64), 64)], z(96, 64), 64, ta=True, tw=True, t64=True)
    elif bad == "lnbwd_D":
        pb.lnbwd(8, 320, z(8, 320), 320, z(8, 320), 320, z(320), z(320), 1e-5, z(8, 320), 320)
    elif bad == "lnbwd_two_stage":
        j_before = len(pb._phases[-1])
        pb.lnbwd2(8, 256, z(8, 256), z(8, 256), None, None, 1e-5, None, None, None, z(8, 256), z(256), z(256), 1e-5, z(8, 256))
        assert len(pb._phases[-1]) == j_before + 1
    else:
        pb.adam(1002, z(1002), z(1002), z(1002), z(1002), z(256), z(1), z(1, dtype=torch.int32), 0.9, 0.999, 1e-8, 0.0, True)
    with pytest.raises(ValueError):
        pb.finalize("cpu")


def test_training_program_runs_the_row_local_chain_xcd_local_and_batch_contractions_last():
    """program_train (host logic): the forward and the whole input-gradient chain are XCD-local phases; every job that
    contracts over the batch (weight gradients, LayerNorm weight/bias gradients, embedding gradient, loss mean) sits in the
    device-wide phase(s) at the very end, behind ONE local -> device-wide seam."""
    from oxford_102_flower_gan_vae_latent_diffusion_amd import modules as M, steps
    from oxford_102_flower_gan_vae_latent_diffusion_amd.program_train import DenoiserTrainProgram
    torch.manual_seed(0)
    un = M.ConditionalUNet()
    fp = steps.FlatParams(un, ("time_projections.4.", "attention_layers.4."))
    tp = DenoiserTrainProgram(un, fp.gviews, 64, "cpu")
    ph, jb = phases_of(tp.prog), jobs_of(tp.prog)
    local = [bool(f & P.PHASE_LOCAL) for f in ph["flags"]]
    n_global = local[::-1].index(True)          # trailing device-wide phases
    assert 1 <= n_global <= 2 and all(local[:len(local) - n_global]) and tp.prog.n_local_phases == len(local) - n_global
    row_kinds, batch_kinds = set(), set()
    for pi, p in enumerate(ph):
        for j in jb[p["job_begin"]:p["job_end"]]:
            key = (int(j["kind"]), bool(j["flags"] & P.PG_TA))
            (row_kinds if local[pi] else batch_kinds).add(key)
    assert batch_kinds == {(P.GEMM, True), (P.LNCOL, False), (P.EMBBWD, False), (P.FIN, False)}
    assert not (row_kinds & batch_kinds) and all(len(range(p["job_begin"], p["job_end"])) <= 64 for p in ph)
    # every input-gradient GEMM of the chain takes the LDS-DMA tile and may stream its (constant) weights ahead
    dg = [j for j in jb if j["kind"] == P.GEMM and j["flags"] & P.PG_TW and not j["flags"] & P.PG_TA]
    assert dg and all(j["flags"] & P.PG_DMA and j["flags"] & P.PG_WCONST for j in dg)
