"""The hand-counted asynchronous loads of the HIP kernels are covered by a wait on every path (tools/check_async_loads.py walks the
gfx950 disassembly with the vmcnt model). The round-3 two-stream fault was one uncovered path: the drain behind the conv kernels'
main loop did not name the staging registers, hipcc scheduled epilogue index arithmetic into one of them above the wait, and a halo
load that landed late overwrote it."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oxford_102_flower_gan_vae_latent_diffusion_amd", "libflowers_hip.so")


def _tool():
    spec = importlib.util.spec_from_file_location("check_async_loads", os.path.join(ROOT, "tools", "check_async_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_lint_model_flags_an_uncovered_load_and_accepts_a_covered_one():
    t = _tool()
    covered = [(None, "global_load_dwordx4", "v[4:7], v[2:3], off"), (None, "global_load_lds_dwordx4", "v[8:9], off"),
               (None, "v_add_u32_e32", "v1, v2, v3"), (None, "s_waitcnt", "vmcnt(1)"), (None, "v_mov_b32_e32", "v10, v5"),
               (None, "s_waitcnt", "vmcnt(0)"), (None, "s_endpgm", "")]
    assert t.check("k", covered) == []
    late = list(covered)
    late[3] = (None, "s_waitcnt", "vmcnt(2)")
    assert t.check("k", late) == [(4, 0)]
    # through a branch: the load is still in flight on the taken path only
    br = [(None, "global_load_dword", "v4, v[2:3], off"), (None, "s_cbranch_scc1", "L1"), (None, "s_waitcnt", "vmcnt(0)"),
          ("L1", "v_and_b32_e32", "v4, 15, v0"), (None, "s_endpgm", "")]
    assert t.check("k", br) == [(3, 0)]
    # a later load into the same registers retires behind the earlier one
    waw = [(None, "global_load_dword", "v4, v[2:3], off"), (None, "global_load_dwordx2", "v[4:5], v[2:3], off"),
           (None, "s_waitcnt", "vmcnt(0)"), (None, "s_endpgm", "")]
    assert t.check("k", waw) == []
    # a wave that ends with an LDS-DMA in flight
    dma = [(None, "global_load_lds_dwordx4", "v[8:9], off"), (None, "s_waitcnt", "vmcnt(1)"), (None, "s_endpgm", "")]
    assert t.check("k", dma) == [(2, 0)]
    dma[1] = (None, "s_waitcnt", "vmcnt(0)")
    assert t.check("k", dma) == []


@pytest.mark.skipif(not os.path.exists(os.path.join("/opt/rocm/lib/llvm/bin", "llvm-objdump")), reason="no llvm-objdump")
def test_every_in_flight_load_of_the_built_library_is_covered():
    assert os.path.exists(LIB), "libflowers_hip.so is not built (python __graft_entry__.py)"
    t = _tool()
    import tempfile
    reports, total = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for obj in t.code_objects(LIB, tmp):
            for name, ins in t.kernels(obj).items():
                total += 1
                reports += [(name, pc, ins[pc]) for pc, _ in t.check(name, ins)]
    assert total > 200
    assert not reports, reports[:5]
