"""CPU-side lint of the SHIPPED library's gfx950 instruction stream (the Makefile lost
`-mllvm -amdgpu-atomic-optimizer-strategy=None` for gemm.o and nothing noticed).  What DESIGN.md §3.1b/c states about the
persistent GEMMs is asserted here on the disassembly of enhancing-transformers_amd/lib/libenh_hip.so itself."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(isa_lint.DEFAULT_SO) or not os.path.exists(isa_lint.LLVM + "/llvm-objdump"),
                                reason="needs the built library and the ROCm llvm tools")


@pytest.fixture(scope="module")
def stats():
    return isa_lint.kernel_stats()


def test_tile_claim_atomic_is_one_in_flight_instruction(stats):
    """every dynamic-schedule instantiation keeps exactly the vmcnt(0) waits of its static twin plus the one behind the
    workgroup's first claim, and the claim is NOT rewritten into a wave reduction (v_mbcnt / s_bcnt1 + a vmcnt(0) drain per tile)."""
    pairs = isa_lint.persistent_gemm_pairs(stats)
    assert len(pairs) >= 16, pairs
    for dyn, sta in pairs:
        d, s = stats[dyn], stats[sta]
        assert d.get("mbcnt", 0) == 0 and d.get("bcnt1", 0) == 0, (dyn, d)
        assert d.get("atomics", 0) == 2, (dyn, d)                      # first claim + in-loop claim
        assert s.get("atomics", 0) == 0, (sta, s)
        assert d.get("vmcnt0", 0) == s.get("vmcnt0", 0) + 1, (dyn, d.get("vmcnt0"), s.get("vmcnt0"))


def test_one_wave_per_simd_kernels_have_no_scratch(stats):
    """a scratch reload is a vector-memory load, i.e. a vmcnt wait on the operand requests in flight"""
    hot = [n for n in stats if re.search(r"gemm_bf16_w256[pr]_kernel|gemm_bf16_w256_kernel<(true|false), (true|false), [01456]>|"
                                         r"conv_igemm_w(256|512)_kernel|conv_wgrad_w256_kernel|attn_(fwd|bwd)", n)]
    assert len(hot) >= 50, len(hot)
    for n in hot:
        s = stats[n]
        assert s.get("scratch_bytes", 0) == 0 and s.get("spills", 0) == 0 and s.get("scratch_ops", 0) == 0, (n, s)
    for n in stats:
        if "gemm_bf16_w256" in n and "lab" not in n:
            assert stats[n].get("agpr", 0) == 256 and stats[n].get("vgpr", 0) <= 512, (n, stats[n])


def test_every_gemm_kernel_is_an_mfma_kernel(stats):
    for n in stats:
        if n.startswith("void gemm_bf16_w256"):
            assert stats[n].get("mfma", 0) >= 64, (n, stats[n].get("mfma"))
