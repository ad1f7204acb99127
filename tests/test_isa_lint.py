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
    hot = [n for n in stats if re.search(r"gemm_w256[pr]_kernel|gemm_w256_kernel<(BF16|F16), (true|false), (true|false), [01456]>|"
                                         r"conv_igemm_w(256|512)_kernel|conv_wgrad_w256_kernel|attn_(fwd|bwd)", n)]
    assert len(hot) >= 100, len(hot)      # both operand types (BF16, F16) of every persistent / one-tile GEMM and attention kernel
    for n in hot:
        s = stats[n]
        assert s.get("scratch_bytes", 0) == 0 and s.get("spills", 0) == 0 and s.get("scratch_ops", 0) == 0, (n, s)
    for n in stats:
        if "gemm_w256" in n and "lab" not in n:
            assert stats[n].get("agpr", 0) == 256 and stats[n].get("vgpr", 0) <= 512, (n, stats[n])


def test_every_gemm_kernel_is_an_mfma_kernel(stats):
    for n in stats:
        if n.startswith("void gemm_w256"):
            assert stats[n].get("mfma", 0) >= 64, (n, stats[n].get("mfma"))


def test_no_inline_asm_vector_instructions_in_the_attention_sources():
    """Round 5: an inline-asm v_max3_f32 behind the S products read the MFMA's destination registers before they were written — inline asm is invisible
    to the compiler's hazard recognizer (no s_nop inserted), the softmax stayed valid and the bits became launch-dependent (profiles/r05_attention_lab.txt).
    Vector arithmetic on MFMA results must be compiler-visible: no `asm("v_...")` in the attention sources (the empty register pin `asm volatile("" : "+v")` is fine)."""
    src = os.path.join(ROOT, "enhancing-transformers_amd", "csrc")
    bad = []
    for name in ("attention.hip", "attention_common.h", "x3.hip"):
        text = open(os.path.join(src, name)).read()
        for m in re.finditer(r'asm\s*(?:volatile)?\s*\(\s*"\s*(v_\w+)', text):
            bad.append((name, m.group(1)))
    assert not bad, bad


def test_attention_row_maximum_is_fused_and_hazard_padded(stats):
    """the nested fmaxf must still become v_max3_f32 (the attention objects are built with -fno-honor-nans), and the kernels keep their occupancy classes"""
    for ot in ("BF16", "F16"):      # both operand types: same instruction stream up to the MFMA / pack opcodes
        fwd = f"void attn_fwd_pre_kernel<{ot}>(unsigned short const*, int, int, int, unsigned short*, float*)"
        assert stats[fwd].get("max3", 0) >= 15, stats[fwd]                                                                      # the 32-way row maximum: v_max3 chains, no canonicalising v_max x, x
        assert stats[fwd]["vgpr"] <= 128                                                                                        # four waves per SIMD
        dkv = [n for n in stats if n.startswith(f"void attn_bwd_dkv_kernel<true, true, {ot}>")]
        dq = [n for n in stats if n.startswith(f"void attn_bwd_dq_kernel<2, {ot}>")]
        assert dkv and dq and stats[dkv[0]]["vgpr"] <= 168 and stats[dq[0]]["vgpr"] <= 168                                          # three waves per SIMD
        for n in dkv + dq + [fwd]:
            assert stats[n].get("scratch_bytes", 0) == 0 and stats[n].get("spills", 0) == 0, (n, stats[n])
