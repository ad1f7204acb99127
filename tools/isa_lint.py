"""ISA lint of the shipped library: disassemble libenh_hip.so's gfx950 code objects and report, per kernel symbol, the facts DESIGN.md
states about the instruction stream (§3.1c: the tile-claim atomic stays ONE in-flight instruction; §3.1b: the one-wave-per-SIMD kernels
do not touch scratch).  Runs on CPU (llvm-objdump / llvm-readelf of the ROCm toolchain); `tests/test_isa_lint.py` asserts on it.

    python tools/isa_lint.py [path/to/libenh_hip.so]      # prints the table
"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_SO = os.path.join(ROOT, "enhancing-transformers_amd", "lib", "libenh_hip.so")


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def code_objects(so_path, workdir):
    """the gfx950 code objects bundled in the .so (llvm-objdump --offloading writes them next to its input: work on a copy)"""
    local = os.path.join(workdir, "lib.so")
    shutil.copy(so_path, local)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "hipv4-amdgcn-amd-amdhsa--gfx950" in f)


def kernel_stats(so_path=DEFAULT_SO):
    """{demangled kernel name: {mbcnt, bcnt1, vmcnt0, atomics, scratch_ops, scratch_bytes, vgpr, agpr, spills, lds}}"""
    stats = collections.defaultdict(lambda: collections.Counter())
    with tempfile.TemporaryDirectory() as wd:
        objs = code_objects(so_path, wd)
        assert objs, f"no gfx950 code object in {so_path}"
        for o in objs:
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", o], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    cur = m.group(1)
                    stats[cur]["_seen"] += 1
                    continue
                if cur is None:
                    continue
                if "v_mbcnt" in line:
                    stats[cur]["mbcnt"] += 1
                if "s_bcnt1" in line:
                    stats[cur]["bcnt1"] += 1
                if "s_waitcnt vmcnt(0)" in line:
                    stats[cur]["vmcnt0"] += 1
                if "global_atomic" in line or "buffer_atomic" in line or "flat_atomic" in line:
                    stats[cur]["atomics"] += 1
                if "scratch_" in line:
                    stats[cur]["scratch_ops"] += 1
                if "v_mfma" in line:
                    stats[cur]["mfma"] += 1
                if "v_max3_f32" in line:
                    stats[cur]["max3"] += 1
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", o], capture_output=True, text=True, check=True).stdout
            for b in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                b = ".agpr_count:" + b
                m = re.search(r"\n\s*\.name:\s*(\S+)\n\s*\.private_segment_fixed_size:\s*(\d+)", b)
                if not m:
                    continue
                k = stats[m.group(1)]
                k["scratch_bytes"] = int(m.group(2))
                for key, pat in (("agpr", r"^\.agpr_count:\s*(\d+)"), ("vgpr", r"\.vgpr_count:\s*(\d+)"), ("spills", r"\.vgpr_spill_count:\s*(\d+)"),
                                 ("lds", r"\.group_segment_fixed_size:\s*(\d+)"), ("sgpr", r"\.sgpr_count:\s*(\d+)")):
                    mm = re.search(pat, b, re.M)
                    if mm:
                        k[key] = int(mm.group(1))
    names = [n for n in stats if "scratch_bytes" in stats[n]]      # kernels only (device functions have no metadata entry)
    dm = _demangle(names)
    return {dm[n]: dict(stats[n]) for n in names}


def persistent_gemm_pairs(stats):
    """(dynamic-schedule symbol, its static-schedule twin) for every persistent GEMM instantiation"""
    pairs = []
    for n in stats:
        m = re.match(r"void (gemm_w256[pr]_kernel)<(.*), true>\(GemmArgs\)$", n)
        if m:
            twin = f"void {m.group(1)}<{m.group(2)}, false>(GemmArgs)"
            assert twin in stats, twin
            pairs.append((n, twin))
    return sorted(pairs)


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_SO
    st = kernel_stats(so)
    print(f"{len(st)} kernels in {so}")
    print(f"{'kernel':84s} vgpr agpr scratch spills vmcnt0 mbcnt bcnt1 atomics")
    for n in sorted(st):
        s = st[n]
        if s.get("agpr", 0) or s.get("scratch_bytes", 0) or "gemm_w256" in n:
            print(f"{n[:84]:84s} {s.get('vgpr', 0):4d} {s.get('agpr', 0):4d} {s.get('scratch_bytes', 0):7d} {s.get('spills', 0):6d} "
                  f"{s.get('vmcnt0', 0):6d} {s.get('mbcnt', 0):5d} {s.get('bcnt1', 0):5d} {s.get('atomics', 0):7d}")
    bad = 0
    for dyn, sta in persistent_gemm_pairs(st):
        d, s = st[dyn], st[sta]
        ok = d.get("mbcnt", 0) == 0 and d.get("bcnt1", 0) == 0 and d.get("vmcnt0", 0) == s.get("vmcnt0", 0) + 1
        bad += not ok
        if not ok:
            print("LINT", dyn, d, "static twin vmcnt0", s.get("vmcnt0", 0))
    print("persistent-GEMM tile-claim lint:", "FAIL" if bad else "ok", f"({bad} bad)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
