"""The C-ABI library loads and exports exactly what include/lmod_hip.h declares (no compute, no GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, "include", "lmod_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\bint\s+(lmod_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, params = m.group(1), m.group(2)
        sig = ""
        for p in params.split(","):
            p = p.strip()
            if "hipStream_t" in p or "*" in p:
                sig += "p"
            elif p.startswith("unsigned long long"):
                sig += "Q"
            elif p.startswith("long long"):
                sig += "q"
            elif p.startswith("float"):
                sig += "f"
            elif p.startswith("int"):
                sig += "i"
            else:
                raise AssertionError(f"unparsed parameter {p!r} in {name}")
        decls[name] = sig
    return decls


def test_header_matches_binding_table():
    from llavamod import _hip
    decls = _header_decls()
    assert set(decls) == set(_hip.SIGNATURES), set(decls) ^ set(_hip.SIGNATURES)
    for name, sig in decls.items():
        assert sig == _hip.SIGNATURES[name], f"{name}: header {sig} != binding {_hip.SIGNATURES[name]}"


def test_library_exports_every_declared_symbol():
    from llavamod import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _hip.load()
    for name in _header_decls():
        assert hasattr(lib, name), name


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from llavamod import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _hip.load()


def test_product_path_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under llava-mod_amd/ may import or execute it, nothing may read
    /root/reference, and there is no CPU fallback module behind the kernels."""
    import re
    root = os.path.join(ROOT, "llava-mod_amd")
    bad = []
    for dp, _, files in os.walk(root):
        for f in files:
            if not f.endswith((".py", ".hip", ".h")):
                continue
            src = open(os.path.join(dp, f), encoding="utf-8").read()
            if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M) or "/root/reference" in src:
                bad.append(os.path.join(dp, f))
    assert not bad, bad
    # bench.py and __graft_entry__.py may use the oracle only inside cpu_baseline() / smoke()
    for name, allowed in (("bench.py", "cpu_baseline"), ("__graft_entry__.py", "smoke")):
        lines = open(os.path.join(ROOT, name), encoding="utf-8").read().split("\n")
        for i, line in enumerate(lines):
            if re.match(r"\s*(from|import)\s+oracle\b", line):
                enclosing = next((l for l in reversed(lines[:i]) if re.match(r"def \w+", l)), "")
                assert enclosing.startswith(f"def {allowed}"), (name, i + 1, enclosing)


@pytest.mark.parametrize("src,agprs,wpe,nkern", [("attn_fwd2.hip", 64, 2, 4), ("attn_bwd2.hip", 256, 1, 10), ("attn_fwd3.hip", 256, 1, 2)])
def test_asm_owned_accumulators_are_not_touched_by_the_compiler(tmp_path, src, agprs, wpe, nkern):
    """attn_fwd2.hip keeps its O^T accumulators in a[0:63], attn_bwd2.hip its dQ / dK / dV accumulators in a[0:255], attn_fwd3.hip its O^T
    strips and Q fragments in a[0:191] (built with the 256-register allocation), through inline asm only (see the file headers).  Audit the ISA hipcc emits: no spills, exactly the accumulator registers the asm
    names, and no instruction outside an asm statement that references an accumulator register."""
    import re
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "llava-mod_amd", "csrc")
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    env = dict(os.environ, KEEP_ISA=str(tmp_path / "k.s"), WPE=str(wpe))
    env["PATH"] = "/opt/rocm/bin:" + env.get("PATH", "")
    subprocess.run([os.path.join(csrc, "hipcc_agpr.sh"), os.path.join(csrc, src), str(tmp_path / "k.o"), str(agprs)],
                   check=True, env=env)
    isa = open(tmp_path / "k.s").read()
    assert re.findall(r"\.vgpr_spill_count:\s+(\d+)", isa) == ["0"] * nkern
    assert re.findall(r"\.private_segment_fixed_size:\s+(\d+)", isa) == ["0"] * nkern
    assert re.findall(r"\.agpr_count:\s+(\d+)", isa) == [str(agprs)] * nkern
    in_asm, bad = False, []
    for line in isa.split("\n"):
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif not in_asm and not line.lstrip().startswith((";", ".")) and re.search(r"(^|[\s,\[])a(\d+|\[\d+:\d+\])", line):
            bad.append(line.strip())
    assert not bad, bad[:5]


def test_attn_bwd_head_split_plan():
    """lmod_attn_bwd_nsplit is a host-side query (no device work): it cuts a KV head's group of query heads only while the dK/dV grid
    (KV heads x key blocks [causal: pairs] x batch) has fewer workgroups than 0.9 x the CU count (256 without a device), never beyond
    the group size or 8, and not at all for multi-head attention."""
    from llavamod import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    f = _hip.load().lmod_attn_bwd_nsplit
    assert f(16, 2048, 14, 2, 64, 1) == 2          # Qwen2-0.5B student, the d2s shell's batch: 128 workgroups -> 256
    assert f(4, 8192, 14, 2, 64, 1) == 2
    assert f(16, 2048, 14, 2, 64, 0) == 1          # non-causal: 8 key blocks, 256 workgroups already
    assert f(16, 2048, 16, 16, 64, 1) == 1         # one query head per KV head: nothing to cut
    assert f(16, 2048, 28, 4, 128, 1) == 1         # Qwen2-7B geometry at B 16: 256 workgroups
    assert f(1, 2048, 28, 4, 128, 1) == 7          # a single sample: down to one query head per workgroup
    assert f(1, 256, 64, 1, 128, 1) == 8           # capped at 8 parts
    assert f(6, 2048, 14, 2, 64, 1) == 4           # 48 workgroups want 5 parts; 7 heads at ceil(7/5) = 2 per part fill only 4: no empty part
    assert f(16, 2048, 14, 2, 96, 1) == 1 and f(0, 2048, 14, 2, 64, 1) == 1 and f(16, 2048, 14, 3, 64, 1) == 1   # outside the envelope


def test_product_library_ships_the_product_not_the_lab():
    """the A/B arms that lost their measurements (round-1 generic attention kernels, the one-wave-per-SIMD
    forward of round 5) are not compiled into liblmod_hip.so; `make LAB=1` builds them into liblmod_hip_lab.so, which exports the
    same C-ABI and which nothing under llava-mod_amd/llavamod loads."""
    from llavamod import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    blob = open(_hip.LIB_PATH, "rb").read()
    for lab_kernel in (b"attn_fwd3_kernel", b"attn_bwd_dq_kernel", b"attn_bwd_dkv_kernel", b"attn_fwd_kernelILi"):
        assert lab_kernel not in blob, lab_kernel
    for product_kernel in (b"attn_fwd2_kernel", b"attn_bwd2_kernel", b"gemm4_kernel", b"gemm4t_kernel"):
        assert product_kernel in blob, product_kernel
    mk = open(os.path.join(ROOT, "llava-mod_amd", "csrc", "Makefile")).read()
    default_srcs = next(l for l in mk.split("\n") if l.startswith("SRCS ="))
    assert "attn_fwd3.hip" not in default_srcs and "attn_lab.hip" not in default_srcs
    for dp, _, files in os.walk(os.path.join(ROOT, "llava-mod_amd", "llavamod")):
        for f in files:
            if f.endswith(".py"):
                assert "liblmod_hip_lab" not in open(os.path.join(dp, f), encoding="utf-8").read(), f
    lab = os.path.join(os.path.dirname(_hip.LIB_PATH), "liblmod_hip_lab.so")
    if os.path.exists(lab):
        import ctypes
        lib = ctypes.CDLL(lab)
        for name in _header_decls():
            assert hasattr(lib, name), name
        assert b"attn_fwd3_kernel" in open(lab, "rb").read()
