"""The generated K loops (CPU-side checks; the numerics are tests/test_kernels_gpu.py's job).

* the tracked headers are what the generators write today (an edit to a generator without regenerating, or a hand edit of a
  header, fails here);
* gemm4t_kernel's LDS image: the staging map of gemm.hip followed by the transposing reads of gemm4t_loop_asm.h hands every lane
  the MFMA operand fragment it must hold, and no half-wave read touches an LDS bank twice (tools/probe/gemm4t_layout.py replays
  both on the CPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "llava-mod_amd", "csrc")


@pytest.mark.parametrize("script,args,header", [
    ("gen_gemm4_loop.py", ["--variant", "b2", "--persist", "1"], "gemm4_loop_asm.h"),
    ("gen_gemm4_loop.py", ["--variant", "b3", "--suffix", "_B3"], "gemm4_loop_asm_b3.h"),
    ("gen_gemm4t_loop.py", [], "gemm4t_loop_asm.h"),
])
def test_tracked_loop_headers_are_current(script, args, header, tmp_path):
    out = tmp_path / header
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *args, "--out", str(out)], check=True, capture_output=True)
    assert out.read_text() == open(os.path.join(CSRC, header)).read(), f"{header} is not what tools/{script} writes: regenerate it"


def test_gemm4t_lds_image_and_transposing_reads():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "gemm4t_layout.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "check out" in r.stdout


def test_gemm4t_staging_constants_match_the_generator():
    src = open(os.path.join(CSRC, "gemm.hip")).read()
    gen = open(os.path.join(ROOT, "tools", "gen_gemm4t_loop.py")).read()
    for name, val in (("G4T_SUB", 8448), ("G4T_PIECE", 1056), ("G4T_OPB", 33792), ("G4T_STAGE", 67584)):
        assert f"#define {name} {val}" in src
    assert "SUB, PIECE, OPB = 8448, 1056, 33792" in gen
