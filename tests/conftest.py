import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "llava-mod_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


# tests run BOTH arms of the GEMM routing switches inside one process (monkeypatch.setenv between launches): ask the library to
# re-read them per launch — the product default reads them once at the first launch (csrc/gemm.hip `gemm_routing`)
os.environ.setdefault("LMOD_GEMM_ENV_DYNAMIC", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
