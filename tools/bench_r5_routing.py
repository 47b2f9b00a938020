"""Round-5 routing switches, each arm in ONE process on ONE box (the switches are read per launch):
  LMOD_GEMM_PERSIST_GROUPED  grouped (MoE, m_valid) launches on the persistent four-wave kernel — fused SwiGLU forward, the experts'
                             down projection and its dgrad — against one tile per workgroup
  LMOD_GEMM_SB4              the dense fused SwiGLU backward on the persistent four-wave kernel against the 8-wave instantiation
Config-2 MoE shapes: micro-batch 16 x 2048 tokens, top-2 of 4 experts (capacity 24576 rows per expert, 65536 live rows in all).
Prints JSON lines (profiles/r05_grouped_persistent.jsonl)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
os.environ.setdefault("LMOD_GEMM_ENV_DYNAMIC", "1")      # both arms of a routing switch in one process
from llavamod import kernels as K  # noqa: E402

BF = torch.bfloat16


def t(fn, it=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


def ab(name, env, fn, flops, check=None):
    out = {}
    res = {}
    for arm in ("0", "1", "0", "1"):
        os.environ[env] = ("2" if env == "LMOD_GEMM_PERSIST_GROUPED" else "1") if arm == "1" else "0"
        dt = t(fn)
        res.setdefault(arm, []).append(flops / dt / 1e12)
        if check is not None:
            out[arm] = check().clone()
    same = None if check is None else bool(torch.equal(out["0"], out["1"]))
    row = {"launch": name, "switch": env, "tf_off": [round(x, 1) for x in res["0"]], "tf_on": [round(x, 1) for x in res["1"]],
           "gain_pct": round((sum(res["1"]) / sum(res["0"]) - 1) * 100, 2), "bit_identical": same}
    print(json.dumps(row), flush=True)
    os.environ.pop(env, None)


E, C, H, I = 4, 24576, 2048, 5504
for tag, rows in (("balanced", [16384] * 4), ("skewed", [24576, 18000, 14000, 8960])):
    mv = torch.tensor(rows, dtype=torch.int32, device="cuda")
    live = sum(rows)
    x = torch.randn(E, C, H, device="cuda").to(BF)
    wgu = torch.randn(E, 2 * I, H, device="cuda").to(BF)
    act = torch.zeros(E, C, I, device="cuda", dtype=BF)
    gu = torch.zeros(E, C, 2 * I, device="cuda", dtype=BF)
    ab(f"grouped fused SwiGLU forward [{live} live x {2 * I} x {H}] ({tag})", "LMOD_GEMM_PERSIST_GROUPED",
       lambda: K.gemm_swiglu(x, wgu, act=act, gu=gu, m_valid=mv), 2.0 * live * 2 * I * H, check=lambda: act)
    wd = torch.randn(E, H, I, device="cuda").to(BF)
    y = torch.zeros(E, C, H, device="cuda", dtype=BF)
    ab(f"grouped down projection [{live} live x {H} x {I}] ({tag})", "LMOD_GEMM_PERSIST_GROUPED",
       lambda: K.gemm_nt(act, wd, out=y, m_valid=mv), 2.0 * live * H * I, check=lambda: y)
    wgut = torch.randn(E, H, 2 * I, device="cuda").to(BF)
    dx = torch.zeros(E, C, H, device="cuda", dtype=BF)
    ab(f"grouped gate/up dgrad [{live} live x {H} x {2 * I}] ({tag})", "LMOD_GEMM_PERSIST_GROUPED",
       lambda: K.gemm_nt(gu, wgut, out=dx, m_valid=mv), 2.0 * live * H * 2 * I, check=lambda: dx)
    del x, wgu, act, gu, wd, y, wgut, dx
T = 32768
gud = torch.randn(T, 2 * I, device="cuda").to(BF)
dy = torch.randn(T, H, device="cuda").to(BF)
wdt = torch.randn(I, H, device="cuda").to(BF)
outd = torch.empty_like(gud)
ab(f"dense fused SwiGLU backward [{T} x {I} x {H}]", "LMOD_GEMM_SB4", lambda: K.gemm_swiglu_bwd(dy, wdt, gud, out=outd, K=H), 2.0 * T * I * H,
   check=lambda: outd)
