"""A/B of library builds on the weight-gradient (TN, both operands reduction-major) launches of the step, one process per build, alternating:
shipped vs alt_libs/liblmod_<name>.so (argv).  TF per shape + a bit-identity check of the results between builds (saved by the first arm)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import json, os, sys, torch
sys.path.insert(0, os.path.join(%r, "llava-mod_amd"))
from llavamod import kernels as K
def t(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
T = 32768
res, sums = {}, {}
for tag, M, N in [("gate+up", 11008, 2048), ("qkv", 6144, 2048), ("down", 2048, 5504), ("o", 2048, 2048), ("teacher-like 4096x4096", 4096, 4096)]:
    g0 = torch.Generator(device="cuda").manual_seed(1)
    dy = torch.randn(T, M, device="cuda", generator=g0).to(torch.bfloat16); x = torch.randn(T, N, device="cuda", generator=g0).to(torch.bfloat16)
    g = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    K.gemm_wgrad(dy, x, g, a_kmajor=True)
    sums[tag] = [float(g.double().sum()), float(g.double().abs().sum()), float(g[17, 33]), float(g[-1, -1])]
    ms = t(lambda: K.gemm_wgrad(dy, x, g, a_kmajor=True))
    res[tag] = round(2.0 * T * M * N / ms / 1e9, 1)
E_, C = 4, 24576
rows = torch.tensor([9000, 12500, 20036, 24000], device="cuda", dtype=torch.int32)
for tag, M, N in [("moe gate+up", 11008, 2048), ("moe down", 2048, 5504)]:
    g0 = torch.Generator(device="cuda").manual_seed(2)
    dy = torch.randn(E_, C, M, device="cuda", generator=g0).to(torch.bfloat16); x = torch.randn(E_, C, N, device="cuda", generator=g0).to(torch.bfloat16)
    g = torch.zeros(E_, M, N, device="cuda", dtype=torch.float32)
    K.gemm_tn(dy, x, out=g, accumulate=True, k_valid=rows)
    sums[tag] = [float(g.double().sum()), float(g.double().abs().sum()), float(g[1, 17, 33]), float(g[-1, -1, -1])]
    ms = t(lambda: K.gemm_tn(dy, x, out=g, accumulate=True, k_valid=rows))
    res[tag] = round(2.0 * float(rows.sum()) * M * N / ms / 1e9, 1)
print("RES " + json.dumps({"tf": res, "sums": sums}))
''' % ROOT
names = [a for a in sys.argv[1:]]
libs = [("shipped", os.path.join(ROOT, "llava-mod_amd", "llavamod", "_lib", "liblmod_hip.so"))] + \
       [(n, os.path.join(ROOT, "alt_libs", f"liblmod_{n}.so")) for n in names]
ref = None
for rnd in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for name, path in libs:
        out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, LMOD_HIP_LIB=path), capture_output=True, text=True, timeout=400)
        line = [l for l in out.stdout.split("\n") if l.startswith("RES ")]
        if not line:
            print(json.dumps({"build": name, "error": out.stderr[-400:]}), flush=True)
            continue
        r = json.loads(line[0][4:])
        if ref is None:
            ref = r["sums"]
        print(json.dumps({"build": name, "round": rnd, **r["tf"], "bit_identical_to_first_arm": r["sums"] == ref}), flush=True)
