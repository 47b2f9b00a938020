"""A/B of attention library builds, one process per build, alternating rounds: shipped vs alt_libs/liblmod_<name>.so (argv).
Forward at the step's shapes (student B16 nh16, teacher B16 nh32, S 2048 causal) + non-causal + S 8192 + ragged; --bwd adds the backward."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = [a for a in sys.argv[1:] if not a.startswith("--")]
bwd = "--bwd" in sys.argv
CODE = r'''
import json, math, os, sys, torch
sys.path.insert(0, os.path.join(%r, "llava-mod_amd")); sys.path.insert(0, os.path.join(%r, "tools"))
from llavamod import kernels as K
from bench_kernels import timeit
BF = torch.bfloat16
res = {}
for tag, B, S, nh, causal, ragged in (("c_b16_nh16", 16, 2048, 16, True, False), ("c_b16_nh32", 16, 2048, 32, True, False), ("full_b8", 8, 2048, 16, False, False),
                                      ("c_s8192", 4, 8192, 16, True, False), ("c_b16_ragged", 16, 2048, 16, True, True)):
    hd = 128; ld = 3 * nh * hd
    qkv = torch.randn(B * S, ld, device="cuda").to(BF)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:2 * nh * hd], qkv[:, 2 * nh * hd:]
    sc = 1 / math.sqrt(hd)
    sl = torch.randint(1175, 2049, (B,), device="cuda", dtype=torch.int32) if ragged else None
    o, lse = K.attn_fwd(q, k, v, B, S, nh, nh, hd, sc, causal, sl)
    fl = 4.0 * B * nh * S * S * hd * (0.5 if causal else 1.0)
    t = timeit(lambda: K.attn_fwd(q, k, v, B, S, nh, nh, hd, sc, causal, sl))
    res["fwd_" + tag] = round(fl / t / 1e12, 1)
    if %r and tag in ("c_b16_nh16", "full_b8", "c_s8192"):
        do = torch.randn(B * S, nh * hd, device="cuda").to(BF); dqkv = torch.empty_like(qkv)
        f = lambda: K.attn_bwd(q, k, v, o, do, lse, dqkv[:, :nh * hd], dqkv[:, nh * hd:2 * nh * hd], dqkv[:, 2 * nh * hd:], B, S, nh, nh, hd, sc, causal, sl)
        t = timeit(f)
        res["bwd_" + tag] = round(2.5 * fl / t / 1e12, 1)
print("RES " + json.dumps(res))
''' % (ROOT, ROOT, bwd)
SHIPPED = os.path.join(ROOT, "llava-mod_amd", "llavamod", "_lib", "liblmod_hip.so")
# an arm is a library build (alt_libs/liblmod_<name>.so) or, spelled env:VAR=value, the shipped library under an environment switch
# (name@VAR=value: that library build under the switch)
def arm(n):
    if n.startswith("env:"):
        return (n, SHIPPED, dict([n[4:].split("=", 1)]))
    if "@" in n:
        lib, ev = n.split("@", 1)
        return (n, os.path.join(ROOT, "alt_libs", f"liblmod_{lib}.so"), dict([ev.split("=", 1)]))
    return (n, os.path.join(ROOT, "alt_libs", f"liblmod_{n}.so"), {})
libs = [("shipped", SHIPPED, {})] + [arm(n) for n in names]
for rnd in range(int(os.environ.get('ATTN_AB_ROUNDS', '3'))):
    for name, path, extra in libs:
        out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, LMOD_HIP_LIB=path, **extra), capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.split("\n") if l.startswith("RES ")]
        print(json.dumps({"build": name, "round": rnd, **(json.loads(line[0][4:]) if line else {"error": out.stderr[-300:]})}), flush=True)
