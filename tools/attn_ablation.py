"""Timing ablations of the attention kernels (WRONG RESULTS by construction — each build removes one class of work from the tile loop):
every alt_libs/liblmod_{f2a*,b2a*}.so in its own process beside the shipped library, B16 S2048 nh16 hd128 causal (the step's
student shape) and B8 S2048 non-causal.  Prints a markdown table (profiles/r05_attn_ablation.md is written from it)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import json, math, os, sys, torch
sys.path.insert(0, os.path.join(%r, "llava-mod_amd")); sys.path.insert(0, os.path.join(%r, "tools"))
from llavamod import kernels as K
from bench_kernels import timeit
BF = torch.bfloat16
res = {}
for tag, B, S, nh, causal in (("causal_b16_s2048", 16, 2048, 16, True), ("full_b8_s2048", 8, 2048, 16, False), ("causal_b4_s8192", 4, 8192, 16, True)):
    hd = 128; ld = 3 * nh * hd
    qkv = torch.randn(B * S, ld, device="cuda").to(BF)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:2 * nh * hd], qkv[:, 2 * nh * hd:]
    sc = 1 / math.sqrt(hd)
    o, lse = K.attn_fwd(q, k, v, B, S, nh, nh, hd, sc, causal, None)
    fl = 4.0 * B * nh * S * S * hd * (0.5 if causal else 1.0)
    t = timeit(lambda: K.attn_fwd(q, k, v, B, S, nh, nh, hd, sc, causal, None))
    res["fwd_" + tag] = round(fl / t / 1e12, 1)
    if sys.argv[1] == "bwd":
        do = torch.randn(B * S, nh * hd, device="cuda").to(BF); dqkv = torch.empty_like(qkv)
        f = lambda: K.attn_bwd(q, k, v, o, do, lse, dqkv[:, :nh * hd], dqkv[:, nh * hd:2 * nh * hd], dqkv[:, 2 * nh * hd:], B, S, nh, nh, hd, sc, causal, None)
        t = timeit(f)
        res["bwd_" + tag] = round(2.5 * fl / t / 1e12, 1)
print("RES " + json.dumps(res))
''' % (ROOT, ROOT)
libs = [("shipped", os.path.join(ROOT, "llava-mod_amd", "llavamod", "_lib", "liblmod_hip.so"))]
libs += [(os.path.basename(f)[len("liblmod_"):-3], f) for f in sorted(glob.glob(os.path.join(ROOT, "alt_libs", "liblmod_[fb]2a*.so")))]
libs.append(("shipped (again)", libs[0][1]))
rows = []
for name, path in libs:
    mode = "fwd" if name.startswith("f2") else "bwd"
    env = dict(os.environ, LMOD_HIP_LIB=path)
    out = subprocess.run([sys.executable, "-c", CODE, mode], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.split("\n") if l.startswith("RES ")]
    rows.append((name, json.loads(line[0][4:]) if line else {"error": out.stderr[-300:]}))
    print(name, rows[-1][1], flush=True)
