"""Same-box, same-process A/B of library builds (alt_libs/*.so + the current one) on GEMM shapes, with the card's socket
power and gfx clock sampled beside every arm (sysfs hwmon, 100 Hz): TF, W, MHz per build — the table VERDICT r02 next #5 asks
for (equal power, one process, one box).  Plain ctypes on lmod_gemm_bf16_nt (no dependence on newer symbols).

    python tools/gemm_variants_ab.py [--seconds 2.0] [--rounds 2] [--shapes qkv,sq8k] [--only name1,name2]
"""
import argparse, ctypes, glob, os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=1.5)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--shapes", default="qkv")
ap.add_argument("--only", default="")
ap.add_argument("--vendor", action="store_true", help="add PyTorch-ROCm F.linear (hipBLASLt) as an arm: the yardstick on the same box, same process")
ap.add_argument("--pads", default="0", help="comma list: extra elements on both operands' leading dimensions (L2 channel striding probe)")
args = ap.parse_args()
SHAPES = {"qkv": (32768, 12288, 4096), "sq8k": (8192, 8192, 8192), "sdown": (32768, 2048, 5504), "sqkv": (32768, 6144, 2048),
          "tdown": (32768, 4096, 11008), "sq4k": (4096, 4096, 4096), "longk": (4096, 8192, 32768)}
libs = {"current": os.path.join(ROOT, "llava-mod_amd", "llavamod", "_lib", "liblmod_hip.so")}
for f in sorted(glob.glob(os.path.join(ROOT, "alt_libs", "*.so"))):
    libs[os.path.basename(f)[len("liblmod_"):-3]] = f
if args.only:
    keep = set(args.only.split(",")) | {"current"}
    libs = {k: v for k, v in libs.items() if k in keep}
P, I, Q = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
fns = {}
for name, path in libs.items():
    f = ctypes.CDLL(path).lmod_gemm_bf16_nt
    f.restype = I
    f.argtypes = [P, P, P, P, I, I, I, I, I, I, I, Q, Q, Q, P, P, I, I, I, P]
    fns[name] = f


def sensors():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(d + "/hwmon/hwmon*")
        if hw:
            p = next((x for x in (hw[0] + "/power1_average", hw[0] + "/power1_input") if os.path.exists(x)), None)
            fq = hw[0] + "/freq1_input"
            if p and os.path.exists(fq):
                out.append((p, fq))
    return out


class Sampler(threading.Thread):
    def __init__(self, cards):
        super().__init__(daemon=True)
        self.cards, self.rows, self.stop = cards, [], False

    def run(self):
        fds = [(open(p), open(f)) for p, f in self.cards]
        while not self.stop:
            row = [time.perf_counter()]
            for fp, ff in fds:
                try:
                    fp.seek(0); ff.seek(0)
                    row += [int(fp.read()) / 1e6, int(ff.read()) / 1e6]
                except Exception:
                    row += [float("nan")] * 2
            self.rows.append(row)
            time.sleep(0.01)


cards = sensors()
smp = Sampler(cards)
smp.start()
res = {}
for sh0 in [f"{x}+{pd}" for x in args.shapes.split(",") for pd in args.pads.split(",")]:
    sh, pad = sh0.split("+")[0], int(sh0.split("+")[1])
    M, N, Kd = SHAPES[sh]
    sh = sh if pad == 0 else f"{sh} ld+{pad}"
    ld = Kd + pad
    a = torch.randn(M, ld, device="cuda").to(torch.bfloat16); b = torch.randn(N, ld, device="cuda").to(torch.bfloat16)
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * Kd
    s = torch.cuda.current_stream().cuda_stream

    def run(f, n):
        for _ in range(n):
            rc = f(a.data_ptr(), b.data_ptr(), o.data_ptr(), None, M, N, Kd, ld, ld, N, 1, 0, 0, 0, None, None, 0, 0, 0, s)
            assert rc == 0
    if args.vendor and "vendor" not in fns:
        fns["vendor"] = None
    def run(f, n, _run=run):
        if f is None:
            for _ in range(n):
                torch.nn.functional.linear(a[:, :Kd], b[:, :Kd])
        else:
            _run(f, n)
    ref = None
    for name, f in fns.items():             # warm up; results of every build against the current one (timing-only builds differ)
        o.zero_(); run(f, 3); torch.cuda.synchronize()
        if ref is None:
            ref = o.clone()
        else:
            if f is not None: print(f"# {sh} {name}: max |out - current| = {(o.float() - ref.float()).abs().max().item():.4g}", flush=True)
    for rnd in range(args.rounds):
        for name, f in fns.items():
            t0 = time.perf_counter(); run(f, 20); torch.cuda.synchronize()
            n = max(20, int(args.seconds / ((time.perf_counter() - t0) / 20)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0 = time.perf_counter()
            e0.record(); run(f, n); e1.record(); torch.cuda.synchronize()
            w1 = time.perf_counter()
            res.setdefault((sh, name), []).append((fl * n / (e0.elapsed_time(e1) * 1e-3) / 1e12, w0, w1))
smp.stop = True
smp.join()
rows = smp.rows
# the card under test: the one with the largest mean clock over the arms
def stat(ci, t0, t1, col):
    xs = [r[1 + 2 * ci + col] for r in rows if t0 + 0.3 <= r[0] <= t1 and r[1 + 2 * ci + col] == r[1 + 2 * ci + col]]
    return sum(xs) / len(xs) if xs else float("nan")
t_all0, t_all1 = min(v[0][1] for v in res.values()), max(v[-1][2] for v in res.values())
ci = max(range(len(cards)), key=lambda c: stat(c, t_all0, t_all1, 0)) if cards else None
print("| shape | build | TF (per round) | mean TF | socket W | gfx MHz | TF per kW |")
print("|---|---|---|---|---|---|---|")
for (sh, name), v in res.items():
    tf = [x[0] for x in v]
    w = sum(stat(ci, x[1], x[2], 0) for x in v) / len(v) if ci is not None else float("nan")
    mhz = sum(stat(ci, x[1], x[2], 1) for x in v) / len(v) if ci is not None else float("nan")
    m = sum(tf) / len(tf)
    print(f"| {sh} | {name} | {' / '.join('%.0f' % x for x in tf)} | {m:.0f} | {w:.0f} | {mhz:.0f} | {m / w * 1000:.0f} |", flush=True)
