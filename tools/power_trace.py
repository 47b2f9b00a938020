"""Power / clock trace (sysfs hwmon, ~100 Hz) beside the kernels whose roofline fraction is argued from the power limit:
register-only MFMA spin (tools/probe/mfma_power), the dominant GEMM, attention forward + backward.
Writes gpurun_out/power/trace.csv and prints a per-phase summary (markdown) for profiles/.

  python tools/power_trace.py            # all phases
"""
import glob, json, math, os, subprocess, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
OUT = os.path.join(ROOT, "gpurun_out", "power")
os.makedirs(OUT, exist_ok=True)


def sensors():
    cards = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(d + "/hwmon/hwmon*")
        if not hw:
            continue
        h = hw[0]
        p = next((f for f in (h + "/power1_average", h + "/power1_input") if os.path.exists(f)), None)
        f = h + "/freq1_input" if os.path.exists(h + "/freq1_input") else None
        if p and f:
            cards.append((d, p, f))
    return cards


class Sampler(threading.Thread):
    def __init__(self, cards, hz=100):
        super().__init__(daemon=True)
        self.cards, self.dt, self.rows, self.stop = cards, 1.0 / hz, [], False

    def run(self):
        fds = [(open(p), open(f)) for _, p, f in self.cards]
        while not self.stop:
            t = time.perf_counter()
            row = [t]
            for fp, ff in fds:
                try:
                    fp.seek(0); ff.seek(0)
                    row += [int(fp.read()) / 1e6, int(ff.read()) / 1e6]
                except Exception:
                    row += [float("nan"), float("nan")]
            self.rows.append(row)
            time.sleep(max(0.0, self.dt - (time.perf_counter() - t)))


def main():
    import torch
    from llavamod import kernels as K
    cards = sensors()
    assert cards, "no hwmon power / clock sensors found"
    smp = Sampler(cards)
    smp.start()
    phases = []

    def phase(name, fn):
        torch.cuda.synchronize()
        time.sleep(1.0)
        t0 = time.perf_counter()
        info = fn()
        torch.cuda.synchronize()
        phases.append((name, t0, time.perf_counter(), info))

    BF = torch.bfloat16
    phase("idle", lambda: time.sleep(1.0))

    def spin():
        exe = os.path.join(ROOT, "tools", "probe", "mfma_power")
        if not os.path.exists(exe):                          # the binary is not tracked: build it from the source beside it
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, exe + ".hip"], check=True)
        out = subprocess.run([exe, "120"], capture_output=True, text=True, timeout=120).stdout.strip().split("\n")
        tf = {16: [], 32: []}
        for l in out:
            w = l.split()
            tf[int(w[1].split("x")[0])].append(float(w[-2]))
        return "; ".join("%dx%d: %.0f / %.0f / %.0f TF (first / median / max of %d)" % (k, k, v[0], sorted(v)[len(v) // 2], max(v), len(v))
                         for k, v in tf.items())
    phase("mfma_spin (registers only, 120 x {16x16x32, 32x32x16} alternating)", spin)

    M, N, Kd = 32768, 12288, 4096
    a = torch.randn(M, Kd, device="cuda").to(BF); b = torch.randn(N, Kd, device="cuda").to(BF)
    o = torch.empty(M, N, device="cuda", dtype=BF)

    def gemm():
        n = 1200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            K.gemm_nt(a, b, out=o)
        e1.record(); torch.cuda.synchronize()
        return "%d launches, %.0f TF" % (n, 2.0 * M * N * Kd * n / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    phase("gemm_nt 32768x12288x4096 (teacher QKV shape)", gemm)

    Bb, S, nh, hd = 16, 2048, 16, 128
    qkv = torch.randn(Bb * S, 3 * nh * hd, device="cuda").to(BF)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:2 * nh * hd], qkv[:, 2 * nh * hd:]
    sc = 1 / math.sqrt(hd)
    oo, lse = K.attn_fwd(q, k, v, Bb, S, nh, nh, hd, sc, True)
    do = torch.randn(Bb * S, nh * hd, device="cuda").to(BF)
    d = torch.empty_like(qkv)
    fl = 4.0 * Bb * nh * S * S * hd * 0.5

    def afwd():
        n = 6000
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            K.attn_fwd(q, k, v, Bb, S, nh, nh, hd, sc, True)
        e1.record(); torch.cuda.synchronize()
        return "%d launches, %.0f TF" % (n, fl * n / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    phase("attention forward B16 S2048 nh16 hd128 causal", afwd)

    def abwd():
        n = 2000
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            K.attn_bwd(q, k, v, oo, do, lse, d[:, :nh * hd], d[:, nh * hd:2 * nh * hd], d[:, 2 * nh * hd:], Bb, S, nh, nh, hd, sc, True)
        e1.record(); torch.cuda.synchronize()
        return "%d launches, %.0f TF algorithmic (2.5x forward)" % (n, 2.5 * fl * n / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    phase("attention backward (same shape)", abwd)

    time.sleep(0.5)
    smp.stop = True
    smp.join()
    rows = smp.rows
    with open(os.path.join(OUT, "trace.csv"), "w") as f:
        f.write("t," + ",".join("card%d_W,card%d_MHz" % (i, i) for i in range(len(cards))) + "\n")
        for r in rows:
            f.write(",".join("%.4f" % x for x in r) + "\n")
    # the card under test: largest clock swing between idle and the GEMM phase
    def stats(ci, t0, t1, col):
        xs = [r[1 + 2 * ci + col] for r in rows if t0 + 0.3 <= r[0] <= t1]
        xs = [x for x in xs if x == x]
        return (sum(xs) / len(xs), min(xs), max(xs), len(xs)) if xs else (float("nan"),) * 3 + (0,)
    gi = [p for p in phases if p[0].startswith("gemm")][0]
    ci = max(range(len(cards)), key=lambda c: stats(c, gi[1], gi[2], 1)[0] - stats(c, phases[0][1] - 0.3, phases[0][2], 1)[0])
    print("# Power and clock beside the MFMA-bound kernels (sysfs hwmon `power1_average` / `freq1_input`, %d Hz sampling, card %s)\n"
          % (round(1 / smp.dt), os.path.basename(os.path.dirname(cards[ci][0]))))
    print("| phase | socket power W (mean / max) | gfx clock MHz (mean / min / max) | samples | measured |")
    print("|---|---|---|---|---|")
    for name, t0, t1, info in phases:
        pw, ck = stats(ci, t0, t1, 0), stats(ci, t0, t1, 1)
        print("| %s | %.0f / %.0f | %.0f / %.0f / %.0f | %d | %s |" % (name, pw[0], pw[2], ck[0], ck[1], ck[2], pw[3], info))
    json.dump([dict(name=n, seconds=t1 - t0, info=i) for n, t0, t1, i in phases], open(os.path.join(OUT, "phases.json"), "w"))


if __name__ == "__main__":
    main()
