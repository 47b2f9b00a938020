"""Socket power and gfx clock (sysfs hwmon, ~100 Hz) while the vendor GEMM (F.linear) and ours run back to back for 2 s each."""
import glob, os, sys, threading, time
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K


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
        super().__init__(daemon=True); self.cards, self.rows, self.stop = cards, [], False
    def run(self):
        while not self.stop:
            row = []
            for p, f in self.cards:
                try: row += [int(open(p).read()) / 1e6, int(open(f).read()) / 1e6]
                except Exception: row += [float("nan")] * 2
            self.rows.append(row); time.sleep(0.01)


M, N, Kd = [int(v) for v in os.environ.get("SHAPE", "32768,12288,4096").split(",")]
FUSED = os.environ.get("FUSED") == "1"            # compare on the fused [gate; up] weight: vendor plain GEMM vs our fused SwiGLU GEMM (same flops)
if FUSED and "SHAPE" not in os.environ:
    N = 22016
g = torch.Generator(device="cuda"); g.manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
x, w = rnd(M, Kd), rnd(N, Kd)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
cards = sensors()
ours = (lambda: K.gemm_swiglu(x, w)) if FUSED else (lambda: K.gemm_nt(x, w, out=out))
arms = [("vendor F.linear", lambda: F.linear(x, w)), ("ours " + os.environ.get("LMOD_GEMM_WAVES", "8") + "-wave" + (" fused SwiGLU" if FUSED else ""), ours)]
with torch.no_grad():
    for rnd_ in range(2):
        for name, f in arms:
            for _ in range(20): f()
            torch.cuda.synchronize()
            s = Sampler(cards); s.start()
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < 2.0:
                for _ in range(20): f()
                torch.cuda.synchronize(); n += 20
            dt = time.perf_counter() - t0
            s.stop = True; s.join()
            rows = s.rows[len(s.rows) // 5:]
            # the card under test is the one drawing the most power
            best = max(range(len(cards)), key=lambda c: sum(r[2 * c] for r in rows))
            W = sum(r[2 * best] for r in rows) / len(rows); MHz = sum(r[2 * best + 1] for r in rows) / len(rows)
            print(f"{name}: {2.0 * M * N * Kd * n / dt / 1e12:.0f} TF  {W:.0f} W  {MHz:.0f} MHz  (card {best} of {len(cards)}, {len(rows)} samples)", flush=True)
