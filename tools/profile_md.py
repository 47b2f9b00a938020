"""Turn rocprofv3 outputs under gpurun_out/<run>/ into the markdown summaries committed under profiles/.
usage: python tools/profile_md.py gpurun_out/final profiles/r01_final"""
import collections, csv, glob, json, os, re, sys

src, dst = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("", "")
TAG = os.path.basename(dst).replace("_", " ")


LAYERS_QKV = 24 + 32          # decoder layers of the config-2 student + teacher: one fused QKV + RoPE launch each per micro-batch


def steps_in_trace(qkv_rope_launches, grad_accum, layers=LAYERS_QKV):
    """Optimizer steps a kernel trace holds, from its own `gemm4_kernel<5>` (fused QKV + RoPE) launch count."""
    per_step = layers * grad_accum
    return qkv_rope_launches / per_step if per_step and qkv_rope_launches else 0


def gemm_algorithmic_bytes(M, N, K, elem=2):
    """A + B + C once (the same formula bench.py prints)."""
    return (M * K + N * K + M * N) * elem


def stats_md():
    rows = list(csv.DictReader(open(f"{src}/bench/a_kernel_stats.csv")))
    trace = list(csv.DictReader(open(f"{src}/bench/a_kernel_trace.csv")))
    line = json.loads(open(f"{src}/bench_line.json").read())
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    # optimizer steps actually inside the trace: timed + warm-up + bench.py's extra in-step-aggregate step (+ anything else that
    # ran the models) — counted from the trace itself instead of trusted from argv: every student layer launches the fused
    # QKV + RoPE GEMM (`gemm4_kernel<5`) once per micro-batch, and so does every teacher layer
    A = line["config"].get("grad_accum", 1)
    qkv_calls = sum(1 for r in trace if "gemm4_kernel<5" in r["Kernel_Name"])
    n_steps = steps_in_trace(qkv_calls, A) or (line["steps"] + line["warmup"])
    out = [f"# rocprofv3 --kernel-trace --stats of the default bench workload ({TAG})", "",
           "Command (MI355X box): `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 3 --warmup 1 "
           "--no-cpu-baseline --no-extras --no-teacher-prefetch` (teacher pipelining off so that kernel durations do not overlap; the "
           "default bench line with pipelining on is in `*_bench_n1.json`).", "",
           f"{n_steps:g} optimizer steps in the trace ({line['warmup']} warm-up + {line['steps']} timed + the in-step-aggregate step; counted from the "
           f"{qkv_calls} fused QKV + RoPE launches = {LAYERS_QKV} layers x {A} micro-batches per step) of {A} micro-batches x "
           f"{line['config']['micro_batch_per_gpu']} samples, config 2.  Sum of kernel "
           f"time {tot / 1e6:.1f} ms = {tot / 1e6 / n_steps:.1f} ms/step; bench wall clock under the profiler "
           f"{line['ms_per_step']} ms/step ({line['value']} samples/s).  Model construction is inside the trace (torch "
           "`distribution_elementwise` / `copyBuffer` rows).", "",
           "| kernel | calls | total ms | % | avg us | min us | max us |", "|---|---|---|---|---|---|---|"]
    for r in rows[:32]:
        out.append(f"| `{r['Name'][:72]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['Percentage']):.1f} | "
                   f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} |")
    # the bench's live-timed launches: gemm_256_kernel<0> with the teacher-QKV grid (128 x 48 tiles)
    gm = line["config"]["micro_batch_per_gpu"] * 2048
    blocks = (gm // 256) * (12288 // 256)
    # (round 3: plain bf16 launches run on the 4-wave kernel, 256 threads per workgroup; older builds / LMOD_GEMM_WAVES=8: 512)
    rl = line["roofline"]
    # round 6: WHICH kernel bench.py times live follows the committed kernel summary (bench.roofline_recipe): the instantiation number is
    # in the label (`gemm4_kernel<1> (fused SwiGLU forward ...`), its persistent form is `gemm4_kernel<N, true, false>` in the trace
    mnum = re.match(r"gemm4_kernel<(\d+)>", rl.get("kernel", ""))
    dnum = mnum.group(1) if mnum else "7"
    if any(f"gemm4_kernel<{dnum}, true, false" in r["Kernel_Name"] for r in trace):
        # round 4: the teacher-QKV launch (24 rounds of the CUs) runs on the PERSISTENT instantiation, whose grid is one workgroup
        # per CU whatever the shape: the bench's own launches are told apart by their duration (+-12 % of the live figure)
        dom = f"gemm4_kernel<{dnum}, true, false"
        # (with the persistent form taken from 4 rounds up several shapes of the step run on this instantiation with the same grid
        # and similar durations: the bench's own launches are the one place where the kernel is dispatched back to back, so they
        # are the LONGEST RUN of consecutive dispatches of it in start order)
        seq = sorted(trace, key=lambda r: int(r["Start_Timestamp"]))
        best, cur = [], []
        for r in seq:
            if dom in r["Kernel_Name"]:
                cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            else:
                if len(cur) > len(best):
                    best = cur
                cur = []
        if len(cur) > len(best):
            best = cur
        d = best
    else:
        dom = "gemm4_kernel<7>" if any("gemm4_kernel<7>" in r["Kernel_Name"] for r in trace) else "gemm_256_kernel<0"
        tpw = 256 if dom.startswith("gemm4") else 512
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trace
             if dom in r["Kernel_Name"] and int(r["Grid_Size_X"]) // tpw == blocks]
    shape_txt = rl.get("kernel", "").split("@")[-1].strip() if "@" in rl.get("kernel", "") else f"teacher QKV [{gm} x 12288 x 4096]"
    out += ["", "## Dominant kernel cross-check", "",
            f"`{dom}{'>' if not dom.endswith('0') else '>'}`: the longest run of back-to-back dispatches in the trace = the launches bench.py times with HIP events "
            f"({shape_txt}): {len(d)} launches (2 untimed + 10 timed), average {sum(d) / len(d):.1f} us, "
            f"min {min(d):.1f}, max {max(d):.1f}.  bench.py's live HIP-event figure in the same run: {rl['launch_ms'] * 1e3:.1f} us "
            f"per launch = {rl['achieved']} TFLOP/s ({rl['frac'] * 100:.1f} % of the 2.5 PFLOP/s bf16 MFMA peak).  {rl.get('dominant_by', '')}"]
    by = collections.defaultdict(lambda: [0, 0.0])
    for r in trace:
        if "gemm_256_kernel" in r["Kernel_Name"] or "gemm4_kernel" in r["Kernel_Name"]:
            k = (r["Kernel_Name"][5:30].split("(")[0], int(r["Grid_Size_X"]) // (256 if "gemm4_kernel" in r["Kernel_Name"] else 512))
            by[k][0] += 1; by[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    ins = rl.get("in_step") or {}
    fams = ins.get("families") or {}
    if fams:
        out += ["", "## The MFMA kernels inside the step (bench.py's in-step aggregate: one extra step, HIP events around every launch)", "",
                "| family (entry point) | launches | ms per step | TFLOP/s (launches with a host-known flop count) | of 2.5 PF |", "|---|---|---|---|---|"]
        for k, v in fams.items():
            out.append(f"| {k} | {v['launches']} | {v['ms']} | {v['achieved']} | {v['frac']} |")
        out += ["", "Top shapes per family:", ""]
        for k, v in fams.items():
            out.append(f"* {k}: " + "; ".join(f"`{sk}` x{sv['launches']} {sv['ms']} ms" + (f" {sv['tflops']} TF" if sv.get('tflops') else "") for sk, sv in v["by_shape"].items()))
    out += ["", "## 256-tile GEMM launches by grid size (workgroups; 256 CUs => `waves` rounds)", "",
            "| kernel | workgroups | rounds | calls | total ms |", "|---|---|---|---|---|"]
    for (k, nb), (n, ms) in sorted(by.items(), key=lambda t: -t[1][1])[:16]:
        out.append(f"| `{k}` | {nb} | {nb / 256:.2f} | {n} | {ms:.1f} |")
    open(dst + "_bench_kernel_stats.md", "w").write("\n".join(out) + "\n")
    dflt = [l for l in open(f"{src}/bench_default.json") if l.startswith("{")]
    json.dump(json.loads(dflt[-1]) if dflt else line, open(dst + "_bench_n1.json", "w"), indent=1)


def pmc(run):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob(f"{src}/{run}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{src}/{run}/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: ({c: v / len(disp[k]) for c, v in cs.items()}, sum(dur[k]) / max(1, len(dur[k]))) for k, cs in agg.items()}


def pmc_md():
    out = [f"# PMC counters (rocprofv3 --pmc, separate passes, --kernel-trace only) — {TAG}", "",
           "Per-dispatch averages.  `SQ_WAVE_CYCLES`, `SQ_WAIT_*`, `SQ_ACTIVE_INST_*` count quad-cycles; "
           "`SQ_VALU_MFMA_BUSY_CYCLES` counts cycles summed over the 1024 SIMDs; `GRBM_GUI_ACTIVE` is summed over the 8 XCDs.", ""]
    g = pmc("gemm_sq"); f = pmc("gemm_fetch"); w = pmc("gemm_write")
    k = next(x for x in g if "gemm4_kernel" in x or "gemm_256" in x)      # whichever kernel tools/gemm_one.py launched
    kname = k.replace("void ", "").split("(")[0]
    c, us = g[k]
    fetch_kb, write_kb = f[k][0]["FETCH_SIZE"], w[k][0]["WRITE_SIZE"]
    meta = json.load(open(f"{src}/gemm_one_meta.json")) if os.path.exists(f"{src}/gemm_one_meta.json") else {"kind": "nt", "shape": [32768, 12288, 4096]}
    M, N, Kd = meta["shape"]
    out_cols = meta.get("output_cols", N)                 # fused SwiGLU forward: the GEMM is N = 2I wide, the result I
    algo_nt = (M * Kd + N * Kd + M * out_cols) * 2
    clk = c["GRBM_GUI_ACTIVE"] / 8 / us / 1e3
    out += [f"## `{kname}` at {meta.get('what', 'the teacher QKV shape')} [{M} x {N} x {Kd}] (`{meta.get('cmd', 'python tools/gemm_one.py')}`)", "",
            f"* duration under the counter passes: {us:.0f} us ({2.0 * M * N * Kd / us / 1e6:.0f} TFLOP/s; counter collection and its lower "
            f"clock cost ~10 % against the un-profiled {2.0 * M * N * Kd / 1e12:.2f} TFLOP launch in bench.py)",
            f"* effective clock: GRBM_GUI_ACTIVE / 8 / duration = **{clk:.2f} GHz** (peak 2.4): the chip is power-limited under this kernel",
            f"* MFMA pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8) = **{c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (c['GRBM_GUI_ACTIVE'] / 8) * 100:.1f} %** of the cycles the chip actually ran",
            f"* LDS: SQ_LDS_BANK_CONFLICT = {c['SQ_LDS_BANK_CONFLICT']:.0f} (conflict-free swizzle), SQ_LDS_IDX_ACTIVE / 256 CUs = "
            f"{c['SQ_LDS_IDX_ACTIVE'] / 256 / (c['GRBM_GUI_ACTIVE'] / 8) * 100:.0f} % of cycles",
            f"* memory side: FETCH_SIZE {fetch_kb / 1e6:.3f} GB x 2 (gfx950 16-byte-lane correction, MI355X_MICROARCH.md §HBM) = "
            f"**{2 * fetch_kb / 1e6:.2f} GB**, WRITE_SIZE **{write_kb / 1e6:.2f} GB** per launch; algorithmic bytes (operands + result once) "
            f"{algo_nt / 1e9:.2f} GB => {(2 * fetch_kb * 1e3 + write_kb * 1e3) / algo_nt:.2f}x.  The fetch figure counts every L2 miss at the fabric, Infinity-Cache hits included: an XCD's 32 "
            f"concurrent tiles (256 operand rows on each side) form a 4 x 8 rectangle that needs 12 operand panels of 256 x {Kd} bf16 per K sweep; "
            f"{-(-(M // 256) * (N // 256) // 256)} sweeps of the 8 XCDs => ~{-(-(M // 256) * (N // 256) // 256) * 8 * 12 * 256 * Kd * 2 / 1e9:.1f} GB by construction; the 256 MB "
            "Infinity Cache absorbs the re-reads, so this is fabric traffic under an MFMA-bound kernel, not HBM over-fetch.",
            "", "| counter | per dispatch |", "|---|---|"]
    out += [f"| {a} | {b:.4g} |" for a, b in sorted(c.items())]
    if glob.glob(f"{src}/wgrad_sq/*counter_collection.csv"):
        t = pmc("wgrad_sq"); tf = pmc("wgrad_fetch"); tw = pmc("wgrad_write")
        tk = next(x for x in t if "gemm4t_kernel" in x)
        c, us = t[tk]
        live = 9000 + 12500 + 20036 + 24000
        flop = 2.0 * live * 11008 * 2048
        algo_tn = (live * (11008 + 2048) * 2 + 4 * 11008 * 2048 * 8)
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        out += ["", "## `gemm4t_kernel<false>` (weight gradients on reduction-major operands) at the MoE gate+up shape: 4 experts x [11008 x 2048], "
                f"{live} live rows (`python tools/wgrad_one.py`)", "",
                f"* duration under the counter passes: {us:.0f} us ({flop / us / 1e6:.0f} TFLOP/s); effective clock **{cyc / us / 1e3:.2f} GHz**",
                f"* MFMA pipe busy: **{c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc * 100:.1f} %** of the cycles the chip actually ran",
                f"* LDS: SQ_LDS_BANK_CONFLICT = {c['SQ_LDS_BANK_CONFLICT']:.0f} against SQ_LDS_IDX_ACTIVE = {c['SQ_LDS_IDX_ACTIVE']:.4g} "
                f"(the padded, XOR-free image; `tools/probe/gemm4t_layout.py` predicts zero), LDS active {c['SQ_LDS_IDX_ACTIVE'] / 256 / cyc * 100:.0f} % of CU cycles "
                "(twice the instructions of the NT kernel's `ds_read_b128` for the same bytes)",
                f"* memory side: FETCH_SIZE x 2 = **{2 * tf[tk][0]['FETCH_SIZE'] / 1e6:.2f} GB**, WRITE_SIZE **{tw[tk][0]['WRITE_SIZE'] / 1e6:.2f} GB** per launch; "
                f"algorithmic bytes (live rows of dY and X once, fp32 C read + written) {algo_tn / 1e9:.2f} GB",
                "", "| counter | per dispatch |", "|---|---|"]
        out += [f"| {a_} | {b_:.4g} |" for a_, b_ in sorted(c.items())]
    a = pmc("attn_sq")
    out += ["", "## attention kernels, B 16, S 2048, 16 heads, hd 128, causal (`A1_B=16 python tools/attn_one.py bwd`; first launches, "
            "clock not yet settled: durations are longer than in the micro-benchmarks)", "",
            "| kernel | us | clock GHz | MFMA busy % | LDS active % of CU cycles | LDS conflict / active | WAIT_ANY / WAVE_CYCLES |", "|---|---|---|---|---|---|---|"]
    for k, (c, us) in a.items():
        if ("attn" not in k and "gemm4t_kernel<2" not in k) or "delta" in k:       # (gemm4t_kernel<2>: dQ from the spilled dS, round 6)
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        out.append(f"| `{k}` | {us:.0f} | {cyc / us / 1e3:.2f} | {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc * 100:.0f} | "
                   f"{c['SQ_LDS_IDX_ACTIVE'] / 256 / cyc * 100:.0f} | {c['SQ_LDS_BANK_CONFLICT'] / max(1, c['SQ_LDS_IDX_ACTIVE']):.2f} | "
                   f"{c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f} |")
    ai = pmc("attn_inst") if glob.glob(f"{src}/attn_inst/*counter_collection.csv") else {}
    if ai:
        out += ["", "Instruction mix per dispatch (millions of wave-instructions; ACTIVE / WAIT in millions of quad-cycles):", "",
                "| kernel | MFMA | VALU | LDS | SALU | VMEM rd | VALU per MFMA | ACTIVE_INST_VALU | ACTIVE_INST_LDS | WAIT_INST_LDS |", "|---|---|---|---|---|---|---|---|---|---|"]
        for k, (c, us) in ai.items():
            if ("attn" not in k and "gemm4t_kernel<2" not in k) or "delta" in k:
                continue
            out.append(f"| `{k}` | {c['SQ_INSTS_MFMA'] / 1e6:.2f} | {c['SQ_INSTS_VALU'] / 1e6:.2f} | {c['SQ_INSTS_LDS'] / 1e6:.2f} | "
                       f"{c['SQ_INSTS_SALU'] / 1e6:.2f} | {c['SQ_INSTS_VMEM_RD'] / 1e6:.2f} | {c['SQ_INSTS_VALU'] / max(1.0, c['SQ_INSTS_MFMA']):.2f} | "
                       f"{c['SQ_ACTIVE_INST_VALU'] / 1e6:.1f} | {c['SQ_ACTIVE_INST_LDS'] / 1e6:.1f} | {c['SQ_WAIT_INST_LDS'] / 1e6:.1f} |")
    open(dst + "_pmc.md", "w").write("\n".join(out) + "\n")
    json.dump({"kernel": kname, "shape": [M, N, Kd], **({"output_cols": out_cols} if out_cols != N else {}), "fetch_bytes_corrected": 2 * fetch_kb * 1e3,
               "write_bytes": write_kb * 1e3, "algorithmic_bytes": algo_nt,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH x2 per MI355X_MICROARCH.md §HBM"},
              open(dst + "_gemm_traffic.json", "w"), indent=1)


if __name__ == "__main__":
    stats_md()
    pmc_md()
    os.system(f"cp {src}/kernel_microbench.jsonl {dst}_kernel_microbench.jsonl; grep -h weighted8 {src}/gemm_shapes.json > {dst}_gemm_shapes.json")
    os.system(f"cp {src}/bench/a_kernel_stats.csv {dst}_bench_kernel_stats.csv")
    if os.path.exists(f"{src}/attn_bench.jsonl"):
        os.system(f"cp {src}/attn_bench.jsonl {dst}_attn_bench.jsonl")
    for name in ("bench_driver_form.json", "routing.jsonl", "pytest_gpu.txt", "attn_hd64.jsonl", "attn_bwd_forms.jsonl"):
        if os.path.exists(f"{src}/{name}"):
            os.system(f"cp {src}/{name} {dst}_{name}")
