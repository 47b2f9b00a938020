"""Per-layer routing state of config 5's student (8 experts, top-2, C 768) on the full-depth test batch, with the grouped launches on the
persistent kernel and off: slots_used per MoE layer and whether the hidden states after the decoder agree bit for bit."""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from llavamod.model import LLaVAMoDQwen2ForCausalLM  # noqa: E402

student = LLaVAMoDQwen2ForCausalLM(bench.student_cfg(8), device="cuda")
student.initialize_moe_modules(bench.moe_model_args(8))
batch = bench.synthetic_batch(1, seed=11)
batch = {k: (v.to("cuda") if torch.is_tensor(v) else v) for k, v in batch.items()}
for m in student.moe_layers():
    m.deterministic = True
student.train()
outs = {}
for arm in ("0", "1", "2"):
    os.environ["LMOD_GEMM_PERSIST_GROUPED"] = arm
    with torch.no_grad():
        hidden, _, _ = student.forward_hidden(**batch)
    outs[arm] = hidden.clone()
    print("PERSIST_GROUPED", arm, "finite", bool(torch.isfinite(hidden.float()).all()))
    for i, m in enumerate(student.moe_layers()):
        st = m.last_state
        print("  layer", i, "exp_counts", st.exp_counts.tolist(), "slots_used", st.slots_used.tolist(), "sum", int(st.slots_used.sum()))
print("hidden equal 0 vs 1:", torch.equal(outs["0"], outs["1"]), " 0 vs 2:", torch.equal(outs["0"], outs["2"]))
