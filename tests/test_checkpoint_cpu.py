"""Checkpoint I/O in the reference's key layout (host logic, no GPU): round trip through HF-style shards, the separate
projector file, dense -> up-cycled loading, key dialects."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "llava-mod_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle.decoder import DecoderConfig  # noqa: E402
from oracle.vision import VisionConfig  # noqa: E402
from tests import _util as U  # noqa: E402


def _tiny(moe, seed):
    from llavamod.model import LLaVAMoDQwen2ForCausalLM, LlavaQwen2ForCausalLM
    vc = VisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, image_size=28,
                      patch_size=14, select_layer=-2)
    dc = DecoderConfig(vocab_size=96, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                       num_key_value_heads=1, moe_layers_idx=[0], num_experts=4, top_k_experts=2, capacity_factor=1.5,
                       min_capacity=0)
    cfg, _ = U.hip_configs(dc, vc, moe=moe)
    cfg.init_seed = seed
    model = (LLaVAMoDQwen2ForCausalLM if moe else LlavaQwen2ForCausalLM)(cfg, device="cpu")
    return model, dc


def test_roundtrip_sharded_and_projector_file(tmp_path):
    from llavamod.checkpoint import INDEX, load_checkpoint, read_state, save_checkpoint
    a, dc = _tiny(True, 1)
    a.initialize_moe_modules(U.moe_args(dc))
    with torch.no_grad():                               # make the experts differ from each other
        for i, (n, p) in enumerate(a.named_parameters()):
            p.add_(0.01 * (i % 7))
    files = save_checkpoint(a, str(tmp_path), max_shard_bytes=64 << 10)
    assert INDEX in files and "mm_projector.bin" in files and sum(f.endswith(".safetensors") for f in files) > 1
    keys = set(read_state(str(tmp_path)))
    assert "model.layers.0.mlp.deepspeed_moe.gate.wg.weight" in keys
    assert "model.layers.0.mlp.deepspeed_moe.experts.deepspeed_experts.3.down_proj.weight" in keys
    assert "model.layers.1.mlp.up_proj.weight" in keys and "lm_head.weight" in keys
    assert any(k.startswith("model.image_tower.image_tower.vision_model.") for k in keys)
    proj = torch.load(os.path.join(tmp_path, "mm_projector.bin"))
    assert set(proj) == {k for k in keys if "mm_projector" in k} and len(proj) == 4
    b, _ = _tiny(True, 2)
    b.initialize_moe_modules(U.moe_args(dc))
    assert load_checkpoint(b, str(tmp_path)) == ([], [])
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    # fused storage stays coherent: the q/k/v parameters are still views of one buffer after loading
    att = b.get_model().layers[0].self_attn
    att._qkv.ensure()
    assert att.q_proj.weight.data_ptr() == att._qkv.w.data_ptr()


def test_dense_checkpoint_upcycles_and_key_dialects(tmp_path):
    from safetensors.torch import save_file
    from llavamod.checkpoint import load_checkpoint, save_checkpoint
    dense, dc = _tiny(True, 3)                          # same class BEFORE initialize_moe_modules: dense FFN keys
    save_checkpoint(dense, str(tmp_path / "dense"), projector_file=False)
    moe, _ = _tiny(True, 4)
    moe.initialize_moe_modules(U.moe_args(dc))
    missing, unexpected = load_checkpoint(moe, str(tmp_path / "dense"), strict=False)
    assert missing == ["model.layers.0.mlp.deepspeed_moe.gate.wg.weight"] and unexpected == []
    ex = moe.get_model().layers[0].mlp.deepspeed_moe.experts.deepspeed_experts
    ref = dense.get_model().layers[0].mlp
    for e in ex:
        assert torch.equal(e.gate_proj.weight, ref.gate_proj.weight) and torch.equal(e.down_proj.weight, ref.down_proj.weight)
    # transformers-5.x CLIP keys (no `vision_model.` level) and DeepSpeed's `module.` prefix are accepted
    sd = {("module." + k).replace("image_tower.image_tower.vision_model.", "image_tower.image_tower."): v.contiguous()
          for k, v in dense.state_dict().items()}
    save_file(sd, str(tmp_path / "dialect.safetensors"))
    other, _ = _tiny(True, 5)
    assert load_checkpoint(other, str(tmp_path / "dialect.safetensors")) == ([], [])
    assert torch.equal(other.lm_head.weight, dense.lm_head.weight)
    with pytest.raises(KeyError):
        load_checkpoint(moe, str(tmp_path / "dense"), strict=True)
    bad = {k: (v[:1].contiguous() if k == "lm_head.weight" else v.contiguous()) for k, v in dense.state_dict().items()}
    save_file(bad, str(tmp_path / "bad.safetensors"))
    with pytest.raises(ValueError):
        load_checkpoint(other, str(tmp_path / "bad.safetensors"))
