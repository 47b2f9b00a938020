"""Shared helpers for the parity tests: build the HIP models from oracle state dicts / configs."""
import json
import os
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

D2S_TRAIN_MODULES = ["mlp.gate_proj", "mlp.up_proj", "mlp.down_proj", "wg", "deepspeed_experts"]


def load_golden(name):
    from safetensors.torch import load_file
    return load_file(os.path.join(GOLD, name))


def load_json(name):
    return json.load(open(os.path.join(GOLD, name)))


def oracle_to_hip_key(k):
    if k.startswith("lm."):
        return k[3:]
    if k.startswith("image_tower."):
        return "model.image_tower.image_tower." + k[len("image_tower."):]
    if k.startswith("mm_projector."):
        return "model." + k
    raise KeyError(k)


def hip_configs(dec_cfg, vis_cfg, moe):
    """oracle DecoderConfig / VisionConfig -> llavamod config objects."""
    from llavamod.model import CLIPVisionConfig, LLaVAMoDQwen2Config, LlavaQwen2Config
    vc = CLIPVisionConfig(hidden_size=vis_cfg.hidden_size, intermediate_size=vis_cfg.intermediate_size,
                          num_hidden_layers=vis_cfg.num_hidden_layers, num_attention_heads=vis_cfg.num_attention_heads,
                          image_size=vis_cfg.image_size, patch_size=vis_cfg.patch_size, layer_norm_eps=vis_cfg.layer_norm_eps)
    kw = dict(vocab_size=dec_cfg.vocab_size, hidden_size=dec_cfg.hidden_size, intermediate_size=dec_cfg.intermediate_size,
              num_hidden_layers=dec_cfg.num_hidden_layers, num_attention_heads=dec_cfg.num_attention_heads,
              num_key_value_heads=dec_cfg.num_key_value_heads, rms_norm_eps=dec_cfg.rms_norm_eps,
              rope_theta=dec_cfg.rope_theta, max_position_embeddings=dec_cfg.max_position_embeddings,
              mm_image_tower=vc, image_projector_type="mlp2x_gelu", mm_hidden_size=vis_cfg.hidden_size,
              mm_vision_select_layer=vis_cfg.select_layer, mm_vision_select_feature="patch")
    return (LLaVAMoDQwen2Config(**kw) if moe else LlavaQwen2Config(**kw)), vc


def moe_args(dec_cfg, train_modules=D2S_TRAIN_MODULES):
    return SimpleNamespace(moe_enable=True, train_modules=list(train_modules), moe_mode="custom",
                           moe_layers_idx=list(dec_cfg.moe_layers_idx), ep_size=1, top_k_experts=dec_cfg.top_k_experts,
                           capacity_factor=dec_cfg.capacity_factor, eval_capacity_factor=dec_cfg.eval_capacity_factor,
                           min_capacity=dec_cfg.min_capacity, use_residual=False,
                           router_aux_loss_coef=dec_cfg.router_aux_loss_coef, num_experts=[dec_cfg.num_experts])


def build_hip_pair(student_sd, teacher_sd, sc, tc, vc, device="cuda", margs=None):
    """HIP student (up-cycled MoE) + dense teacher carrying the oracle's weights (cast to bf16; router fp32).
    margs: the `model_args` handed to `initialize_moe_modules` (default: explicit layer indices from `sc`)."""
    from llavamod.model import LLaVAMoDQwen2ForCausalLM, LlavaQwen2ForCausalLM
    scfg, _ = hip_configs(sc, vc, moe=True)
    tcfg, _ = hip_configs(tc, vc, moe=False)
    student = LLaVAMoDQwen2ForCausalLM(scfg, device=device)
    student.initialize_moe_modules(margs if margs is not None else moe_args(sc))
    for p in student.get_model().mm_projector.parameters():       # initialize_vision_modules re-enables these
        p.requires_grad = True
    teacher = LlavaQwen2ForCausalLM(tcfg, device=device)
    for model, sd in ((student, student_sd), (teacher, teacher_sd)):
        mapped = {oracle_to_hip_key(k): v for k, v in sd.items()}
        own = model.state_dict()
        assert set(mapped) == set(own), (sorted(set(mapped) ^ set(own))[:10])
        with torch.no_grad():
            for k, t in own.items():
                t.copy_(mapped[k].to(device=t.device, dtype=t.dtype))
    return student, teacher


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
