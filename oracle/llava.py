"""Oracle: LLaVA wrapper (vision tower + projector + splice + decoder) and the two training steps.
TEST INFRASTRUCTURE ONLY.

  LlavaQwen2ForCausalLM.forward (teacher)        language_model/llava_qwen2.py:57-108
  LLaVAMoDQwen2ForCausalLM.forward (student)     language_model/llava_qwen2_moe.py:357-451
  LlavaMetaForCausalLM.encode_images             llava_arch.py:143-148
  AlignTrainer.compute_loss / DPOTrainer.compute_loss   see oracle/losses.py
"""
import copy

import torch
import torch.nn as nn

from . import losses
from .decoder import CausalLM, DecoderConfig
from .vision import Projector, VisionConfig, VisionTower, splice


class LlavaOracle(nn.Module):
    """state_dict keys follow the reference: model.{embed_tokens,layers,norm,image_tower.image_tower.vision_model,
    mm_projector.image_spatial_proj}, lm_head."""

    def __init__(self, cfg: DecoderConfig, vcfg: VisionConfig, moe: bool):
        super().__init__()
        self.cfg, self.vcfg = cfg, vcfg
        self.lm = CausalLM(cfg)
        self.image_tower = VisionTower(vcfg)
        self.mm_projector = Projector(vcfg.hidden_size, cfg.hidden_size)
        if moe:
            self.lm.upcycle()

    def encode_images(self, images):                          # llava_arch.py:143-148
        return self.mm_projector.forward_image(self.image_tower(images))

    def forward(self, input_ids, attention_mask=None, labels=None, images=None):
        if images is not None:
            imgs = torch.stack(list(images)) if isinstance(images, (list, tuple)) else images
            feats = self.encode_images(imgs.to(self.lm.lm_head.weight.dtype))
            embeds, attention_mask, labels = splice(self.lm.model.embed_tokens, feats, input_ids, attention_mask, labels)
        else:
            embeds = self.lm.model.embed_tokens(input_ids)
        return self.lm.forward_embeds(embeds, attention_mask, labels)

    def set_gate_noise(self, noises):
        """noises: list (one per MoE layer, in layer order) of [T,E] tensors or None."""
        mo = [l.mlp for l in self.lm.model.layers if hasattr(l.mlp, "deepspeed_moe")]
        for m, n in zip(mo, noises if noises is not None else [None] * len(mo)):
            m.noise = n


def hip_to_oracle_key(k):
    """State-dict key of the product models (= the reference's layout) -> key of `LlavaOracle` (inverse of the mapping the
    parity tests use to load oracle weights into the product models)."""
    if k.startswith("model.image_tower.image_tower."):
        return "image_tower." + k[len("model.image_tower.image_tower."):]
    if k.startswith("model.mm_projector."):
        return k[len("model."):]
    return "lm." + k


def load_from_product_state(oracle_model: "LlavaOracle", state):
    """Copy a product model's state dict (bf16 device tensors) into the fp32 oracle, key by key, strictly."""
    own = oracle_model.state_dict()
    mapped = {hip_to_oracle_key(k): v for k, v in state.items()}
    assert set(mapped) == set(own), sorted(set(mapped) ^ set(own))[:8]
    with torch.no_grad():
        for k, t in own.items():
            t.copy_(mapped[k].detach().to(device="cpu", dtype=t.dtype))
    return oracle_model


def freeze_like_d2s(student: LlavaOracle):
    """Trainable set of the dense-to-sparse stage: FFNs (dense + experts) + routers `wg`
    + mm_projector; everything else frozen (llava_qwen2_moe.py:501-506, llava_arch.py:115-120)."""
    keys = ("mlp.gate_proj", "mlp.up_proj", "mlp.down_proj", "wg", "deepspeed_experts", "mm_projector")
    for n, p in student.named_parameters():
        p.requires_grad_(any(k in n for k in keys) and "image_tower" not in n)
    return student


def mimic_step(student, teacher, batch, loss_type="only_kd", moe_loss_enable=True, align_vocab=losses.ALIGN_VOCAB,
               distill_all_tokens=False):
    """One AlignTrainer.training_step worth of math: teacher fwd (no_grad), student fwd, loss, backward.
    distill_all_tokens: `--distill_all_tokens` (config/args.py:110; align_trainer.py:516-520: the KD mask is all ones,
    padding rows included)."""
    with torch.no_grad():
        t_out = teacher(**batch)
    s_out = student(**batch)
    loss, logs = losses.mimic_loss(s_out, t_out, loss_type, moe_loss_enable, distill_all_tokens=distill_all_tokens,
                                   align_vocab=align_vocab)
    loss.backward()
    return loss.detach(), logs, s_out, t_out


def dpo_step(student, teacher, batch, beta=0.1, loss_type="sigmoid", moe_loss_enable=True):
    ch = dict(input_ids=batch["chosen_input_ids"], labels=batch["chosen_labels"],
              attention_mask=batch["chosen_attention_mask"], images=batch.get("images"))
    rj = dict(input_ids=batch["rejected_input_ids"], labels=batch["rejected_labels"],
              attention_mask=batch["rejected_attention_mask"], images=batch.get("images"))
    s_ch, s_rj = student(**ch), student(**rj)
    with torch.no_grad():
        t_ch, t_rj = teacher(**ch), teacher(**rj)
    loss, logs = losses.preference_loss(s_ch, s_rj, t_ch, t_rj, beta, 0.0, loss_type, moe_loss_enable)
    loss.backward()
    return loss.detach(), logs


def init_weights(module, seed, std=0.02):
    """N(0, std) for matrices/embeddings/biases, ones for norm scales (config 1 recipe)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if ("norm" in n or "layrnorm" in n) and n.endswith("weight"):
                p.fill_(1.0)
            elif ("norm" in n or "layrnorm" in n) and n.endswith("bias"):
                p.zero_()
            else:
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return module


def sync_experts_from_dense(student: LlavaOracle):
    """After init_weights the experts of each MoE layer must again be identical copies of one FFN
    (up-cycling invariant asserted at llava_qwen2_moe.py:547-550)."""
    for l in student.lm.model.layers:
        if hasattr(l.mlp, "deepspeed_moe"):
            ex = l.mlp.deepspeed_moe.experts.deepspeed_experts
            for e in ex[1:]:
                e.load_state_dict(copy.deepcopy(ex[0].state_dict()))
    return student
