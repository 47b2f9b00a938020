"""LlavaQwen2ForCausalLM — the dense teacher API (reference language_model/llava_qwen2.py:31-130, which wraps
the vendored Qwen2ForCausalLM, qwen2/modeling_qwen2.py:1129-1190).  Same constructor-from-config,
`get_model()`, `forward(input_ids, attention_mask, …, labels, images, return_dict)` and output fields
(`loss`, `logits` fp32 [B,S',V], `labels` after the image splice); execution is on the HIP kernels.
`LlavaQwen1_5ForCausalLM` (llava_qwen1_5.py) is the same class: the two files differ in names only.
"""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from ... import kernels as K
from ... import ops
from ...constants import IGNORE_INDEX
from ...ops import FusedWeight
from ..llava_arch import LlavaMetaForCausalLM, LlavaMetaModel
from ..utils import CausalLMOutputWithPast
from .qwen2_hip import Qwen2Config, Qwen2Model, _Linear, init_normal_

BF16 = torch.bfloat16


class LlavaQwen2Config(Qwen2Config):
    model_type = "llava_qwen2"


class LlavaQwen2Model(LlavaMetaModel, Qwen2Model):
    config_class = LlavaQwen2Config

    def __init__(self, config, device="cuda"):
        Qwen2Model.__init__(self, config, device)
        self._init_vision(config, device)


def build_loss_plan(labels_np, lens_np, kd_rows=True, ce_rows=True, distill_all_tokens=False, align_vocab=None,
                    device="cuda"):
    """Which rows of the flattened [B*S'] hidden states carry loss, in sample-major order.
      KD row t  : labels[b,t] != -100          (UNshifted mask, align_trainer.py:522; or every row incl. pads
                                               when distill_all_tokens, :516-520)
      CE row t  : labels[b,t+1] != -100        (shifted LM / DPO rows, llava_qwen2_moe.py:413-421, dpo_trainer.py:483-485)
    """
    B, S = labels_np.shape
    valid = labels_np != IGNORE_INDEX
    kd = (np.ones_like(valid) if distill_all_tokens else valid) if kd_rows else np.zeros_like(valid)
    ce = np.zeros_like(valid)
    if ce_rows:
        ce[:, :-1] = valid[:, 1:]
    ce_label = np.full((B, S), -1, dtype=np.int32)
    ce_label[:, :-1] = np.where(valid[:, 1:], labels_np[:, 1:], -1)
    need = kd | ce
    bi, ti = np.nonzero(need)                                   # row-major => sample-major order
    rows = (bi * S + ti).astype(np.int32)
    seg_off = np.zeros(B + 1, dtype=np.int32)
    np.cumsum(np.bincount(bi, minlength=B), out=seg_off[1:])
    inv = np.full(B * S, -1, dtype=np.int32)
    inv[rows] = np.arange(len(rows), dtype=np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return SimpleNamespace(R=len(rows), row_idx=t(rows), inv_row_idx=t(inv), kd_w=t(kd[bi, ti].astype(np.float32)),
                           ce_w=t(ce[bi, ti].astype(np.float32)), ce_label=t(ce_label[bi, ti]),
                           seg_off=t(seg_off), seg_id=t(bi.astype(np.int32)), align_vocab=align_vocab, shape=(B, S),
                           labels_np=labels_np,
                           n_kd=int(kd.sum()), n_ce=int(ce.sum()))


class _CausalLMBase(nn.Module, LlavaMetaForCausalLM):
    """Shared forward machinery of the dense teacher and the MoE student."""

    def _setup_head(self, config, device):
        self.vocab_size = config.vocab_size
        self.lm_head = _Linear(config.hidden_size, config.vocab_size, False, device)
        self._head = FusedWeight([[self.lm_head.weight]])

    def get_model(self):
        return self.model

    def head(self):
        return self._head.ensure()

    # ---- internal fast path -------------------------------------------------------------------
    def forward_hidden(self, input_ids=None, attention_mask=None, labels=None, images=None, inputs_embeds=None,
                       plan_fn=None):
        """Splice + decoder.  Returns (hidden [B*S', H], moe_loss_list, info) — no logits.
        plan_fn(info) -> loss plan: the decoder then returns just the plan's rows ([R, H], `info.plan.pregathered`),
        letting a dense last layer skip (forward and backward) the rows nobody reads."""
        if inputs_embeds is None:
            _, _, attention_mask, _, inputs_embeds, labels = self.prepare_inputs_labels_for_multimodal(
                input_ids, None, attention_mask, None, labels, images)
            plan = self._plan
            if inputs_embeds is None:                          # text-only batch
                dev = self.model.embed_tokens.weight.device
                B, S = input_ids.shape
                idx = input_ids.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
                inputs_embeds = K.gather_rows(self.model.embed_tokens.weight, None, idx,
                                              self.model.embed_tokens.weight.shape[1]).view(B, S, -1)
        else:
            plan = None
        B, S, H = inputs_embeds.shape
        seqlens = None
        if plan is not None:
            seqlens = plan.seqlens
        elif attention_mask is not None:
            lens = attention_mask.to(torch.int32).sum(1)
            if bool((lens != S).any()):
                seqlens = lens.to(device=inputs_embeds.device, dtype=torch.int32).contiguous()
        if labels is not None:
            labels_np = plan.labels_np if plan is not None else labels.detach().cpu().numpy()
        else:
            labels_np = None
        lens_np = plan.lens_np if plan is not None else None
        info = SimpleNamespace(B=B, S=S, labels=labels, labels_np=labels_np, lens_np=lens_np,
                               attention_mask=attention_mask, plan=None)
        out_rows = inv_rows = None
        if plan_fn is not None and labels_np is not None:
            info.plan = plan_fn(info)
            info.plan.pregathered = True
            out_rows, inv_rows = info.plan.row_idx, info.plan.inv_row_idx
        hidden, moe_list = self.model(inputs_embeds.reshape(B * S, H), B, S, seqlens, out_rows=out_rows,
                                      inv_rows=inv_rows)
        return hidden, moe_list, info

    def lm_loss_from_hidden(self, hidden, info):
        """Shifted CrossEntropyLoss() of the reference forward (mean over non-ignored), loss rows only."""
        plan = build_loss_plan(info.labels_np, info.lens_np, kd_rows=False, ce_rows=True, device=hidden.device)
        _, _, ce_sum, ce_cnt = ops.DistillHead.apply(hidden, self.head(), plan, None, *self._head_trainable())
        return ce_sum.sum() / ce_cnt.sum()

    def _head_trainable(self):
        return [self.lm_head.weight] if self.lm_head.weight.requires_grad else []

    def full_logits(self, hidden, B, S):
        lg = ops.Linear.apply(hidden, self.head(), *self._head_trainable())
        return lg.view(B, S, -1).float()                        # `logits = logits.float()` (modeling_qwen2.py:1164)


class LlavaQwen2ForCausalLM(_CausalLMBase):
    config_class = LlavaQwen2Config

    def __init__(self, config, device="cuda"):
        super().__init__()
        self.config = config
        self.model = LlavaQwen2Model(config, device)
        self._setup_head(config, device)
        self._plan = None
        init_normal_(self, getattr(config, "initializer_range", 0.02), getattr(config, "init_seed", 0))   # post_init()

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                images=None, return_dict=None):
        hidden, _, info = self.forward_hidden(input_ids, attention_mask, labels, images, inputs_embeds)
        logits = self.full_logits(hidden, info.B, info.S)
        loss = self.lm_loss_from_hidden(hidden, info) if info.labels is not None else None
        return CausalLMOutputWithPast(loss=loss, logits=logits, labels=info.labels)


LlavaQwen1_5Config = LlavaQwen2Config
LlavaQwen1_5ForCausalLM = LlavaQwen2ForCausalLM
