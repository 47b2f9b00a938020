"""Oracle: Qwen2-style decoder in plain PyTorch (CPU, any float dtype).  TEST INFRASTRUCTURE ONLY.

Restates the vendored decoder the dense teacher runs and the HF Qwen2Model the MoE student
subclasses (arithmetic identical): /root/reference/llavamod/model/language_model/qwen2/modeling_qwen2.py
  RMSNorm :83-97, rotary :101-171, MLP :175-187, repeat_kv :191-200, eager attention :203-325,
  decoder layer :725-799, model loop :950-1094, CausalLM head + shifted CE :1129-1190;
MoE-aware layer/model/CausalLM forward: language_model/llava_qwen2_moe.py:112-181,184-339,357-451.
Parameter names follow HF so state dicts interchange with the reference and the product model.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class DecoderConfig:
    vocab_size: int = 512
    hidden_size: int = 64
    intermediate_size: int = 128
    num_hidden_layers: int = 2
    num_attention_heads: int = 4
    num_key_value_heads: int = 4
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    max_position_embeddings: int = 4096
    # MoE (student only); layout of the reference's config.moe dict (llava_qwen2_moe.py:63-78)
    moe_layers_idx: List[int] = field(default_factory=list)
    num_experts: int = 4
    top_k_experts: int = 2
    capacity_factor: float = 1.5
    eval_capacity_factor: float = 2.0
    min_capacity: int = 0
    use_residual: bool = False
    router_aux_loss_coef: float = 0.01

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads


class RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):                                    # modeling_qwen2.py:92-97
        dt = x.dtype
        x = x.to(torch.float32)
        var = x.pow(2).mean(-1, keepdim=True)
        x = x * torch.rsqrt(var + self.variance_epsilon)
        return self.weight * x.to(dt)


def rope_tables(head_dim, max_pos, theta, dtype):           # modeling_qwen2.py:101-134
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(max_pos, dtype=inv_freq.dtype)
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):                                         # modeling_qwen2.py:138-142
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin, position_ids):               # modeling_qwen2.py:146-171
    cos = cos[position_ids].unsqueeze(1)
    sin = sin[position_ids].unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def causal_padding_mask(attention_mask, B, S, dtype):
    """Additive [B,1,S,S] mask = causal + key padding (what _prepare_4d_causal_attention_mask builds,
    modeling_qwen2.py:1019-1036)."""
    neg = torch.finfo(dtype).min
    m = torch.full((S, S), neg, dtype=dtype).triu(1)[None, None].expand(B, 1, S, S).clone()
    if attention_mask is not None:
        pad = ~attention_mask.bool()
        m = m.masked_fill(pad[:, None, None, :], neg)
    return m


class MLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.gate_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=False)
        self.up_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=False)
        self.down_proj = nn.Linear(cfg.intermediate_size, cfg.hidden_size, bias=False)

    def forward(self, x):                                    # modeling_qwen2.py:186-187
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x))


class Attention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        hd = cfg.head_dim
        self.q_proj = nn.Linear(cfg.hidden_size, cfg.num_attention_heads * hd, bias=True)
        self.k_proj = nn.Linear(cfg.hidden_size, cfg.num_key_value_heads * hd, bias=True)
        self.v_proj = nn.Linear(cfg.hidden_size, cfg.num_key_value_heads * hd, bias=True)
        self.o_proj = nn.Linear(cfg.num_attention_heads * hd, cfg.hidden_size, bias=False)

    def forward(self, x, mask, position_ids, cos, sin):      # eager path, modeling_qwen2.py:240-325
        B, S, _ = x.shape
        c = self.cfg
        hd = c.head_dim
        q = self.q_proj(x).view(B, S, c.num_attention_heads, hd).transpose(1, 2)
        k = self.k_proj(x).view(B, S, c.num_key_value_heads, hd).transpose(1, 2)
        v = self.v_proj(x).view(B, S, c.num_key_value_heads, hd).transpose(1, 2)
        q, k = apply_rope(q, k, cos.to(x.dtype), sin.to(x.dtype), position_ids)
        rep = c.num_attention_heads // c.num_key_value_heads
        if rep > 1:                                          # repeat_kv :191-200
            k = k[:, :, None].expand(B, c.num_key_value_heads, rep, S, hd).reshape(B, c.num_attention_heads, S, hd)
            v = v[:, :, None].expand(B, c.num_key_value_heads, rep, S, hd).reshape(B, c.num_attention_heads, S, hd)
        w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(hd)
        w = w + mask
        w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)   # :307
        o = torch.matmul(w, v).transpose(1, 2).contiguous().reshape(B, S, c.num_attention_heads * hd)
        return self.o_proj(o)


class DecoderLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self_attn = Attention(cfg)
        self.mlp = MLP(cfg)
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)

    def forward(self, h, mask, position_ids, cos, sin):      # llava_qwen2_moe.py:143-179
        res = h
        h = self.input_layernorm(h)
        h = self.self_attn(h, mask, position_ids, cos, sin)
        h = res + h
        res = h
        h = self.post_attention_layernorm(h)
        h = self.mlp(h)
        moe_losses = []
        if isinstance(h, tuple) and len(h) == 3:             # deepspeed MoE returns (out, l_aux, exp_counts)
            moe_losses.append(h[1])
            h = h[0]
        return res + h, moe_losses


class DecoderModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.layers = nn.ModuleList([DecoderLayer(cfg) for _ in range(cfg.num_hidden_layers)])
        self.norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)

    def forward(self, inputs_embeds, attention_mask=None, position_ids=None):   # llava_qwen2_moe.py:198-338
        B, S, _ = inputs_embeds.shape
        if position_ids is None:
            position_ids = torch.arange(S, dtype=torch.long).unsqueeze(0)
        mask = causal_padding_mask(attention_mask, B, S, inputs_embeds.dtype)
        cos, sin = rope_tables(self.cfg.head_dim, max(self.cfg.max_position_embeddings, S), self.cfg.rope_theta,
                               inputs_embeds.dtype)
        h = inputs_embeds
        all_moe = []
        for layer in self.layers:
            h, ml = layer(h, mask, position_ids, cos, sin)
            all_moe.extend(ml)
        return self.norm(h), all_moe


@dataclass
class CausalLMOut:
    loss: Optional[torch.Tensor]
    logits: torch.Tensor
    labels: Optional[torch.Tensor]
    moe_loss: Optional[torch.Tensor] = None
    moe_loss_list: Optional[list] = None


class CausalLM(nn.Module):
    """Dense teacher (no MoE layers) or MoE student (after `upcycle`)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.model = DecoderModel(cfg)
        self.lm_head = nn.Linear(cfg.hidden_size, cfg.vocab_size, bias=False)
        self.router_aux_loss_coef = cfg.router_aux_loss_coef

    def upcycle(self):
        """initialize_moe_modules (llava_qwen2_moe.py:475-561): every listed layer's dense FFN becomes a
        MoE whose experts are copies of it."""
        from .moe import OracleMoE
        c = self.cfg
        for li in c.moe_layers_idx:
            dense = self.model.layers[li].mlp
            self.model.layers[li].mlp = OracleMoE(c.hidden_size, dense, c.num_experts, 1, c.top_k_experts,
                                                  c.capacity_factor, c.eval_capacity_factor, c.min_capacity,
                                                  c.use_residual)
        return self

    def forward_embeds(self, inputs_embeds, attention_mask=None, labels=None):
        h, moe_list = self.model(inputs_embeds, attention_mask)
        logits = self.lm_head(h).float()                     # llava_qwen2_moe.py:407-408
        loss = None
        if labels is not None:                               # :411-421
            sl = logits[..., :-1, :].contiguous().view(-1, self.cfg.vocab_size)
            tl = labels[..., 1:].contiguous().view(-1)
            loss = F.cross_entropy(sl, tl)
        moe_loss = None
        if len(moe_list) > 0:                                # :423-434 (loss += moe_loss)
            moe_loss = self.router_aux_loss_coef * sum(moe_list)
            if labels is not None:
                loss = loss + moe_loss
        return CausalLMOut(loss=loss, logits=logits, labels=labels, moe_loss=moe_loss, moe_loss_list=moe_list)
