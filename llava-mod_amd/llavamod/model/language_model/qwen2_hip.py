"""Qwen2-style decoder whose forward/backward run on the HIP kernels (llavamod.ops / llavamod.kernels).

Module tree and parameter names are the reference's (vendored qwen2/modeling_qwen2.py:203-325,175-187,
725-799,919-1094 == HF Qwen2Model that the MoE student subclasses, llava_qwen2_moe.py:23,84), so state
dicts interchange:  model.embed_tokens, model.layers.N.{self_attn.{q,k,v,o}_proj, mlp.{gate,up,down}_proj,
input_layernorm, post_attention_layernorm}, model.norm, lm_head.

Activations are flat [T = B*S, H] bf16 buffers.  The residual stream is carried as (res, delta) and
the add is fused into the next RMSNorm.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn

from ... import kernels as K
from ... import ops
from ...ops import FusedWeight

BF16 = torch.bfloat16


class Qwen2Config:
    """Plain config object with the HF Qwen2Config field names the reference reads."""
    model_type = "qwen2"

    def __init__(self, vocab_size=151936, hidden_size=2048, intermediate_size=5504, num_hidden_layers=24,
                 num_attention_heads=16, num_key_value_heads=None, rms_norm_eps=1e-6, rope_theta=10000.0,
                 max_position_embeddings=4096, **kwargs):
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads or num_attention_heads
        self.rms_norm_eps = rms_norm_eps
        self.rope_theta = rope_theta
        self.max_position_embeddings = max_position_embeddings
        self.use_return_dict = True
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads


class _Linear(nn.Module):
    """Parameter holder with nn.Linear's names/shapes; the math happens in fused blocks."""

    def __init__(self, in_f, out_f, bias, device, dtype=BF16):
        super().__init__()
        self.in_features, self.out_features = in_f, out_f
        self.weight = nn.Parameter(torch.empty((out_f, in_f), device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.empty(out_f, device=device, dtype=dtype)) if bias else None


class KVCache:
    """Per-layer key/value cache for generation (the reference relies on HF's DynamicCache through
    prepare_inputs_for_generation, llava_qwen2_moe.py:453-473).  k/v: [L][B, Smax, nkv*hd] bf16, post-RoPE keys;
    lens [B] int32 = number of valid positions per sample (right-padded prompts keep their own length)."""

    def __init__(self, n_layers, B, smax, width, device):
        self.k = [torch.zeros((B, smax, width), device=device, dtype=BF16) for _ in range(n_layers)]
        self.v = [torch.zeros((B, smax, width), device=device, dtype=BF16) for _ in range(n_layers)]
        self.lens = torch.zeros(B, device=device, dtype=torch.int32)
        self.smax = smax

    def get_seq_length(self):
        return int(self.lens.max())


class Qwen2RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps, device):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=BF16))
        self.variance_epsilon = eps


class Qwen2Attention(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        hd = cfg.head_dim
        self.q_proj = _Linear(cfg.hidden_size, cfg.num_attention_heads * hd, True, device)
        self.k_proj = _Linear(cfg.hidden_size, cfg.num_key_value_heads * hd, True, device)
        self.v_proj = _Linear(cfg.hidden_size, cfg.num_key_value_heads * hd, True, device)
        self.o_proj = _Linear(cfg.num_attention_heads * hd, cfg.hidden_size, False, device)
        self._qkv = FusedWeight([[self.q_proj.weight, self.k_proj.weight, self.v_proj.weight]],
                                [[self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]])
        self._o = FusedWeight([[self.o_proj.weight]])
        self.nh, self.nkv, self.hd = cfg.num_attention_heads, cfg.num_key_value_heads, hd

    def trainable(self):
        return [p for p in self.parameters() if p.requires_grad]

    def res_fusable(self, x, res):
        """Can forward() take the layer's residual stream into the o projection's epilogue?"""
        return ops.res_fusable(x.shape[0], self._o.ensure(), res)

    def forward(self, x, rt, rows=None, inv_rows=None, res=None):
        """res: the residual stream [T, H]; the result is then res + attention(x) (callers check res_fusable first)."""
        spec = SimpleNamespace(qkv=self._qkv.ensure(), o=self._o.ensure(), B=rt.B, S=rt.S, nh=self.nh, nkv=self.nkv,
                               hd=self.hd, cos=rt.cos, sin=rt.sin, pos=rt.pos, scale=1.0 / math.sqrt(self.hd),
                               seqlens=rt.seqlens, rows=rows, inv_rows=inv_rows, cu=getattr(rt, "cu", None),
                               kv_out=(rt.cache.k[rt.layer], rt.cache.v[rt.layer]) if getattr(rt, "cache", None) is not None else None)
        if res is not None:
            assert rows is None
            return ops.AttnBlockRes.apply(x, res, spec, *self.trainable())
        return ops.AttnBlock.apply(x, spec, *self.trainable())

    def forward_decode(self, x, rt):
        """One new token per sample against the cache: x [B, H] -> [B, H] (no autograd)."""
        nh, nkv, hd = self.nh, self.nkv, self.hd
        qkv = ops.linear_fwd(x, self._qkv.ensure())
        K.rope_(qkv, rt.cos, rt.sin, rt.pos, nh + nkv, hd)                       # position = current length of the sample
        kc, vc = rt.cache.k[rt.layer], rt.cache.v[rt.layer]
        bi = rt.batch_index
        kc[bi, rt.pos_long] = qkv[:, nh * hd:(nh + nkv) * hd]                    # append (indexed copy, no arithmetic)
        vc[bi, rt.pos_long] = qkv[:, (nh + nkv) * hd:]
        o = K.attn_decode(qkv[:, :nh * hd], kc, vc, rt.lens_after, nh, nkv, hd, 1.0 / math.sqrt(hd))
        return ops.linear_fwd(o, self._o.ensure())


class Qwen2MLP(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        self.hidden_size, self.intermediate_size = cfg.hidden_size, cfg.intermediate_size
        self.gate_proj = _Linear(cfg.hidden_size, cfg.intermediate_size, False, device)
        self.up_proj = _Linear(cfg.hidden_size, cfg.intermediate_size, False, device)
        self.down_proj = _Linear(cfg.intermediate_size, cfg.hidden_size, False, device)
        self._gu = FusedWeight([[self.gate_proj.weight, self.up_proj.weight]])
        self._down = FusedWeight([[self.down_proj.weight]])

    def trainable(self):
        return [p for p in self.parameters() if p.requires_grad]

    def fused_weights(self):
        return [self._gu, self._down]

    def res_fusable(self, x, res):
        return ops.res_fusable(x.shape[0], self._down.ensure(), res)

    def forward(self, x, rt=None, res=None):
        spec = SimpleNamespace(gu=self._gu.ensure(), down=self._down.ensure())
        if res is not None:         # res + mlp(x), the add in the down projection's epilogue (callers check res_fusable first)
            return ops.MLPBlockRes.apply(x, res, spec, *self.trainable())
        return ops.MLPBlock.apply(x, spec, *self.trainable())


class Qwen2DecoderLayer(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        self.self_attn = Qwen2Attention(cfg, device)
        self.mlp = Qwen2MLP(cfg, device)
        self.input_layernorm = Qwen2RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device)
        self.post_attention_layernorm = Qwen2RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device)

    def forward(self, delta, res, rt):
        """MoEQwen2DecoderLayer_forward (llava_qwen2_moe.py:143-179) on the (res, delta) stream."""
        n1, h = ops.AddRMSNorm.apply(delta, res, self.input_layernorm.weight, self.input_layernorm.variance_epsilon)
        # the two residual adds of the layer (hidden = residual + attn / + mlp) run in the epilogues of the o and down projections
        # where the 4-wave GEMM takes the shape (same roundings: bf16(res + bf16(acc))); the norm that follows then reads ONE tensor
        # and writes one.  A sum that is already formed travels on as (delta = sum, res = None).
        if self.self_attn.res_fusable(n1, h):
            a, h = self.self_attn(n1, rt, res=h), None
        else:
            a = self.self_attn(n1, rt)
        n2, h2 = ops.AddRMSNorm.apply(a, h, self.post_attention_layernorm.weight,
                                      self.post_attention_layernorm.variance_epsilon)
        if isinstance(self.mlp, Qwen2MLP) and self.mlp.res_fusable(n2, h2):
            return self.mlp(n2, rt, res=h2), None, []
        m = self.mlp(n2, rt)
        moe_losses = []
        if isinstance(m, tuple) and len(m) == 3:      # MoE returns (out, l_aux, exp_counts)  (:161-164)
            moe_losses.append(m[1])
            m = m[0]
        return m, h2, moe_losses

    def forward_decode(self, delta, res, rt):
        n1, _, h = K.rmsnorm_fwd(delta, self.input_layernorm.weight, self.input_layernorm.variance_epsilon, res=res)
        a = self.self_attn.forward_decode(n1, rt)
        n2, _, h2 = K.rmsnorm_fwd(a, self.post_attention_layernorm.weight, self.post_attention_layernorm.variance_epsilon, res=h)
        m = self.mlp(n2, rt)
        if isinstance(m, tuple):
            m = m[0]
        return m, h2

    def forward_rows(self, delta, res, rt, rows, inv_rows):
        """Same layer, but everything after the attention core runs only on token rows `rows` (int32 [R]; inv_rows[t] =
        position of t in rows or -1): the LAST layer of a model whose consumer reads R << T rows (the loss rows).  The
        values on those rows, and every gradient, are the same as computing all rows.  Dense MLP only (a MoE layer's
        capacity and l_aux depend on every token)."""
        n1, h = ops.AddRMSNorm.apply(delta, res, self.input_layernorm.weight, self.input_layernorm.variance_epsilon)
        a = self.self_attn(n1, rt, rows=rows, inv_rows=inv_rows)                    # [R, H]
        h_r = ops.RowGather.apply(h, rows, inv_rows)
        n2, h2 = ops.AddRMSNorm.apply(a, h_r, self.post_attention_layernorm.weight,
                                      self.post_attention_layernorm.variance_epsilon)
        return self.mlp(n2, rt), h2


_ROPE_CACHE = {}


def rope_tables(hd, max_pos, theta, device):
    """cos/sin tables cast to the activation dtype (qwen2/modeling_qwen2.py:119-134)."""
    key = (hd, max_pos, float(theta), str(device))
    if key not in _ROPE_CACHE:
        inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        fr = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv)
        emb = torch.cat((fr, fr), dim=-1)
        _ROPE_CACHE[key] = (emb.cos().to(BF16).to(device), emb.sin().to(BF16).to(device))
    return _ROPE_CACHE[key]


class Qwen2Model(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        self.config = cfg
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.hidden_size, device=device, dtype=BF16)
        self.layers = nn.ModuleList([Qwen2DecoderLayer(cfg, device) for _ in range(cfg.num_hidden_layers)])
        self.norm = Qwen2RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device)

    def runtime(self, B, S, seqlens, device, cu=None, pos=None):
        cos, sin = rope_tables(self.config.head_dim, max(self.config.max_position_embeddings, S),
                               self.config.rope_theta, device)
        if pos is None:
            pos = torch.arange(S, device=device, dtype=torch.int32).repeat(B)      # position_ids = arange(S')
        return SimpleNamespace(B=B, S=S, cos=cos, sin=sin, pos=pos, seqlens=seqlens, cu=cu)

    def forward_decode(self, inputs_embeds, cache):
        """inputs_embeds [B, H]: the embeddings of ONE new token per sample; appends to `cache` and returns the
        final-normed hidden states [B, H]."""
        dev = inputs_embeds.device
        B = inputs_embeds.shape[0]
        cos, sin = rope_tables(self.config.head_dim, max(self.config.max_position_embeddings, cache.smax),
                               self.config.rope_theta, dev)
        pos = cache.lens.clone()                                                   # position_ids of the new tokens
        rt = SimpleNamespace(B=B, S=1, cos=cos, sin=sin, pos=pos, seqlens=None, cache=cache, layer=0,
                             pos_long=pos.long(), batch_index=torch.arange(B, device=dev), lens_after=pos + 1)
        delta, res = inputs_embeds, None
        with torch.no_grad():
            for i, layer in enumerate(self.layers):
                rt.layer = i
                delta, res = layer.forward_decode(delta, res, rt)
            y, _, _ = K.rmsnorm_fwd(delta, self.norm.weight, self.norm.variance_epsilon, res=res)
        cache.lens += 1
        return y

    def forward(self, inputs_embeds, B, S, seqlens=None, out_rows=None, inv_rows=None, cache=None, cu=None, pos=None):
        """inputs_embeds: [B*S, H].  Returns (final-normed hidden [B*S, H], list of l_aux).  out_rows (int32 [R]) with
        inv_rows (int32 [B*S]): return just those rows, [R, H] — a dense last layer then skips o_proj / MLP / norm work
        (forward and backward) on every other row."""
        rt = self.runtime(B, S, seqlens, inputs_embeds.device, cu=cu, pos=pos)   # cu/pos: packed rows (unpadded execution)
        rt.cache = cache                       # generation prefill: every layer stores its post-RoPE K and V
        delta, res = inputs_embeds, None
        all_moe = []
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            rt.layer = i
            if i == last and out_rows is not None and type(layer.mlp) is Qwen2MLP:
                delta, res = layer.forward_rows(delta, res, rt, out_rows, inv_rows)
                out_rows = None
                continue
            delta, res, ml = layer(delta, res, rt)
            all_moe.extend(ml)
        if out_rows is not None:               # sparse last layer: every row is needed inside it; select at the end
            delta = ops.RowGather.apply(delta, out_rows, inv_rows)
            res = ops.RowGather.apply(res, out_rows, inv_rows) if res is not None else None
        y, _ = ops.AddRMSNorm.apply(delta, res, self.norm.weight, self.norm.variance_epsilon)
        return y, all_moe


def init_normal_(module, std=0.02, seed=0):
    """Random-init of the named architecture (no network for checkpoints): N(0, std) everywhere except
    norm scales (1) — config 1/2 recipe of SURVEY.md §8d."""
    dev = next(module.parameters()).device
    if dev.type == "meta":          # shape-only construction (bench.py --launch-check plans the exchange without weights)
        return module
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if "norm" in n and n.endswith("weight"):
                p.fill_(1.0)
            elif "norm" in n and n.endswith("bias"):
                p.zero_()
            else:
                p.normal_(0.0, std, generator=g)
    return module
