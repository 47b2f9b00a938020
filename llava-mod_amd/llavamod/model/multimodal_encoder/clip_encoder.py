"""CLIPVisionTower on HIP kernels — the frozen ViT-L/14-336 front end.

Reference wrapper: multimodal_encoder/clip_encoder.py:7-84 (load :24-33, feature_select :35-43,
forward under no_grad :45-57).  The arithmetic the reference delegates to HF `CLIPVisionModel`
(patch conv 14x14/14 without bias, class token, position embedding, pre-LN, encoder layers with
biased q/k/v/out projections and quick_gelu MLP) runs here as: im2col + MFMA GEMM, embed-assembly,
LayerNorm, fused-QKV GEMM, non-causal flash attention (hd 64), GEMM(+bias)(+quick_gelu epilogue).
Only the layers up to `select_layer` are executed (hidden_states[-2] => 23 of 24).

Parameter names are HF 4.37's (`image_tower.vision_model.…`) so LLaVA-MoD checkpoints load by name.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

from ... import kernels as K
from ...ops import FusedWeight

BF16 = torch.bfloat16


class CLIPVisionConfig:
    def __init__(self, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                 image_size=336, patch_size=14, layer_norm_eps=1e-5, **kw):
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.image_size, self.patch_size, self.layer_norm_eps = image_size, patch_size, layer_norm_eps


def _lin(i, o, device):
    m = nn.Module()
    m.weight = nn.Parameter(torch.empty((o, i), device=device, dtype=BF16))
    m.bias = nn.Parameter(torch.empty(o, device=device, dtype=BF16))
    return m


def _ln(d, device):
    m = nn.Module()
    m.weight = nn.Parameter(torch.ones(d, device=device, dtype=BF16))
    m.bias = nn.Parameter(torch.zeros(d, device=device, dtype=BF16))
    return m


class _Layer(nn.Module):
    def __init__(self, c, device):
        super().__init__()
        d = c.hidden_size
        self.self_attn = nn.Module()
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            setattr(self.self_attn, n, _lin(d, d, device))
        self.layer_norm1 = _ln(d, device)
        self.mlp = nn.Module()
        self.mlp.fc1 = _lin(d, c.intermediate_size, device)
        self.mlp.fc2 = _lin(c.intermediate_size, d, device)
        self.layer_norm2 = _ln(d, device)
        a = self.self_attn
        self._qkv = FusedWeight([[a.q_proj.weight, a.k_proj.weight, a.v_proj.weight]],
                                [[a.q_proj.bias, a.k_proj.bias, a.v_proj.bias]])
        self._out = FusedWeight([[a.out_proj.weight]], [[a.out_proj.bias]])
        self._fc1 = FusedWeight([[self.mlp.fc1.weight]], [[self.mlp.fc1.bias]])
        self._fc2 = FusedWeight([[self.mlp.fc2.weight]], [[self.mlp.fc2.bias]])


class _Embeddings(nn.Module):
    def __init__(self, c, device):
        super().__init__()
        d, p = c.hidden_size, c.patch_size
        self.class_embedding = nn.Parameter(torch.empty(d, device=device, dtype=BF16))
        self.patch_embedding = nn.Module()
        self.patch_embedding.weight = nn.Parameter(torch.empty((d, 3, p, p), device=device, dtype=BF16))
        self.position_embedding = nn.Module()
        n = (c.image_size // p) ** 2 + 1
        self.position_embedding.weight = nn.Parameter(torch.empty((n, d), device=device, dtype=BF16))


class _VisionModel(nn.Module):
    def __init__(self, c, device):
        super().__init__()
        self.embeddings = _Embeddings(c, device)
        self.pre_layrnorm = _ln(c.hidden_size, device)          # (sic) HF attribute name
        self.encoder = nn.Module()
        self.encoder.layers = nn.ModuleList([_Layer(c, device) for _ in range(c.num_hidden_layers)])
        self.post_layernorm = _ln(c.hidden_size, device)


class _CLIPVisionModel(nn.Module):
    def __init__(self, c, device):
        super().__init__()
        self.config = c
        self.vision_model = _VisionModel(c, device)


class CLIPVisionTower(nn.Module):
    def __init__(self, image_tower, args, delay_load=False, cache_dir="./cache_dir", device="cuda"):
        super().__init__()
        self.is_loaded = False
        self.image_tower_name = image_tower
        self.select_layer = getattr(args, "mm_vision_select_layer", -2)
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        self.cfg_only = image_tower if isinstance(image_tower, CLIPVisionConfig) else CLIPVisionConfig()
        self._device = device
        self._patch_w = None
        self._patch_ver = None
        if not delay_load:
            self.load_model()

    def load_model(self):
        if self.is_loaded:
            return
        self.image_tower = _CLIPVisionModel(self.cfg_only, self._device)
        self.image_tower.requires_grad_(False)               # clip_encoder.py:31
        self.is_loaded = True

    @property
    def config(self):
        return self.cfg_only

    @property
    def hidden_size(self):
        return self.cfg_only.hidden_size

    @property
    def num_patches(self):
        return (self.cfg_only.image_size // self.cfg_only.patch_size) ** 2

    @property
    def dtype(self):
        return BF16

    @property
    def device(self):
        return self.image_tower.vision_model.pre_layrnorm.weight.device

    def _n_layers_run(self):
        L = self.cfg_only.num_hidden_layers
        idx = self.select_layer if self.select_layer >= 0 else L + 1 + self.select_layer   # index into hidden_states
        return idx                                                                          # hidden_states[idx] = after idx layers

    def _patch_weight(self):
        w = self.image_tower.vision_model.embeddings.patch_embedding.weight
        if self._patch_w is None or self._patch_ver != w._version or self._patch_w.device != w.device:
            d = w.shape[0]
            k = w[0].numel()
            kp = (k + 7) // 8 * 8
            pw = torch.zeros((d, kp), device=w.device, dtype=BF16)
            pw[:, :k].copy_(w.reshape(d, k))
            self._patch_w, self._patch_ver = pw, w._version
        return self._patch_w

    @torch.no_grad()                                         # clip_encoder.py:45
    def forward(self, images):
        """images [B,3,S,S] -> hidden_states[select_layer] INCLUDING the CLS row: [B, 1+P, D] bf16.
        The 'patch' feature selection (drop CLS, clip_encoder.py:36-38) is folded into the projector
        GEMM's batch stride; `select_patches()` gives the explicit [B, P, D] tensor."""
        if self.select_feature != "patch":
            raise ValueError(f"Unexpected select feature: {self.select_feature}")
        c = self.cfg_only
        vm = self.image_tower.vision_model
        B = images.shape[0]
        D, P = c.hidden_size, self.num_patches
        nh = c.num_attention_heads
        hd = D // nh
        x = images.to(device=self.device, dtype=BF16).contiguous()
        pw = self._patch_weight()
        cols = K.im2col_patch(x, c.patch_size, pw.shape[1])
        pe = K.gemm_nt(cols, pw)                                                   # [B*P, D]
        tok = K.vit_embed(pe, vm.embeddings.class_embedding.data, vm.embeddings.position_embedding.weight.data, B, P)
        h = K.layernorm_fwd(tok, vm.pre_layrnorm.weight.data, vm.pre_layrnorm.bias.data, c.layer_norm_eps)
        S = P + 1
        for layer in list(vm.encoder.layers)[:self._n_layers_run()]:
            n1 = K.layernorm_fwd(h, layer.layer_norm1.weight.data, layer.layer_norm1.bias.data, c.layer_norm_eps)
            qkv = K.gemm_nt(n1, layer._qkv.ensure().w, bias=layer._qkv.b)
            o, _ = K.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, S, nh, nh, hd, hd ** -0.5, False,
                              None, want_lse=False)
            a = K.gemm_nt(o, layer._out.ensure().w, bias=layer._out.b)
            h = K.add(h, a)
            n2 = K.layernorm_fwd(h, layer.layer_norm2.weight.data, layer.layer_norm2.bias.data, c.layer_norm_eps)
            f = K.gemm_nt(n2, layer._fc1.ensure().w, bias=layer._fc1.b, act=2)     # quick_gelu in the epilogue
            m = K.gemm_nt(f, layer._fc2.ensure().w, bias=layer._fc2.b)
            h = K.add(h, m)
        return h.view(B, S, D)

    def select_patches(self, feats):
        return feats[:, 1:]
