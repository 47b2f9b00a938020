"""Oracle: CLIP vision tower wrapper, mlp2x_gelu projector, multimodal splice.  TEST INFRASTRUCTURE ONLY.

  CLIPVisionTower.forward / feature_select   multimodal_encoder/clip_encoder.py:35-57
      (the ViT arithmetic itself is HF transformers.CLIPVisionModel — third-party; restated here and
       pinned against the installed transformers implementation by validate_vs_reference.py)
  build_image_projector (mlpNx_gelu)         multimodal_projector/builder.py:54-61,148-149
  prepare_inputs_labels_for_multimodal       llava_arch.py:155-334
"""
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

IGNORE_INDEX = -100        # llavamod/constants.py:6
IMAGE_TOKEN_INDEX = -200   # llavamod/constants.py:8


@dataclass
class VisionConfig:
    hidden_size: int = 32
    intermediate_size: int = 64
    num_hidden_layers: int = 3
    num_attention_heads: int = 4
    image_size: int = 28
    patch_size: int = 14
    layer_norm_eps: float = 1e-5
    select_layer: int = -2

    @property
    def num_patches(self):
        return (self.image_size // self.patch_size) ** 2


class _Emb(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(c.hidden_size))
        self.patch_embedding = nn.Conv2d(3, c.hidden_size, c.patch_size, c.patch_size, bias=False)
        self.position_embedding = nn.Embedding(c.num_patches + 1, c.hidden_size)

    def forward(self, pixels):
        B = pixels.shape[0]
        p = self.patch_embedding(pixels.to(self.patch_embedding.weight.dtype)).flatten(2).transpose(1, 2)
        cls = self.class_embedding.expand(B, 1, -1)
        x = torch.cat([cls, p], dim=1)
        return x + self.position_embedding.weight[None]


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        d = c.hidden_size
        self.nh = c.num_attention_heads
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))

    def forward(self, x):
        B, S, D = x.shape
        hd = D // self.nh
        q = self.q_proj(x).view(B, S, self.nh, hd).transpose(1, 2)
        k = self.k_proj(x).view(B, S, self.nh, hd).transpose(1, 2)
        v = self.v_proj(x).view(B, S, self.nh, hd).transpose(1, 2)
        w = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5)
        w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(w, v).transpose(1, 2).reshape(B, S, D)
        return self.out_proj(o)


class _Mlp(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.fc1 = nn.Linear(c.hidden_size, c.intermediate_size)
        self.fc2 = nn.Linear(c.intermediate_size, c.hidden_size)

    def forward(self, x):
        h = self.fc1(x)
        return self.fc2(h * torch.sigmoid(1.702 * h))        # quick_gelu


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self_attn = _Attn(c)
        self.layer_norm1 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.mlp = _Mlp(c)
        self.layer_norm2 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)

    def forward(self, x):
        x = x + self.self_attn(self.layer_norm1(x))
        return x + self.mlp(self.layer_norm2(x))


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(c) for _ in range(c.num_hidden_layers)])


class _VisionModel(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.embeddings = _Emb(c)
        self.pre_layrnorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)   # (sic) HF attribute name
        self.encoder = _Encoder(c)
        self.post_layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class VisionTower(nn.Module):
    """Parameter names match transformers.CLIPVisionModel (`vision_model.*`)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.vision_model = _VisionModel(cfg)
        self.requires_grad_(False)                            # clip_encoder.py:31

    @torch.no_grad()                                          # clip_encoder.py:45
    def forward(self, images):
        vm = self.vision_model
        x = vm.pre_layrnorm(vm.embeddings(images))
        hs = [x]
        for layer in vm.encoder.layers:
            x = layer(x)
            hs.append(x)
        feats = hs[self.cfg.select_layer][:, 1:]              # feature_select 'patch', clip_encoder.py:35-43
        return feats.to(images.dtype)


class Projector(nn.Module):
    """`mlp2x_gelu`; parameter path mm_projector.image_spatial_proj.{0,2}.{weight,bias} (builder.py:130,57-61)."""

    def __init__(self, mm_hidden, hidden, depth=2):
        super().__init__()
        mods = [nn.Linear(mm_hidden, hidden)]
        for _ in range(1, depth):
            mods += [nn.GELU(), nn.Linear(hidden, hidden)]
        self.image_spatial_proj = nn.Sequential(*mods)

    def forward_image(self, x):
        return self.image_spatial_proj(x)


def splice(embed_tokens, image_features, input_ids, attention_mask, labels):
    """prepare_inputs_labels_for_multimodal, image-only path, right padding (llava_arch.py:213-334).
    image_features: [n_images, P, H].  Returns (inputs_embeds [B,S',H], attention_mask [B,S'], labels [B,S'])."""
    had_labels, had_mask = labels is not None, attention_mask is not None
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    else:
        attention_mask = attention_mask.bool()
    if labels is None:
        labels = torch.full_like(input_ids, IGNORE_INDEX)
    ids = [x[m] for x, m in zip(input_ids, attention_mask)]            # :228-230 strip pads
    lbs = [x[m] for x, m in zip(labels, attention_mask)]
    new_embeds, new_labels, cur = [], [], 0
    for bi, cur_ids in enumerate(ids):
        n_img = int((cur_ids == IMAGE_TOKEN_INDEX).sum())
        if n_img == 0:                                                  # :238-246
            new_embeds.append(torch.cat([embed_tokens(cur_ids), image_features[cur][0:0]], dim=0))
            new_labels.append(lbs[bi])
            cur += 1
            continue
        pos = [-1] + torch.where(cur_ids == IMAGE_TOKEN_INDEX)[0].tolist() + [cur_ids.shape[0]]
        id_chunks = [cur_ids[pos[i] + 1:pos[i + 1]] for i in range(len(pos) - 1)]
        lb_chunks = [lbs[bi][pos[i] + 1:pos[i + 1]] for i in range(len(pos) - 1)]
        emb = torch.split(embed_tokens(torch.cat(id_chunks)), [c.shape[0] for c in lb_chunks], dim=0)
        e_parts, l_parts = [], []
        for i in range(n_img + 1):                                      # :265-275
            e_parts.append(emb[i]); l_parts.append(lb_chunks[i])
            if i < n_img:
                f = image_features[cur]; cur += 1
                e_parts.append(f)
                l_parts.append(torch.full((f.shape[0],), IGNORE_INDEX, dtype=lbs[bi].dtype))
        new_embeds.append(torch.cat(e_parts)); new_labels.append(torch.cat(l_parts))
    max_len = max(x.shape[0] for x in new_embeds)                       # :286-318 right-pad
    B = len(new_embeds)
    lab = torch.full((B, max_len), IGNORE_INDEX, dtype=new_labels[0].dtype)
    am = torch.zeros((B, max_len), dtype=torch.bool)
    emb_p = []
    for i, (e, l) in enumerate(zip(new_embeds, new_labels)):
        n = e.shape[0]
        emb_p.append(torch.cat((e, torch.zeros((max_len - n, e.shape[1]), dtype=e.dtype)), dim=0))
        if n > 0:
            lab[i, :n] = l
            am[i, :n] = True
    return torch.stack(emb_p, dim=0), (am if had_mask else None), (lab if had_labels else None)
