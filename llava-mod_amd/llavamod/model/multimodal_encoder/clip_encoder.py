"""CLIPVisionTower on HIP kernels — the frozen ViT-L/14-336 front end.

Reference wrapper: multimodal_encoder/clip_encoder.py:7-84 (load :24-33, feature_select :35-43,
forward under no_grad :45-57).  The arithmetic the reference delegates to HF `CLIPVisionModel`
(patch conv 14x14/14 without bias, class token, position embedding, pre-LN, encoder layers with
biased q/k/v/out projections and quick_gelu MLP) runs here as: im2col + MFMA GEMM, embed-assembly,
LayerNorm, fused-QKV GEMM, non-causal flash attention (hd 64), GEMM(+bias)(+quick_gelu epilogue).
Only the layers up to `select_layer` are executed (hidden_states[-2] => 23 of 24).

Parameter names are HF 4.37's (`image_tower.vision_model.…`) so LLaVA-MoD checkpoints load by name.
"""

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


# geometry of the hub names the reference's shells pass as --image_tower (dense2sparse_distillation.sh:23); used only to
# build the ARCHITECTURE when the weights then come from a checkpoint's own `model.image_tower.*` keys
KNOWN_GEOMETRY = {
    "openai/clip-vit-large-patch14-336": dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                              num_attention_heads=16, image_size=336, patch_size=14),
    "openai/clip-vit-large-patch14": dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                                          num_attention_heads=16, image_size=224, patch_size=14),
    "openai/clip-vit-base-patch16": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                         num_attention_heads=12, image_size=224, patch_size=16),
    "openai/clip-vit-base-patch32": dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                         num_attention_heads=12, image_size=224, patch_size=32),
}
_WEIGHT_FILES = ("model.safetensors", "pytorch_model.bin")


def resolve_tower_dir(name, search=()):
    """A local directory holding the tower (`config.json` + weights): the name itself, or the name relative to one of
    `search` (the main checkpoint's directory, cache_dir).  None if there is none — hub names cannot be fetched here."""
    import os
    if not isinstance(name, str):
        return None
    for base in ("",) + tuple(s for s in search if s):
        d = os.path.join(base, name) if base else name
        if os.path.isdir(d) and os.path.exists(os.path.join(d, "config.json")):
            return d
    return None


def read_clip_config(path):
    """`CLIPVisionConfig.from_pretrained(dir)`: a CLIPVisionModel config, or the `vision_config` of a full CLIPModel one."""
    import json
    import os
    raw = json.load(open(os.path.join(path, "config.json")))
    if isinstance(raw.get("vision_config"), dict):
        raw = raw["vision_config"]
    if raw.get("hidden_act", "quick_gelu") != "quick_gelu":
        raise NotImplementedError(f"CLIP tower with hidden_act={raw['hidden_act']!r}: only quick_gelu (OpenAI CLIP) is on this path")
    keys = ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "image_size", "patch_size",
            "layer_norm_eps")
    return CLIPVisionConfig(**{k: raw[k] for k in keys if k in raw})


def normalise_clip_keys(sd):
    """Tower weights by the names of this module tree (`vision_model.…`, HF 4.37).  Accepts a CLIPVisionModel or full
    CLIPModel state dict in either key dialect (transformers 5.x drops the `vision_model.` level of CLIPVisionModel);
    text tower, projections, logit scale and position-id buffers are dropped."""
    out = {}
    for k, v in sd.items():
        if k.startswith(("text_model.", "text_projection", "visual_projection", "logit_scale")) or k.endswith("position_ids"):
            continue
        if not k.startswith("vision_model."):
            if k.split(".")[0] in ("embeddings", "pre_layrnorm", "encoder", "post_layernorm"):
                k = "vision_model." + k
            else:
                continue
        out[k] = v
    return out


def geometry_from_state(sd, heads=None):
    """CLIPVisionConfig fields from the tensor shapes of a (normalised) tower state dict.  The head count is not in any
    shape: `heads` if given, else hidden/64 (every OpenAI CLIP ViT has 64-wide heads)."""
    d = sd["vision_model.embeddings.class_embedding"].shape[0]
    pw = sd["vision_model.embeddings.patch_embedding.weight"]
    n_pos = sd["vision_model.embeddings.position_embedding.weight"].shape[0]
    L = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("vision_model.encoder.layers."))
    side = int(round((n_pos - 1) ** 0.5))
    return CLIPVisionConfig(hidden_size=d, intermediate_size=sd["vision_model.encoder.layers.0.mlp.fc1.weight"].shape[0],
                            num_hidden_layers=L, num_attention_heads=heads or max(1, d // 64),
                            image_size=side * pw.shape[-1], patch_size=pw.shape[-1])


class CLIPVisionTower(nn.Module):
    """`image_tower`: a checkpoint directory / hub name (the reference's only form, clip_encoder.py:8-33) or — for
    synthetic-weight benchmarks and tests — a `CLIPVisionConfig` object, which builds a RANDOMLY INITIALISED tower of that
    geometry.  A string never random-initialises: it loads from a local directory or raises."""

    def __init__(self, image_tower, args, delay_load=False, cache_dir="./cache_dir", device="cuda", search=()):
        super().__init__()
        self.is_loaded = False
        self.image_tower_name = image_tower
        self.select_layer = getattr(args, "mm_vision_select_layer", -2)
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        self.cache_dir = cache_dir
        self._search = tuple(search)
        self._device = device
        self._patch_w = None
        self._patch_ver = None
        self.weights_source = None
        if isinstance(image_tower, CLIPVisionConfig):
            self.cfg_only = image_tower
        else:
            d = resolve_tower_dir(image_tower, self._search + (cache_dir,))
            # delay_load in the reference reads only the config (clip_encoder.py:21-22)
            self.cfg_only = read_clip_config(d) if d is not None else (
                CLIPVisionConfig(**KNOWN_GEOMETRY[image_tower]) if image_tower in KNOWN_GEOMETRY else None)
        if not delay_load:
            self.load_model()

    def load_model(self, search=()):
        """clip_encoder.py:24-33: `CLIPVisionModel.from_pretrained(image_tower_name)`, frozen."""
        if self.is_loaded:
            return
        if isinstance(self.image_tower_name, CLIPVisionConfig):
            self.image_tower = _CLIPVisionModel(self.cfg_only, self._device)      # explicit random-init architecture
            self.weights_source = "random-init (CLIPVisionConfig object)"
        else:
            d = resolve_tower_dir(self.image_tower_name, tuple(search) + self._search + (self.cache_dir,))
            if d is None:
                raise FileNotFoundError(
                    f"image tower {self.image_tower_name!r} is not a local checkpoint directory (config.json + "
                    f"{' / '.join(_WEIGHT_FILES)}); hub names cannot be downloaded on this path and the tower is never "
                    f"randomly initialised from a name — pass a directory, or load a checkpoint that carries "
                    f"`model.image_tower.*` weights through from_pretrained")
            self.load_from_dir(d)
        self.image_tower.requires_grad_(False)               # clip_encoder.py:31
        self.is_loaded = True

    def load_from_dir(self, d):
        import os
        from ...checkpoint import _read
        f = next((os.path.join(d, n) for n in _WEIGHT_FILES if os.path.exists(os.path.join(d, n))), None)
        if f is None:
            raise FileNotFoundError(f"no {' / '.join(_WEIGHT_FILES)} under {d}")
        self.cfg_only = read_clip_config(d)
        self.load_from_state(_read(f), source=f)

    def load_from_state(self, sd, source="state dict", heads=None):
        """Build the tower (geometry from `cfg_only`, else from the tensor shapes) and copy `sd` into it; every tensor of
        the layers that are executed must be present with the right shape."""
        sd = normalise_clip_keys(sd)
        if self.cfg_only is None:
            self.cfg_only = geometry_from_state(sd, heads)
        tower = _CLIPVisionModel(self.cfg_only, self._device)
        own = tower.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"image tower weights from {source}: missing {missing[:4]}{'...' if len(missing) > 4 else ''}")
        with torch.no_grad():
            for k, t in own.items():
                if tuple(sd[k].shape) != tuple(t.shape):
                    raise ValueError(f"{k}: {tuple(sd[k].shape)} in {source} != {tuple(t.shape)} of the configured geometry")
                t.copy_(sd[k].to(device=t.device, dtype=t.dtype))
        self.image_tower = tower
        self.image_tower.requires_grad_(False)
        self.weights_source = str(source)
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
        return self.image_tower.vision_model.pre_layrnorm.weight.device if self.is_loaded else torch.device(self._device)

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
        if not self.is_loaded:
            raise RuntimeError(f"image tower {self.image_tower_name!r} has no weights yet: call load_model() "
                               f"(initialize_vision_modules) or from_pretrained first")
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
