"""Checkpoint I/O in the reference's key layout.

The reference saves through HF `save_pretrained` (sharded safetensors + `model.safetensors.index.json`) and writes the
trainable projector separately as `mm_projector.bin` (`train/align_train.py:623-631`, read back by
`llava_arch.py:122-128`).  The module tree here already carries the reference's parameter names
(`model.layers.N.mlp.deepspeed_moe.gate.wg.weight`, `…deepspeed_experts.E.{gate,up,down}_proj.weight`,
`model.mm_projector.image_spatial_proj.{0,2}.*`, `model.image_tower.image_tower.vision_model.*`), so this is plain
name -> tensor I/O plus the two key dialects a real checkpoint may use.
"""
import json
import os
import re

import torch
from safetensors import safe_open
from safetensors.torch import save_file

INDEX = "model.safetensors.index.json"


def _normalise(key):
    """Accept HF-transformers-5.x CLIP keys (no `vision_model.` level) and DeepSpeed's `module.` prefix."""
    if key.startswith("module."):
        key = key[len("module."):]
    m = re.match(r"(model\.image_tower\.image_tower\.)(?!vision_model\.)(.*)", key)
    if m:
        key = m.group(1) + "vision_model." + m.group(2)
    return key


def save_checkpoint(model, out_dir, max_shard_bytes=5 << 30, projector_file=True):
    """Write `model-XXXXX-of-YYYYY.safetensors` + index (HF layout) and, like the reference's trainer, the projector
    weights alone as `mm_projector.bin`.  Returns the list of files written."""
    os.makedirs(out_dir, exist_ok=True)
    from .engine import fused_weights_of
    for fw in fused_weights_of(model):          # orders this stream after any optimizer update still in flight
        if fw.w is not None:
            fw.ensure()
    sd = {k: v.detach().to("cpu").contiguous() for k, v in model.state_dict().items()}
    shards, cur, size = [], {}, 0
    for k, v in sd.items():
        n = v.numel() * v.element_size()
        if cur and size + n > max_shard_bytes:
            shards.append(cur); cur, size = {}, 0
        cur[k] = v; size += n
    if cur:
        shards.append(cur)
    files, weight_map = [], {}
    for i, sh in enumerate(shards):
        name = "model.safetensors" if len(shards) == 1 else f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(out_dir, name), metadata={"format": "pt"})
        files.append(name)
        weight_map.update({k: name for k in sh})
    if len(shards) > 1:
        total = sum(v.numel() * v.element_size() for v in sd.values())
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, open(os.path.join(out_dir, INDEX), "w"), indent=1)
        files.append(INDEX)
    if projector_file:
        proj = {k: v for k, v in sd.items() if "mm_projector" in k}
        if proj:
            torch.save(proj, os.path.join(out_dir, "mm_projector.bin"))
            files.append("mm_projector.bin")
    return files


def _read(path):
    if path.endswith(".safetensors"):
        with safe_open(path, framework="pt", device="cpu") as f:
            return {k: f.get_tensor(k) for k in f.keys()}
    return torch.load(path, map_location="cpu")


def read_state(path):
    """A directory (HF shards / single file / *.bin), a .safetensors file or a torch-pickled state dict -> {name: tensor}."""
    if os.path.isdir(path):
        idx = os.path.join(path, INDEX)
        if os.path.exists(idx):
            names = sorted(set(json.load(open(idx))["weight_map"].values()))
        else:
            # HF's resolution order: the single file, then numbered shards, then the torch-pickled forms
            ls = os.listdir(path)
            names = (["model.safetensors"] if "model.safetensors" in ls else []) or \
                    sorted(n for n in ls if re.fullmatch(r"model-\d+-of-\d+\.safetensors", n)) or \
                    sorted(n for n in ls if re.fullmatch(r"pytorch_model.*\.bin", n))
        if not names:
            raise FileNotFoundError(f"no model weights under {path}")
        out = {}
        for n in names:
            out.update(_read(os.path.join(path, n)))
        return out
    return _read(path)


def load_checkpoint(model, path, strict=True, ignore_prefixes=(), state=None):
    """Copy tensors into `model` by (normalised) name, casting to each parameter's dtype/device.  Checkpoints saved
    BEFORE up-cycling (dense `mlp.{gate,up,down}_proj`) load into an up-cycled model the way the reference's
    `initialize_moe_modules` does: every expert receives the dense FFN (`llava_qwen2_moe.py:547-556`).
    ignore_prefixes: checkpoint AND model tensors under these names are left alone (the image tower when it was loaded
    from its own directory).  state: the already-read {name: tensor} of `path`.  Non-persistent HF buffers
    (`rotary_emb.inv_freq`, `position_ids`) in a checkpoint are not parameters of this tree and are skipped.
    Returns (missing, unexpected)."""
    raw = state if state is not None else read_state(path)
    src = {_normalise(k): v for k, v in raw.items()
           if not (k.endswith("rotary_emb.inv_freq") or k.endswith(".position_ids"))}
    src = {k: v for k, v in src.items() if not k.startswith(tuple(ignore_prefixes))} if ignore_prefixes else src
    own = model.state_dict()
    used, missing, plan = set(), [], []
    for k, t in own.items():                 # validate everything BEFORE touching the model: a failed load leaves it intact
        if ignore_prefixes and k.startswith(tuple(ignore_prefixes)):
            continue
        v = src.get(k)
        if v is None:
            m = re.match(r"(.*\.mlp\.)deepspeed_moe\.experts\.deepspeed_experts\.\d+\.(.*)", k)
            if m and (m.group(1) + m.group(2)) in src:
                v = src[m.group(1) + m.group(2)]
                used.add(m.group(1) + m.group(2))
        else:
            used.add(k)
        if v is None:
            missing.append(k)
            continue
        if tuple(v.shape) != tuple(t.shape):
            raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != model shape {tuple(t.shape)}")
        plan.append((t, v))
    unexpected = sorted(set(src) - used)
    if strict and (missing or unexpected):
        raise KeyError(f"missing {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                       f"unexpected {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
    with torch.no_grad():
        for t, v in plan:
            t.copy_(v.to(device=t.device, dtype=t.dtype))
    # weights changed behind the caches' back: fused-weight transposes are keyed on parameter versions (copy_ bumps them);
    # optimizers holding fp32 masters must be told (HipAdamW.resync_master()); trainers re-check tower sharing
    model._weights_epoch = getattr(model, "_weights_epoch", 0) + 1
    return missing, unexpected


def save_optimizer(opt, out_dir):
    """`HipAdamW.state_dict()` of THIS rank -> `optimizer_rank{r}_of{w}.pt` (atomic).  With the ZeRO-2 style optimizer the
    fp32 master / m / v exist only as per-rank shards, so every rank saves its own file (DeepSpeed's
    `zero_pp_rank_*_optim_states.pt` convention)."""
    os.makedirs(out_dir, exist_ok=True)
    st = opt.state_dict()
    path = os.path.join(out_dir, f"optimizer_rank{st['rank']}_of{st['world']}.pt")
    tmp = f"{path}.tmp{os.getpid()}"
    torch.save(st, tmp)
    os.replace(tmp, path)
    return path


def load_optimizer(opt, in_dir):
    """Inverse of `save_optimizer` for this rank; raises if the shard layout differs (exact resume needs the same world
    size / ZeRO-2 setting / trainable set)."""
    world = opt.dp.world if opt.dp is not None else 1
    rank = opt.dp.rank if opt.dp is not None else 0
    path = os.path.join(in_dir, f"optimizer_rank{rank}_of{world}.pt")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: no optimizer state for rank {rank} of {world}")
    # tensors, numbers, strings, lists and dicts only (HipAdamW.state_dict): no pickled code is ever executed on resume
    opt.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    return path
