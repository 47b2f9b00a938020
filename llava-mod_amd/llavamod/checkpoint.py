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


_EXPERT_KEY = re.compile(r"(.*)\.deepspeed_moe\.experts\.deepspeed_experts\.(\d+)\.(.*)")


def _dist_rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def job_rank(args=None):
    """This process's rank in the whole job, for "ONE rank writes" decisions.  torch.distributed when it is initialised; otherwise
    (INTEGRATION.md's native-communicator hosts, plain RANK / LOCAL_RANK launches) what the launcher told the process: HF's
    `args.process_index`, then the RANK environment variable, then a non-negative `args.local_rank`.  Without this fallback every
    process of such a run sees (0, 1) from `_dist_rank_world` and all of them write the same shards, index and config."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    pi = getattr(args, "process_index", None)
    if isinstance(pi, int) and pi >= 0:
        return pi
    if os.environ.get("RANK", "").isdigit():
        return int(os.environ["RANK"])
    lr = getattr(args, "local_rank", None)
    return lr if isinstance(lr, int) and lr > 0 else 0


def expert_parallel_layout(model):
    """{MoE module name: (ep_size, local experts, this rank's index in its expert-parallel group)} for the layers whose experts
    are sharded over ranks (`ep_size` > 1).  A rank's `deepspeed_experts.{i}` is then GLOBAL expert `ep_rank * n_local + i`
    (DeepSpeed numbers them the same way: `global_expert_id = ep_rank * num_local_experts + local_id`)."""
    from .model.moe_layer import MoE
    rank, _ = _dist_rank_world()
    return {n: (m.ep_size, m.num_local_experts, rank % m.ep_size) for n, m in model.named_modules()
            if isinstance(m, MoE) and m.ep_size > 1}


def _atomic(write, path):
    """`write(tmp)` then rename: a reader (or a crash) never sees a half-written shard / index / projector file."""
    tmp = f"{path}.tmp{os.getpid()}"
    write(tmp)
    os.replace(tmp, path)


def full_state_dict(model, writer_rank=0):
    """The model's state dict on the host with GLOBAL expert names.  Without expert parallelism this is `model.state_dict()`.
    With `ep_size` > 1 a rank only holds `n_local` of a layer's experts under local indices: the ranks of the WRITER's
    expert-parallel group all-gather every expert tensor (a collective: each of them must call this) and the result carries
    `deepspeed_experts.{ep_rank * n_local + i}` — the layout an `ep_size` = 1 model (and the reference's FineTune / Eval classes)
    load.  Returns None on ranks that are not the writer."""
    import torch.distributed as dist
    rank, world = _dist_rank_world()
    layout = expert_parallel_layout(model)
    own = model.state_dict()
    if not layout:
        return {k: v.detach().to("cpu").contiguous() for k, v in own.items()} if rank == writer_rank else None
    ep = next(iter(layout.values()))[0]
    # Group creation first, on EVERY rank: `expert_parallel_group` walks dist.new_group over all expert-parallel groups, which the
    # whole world must call together.  A save before the first MoE forward (step-0 save, a conversion or resume script) finds the
    # group uncached; leaving through the early return below first would have only the writer's group create it and hang.
    from .engine import expert_parallel_group
    group = expert_parallel_group(ep)
    if rank // ep != writer_rank // ep:          # another replica of the same experts: nothing to contribute
        return None
    out = {}
    for k, v in own.items():
        m = _EXPERT_KEY.match(k)
        lay = layout.get(m.group(1)) if m else None
        if lay is None:
            if rank == writer_rank:
                out[k] = v.detach().to("cpu").contiguous()
            continue
        ep_size, n_local, _ = lay
        parts = [torch.empty_like(v) for _ in range(ep_size)]
        dist.all_gather(parts, v.detach().contiguous(), group=group)
        if rank == writer_rank:
            for r, t in enumerate(parts):
                out[f"{m.group(1)}.deepspeed_moe.experts.deepspeed_experts.{r * n_local + int(m.group(2))}.{m.group(3)}"] = \
                    t.to("cpu").contiguous()
    return out if rank == writer_rank else None


def save_checkpoint(model, out_dir, max_shard_bytes=5 << 30, projector_file=True, writer_rank=0):
    """Write `model-XXXXX-of-YYYYY.safetensors` + index (HF layout) and, like the reference's trainer, the projector
    weights alone as `mm_projector.bin`; every file through a temporary name + rename.  ONE rank writes (`writer_rank`, the
    global rank: on a multi-node run every node's local rank 0 would otherwise race on the same files).  Expert-parallel
    models: collective over the writer's expert-parallel group, experts saved under global indices (`full_state_dict`).
    Returns the list of files written (empty on the other ranks)."""
    from .engine import fused_weights_of
    for fw in fused_weights_of(model):          # orders this stream after any optimizer update still in flight
        if fw.w is not None:
            fw.ensure()
    sd = full_state_dict(model, writer_rank)
    if sd is None:
        return []
    os.makedirs(out_dir, exist_ok=True)
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
        _atomic(lambda t, sh=sh: save_file(sh, t, metadata={"format": "pt"}), os.path.join(out_dir, name))
        files.append(name)
        weight_map.update({k: name for k in sh})
    if len(shards) > 1:
        total = sum(v.numel() * v.element_size() for v in sd.values())
        def write_index(t):
            with open(t, "w") as f:
                json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=1)
        _atomic(write_index, os.path.join(out_dir, INDEX))
        files.append(INDEX)
    if projector_file:
        proj = {k: v for k, v in sd.items() if "mm_projector" in k}
        if proj:
            _atomic(lambda t: torch.save(proj, t), os.path.join(out_dir, "mm_projector.bin"))
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
    layout = expert_parallel_layout(model)      # expert-parallel model: local expert i is the checkpoint's global ep_rank*n_local+i
    ep_owned = set()
    for k, t in own.items():                 # validate everything BEFORE touching the model: a failed load leaves it intact
        if ignore_prefixes and k.startswith(tuple(ignore_prefixes)):
            continue
        m = _EXPERT_KEY.match(k) if layout else None
        if m and m.group(1) in layout:
            ep_size, n_local, ep_rank = layout[m.group(1)]
            gk = f"{m.group(1)}.deepspeed_moe.experts.deepspeed_experts.{ep_rank * n_local + int(m.group(2))}.{m.group(3)}"
            for r in range(ep_size * n_local):   # the other ranks' experts of this layer are not "unexpected" here
                ep_owned.add(f"{m.group(1)}.deepspeed_moe.experts.deepspeed_experts.{r}.{m.group(3)}")
            v = src.get(gk)
            if v is not None:
                used.add(gk)
        else:
            v = src.get(k)
            if v is not None:
                used.add(k)
        if v is None:
            m = re.match(r"(.*\.mlp\.)deepspeed_moe\.experts\.deepspeed_experts\.\d+\.(.*)", k)
            if m and (m.group(1) + m.group(2)) in src:
                v = src[m.group(1) + m.group(2)]
                used.add(m.group(1) + m.group(2))
        if v is None:
            missing.append(k)
            continue
        if tuple(v.shape) != tuple(t.shape):
            raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != model shape {tuple(t.shape)}")
        plan.append((t, v))
    unexpected = sorted(set(src) - used - ep_owned)
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
