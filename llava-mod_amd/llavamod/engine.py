"""Step engine around the model: flat fp32 gradient buffer (wgrad GEMMs accumulate into it directly),
data-parallel gradient exchange over RCCL, fused AdamW on fp32 masters with bf16 working weights,
warmup-cosine LR.  This replaces what the reference gets from HF Trainer + DeepSpeed ZeRO-2
(config/dpconfig/zero2*.json, train/align_trainer.py:326-434): only the semantics the step needs —
mean of the trainable-parameter gradients over DP ranks once per optimizer step, AdamW — are kept;
no CPU offload (288 GB of HBM holds teacher + student + fp32 optimizer state outright).
"""
import math
import os

import torch
import torch.distributed as dist

from . import kernels as K
from .ops import FusedWeight

BF16 = torch.bfloat16


_EP_GROUPS = {}


def expert_parallel_group(ep_size):
    """Ranks [k*ep, (k+1)*ep) form one expert-parallel group (DeepSpeed groups._create_expert_and_data_parallel)."""
    if ep_size not in _EP_GROUPS:
        world, rank = dist.get_world_size(), dist.get_rank()
        assert world % ep_size == 0, "world size must be divisible by ep_size"
        mine = None
        for k in range(world // ep_size):
            g = dist.new_group(list(range(k * ep_size, (k + 1) * ep_size)))
            if k == rank // ep_size:
                mine = g
        _EP_GROUPS[ep_size] = mine
    return _EP_GROUPS[ep_size]


_EDP_GROUPS = {}


def expert_data_parallel_group(ep_size):
    """Ranks holding the SAME experts (rank % ep_size equal): the only ranks an expert gradient is reduced over."""
    if ep_size not in _EDP_GROUPS:
        world, rank = dist.get_world_size(), dist.get_rank()
        mine = None
        for j in range(ep_size):
            g = dist.new_group(list(range(j, world, ep_size)))
            if j == rank % ep_size:
                mine = g
        _EDP_GROUPS[ep_size] = mine
    return _EDP_GROUPS[ep_size]


def fused_weights_of(model):
    seen, out = set(), []
    for m in model.modules():
        for v in vars(m).values():
            if isinstance(v, FusedWeight) and id(v) not in seen:
                seen.add(id(v))
                out.append(v)
    return out


class GradBuffer:
    """One flat fp32 buffer holding every trainable parameter's gradient.  Fused weights (q/k/v,
    gate/up, stacked experts) get one contiguous span so a single wgrad GEMM fills them."""

    def __init__(self, model):
        self.model = model
        spans = []          # (kind, obj, numel)
        covered = set()
        for fw in fused_weights_of(model):
            ps = fw.params
            # expert copies keep a private (unused) FusedWeight of their own; the MoE layer's stacked one wins
            if any(id(p) in covered for p in ps):
                continue
            if fw.requires_grad:
                fw.ensure()
                spans.append(("w", fw, fw.w.numel()))
                covered.update(id(p) for p in ps)
            if fw.bias_groups is not None:
                bs = fw.bias_groups[0]
                if any(b.requires_grad for b in bs):
                    fw.ensure()
                    spans.append(("b", fw, fw.b.numel()))
                covered.update(id(b) for b in bs)
        # stacked MoE weights must win over per-expert private FusedWeights: order by size desc was not
        # needed because MoE modules are visited before their expert children in model.modules().
        for n, p in model.named_parameters():
            if p.requires_grad and id(p) not in covered:
                spans.append(("p", p, p.numel()))
                covered.add(id(p))
        # dense (replicated) parameters first, expert-parallel-sharded expert weights last: two contiguous regions
        is_exp = lambda sp: bool(getattr(sp[1], "is_expert", False)) and sp[0] == "w"
        spans = [sp for sp in spans if not is_exp(sp)] + [sp for sp in spans if is_exp(sp)]
        self.n_dense = sum(n for sp in spans if not is_exp(sp) for n in [sp[2]])
        self.spans = spans
        total = sum(n for _, _, n in spans)
        dev = next(model.parameters()).device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        self.offsets = []
        for kind, obj, n in spans:
            view = self.flat[off:off + n]
            if kind == "w":
                obj.set_grad_buffer(view.view(obj.w.shape))
            elif kind == "b":
                obj.set_bias_grad_buffer(view.view(obj.b.shape))
            else:
                obj.main_grad = view.view(obj.shape)
            self.offsets.append(off)
            off += n
        self.numel = total

    def zero(self):
        if getattr(self, "clean", False):        # the optimizer cleared every span in its own pass (HipAdamW overlap)
            self.clean = False
        else:
            self.flat.zero_()
        for kind, obj, _ in self.spans:          # forget uses that never saw a backward (eval-style forwards)
            if kind == "w":
                obj.pending = 0


class DataParallel:
    """Sample-sharded data parallelism: every rank runs teacher + student on its own micro-batches; the
    only exchange is the SUM all-reduce of the flat gradient buffer at the optimizer boundary (the
    division by world size is folded into AdamW's grad scale).  Loss normalisation stays per-rank, as
    in the reference (align_trainer.py:526 divides by the local mask sum; DeepSpeed then averages).

    Overlap: `attach(gb)` hooks every fused weight; as soon as a weight's last wgrad GEMM of the backward
    has been enqueued its span is all-reduced asynchronously (RCCL runs on its own stream, ordered after
    the compute stream by an event), so the exchange hides under the rest of backward.  `finish()` reduces
    whatever has no hook (router weights, biases) and waits.  Expert weights sharded over an expert-
    parallel group are reduced only over the ranks that hold the same experts."""

    def __init__(self, bucket_bytes=512 << 20, overlap=True):
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.enabled else 1
        self.bucket = bucket_bytes // 4
        self.overlap = overlap
        self.armed = True                 # set False on non-final micro-batches of a gradient-accumulation window
        self._handles, self._done, self._map = [], set(), {}
        self.gb, self.ep_size = None, 1

    def attach(self, gb, ep_size=1):
        self.gb, self.ep_size = gb, ep_size
        for (kind, obj, n), off in zip(gb.spans, gb.offsets):
            if kind == "w":
                self._map[id(obj)] = (off, n, bool(getattr(obj, "is_expert", False)))
                obj.grad_ready_hook = self._on_ready
        return self

    def _group_for(self, is_expert):
        if is_expert and self.ep_size > 1:
            return expert_data_parallel_group(self.ep_size) if self.world // self.ep_size > 1 else "skip"
        return None

    def _reduce(self, lo, hi, is_expert):
        g = self._group_for(is_expert)
        if g == "skip":
            return
        for a in range(lo, hi, self.bucket):
            self._handles.append(dist.all_reduce(self.gb.flat[a:min(a + self.bucket, hi)], op=dist.ReduceOp.SUM,
                                                 group=g, async_op=True))

    def _on_ready(self, fw):
        if not (self.enabled and self.overlap and self.armed) or id(fw) in self._done:
            return
        off, n, is_exp = self._map[id(fw)]
        self._reduce(off, off + n, is_exp)
        self._done.add(id(fw))

    def finish(self):
        """Reduce every span that was not already sent by a hook, then wait for all of it."""
        if self.enabled:
            for (kind, obj, n), off in zip(self.gb.spans, self.gb.offsets):
                if id(obj) in self._done:
                    continue
                self._reduce(off, off + n, kind == "w" and bool(getattr(obj, "is_expert", False)))
            for h in self._handles:
                h.wait()
        self._handles, self._done = [], set()

    def all_reduce(self, flat, n_dense=None, ep_size=1):
        """Non-overlapped form: SUM over the DP world for the first n_dense elements (replicated parameters);
        the expert region [n_dense:] is reduced only over the ranks that hold the same experts."""
        if not self.enabled:
            return
        n_dense = flat.numel() if (n_dense is None or ep_size == 1) else n_dense
        handles = []
        for lo in range(0, n_dense, self.bucket):
            handles.append(dist.all_reduce(flat[lo:min(lo + self.bucket, n_dense)], op=dist.ReduceOp.SUM, async_op=True))
        if n_dense < flat.numel() and self.world // ep_size > 1:
            g = expert_data_parallel_group(ep_size)
            for lo in range(n_dense, flat.numel(), self.bucket):
                handles.append(dist.all_reduce(flat[lo:lo + self.bucket], op=dist.ReduceOp.SUM, group=g, async_op=True))
        for h in handles:
            h.wait()


class HipAdamW:
    """torch.optim.AdamW arithmetic (HF `adamw_torch`, reference config/args.py:78) as one fused kernel per
    span: fp32 master / m / v, bf16 working copy refreshed in the same pass."""

    def __init__(self, gb: GradBuffer, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.gb = gb
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        dev = gb.flat.device
        self.master = torch.empty(gb.numel, device=dev, dtype=torch.float32)
        self.m = torch.zeros(gb.numel, device=dev, dtype=torch.float32)
        self.v = torch.zeros(gb.numel, device=dev, dtype=torch.float32)
        self.step_count = 0
        self._scratch = {}
        self._stream = None
        for (kind, obj, n), off in zip(gb.spans, gb.offsets):
            src = obj.w if kind == "w" else obj.b if kind == "b" else obj.data
            self.master[off:off + n].copy_(src.reshape(-1).float())

    def _update(self, kind, obj, off, n, lr, grad_scale, zero_grad):
        tgt = obj.w if kind == "w" else obj.b if kind == "b" else obj.data
        if tgt.dtype == BF16:
            pb = tgt
        else:                                   # fp32 parameter (router wg): bf16 copy goes to scratch
            pb = self._scratch.setdefault(n, torch.empty(n, device=tgt.device, dtype=BF16))
        K.adamw_step(self.master[off:off + n], pb, self.gb.flat[off:off + n], self.m[off:off + n],
                     self.v[off:off + n], lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count,
                     grad_scale, zero_grad=zero_grad)
        if tgt.dtype != BF16:
            tgt.reshape(-1).copy_(self.master[off:off + n])
        if kind == "w":
            obj._wt_version = None      # the kernel wrote behind torch's back: invalidate the cached W^T

    def step(self, grad_scale=1.0, lr=None, overlap=False, clear_grads=False):
        """One AdamW update of every span.  overlap=True is the just-in-time form: the fused weights (99.9 % of the
        bytes) are updated on the optimizer's own stream in the order the next forward will touch them, each one
        publishing an event that `FusedWeight.ensure()` waits on, so this HBM-bound pass runs under the next step's
        forward GEMMs instead of in front of them; every span also clears its gradient in the same pass (GradBuffer.zero()
        then skips the memset).  Arithmetic and results are identical to the serial form.  Call `sync()` before reading
        weights outside a forward."""
        self.step_count += 1
        lr = self.lr if lr is None else lr
        spans = list(zip(self.gb.spans, self.gb.offsets))
        if not overlap:
            # clear_grads: every span zeroes its gradient in the same pass, so the next GradBuffer.zero() skips its
            # 8 GB memset (optimizer.zero_grad() folded into the update)
            for (kind, obj, n), off in spans:
                self._update(kind, obj, off, n, lr, grad_scale, clear_grads)
            if clear_grads:
                self.gb.clean = True
            return
        deferred, inline = {}, []
        for (kind, obj, n), off in spans:
            if kind in ("w", "b") and obj._use_seq is not None:
                deferred.setdefault(id(obj), (obj, []))[1].append((kind, off, n))
            else:
                inline.append((kind, obj, off, n))
        for kind, obj, off, n in inline:                     # small / not-yet-ordered spans: in line
            self._update(kind, obj, off, n, lr, grad_scale, True)
        cur = torch.cuda.current_stream()
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.gb.flat.device, priority=cur.priority)
        self._stream.wait_stream(cur)                        # gradients (and their all-reduce) are final on `cur`
        with torch.cuda.stream(self._stream):
            for obj, parts in sorted(deferred.values(), key=lambda t: t[0]._use_seq):
                for kind, off, n in parts:
                    self._update(kind, obj, off, n, lr, grad_scale, True)
                ev = torch.cuda.Event()
                ev.record(self._stream)
                obj._ready = ev
        self.gb.clean = True

    def sync(self):
        """Make the current stream wait for an overlapped step (before reading weights outside a forward)."""
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)


def warmup_cosine(step, total_steps, base_lr, warmup_ratio=0.03):
    """`--lr_scheduler_type cosine --warmup_ratio 0.03` (dense2sparse_distillation.sh:78-80)."""
    warm = max(1, int(math.ceil(total_steps * warmup_ratio)))
    if step < warm:
        return base_lr * (step + 1) / warm
    prog = (step - warm) / max(1, total_steps - warm)
    return base_lr * 0.5 * (1.0 + math.cos(math.pi * min(1.0, prog)))


def init_distributed():
    """One process per GPU; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the launcher.  backend "nccl" is RCCL."""
    if "RANK" not in os.environ or (int(os.environ.get("WORLD_SIZE", "1")) <= 1 and not os.environ.get("LMOD_FORCE_DIST")):
        if torch.cuda.is_available():
            torch.cuda.set_device(0)
        return 0, 0, 1
    rank, local, world = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        # RCCL's communication streams at HIGH priority: the step's compute stream is a high-priority stream too (the
        # prefetched teacher pass runs below both), and a default-priority all-reduce would starve behind it
        opts = None
        try:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except Exception:                                     # older torch builds: fall back to the environment switch
            os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), pg_options=opts)
    else:
        dist.init_process_group("gloo")
    return rank, local, world
