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


# name -> [calls, payload bytes handed to the collective on this rank] since process start (bench.py reports them per step
# in its `exchange` object, next to the static plan: evidence of what RCCL was actually asked to move)
COMM = {}


def comm_count(name, tensor):
    c = COMM.setdefault(name, [0, 0])
    c[0] += 1
    c[1] += tensor.numel() * tensor.element_size()


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
                # parameters without a fused weight: router, Residual-MoE head, decoder norm scales, the embedding table
                # (all with a weight-gradient kernel).  Anything else (the CLIP tower, which the reference keeps frozen under
                # no_grad, clip_encoder.py:31,45) has no gradient on this path: fail loudly, not silently.
                ok = (n.endswith("wg.weight") or "wg." in n or "coefficient." in n or n.endswith("embed_tokens.weight")
                      or (("layernorm.weight" in n or n.endswith("model.norm.weight") or n.endswith(".norm.weight"))
                          and "image_tower" not in n))
                if not ok:
                    raise NotImplementedError(
                        f"parameter {n!r} is marked trainable but this path computes no gradient for it (supported: "
                        f"attention / MLP / expert / projector / lm_head linears, MoE router, Residual-MoE coefficient head, decoder "
                        f"norm scales, embed_tokens); freeze it or list only supported modules in train_modules")
                spans.append(("p", p, p.numel()))
                covered.add(id(p))
        # dense (replicated) parameters first, expert-parallel-sharded expert weights last: two contiguous regions
        is_exp = lambda sp: bool(getattr(sp[1], "is_expert", False)) and sp[0] == "w"
        spans = [sp for sp in spans if not is_exp(sp)] + [sp for sp in spans if is_exp(sp)]
        self.n_dense = sum((n + 7) // 8 * 8 for sp in spans if not is_exp(sp) for n in [sp[2]])
        self.spans = spans
        al = lambda x: (x + 7) // 8 * 8     # span starts on 8-element boundaries: 16-byte vector access in fp32 and bf16
        total = sum(al(n) for _, _, n in spans)
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
            off += al(n)
        self.numel = total

    def flatten_params(self):
        """Move every bf16 span's working weights into ONE flat bf16 buffer laid out like the gradient buffer (the
        nn.Parameters stay views of it): the in-place all-gather target of the ZeRO-2 style optimizer."""
        if getattr(self, "pflat", None) is not None:
            return self.pflat
        self.pflat = torch.zeros(self.numel, device=self.flat.device, dtype=BF16)
        for (kind, obj, n), off in zip(self.spans, self.offsets):
            view = self.pflat[off:off + n]
            if kind == "w":
                new = view.view(obj.w.shape)
                new.copy_(obj.w)
                for p, v in obj._views(new):
                    p.data = v
                obj.w, obj.wt, obj._wt_version = new, None, None
            elif kind == "b":
                new = view.view(obj.b.shape)
                new.copy_(obj.b)
                r = 0
                for bp in obj.bias_groups[0]:
                    bp.data = new[r:r + bp.shape[0]]
                    r += bp.shape[0]
                obj.b = new
            elif obj.dtype == BF16:
                new = view.view(obj.shape)
                new.copy_(obj.data)
                obj.data = new
        return self.pflat

    def zero(self):
        if getattr(self, "clean", False):        # the optimizer cleared every span in its own pass (HipAdamW overlap)
            self.clean = False
        else:
            self.flat.zero_()
        for kind, obj, _ in self.spans:          # forget uses that never saw a backward (eval-style forwards)
            if kind == "w":
                obj.pending = 0


class DataParallel:
    """Sample-sharded data parallelism: every rank runs teacher + student on its own micro-batches; the only exchange is
    the SUM of the flat gradient buffer at the optimizer boundary (the division by world size is folded into AdamW's grad
    scale).  Loss normalisation stays per-rank, as in the reference (align_trainer.py:526 divides by the local mask sum;
    DeepSpeed then averages).

    Two exchange regimes:
      * all-reduce (zero2=False): every rank ends with the full summed gradient and runs the full optimizer.
      * ZeRO-2 style (zero2=True; the reference's regime, config/dpconfig/zero2_offload.json): a span is REDUCE-SCATTERED
        in place — rank r of the span's group ends with the sum of chunk r only — `HipAdamW` updates that chunk of the fp32
        master / m / v (1/N of the optimizer state and work per rank) and the updated bf16 weights are ALL-GATHERED in
        place into the flat parameter buffer.  Per optimizer step a rank moves (N-1)/N x (grad bytes + bf16 param bytes)
        instead of 2 (N-1)/N x grad bytes.  Spans too small or not divisible by the group size stay replicated.
      grad_dtype=torch.bfloat16 exchanges gradients in bf16 (what the reference's bf16 engine does): the fp32 span is cast
      into a bf16 staging buffer, reduced there, and the owned chunk cast back.

    Overlap: `attach(gb)` hooks every fused weight; as soon as a weight's last wgrad GEMM of the backward has been enqueued
    its span is exchanged asynchronously (RCCL runs on its own stream, ordered after the compute stream by an event), so
    the exchange hides under the rest of backward.  `finish()` sends whatever has no hook (router weights, lone biases)
    and waits.  Expert weights sharded over an expert-parallel group are reduced only over the ranks that hold the same
    experts.  `armed=False` (non-final micro-batches of a gradient-accumulation window) defers everything to `finish()`
    of the final micro-batch."""

    def __init__(self, bucket_bytes=512 << 20, overlap=True, zero2=False, grad_dtype=torch.float32, min_shard_numel=1 << 16,
                 native=None):
        # LMOD_DP_FORCE=1: run the whole exchange path (RCCL collectives, shard plan, all-gather) even in a world of ONE rank —
        # how the N>1 code is exercised on a single-GPU box (tests/test_step_parity_gpu.py)
        self.enabled = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size() > 1 or os.environ.get("LMOD_DP_FORCE") == "1")
        self.world = dist.get_world_size() if self.enabled else 1
        self.rank = dist.get_rank() if self.enabled else 0
        self.bucket = bucket_bytes // 4
        self.overlap = overlap
        self.zero2 = bool(zero2) and self.enabled
        self.grad_dtype = grad_dtype
        self.min_shard = min_shard_numel
        # native: exchanges over the WORLD go through the C-ABI collectives of the kernel library (csrc/comm.hip: lmod_allreduce_grads /
        # lmod_reduce_scatter_grads / lmod_allgather_params, one RCCL communicator created from an id that rank 0 broadcasts) on a side
        # stream ordered behind the compute stream by an event — the boundary a host without torch.distributed binds,
        # here driven by the same engine.  Sub-group exchanges (expert-data-parallel spans) stay on torch.distributed.  LMOD_DP_NATIVE=1.
        self.native = (os.environ.get("LMOD_DP_NATIVE") == "1") if native is None else bool(native)
        self._ncomm = self._nstream = None
        self.armed = True                 # set False on non-final micro-batches of a gradient-accumulation window
        self._handles, self._done, self._map, self._bias = [], set(), {}, {}   # _done holds (id(obj), kind) of sent spans
        self._castback = []               # bf16 exchange: (lo, hi) chunks to cast back to fp32 once the collective is done
        self.gb, self.ep_size, self.plan, self._gbf = None, 1, {}, None

    # ---- layout ----------------------------------------------------------------------------------------------------
    def _group_of(self, is_expert):
        """(process group or None for the world, group size, this rank's index in it); None if nothing to exchange."""
        if is_expert and self.ep_size > 1:
            gsize = self.world // self.ep_size
            if gsize <= 1:
                return None
            return expert_data_parallel_group(self.ep_size), gsize, self.rank // self.ep_size
        return None, self.world, self.rank

    def attach(self, gb, ep_size=1):
        self.gb, self.ep_size = gb, ep_size
        for (kind, obj, n), off in zip(gb.spans, gb.offsets):
            is_exp = kind == "w" and bool(getattr(obj, "is_expert", False))
            grp = self._group_of(is_exp) if self.enabled else None
            tgt = obj.w if kind == "w" else obj.b if kind == "b" else obj
            info = dict(off=off, n=n, is_expert=is_exp, group=None, gsize=1, grank=0, sharded=False, lo=off, hi=off + n)
            if grp is not None:
                g, gsize, grank = grp
                sharded = self.zero2 and tgt.dtype == BF16 and n % (8 * gsize) == 0 and n >= self.min_shard   # chunks stay 16-byte aligned
                info.update(group=g, gsize=gsize, grank=grank, sharded=sharded)
                if sharded:
                    c = n // gsize
                    info.update(lo=off + grank * c, hi=off + (grank + 1) * c)
            info["exchange"] = grp is not None
            self.plan[(id(obj), kind)] = info
            if kind == "w":
                self._map[id(obj)] = (off, n, is_exp)
                obj.grad_ready_hook = self._on_ready
            elif kind == "b":                 # a fused weight's bias span travels with its weight span (same ready hook)
                self._bias[id(obj)] = (off, n)
        if self.zero2:
            gb.flatten_params()
        if self.enabled and self.grad_dtype == BF16:
            self._gbf = torch.empty(gb.numel, device=gb.flat.device, dtype=BF16)
        if self.enabled and getattr(self, "native", False) and gb.flat.is_cuda:
            self.native_comm()                   # eagerly, HERE: every rank attaches at the same point of its set-up (the id broadcast is a
                                                 # collective); creating it lazily inside an autograd hook is the deadlock hazard ADVICE r05 names
        return self

    def owned(self, obj, kind):
        """[lo, hi) of the flat buffers this rank's optimizer updates for the span, and whether it is the rank that
        counts the span in global reductions (a replicated span is counted once per group)."""
        info = self.plan.get((id(obj), kind))
        if info is None:
            off = self.gb.offsets[[i for i, sp in enumerate(self.gb.spans) if sp[1] is obj and sp[0] == kind][0]]
            n = [sp[2] for sp in self.gb.spans if sp[1] is obj and sp[0] == kind][0]
            return off, off + n, True
        return info["lo"], info["hi"], (info["sharded"] or info["grank"] == 0)

    # ---- exchange --------------------------------------------------------------------------------------------------
    def native_comm(self):
        """The process's C-ABI communicator over the world (created on first use: rank 0 draws the id, everybody receives it over the
        existing process group) and the side stream its collectives are enqueued on."""
        if self._ncomm is None:
            from . import comm
            self._ncomm = comm.shared_world_comm()       # ONE communicator per process, shared with ops.AllToAllRows
            self._nstream = torch.cuda.Stream()
        return self._ncomm, self._nstream

    class _StreamHandle:
        """`wait()` of a collective enqueued on the native side stream: the current stream waits for everything enqueued there so far."""

        def __init__(self, stream):
            self.ev = torch.cuda.Event()
            self.ev.record(stream)

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def _send_native(self, flat, info):
        nc, ns = self.native_comm()
        off, n = info["off"], info["n"]
        ev = torch.cuda.Event()
        ev.record()                                           # behind the weight-gradient GEMMs enqueued so far on the compute stream
        with torch.cuda.stream(ns):
            ns.wait_event(ev)
            if info["sharded"]:
                comm_count("reduce_scatter", flat[off:off + n])
                nc.reduce_scatter_(flat[off:off + n])         # chunk `rank` of the sum lands in place at off + rank * n / world = info["lo"]
            else:
                for a in range(off, off + n, self.bucket):
                    comm_count("all_reduce", flat[a:min(a + self.bucket, off + n)])
                    nc.allreduce_(flat[a:min(a + self.bucket, off + n)])
        self._handles.append(DataParallel._StreamHandle(ns))

    def _send(self, obj, kind):
        info = self.plan[(id(obj), kind)]
        self._done.add((id(obj), kind))
        if not info["exchange"]:
            return
        off, n, g = info["off"], info["n"], info["group"]
        flat = self.gb.flat
        if self.grad_dtype == BF16:
            K.cast_f32_bf16(flat[off:off + n], self._gbf[off:off + n])
            flat = self._gbf
            self._castback.append((info["lo"], info["hi"]))
        if self.native and g is None and flat.is_cuda:
            return self._send_native(flat, info)
        if info["sharded"]:
            comm_count("reduce_scatter", flat[off:off + n])
            self._handles.append(dist.reduce_scatter_tensor(flat[info["lo"]:info["hi"]], flat[off:off + n],
                                                            op=dist.ReduceOp.SUM, group=g, async_op=True))
        else:
            for a in range(off, off + n, self.bucket):
                comm_count("all_reduce", flat[a:min(a + self.bucket, off + n)])
                self._handles.append(dist.all_reduce(flat[a:min(a + self.bucket, off + n)], op=dist.ReduceOp.SUM, group=g,
                                                     async_op=True))

    def _on_ready(self, fw):
        if not (self.enabled and self.overlap and self.armed) or (id(fw), "w") in self._done:
            return
        self._send(fw, "w")
        if id(fw) in self._bias:              # linear_wgrad adds the bias gradient before it calls grad_done()
            self._send(fw, "b")

    def finish(self):
        """Exchange every span that was not already sent by a hook, then wait for all of it."""
        if self.enabled and self.armed:
            for kind, obj, _ in self.gb.spans:
                if (id(obj), kind) not in self._done:
                    self._send(obj, kind)
            for h in self._handles:
                h.wait()
            for lo, hi in self._castback:
                K.cast_f32_bf16(self._gbf[lo:hi], self.gb.flat[lo:hi])
        self._handles, self._done, self._castback = [], set(), []

    def exchange_plan(self):
        """Static summary of what one optimizer step exchanges on this rank (from the span plan alone — computable on `meta`
        tensors): per collective the number of calls and the payload bytes this rank hands to RCCL."""
        gsz = 2 if self.grad_dtype == BF16 else 4
        out = {"reduce_scatter": [0, 0], "all_reduce": [0, 0], "all_gather": [0, 0]}
        sharded_elems = replicated_elems = local_elems = 0
        for (kind, obj, n) in self.gb.spans:
            info = self.plan[(id(obj), kind)]
            if not info["exchange"]:
                local_elems += n
                continue
            if info["sharded"]:
                out["reduce_scatter"][0] += 1
                out["reduce_scatter"][1] += n * gsz
                out["all_gather"][0] += 1
                out["all_gather"][1] += n * 2                  # bf16 working weights, in place
                sharded_elems += n
            else:
                out["all_reduce"][0] += -(-n // self.bucket)
                out["all_reduce"][1] += n * gsz
                replicated_elems += n
        return {"collectives_per_step": {k: {"calls": v[0], "bytes": v[1]} for k, v in out.items()},
                "spans": len(self.gb.spans), "sharded_params": sharded_elems, "replicated_params": replicated_elems,
                "rank_local_params": local_elems,
                "optimizer_state_params_this_rank": sum(i["hi"] - i["lo"] for i in self.plan.values())}

    def all_reduce(self, flat, n_dense=None, ep_size=1):
        """Non-overlapped all-reduce form: SUM over the DP world for the first n_dense elements (replicated parameters);
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
    """torch.optim.AdamW arithmetic (HF `adamw_torch`, reference config/args.py:78) as one fused kernel per span: fp32
    master / m / v, bf16 working copy refreshed in the same pass.

    dp (a `DataParallel` with zero2=True): optimizer state exists only for the chunks this rank owns after the
    reduce-scatter; after the update the bf16 chunks are all-gathered in place into `gb.pflat` (ZeRO-2 partitioning,
    the reference's regime — config/dpconfig/zero2_offload.json — without the CPU offload: 288 GB of HBM hold it).
    max_grad_norm: global-norm gradient clipping with torch.nn.utils.clip_grad_norm_ arithmetic (HF Trainer's
    max_grad_norm = 1.0 reaches DeepSpeed through `"gradient_clipping": "auto"`); norm and coefficient stay on the device."""

    def __init__(self, gb: GradBuffer, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, dp=None, max_grad_norm=None):
        self.gb, self.dp = gb, dp
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        dev = gb.flat.device
        if dp is not None and dp.enabled and dp.gb is not gb:
            # the shard layout (who owns which chunk of the master / m / v) comes from dp.attach(gb): an optimizer built
            # before it would keep full-span state and update it from reduce-scattered (partially un-reduced) gradients
            raise RuntimeError("HipAdamW(dp=...): call dp.attach(gb) on THIS GradBuffer before constructing the optimizer")
        self._weights_epoch = getattr(gb.model, "_weights_epoch", 0)
        self.items = []                         # (kind, obj, lo, hi, state offset, counted in global reductions)
        soff = 0
        for (kind, obj, n), off in zip(gb.spans, gb.offsets):
            lo, hi, counted = (dp.owned(obj, kind) if dp is not None and dp.gb is gb else (off, off + n, True))
            self.items.append((kind, obj, lo, hi, soff, counted, off))
            soff += hi - lo
        self.n_state = soff
        self.master = torch.empty(soff, device=dev, dtype=torch.float32)
        self.m = torch.zeros(soff, device=dev, dtype=torch.float32)
        self.v = torch.zeros(soff, device=dev, dtype=torch.float32)
        self.step_count = 0
        self._scratch = {}
        self._stream = None
        self._ss = torch.zeros(1, device=dev, dtype=torch.float32)
        self._coef = torch.ones(1, device=dev, dtype=torch.float32)
        self._part = torch.empty(1024, device=dev, dtype=torch.float32)
        self.grad_norm = torch.zeros(1, device=dev, dtype=torch.float32)      # norm of the MEAN gradient at the last step
        self.resync_master()

    @staticmethod
    def _target(kind, obj):
        return obj.w if kind == "w" else obj.b if kind == "b" else obj.data

    def resync_master(self):
        """fp32 masters <- current weights (construction; after a checkpoint was loaded into the model)."""
        self._masters_from_file = False
        self._weights_epoch = getattr(self.gb.model, "_weights_epoch", 0)
        for kind, obj, lo, hi, soff, _, off in self.items:
            src = self._target(kind, obj).reshape(-1)[lo - off:hi - off]
            self.master[soff:soff + hi - lo].copy_(src.float())

    def _update(self, kind, obj, lo, hi, soff, off, lr, grad_scale, zero_grad, dev_scale):
        tgt = self._target(kind, obj)
        n = hi - lo
        if n == 0:
            return
        if tgt.dtype == BF16:
            pb = tgt.reshape(-1)[lo - off:hi - off]
        else:                                   # fp32 parameter (router wg): bf16 copy goes to scratch
            pb = self._scratch.setdefault(n, torch.empty(n, device=tgt.device, dtype=BF16))
        K.adamw_step(self.master[soff:soff + n], pb, self.gb.flat[lo:hi], self.m[soff:soff + n], self.v[soff:soff + n],
                     lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, grad_scale,
                     zero_grad=zero_grad, dev_scale=dev_scale)
        if tgt.dtype != BF16:
            tgt.reshape(-1)[lo - off:hi - off].copy_(self.master[soff:soff + n])
        if kind == "w":
            obj._wt_version = None      # the kernel wrote behind torch's back: invalidate the cached W^T

    def _clip_coef(self, grad_scale):
        """Device-side clipping coefficient for the gradient SUM in gb.flat scaled by grad_scale (= the mean gradient)."""
        first = True
        for kind, obj, lo, hi, soff, counted, off in self.items:
            if counted and hi > lo:
                K.sumsq(self.gb.flat[lo:hi], self._ss, self._part, accumulate=not first)
                first = False
        if first:
            self._ss.zero_()
        if self.dp is not None and self.dp.enabled:
            comm_count("all_reduce_scalar", self._ss)
            dist.all_reduce(self._ss, op=dist.ReduceOp.SUM)
        K.clip_coef(self._ss, grad_scale, self.max_grad_norm, self._coef, self.grad_norm)
        return self._coef

    def step(self, grad_scale=1.0, lr=None, overlap=False, clear_grads=False):
        """One AdamW update of every span this rank owns.  overlap=True (unsharded only) is the just-in-time form: the fused
        weights (99.9 % of the bytes) are updated on the optimizer's own stream in the order the next forward will touch
        them, each one publishing an event that `FusedWeight.ensure()` waits on; every span also clears its gradient in the
        same pass (GradBuffer.zero() then skips the memset).  Arithmetic and results are identical to the serial form.
        Call `sync()` before reading weights outside a forward."""
        ep = getattr(self.gb.model, "_weights_epoch", 0)
        if ep != self._weights_epoch:
            # a checkpoint was loaded into the model after this optimizer took its fp32 masters (checkpoint.load_checkpoint
            # bumps the epoch): stepping from the stale masters would overwrite the just-loaded weights
            if getattr(self, "_masters_from_file", False):
                # ... but THESE masters came from an optimizer file: re-deriving them from the bf16 weights would silently
                # turn an exact resume into an inexact one.  The order is weights first, optimizer second.
                raise RuntimeError("model weights were loaded AFTER load_state_dict() restored the fp32 masters: load the model "
                                   "checkpoint first and the optimizer state second, or call resync_master() to accept "
                                   "bf16-rounded masters")
            self.resync_master()
            self._weights_epoch = ep
        self.step_count += 1
        lr = self.lr if lr is None else lr
        zero2 = self.dp is not None and self.dp.zero2
        dev_scale = self._clip_coef(grad_scale) if self.max_grad_norm else None
        if zero2 or not overlap:
            # clear_grads: every span zeroes its gradient in the same pass, so the next GradBuffer.zero() skips its
            # 8 GB memset (optimizer.zero_grad() folded into the update)
            for kind, obj, lo, hi, soff, _, off in self.items:
                self._update(kind, obj, lo, hi, soff, off, lr, grad_scale, clear_grads and not zero2, dev_scale)
            if zero2:
                self._gather_params()
                if clear_grads:               # chunks of other ranks hold un-reduced partial sums: one memset of the buffer
                    self.gb.flat.zero_()
            if clear_grads:
                self.gb.clean = True
            return
        deferred, inline = {}, []
        for kind, obj, lo, hi, soff, _, off in self.items:
            if kind in ("w", "b") and obj._use_seq is not None:
                deferred.setdefault(id(obj), (obj, []))[1].append((kind, lo, hi, soff, off))
            else:
                inline.append((kind, obj, lo, hi, soff, off))
        for kind, obj, lo, hi, soff, off in inline:          # small / not-yet-ordered spans: in line
            self._update(kind, obj, lo, hi, soff, off, lr, grad_scale, True, dev_scale)
        cur = torch.cuda.current_stream()
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.gb.flat.device, priority=cur.priority)
        self._stream.wait_stream(cur)                        # gradients (and their exchange) are final on `cur`
        with torch.cuda.stream(self._stream):
            for obj, parts in sorted(deferred.values(), key=lambda t: t[0]._use_seq):
                for kind, lo, hi, soff, off in parts:
                    self._update(kind, obj, lo, hi, soff, off, lr, grad_scale, True, dev_scale)
                ev = torch.cuda.Event()
                ev.record(self._stream)
                obj._ready = ev
        self.gb.clean = True

    def _gather_params(self):
        """ZeRO-2: every rank publishes its updated bf16 chunks; in-place all-gather into the flat parameter buffer."""
        handles = []
        for kind, obj, lo, hi, soff, _, off in self.items:
            info = self.dp.plan[(id(obj), kind)]
            if info["sharded"]:
                full = self.gb.pflat[off:off + info["n"]]
                comm_count("all_gather", full)
                if self.dp.native and info["group"] is None and full.is_cuda:      # the C-ABI collective (lmod_allgather_params), in place
                    nc, ns = self.dp.native_comm()
                    ev = torch.cuda.Event()
                    ev.record()
                    with torch.cuda.stream(ns):
                        ns.wait_event(ev)
                        nc.allgather_(full)
                    handles.append(DataParallel._StreamHandle(ns))
                    continue
                handles.append(dist.all_gather_into_tensor(full, self.gb.pflat[lo:hi], group=info["group"], async_op=True))
        for h in handles:
            h.wait()

    # ---- exact resume ------------------------------------------------------------------------------------------------
    def _layout(self):
        return [[kind, int(lo), int(hi), int(soff)] for kind, _, lo, hi, soff, _, _ in self.items]

    def _moes(self):
        return [m for m in self.gb.model.modules() if hasattr(m, "noise_state")]

    def state_dict(self):
        """This rank's optimizer state: fp32 master / m / v of the chunks it owns (all of them without ZeRO-2), the step
        count, the shard layout they belong to, and the MoE layers' gating-noise counters (so a resumed run continues the
        noise stream).  With ZeRO-2 every rank saves its own (`checkpoint.save_optimizer` names the file by rank).  Plain
        tensors, ints, bools, strings, lists and dicts only: the file loads under `torch.load(weights_only=True)`."""
        self.sync()
        return {"step_count": self.step_count, "master": self.master.detach().cpu(), "m": self.m.detach().cpu(),
                "v": self.v.detach().cpu(), "layout": self._layout(), "n_state": self.n_state,
                "world": self.dp.world if self.dp is not None else 1, "rank": self.dp.rank if self.dp is not None else 0,
                "zero2": bool(self.dp is not None and self.dp.zero2), "moe_noise": [m.noise_state() for m in self._moes()]}

    def load_state_dict(self, st):
        """Exact resume.  Everything that decides which numbers the file holds is compared BEFORE anything is copied: the span /
        shard layout, the world size, this rank's index, the ZeRO-2 setting (without it the layout is identical on every rank,
        so a file of another rank or another world size would load silently) and the MoE layer set (count and layer ids)."""
        if st["n_state"] != self.n_state or [list(x) for x in st["layout"]] != self._layout():
            raise ValueError("optimizer state was saved under a different span / shard layout (world size, ZeRO-2 setting "
                             "or trainable set changed): it cannot be resumed exactly")
        mine = {"world": self.dp.world if self.dp is not None else 1, "rank": self.dp.rank if self.dp is not None else 0,
                "zero2": bool(self.dp is not None and self.dp.zero2)}
        for k, v in mine.items():
            if k in st and type(v)(st[k]) != v:
                raise ValueError(f"optimizer state was saved with {k} = {st[k]}, this run has {k} = {v}: not the same shard")
        moes, noise = self._moes(), list(st.get("moe_noise", []))
        if len(noise) != len(moes):
            raise ValueError(f"optimizer state holds gating-noise counters of {len(noise)} MoE layers, the model has {len(moes)}")
        for m, ns in zip(moes, noise):
            if int(ns.get("layer_id", m.noise_state()["layer_id"])) != int(m.noise_state()["layer_id"]):
                raise ValueError(f"gating-noise counters of MoE layer {ns.get('layer_id')} offered to layer {m.noise_state()['layer_id']}")
        self.step_count = int(st["step_count"])
        for name in ("master", "m", "v"):
            getattr(self, name).copy_(st[name].to(getattr(self, name).device))
        for m, ns in zip(moes, noise):
            m.load_noise_state(ns)
        self._weights_epoch = getattr(self.gb.model, "_weights_epoch", 0)    # masters come from the file, not the model
        self._masters_from_file = True

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
    if torch.cuda.is_available() and os.environ.get("LMOD_DIST_BACKEND", "nccl") != "gloo":   # gloo: CPU multi-process tests
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
        if torch.cuda.is_available():          # functional runs with more ranks than GPUs (ranks share devices, exchange through gloo)
            local %= torch.cuda.device_count()
            torch.cuda.set_device(local)
    return rank, local, world
