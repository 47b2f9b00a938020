"""`MoE` — drop-in for `deepspeed.moe.layer.MoE` as the reference uses it (constructor call
llava_qwen2_moe.py:536-546; result 3-tuple consumed at :161-167; `.deepspeed_moe.experts.deepspeed_experts`
read at :547; router parameter name contains `wg`, matched by substring at :501-506,619-626).

Semantics restate DeepSpeed 0.9.5 (top-1/top-2 gating with capacity, token-order slots, drops,
renormalised top-2 weights, l_aux) but execute sparsely on HIP kernels: see llavamod.ops.MoEBlock.
Expert parallelism: `ep_size` > 1 shards the experts over ranks and exchanges capacity slabs with
RCCL all-to-alls (`MoE._forward_expert_parallel` below over `llavamod.engine.expert_parallel_group`; gradients of the
sharded experts are reduced over `engine.expert_data_parallel_group`); ep_size == 1 is the reference shells' setting.
"""
import copy
import math
from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import ops
from ..ops import FusedWeight


class _Gate(nn.Module):
    def __init__(self, hidden, num_experts, device):
        super().__init__()
        # TopKGate keeps wg in fp32 and feeds it x.float() (sharded_moe.TopKGate.forward)
        self.wg = nn.Linear(hidden, num_experts, bias=False, device=device, dtype=torch.float32)


class _Experts(nn.Module):
    def __init__(self, expert, n_local, ep_size):
        super().__init__()
        self.deepspeed_experts = nn.ModuleList([copy.deepcopy(expert) for _ in range(n_local)])
        for e in self.deepspeed_experts:
            for p in e.parameters():
                p.allreduce = False          # deepspeed.moe.experts.Experts tags expert params this way
                p.group_name = f"ep_size_{ep_size}"


class _DSMoE(nn.Module):
    def __init__(self, hidden, expert, num_experts, ep_size, device):
        super().__init__()
        self.gate = _Gate(hidden, num_experts, device)
        self.experts = _Experts(expert, num_experts // ep_size, ep_size)   # num_local_experts (layer.MoE.__init__)


class MoE(nn.Module):
    def __init__(self, hidden_size, expert, num_experts=1, ep_size=1, k=1, capacity_factor=1.0,
                 eval_capacity_factor=1.0, min_capacity=4, use_residual=False, noisy_gate_policy=None,
                 drop_tokens=True, use_rts=True, use_tutel=False, enable_expert_tensor_parallelism=False):
        super().__init__()
        if k not in (1, 2):
            raise ValueError("Only top-1 and top-2 gatings are supported (DeepSpeed 0.9.5 TopKGate)")
        if num_experts % ep_size:
            raise ValueError(f"Number of experts ({num_experts}) should be divisible by expert parallel size ({ep_size})")
        if num_experts > 32:
            # csrc/moe.hip keeps one value per expert in registers: instantiations for 8 / 16 / 32 slots (LMOD_MAX_EXPERTS)
            raise NotImplementedError("the routing kernels are compiled for <= 32 experts per layer")
        self.hidden_size, self.num_experts, self.ep_size, self.k = hidden_size, num_experts, ep_size, k
        self.capacity_factor, self.eval_capacity_factor, self.min_capacity = capacity_factor, eval_capacity_factor, min_capacity
        self.use_rts = use_rts
        device = next(expert.parameters()).device
        self.num_local_experts = num_experts // ep_size
        self.deepspeed_moe = _DSMoE(hidden_size, expert, num_experts, ep_size, device)
        ex = self.deepspeed_moe.experts.deepspeed_experts
        self._gu = FusedWeight([[e.gate_proj.weight, e.up_proj.weight] for e in ex])
        self._down = FusedWeight([[e.down_proj.weight] for e in ex])
        self._gu.is_expert = self._down.is_expert = True        # engine: not all-reduced across the EP group
        self.use_residual = use_residual
        if use_residual:            # Residual-MoE (layer.MoE.__init__): a dense copy of the expert + a 2-way mixing head
            self.mlp = copy.deepcopy(expert)
            self.coefficient = nn.Linear(hidden_size, 2, device=device, dtype=next(expert.parameters()).dtype)
        self.ep_group = None        # torch.distributed group of size ep_size (engine.expert_parallel_group)
        self.force_decomposed = False   # tests: run the EP code path with ep_size == 1 (identity exchange)
        # Gating noise.  Reference: top-2 adds Gumbel noise to the logits for the 2nd pick (gumbel_rsample); top-1 with
        # use_rts=True (the DeepSpeed default) draws uniform priorities for random token selection.  Here both are drawn
        # INSIDE the gating kernel (Philox keyed by torch's seed, this layer's id and a call counter): no torch math on the
        # MoE path.  gate_noise: explicit [T, E] noise for the next forward instead (parity tests feed the oracle the same).
        self.gate_noise = None
        self.deterministic = False  # True: no noise at all (top-2: plain 2nd argmax; top-1: token-order selection)
        # Noise stream identity: (torch seed, layer id, call counter).  The layer id is the layer's INDEX in its model
        # (`_build_moe_layers` sets it; a layer built on its own takes a construction counter), so it does not depend on
        # what else the process built before; training and eval forwards draw from separate streams with separate
        # counters (eval / generate calls between train steps do not shift the training noise); the counters are saved
        # and restored with the optimizer state (`noise_state`), so a resumed run continues the stream instead of
        # replaying step 0's noise.
        MoE._LAYERS[0] += 1
        self._layer_id = MoE._LAYERS[0]
        self._calls = 0
        self._eval_calls = 0

    _LAYERS = [0]

    def noise_state(self):
        return {"layer_id": self._layer_id, "calls": self._calls, "eval_calls": self._eval_calls}

    def load_noise_state(self, st):
        self._calls, self._eval_calls = int(st["calls"]), int(st.get("eval_calls", 0))

    def _noise(self, T, device):
        """(explicit noise tensor or None, Philox seed or None, counter offset) of this forward."""
        if self.deterministic or (self.k == 1 and not self.use_rts):
            return None, None, 0
        if self.gate_noise is not None:
            return self.gate_noise.to(device=device, dtype=torch.float32).contiguous(), None, 0
        seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._layer_id * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        if self.training:
            off = self._calls * (1 << 24)        # a fresh counter range per call (T < 2^24 tokens)
            self._calls += 1
        else:                                    # DeepSpeed's gate draws noise in eval too: its own stream
            seed ^= 0xA5A5A5A55A5A5A5A
            off = self._eval_calls * (1 << 24)
            self._eval_calls += 1
        return None, seed, off

    def capacity(self, T):
        cf = self.capacity_factor if self.training else self.eval_capacity_factor
        if self.k == 2:
            cf = cf * 2
        return max(int(math.ceil((T / self.num_experts) * cf)), int(self.min_capacity), 1)

    def fused_weights(self):
        return [self._gu, self._down]

    def trainable(self):
        return [p for p in self.parameters() if p.requires_grad]

    def forward(self, hidden_states, rt=None):
        shp = hidden_states.shape
        x = hidden_states.reshape(-1, shp[-1])
        T = x.shape[0]
        wg = self.deepspeed_moe.gate.wg.weight
        noise, seed, off = self._noise(T, x.device)
        spec = SimpleNamespace(E=self.num_experts, k=self.k, capacity=self.capacity, wg=wg,
                               gu=self._gu.ensure(), down=self._down.ensure(), seed=seed, offset=off)
        expert_side = [q for q in self.deepspeed_moe.parameters() if q.requires_grad]
        if self.ep_size == 1 and not self.force_decomposed:
            out, l_aux, counts = ops.MoEBlock.apply(x, spec, noise, *expert_side)
        else:
            out, l_aux, counts = self._forward_expert_parallel(x, spec, noise)
        self.last_state = spec.last_state          # routing maps of this call (inspection / tests)
        if self.use_residual:                      # out * coef[..., 0:1] + mlp(x) * coef[..., 1:]  (layer.MoE.forward)
            res = self.mlp(x)
            cw = [q for q in self.coefficient.parameters() if q.requires_grad]
            out = ops.ResidualMix.apply(out, res, x, self.coefficient, *cw)
        return out.reshape(shp), l_aux.reshape(()), counts

    def _forward_expert_parallel(self, x, spec, noise):
        """MOELayer.forward with its two all-to-alls (deepspeed.moe.sharded_moe): route locally over all E experts,
        exchange the capacity slabs, run the local experts on every source rank's slab, exchange back, combine.

        What travels (`ep_live_rows`, default): only the LIVE rows of each slab.  The ranks first exchange their per-expert
        live counts (`slots_used`, ep*E_local ints), then one unequal-split `all_to_all_single` carries the packed live
        rows each way.  With top-2 of E experts at capacity factor 1.5 a slab holds C = 3T/E slots but on average 2T/E
        routed tokens, so about a third of DeepSpeed's [E, C, H] bytes were dead slots; with imbalanced routers the busiest
        expert still ships at most C rows.  Cost: one host read-back of 2*E counts per MoE layer per direction of the step
        (DeepSpeed's own gate syncs `exp_counts` to the host at the same point).  `ep_live_rows = False` (or
        LMOD_EP_FULL_SLABS=1) ships whole [E_local*C, H] slabs like the reference; results are identical
        (tests/test_dp_gloo.py, tests/test_moe_ep_gpu.py)."""
        import os
        import torch.distributed as dist
        H = x.shape[1]
        ep, El = self.ep_size, self.num_local_experts
        if ep > 1 and self.ep_group is None:
            from ..engine import expert_parallel_group
            self.ep_group = expert_parallel_group(ep)
        group = self.ep_group if ep > 1 else None
        if group is None and os.environ.get("LMOD_FORCE_DIST") == "1" and dist.is_available() and dist.is_initialized():
            group = dist.group.WORLD      # a world of ONE rank: the exchange is the identity but goes through the backend's own
                                          # all_to_all_single (RCCL on a GPU box) — the N > 1 code path on the hardware there is
        router_params = [wg_p for wg_p in [self.deepspeed_moe.gate.wg.weight] if wg_p.requires_grad]
        disp, w1, w2, l_aux, counts = ops.MoERoute.apply(x, spec, noise, *router_params)
        st = spec.last_state
        C = st.C
        rows = st.slots_used.view(ep, El)
        if group is not None:          # live rows of each incoming slab (tiny int exchange)
            rr = torch.empty_like(rows)
            dist.all_to_all_single(rr, rows.contiguous(), group=group)
        else:
            rr = rows
        expert_params = [q for q in self.deepspeed_moe.experts.parameters() if q.requires_grad]
        live = getattr(self, "ep_live_rows", True) and os.environ.get("LMOD_EP_FULL_SLABS") != "1"
        if not live:
            recv = ops.AllToAll.apply(disp.view(ep, El * C, H), group)
            y = ops.ExpertFFN.apply(recv.view(ep, El, C, H), spec, rr.contiguous(), *expert_params)
            back = ops.AllToAll.apply(y.view(ep, El * C, H), group)
            return ops.MoECombine.apply(back.reshape(self.num_experts * C, H), w1, w2, st), l_aux, counts
        host = torch.stack([rows.reshape(-1), rr.reshape(-1)]).cpu().numpy()      # the one host sync of the exchange
        pl = ops.ep_live_row_plan(host[0].reshape(ep, El), host[1].reshape(ep, El), C, x.device)
        self.last_ep_plan = pl
        packed = ops.RowGather.apply(disp, pl.send_idx, pl.send_inv)                         # [Ls, H] live rows only
        nchunk = int(getattr(self, "ep_chunks", 0) or os.environ.get("LMOD_EP_CHUNKS", "1"))
        if El == 1 and nchunk > 1 and group is not None:
            # ONE local expert, pipelined: the exchange of chunk c+1 under the expert GEMMs of chunk c, both directions, forward and
            # backward (ops.chunked_expert_exchange).  Every chunk runs the block (or its empty stand-in) so that the weight-gradient
            # hooks fire the same number of times on every rank.
            blk = lambda rows: (ops.MLPBlock if rows.shape[0] > 0 else ops.EmptyExpertPass).apply(rows, spec, *expert_params)
            back_p = ops.chunked_expert_exchange(packed, pl.in_splits, pl.out_splits, group, blk, nchunk)
            back = ops.RowGather.apply(back_p, pl.send_inv, pl.send_idx)
            return ops.MoECombine.apply(back, w1, w2, st), l_aux, counts
        recv_p = ops.AllToAllRows.apply(packed, pl.in_splits, pl.out_splits, group)          # [Lr, H]
        if El == 1:
            # ONE local expert (config 5: 8 experts on 8 ranks): the packed live rows ARE its input — no capacity slabs on the
            # receiving side, no row masks, no zero-filled dead rows: the dense fused-SwiGLU block on [Lr, H] (plain GEMM launches,
            # K = Lr weight gradients).  A rank that received nothing launches nothing, but keeps the block's place in the backward
            # (`ops.EmptyExpertPass`: the same wgrad-ready hooks in the same order, so the expert-data-parallel collectives are
            # issued identically on every rank of the group; its expert's gradient span stays zero).
            y_p = (ops.MLPBlock if recv_p.shape[0] > 0 else ops.EmptyExpertPass).apply(recv_p, spec, *expert_params)
        else:
            recv = ops.RowGather.apply(recv_p, pl.recv_slab, pl.recv_inv)                    # slabs [ep*El*C, H]
            y = ops.ExpertFFN.apply(recv.view(ep, El, C, H), spec, rr.contiguous(), *expert_params)
            y_p = ops.RowGather.apply(y.view(ep * El * C, H), pl.recv_inv, pl.recv_slab)     # live outputs [Lr, H]
        back_p = ops.AllToAllRows.apply(y_p, pl.out_splits, pl.in_splits, group)             # [Ls, H]
        back = ops.RowGather.apply(back_p, pl.send_inv, pl.send_idx)                         # [E*C, H], zero on dead slots
        out = ops.MoECombine.apply(back, w1, w2, st)
        return out, l_aux, counts
