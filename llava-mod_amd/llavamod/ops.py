"""Autograd shim: fused blocks of the decoder expressed as torch.autograd.Functions whose forward and
backward are sequences of C-ABI kernel launches (llavamod.kernels).  torch supplies the tape, device
memory and the stream — no torch math runs on [T,H]-sized data here.

Weight-gradient convention: trainable weights own an fp32 `main_grad` tensor (views into one flat
buffer, see llavamod.engine.GradBuffer); wgrad GEMMs accumulate straight into it from the GEMM
epilogue and the Function returns None for that input.  That is the layout the RCCL gradient
exchange and the fused AdamW consume.
"""
import os

import torch

from . import kernels as K

BF16 = torch.bfloat16


# ------------------------------------------------------------------------------------------ weights
class FusedWeight:
    """A [N, K] bf16 weight matrix — possibly the row-concatenation of several nn.Parameters (q/k/v,
    gate/up), or with a leading expert dim [E, N, K] — plus the K-contiguous transpose that the dgrad
    GEMM needs (refreshed when the parameters' version counters change, i.e. after an optimizer
    step) and an fp32 main_grad of the same shape.  The nn.Parameters are re-pointed at slices of
    the fused storage, so state_dict names/shapes stay the reference's."""

    def __init__(self, groups, bias_groups=None):
        # groups: list over experts (len 1 for a plain weight) of lists of params concatenated by rows
        self.groups = [list(g) for g in groups]
        self.bias_groups = [list(g) for g in bias_groups] if bias_groups is not None else None
        self.stacked = len(self.groups) > 1
        self.w = self.b = self.wt = None
        self._wt_version = None
        self.main_grad = None
        self.bias_main_grad = None
        self.pending = 0                 # wgrad contributions still to come in the current backward
        self.grad_ready_hook = None      # engine: called once the last contribution has been enqueued
        self._ready = None               # event of an optimizer update still running on the optimizer's stream
        self._use_seq = None             # position in the forward's first-use order (for the just-in-time optimizer)

    def note_use(self):
        """A forward that will later accumulate into main_grad (called by the fused blocks)."""
        self.pending += 1

    def grad_done(self):
        self.pending = max(0, self.pending - 1)
        if self.pending == 0 and self.grad_ready_hook is not None:
            self.grad_ready_hook(self)

    def __deepcopy__(self, memo):
        # deep copies (MoE up-cycling copies the dense FFN per expert) share nothing and start unfused
        import copy
        groups = [[copy.deepcopy(p, memo) for p in g] for g in self.groups]
        biases = [[copy.deepcopy(b, memo) for b in g] for g in self.bias_groups] if self.bias_groups is not None else None
        return FusedWeight(groups, biases)

    def _views(self, buf):
        for e, g in enumerate(self.groups):
            r = 0
            for p in g:
                yield p, (buf[e, r:r + p.shape[0]] if self.stacked else buf[r:r + p.shape[0]])
                r += p.shape[0]

    _SEQ = [0]

    def ensure(self):
        """Called by every forward right before the weight is used: (re)build the fused storage if needed and, if the
        optimizer is still updating this weight on its own stream (engine.HipAdamW overlap), order this stream after it."""
        ev = self._ready
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._ready = None
        if self._use_seq is None:
            FusedWeight._SEQ[0] += 1
            self._use_seq = FusedWeight._SEQ[0]
        p0 = self.groups[0][0]
        ok = self.w is not None and self.w.device == p0.device
        if ok:
            ok = all(p.data_ptr() == v.data_ptr() for p, v in self._views(self.w))
        if not ok:
            rows = sum(p.shape[0] for p in self.groups[0])
            shape = (len(self.groups), rows, p0.shape[1]) if self.stacked else (rows, p0.shape[1])
            w = torch.empty(shape, device=p0.device, dtype=BF16)
            for p, v in self._views(w):
                v.copy_(p.data)
                p.data = v
            self.w, self.wt, self.main_grad = w, None, None
            if self.bias_groups is not None:
                bs = self.bias_groups[0]
                b = torch.empty(rows, device=p0.device, dtype=BF16)
                r = 0
                for bp in bs:
                    b[r:r + bp.shape[0]].copy_(bp.data)
                    bp.data = b[r:r + bp.shape[0]]
                    r += bp.shape[0]
                self.b, self.bias_main_grad = b, None
        return self

    @property
    def params(self):
        return [p for g in self.groups for p in g]

    @property
    def requires_grad(self):
        return any(p.requires_grad for p in self.params)

    @property
    def bias_requires_grad(self):
        return self.bias_groups is not None and any(b.requires_grad for b in self.bias_groups[0])

    def transposed(self):
        ver = tuple(p._version for p in self.params)
        if self.wt is None or self._wt_version != ver:
            self.wt = K.transpose(self.w, out=self.wt)
            self._wt_version = ver
        return self.wt

    def grad_buffer(self):
        if self.main_grad is None:
            self.set_grad_buffer(torch.zeros(self.w.shape, device=self.w.device, dtype=torch.float32))
        return self.main_grad

    def set_grad_buffer(self, buf):
        self.main_grad = buf
        for p, v in self._views(buf):
            p.main_grad = v

    def bias_grad_buffer(self):
        if self.bias_main_grad is None:
            self.set_bias_grad_buffer(torch.zeros(self.b.shape, device=self.b.device, dtype=torch.float32))
        return self.bias_main_grad

    def set_bias_grad_buffer(self, buf):
        self.bias_main_grad = buf
        r = 0
        for bp in self.bias_groups[0]:
            bp.main_grad = buf[r:r + bp.shape[0]]
            r += bp.shape[0]


def _need(ctx):
    """Might this Function's backward run?  (Grad mode is always off inside Function.forward, so under an outer
    torch.no_grad() this can still say True for unfrozen parameters; the saved tensors are then simply dropped.)"""
    return any(ctx.needs_input_grad)


_ONES = {}


def _ones_buf(n, device):
    """ONE flat buffer of ones per device, grown to the largest token count seen (packed / varlen batches change the row count
    nearly every step: a tensor per distinct count would grow device memory without bound over a long run)."""
    key = str(device)
    buf = _ONES.get(key)
    need = 8 * n
    if buf is None or buf.numel() < need:
        buf = _ONES[key] = torch.ones(max(need, 8 * 4096), device=device, dtype=BF16)
    return buf


def _ones(n, device):
    """[8, n] bf16 ones (the bias-gradient GEMM's second operand on transposed copies)."""
    return _ones_buf(n, device)[:8 * n].view(8, n)


def _ones_col(n, device):
    """[n, 8] bf16 ones (the same operand on the tensors as autograd holds them)."""
    return _ones_buf(n, device)[:8 * n].view(n, 8)


def linear_fwd(x, fw, act=0, out=None):
    """y = act(x @ W^T + b)."""
    return K.gemm_nt(x, fw.w, bias=fw.b, act=act, out=out)


def linear_dgrad(dy, fw):
    """dx = dy @ W  (NT GEMM against the cached W^T [K_in, N_pad]).
    Few output tiles over a very long reduction with a nearly empty tail round — the student's lm_head dgrad: [8208 loss
    rows x 2048] over K = 151936 is 33 x 8 = 264 tiles of 256x256, i.e. one full round of the 256 CUs plus 8 workgroups that
    run a second, 3 ms round alone — go through the deterministic split-K entry point into an fp32 image and are cast back
    (6.3 -> ~4 ms)."""
    wt = fw.transposed()
    M, N, Kd = dy.shape[0], wt.shape[0], fw.w.shape[0]
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if Kd >= 32768 and 256 < tiles < 512 and tiles % 256 <= 64 and M >= 256 and N >= 256 and N % 16 == 0:
        acc = torch.zeros(M, N, device=dy.device, dtype=torch.float32)
        K.gemm_wgrad(dy if dy.shape[1] == Kd else dy[:, :Kd], wt if wt.shape[1] == Kd else wt[:, :Kd], acc)
        out = torch.empty(M, N, device=dy.device, dtype=BF16)
        K.cast_f32_bf16(acc.view(-1), out.view(-1))
        return out
    return K.gemm_nt(dy, wt, M=M, N=N, K=Kd, lda=dy.stride(0), ldb=wt.stride(0))


def wgrad_tn():
    """Weight gradients on the tensors as autograd holds them (dW = dY^T X through the TN kernel's transposing LDS reads) instead of
    NT GEMMs on transposed copies.  LMOD_WGRAD_TN=0 restores the copies (A/B runs); read per call."""
    return os.environ.get("LMOD_WGRAD_TN", "1") != "0"


def _tn_operand(t):
    return t.dim() >= 2 and t.stride(-1) == 1 and t.shape[-1] % 8 == 0 and t.stride(-2) % 8 == 0 and t.data_ptr() % 16 == 0


def linear_wgrad(dy, x, fw):
    """main_grad += dy^T @ x ; bias main_grad += colsum(dy).
    (Measured and rejected: running the transposes, or the whole weight gradient, on a side stream.  Two GEMMs sharing
    the chip run slower than back to back (-2 %), the transposes alone did not hide under the dgrad GEMM, and with the
    compute stream at high priority and the weight gradients at default priority the step was still 0.4 % slower.)"""
    if wgrad_tn() and _tn_operand(dy) and _tn_operand(x):
        K.gemm_wgrad(dy, x, fw.grad_buffer(), a_kmajor=True)
        if fw.bias_requires_grad:
            tmp = K.gemm_tn(dy, _ones_col(dy.shape[0], dy.device), out_f32=True)       # [N, 8]; every column = token sum
            fw.bias_grad_buffer().add_(tmp[:, 0])
        fw.grad_done()
        return
    dyt = K.transpose(dy)                     # [N, Tpad]
    xt = K.transpose(x)                       # [K, Tpad]
    K.gemm_wgrad(dyt, xt, fw.grad_buffer())
    if fw.bias_requires_grad:
        tmp = K.gemm_nt(dyt, _ones(dyt.shape[1], dy.device), out_f32=True)      # [N, 8]; every column = token sum
        fw.bias_grad_buffer().add_(tmp[:, 0])
    fw.grad_done()


# ------------------------------------------------------------------------------------------ norm
class AddRMSNorm(torch.autograd.Function):
    """(delta, res) -> (y, h):  h = res + delta (bf16), y = RMSNorm(h) * w.  res may be None (h = delta).
    Qwen2RMSNorm (qwen2/modeling_qwen2.py:83-97) fused with the residual add of the decoder layer."""

    @staticmethod
    def forward(ctx, delta, res, w, eps):
        y, rstd, h = K.rmsnorm_fwd(delta, w, eps, res=res)
        ctx.save_for_backward(h, w, rstd)
        ctx.has_res = res is not None
        return y, h

    @staticmethod
    def backward(ctx, dy, dh):
        h, w, rstd = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros_like(h)
        dy = dy.contiguous()
        if w.requires_grad:                    # trainable norm scale (not in the distillation shells): fp32 main_grad += dw
            if getattr(w, "main_grad", None) is None:
                w.main_grad = torch.zeros(w.shape, device=w.device, dtype=torch.float32)
            K.rmsnorm_dw(dy, h, rstd, w.main_grad, accumulate=True)
        g = K.rmsnorm_bwd(dy, h, w, rstd, dres=dh.contiguous() if dh is not None else None)
        return g, (g if ctx.has_res else None), None, None


# ------------------------------------------------------------------------------------------ row selection
class RowGather(torch.autograd.Function):
    """x[T,H] -> x[rows] ([R,H]); backward scatters into zeros (inv[t] = position of row t in `rows`, or -1)."""

    @staticmethod
    def forward(ctx, x, rows, inv):
        ctx.inv = inv
        return K.gather_rows(x, None, rows, x.shape[1])

    @staticmethod
    def backward(ctx, dy):
        return K.gather_rows(dy.contiguous(), None, ctx.inv, dy.shape[1]), None, None


# ------------------------------------------------------------------------------------------ attention block
class AttnBlock(torch.autograd.Function):
    """x[T,H] -> o_proj(attention(rope(qkv_proj(x)))).  Decoder self-attention of
    qwen2/modeling_qwen2.py:631-715 as four launches: fused QKV GEMM(+bias), in-place RoPE,
    flash attention, O GEMM.  `params` are the trainable tensors (only so autograd schedules backward)."""

    @staticmethod
    def forward(ctx, x, spec, *params):
        return AttnBlock._fwd(ctx, x, spec, None)

    @staticmethod
    def _fwd(ctx, x, spec, res):
        """res (AttnBlockRes): the layer's residual stream, added in the o projection's epilogue."""
        B, S, nh, nkv, hd = spec.B, spec.S, spec.nh, spec.nkv, spec.hd
        if K.qkv_rope_fusable(x, spec.qkv.w, nh + nkv, hd):        # rotary embedding in the QKV GEMM's epilogue (one launch)
            qkv = K.gemm_qkv_rope(x, spec.qkv.w, spec.qkv.b, spec.cos, spec.sin, spec.pos, nh + nkv)
        else:
            qkv = linear_fwd(x, spec.qkv)
            K.rope_(qkv, spec.cos, spec.sin, spec.pos, nh + nkv, hd)
        q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
        kv_out = getattr(spec, "kv_out", None)
        if kv_out is not None:                  # generation prefill: keep the post-RoPE keys and the values
            kv_out[0][:, :S].copy_(k.view(B, S, nkv * hd))
            kv_out[1][:, :S].copy_(v.view(B, S, nkv * hd))
        rows = getattr(spec, "rows", None)      # last layer of a model whose consumer reads only these token rows
        need = _need(ctx)
        if need:
            for fw in (spec.qkv, spec.o):
                if fw.requires_grad:
                    fw.note_use()
        o, lse = K.attn_fwd(q, k, v, B, S, nh, nkv, hd, spec.scale, True, spec.seqlens, want_lse=need,
                            cu=getattr(spec, "cu", None))
        o_in = K.gather_rows(o, None, rows, o.shape[1]) if rows is not None else o    # o_proj on [R, nh*hd] only
        out = linear_fwd(o_in, spec.o) if res is None else K.gemm_nt_res(o_in, spec.o.w, res)
        ctx.spec = spec
        if need:
            ctx.save_for_backward(x if spec.qkv.requires_grad else None, qkv, o, lse, o_in if rows is not None else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        sp = ctx.spec
        x, qkv, o, lse, o_rows = ctx.saved_tensors
        nh, nkv, hd = sp.nh, sp.nkv, sp.hd
        dout = dout.contiguous()
        do = linear_dgrad(dout, sp.o)
        if sp.o.requires_grad:
            linear_wgrad(dout, o if o_rows is None else o_rows, sp.o)
        if o_rows is not None:                # scatter the row gradients back to [T, nh*hd] (zeros elsewhere)
            do = K.gather_rows(do, None, sp.inv_rows, do.shape[1])
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
        fused = K.attn_bwd_rope_fusable(hd)       # RoPE's gradient map in the backward kernels' epilogues (no pass over dqkv)
        K.attn_bwd(q, k, v, o, do, lse, dqkv[:, :nh * hd], dqkv[:, nh * hd:(nh + nkv) * hd],
                   dqkv[:, (nh + nkv) * hd:], sp.B, sp.S, nh, nkv, hd, sp.scale, True, sp.seqlens,
                   cu=getattr(sp, "cu", None), rope=(sp.cos, sp.sin, sp.pos) if fused else None)
        if not fused:
            K.rope_(dqkv, sp.cos, sp.sin, sp.pos, nh + nkv, hd, backward=True)
        dx = linear_dgrad(dqkv, sp.qkv)
        if sp.qkv.requires_grad:
            linear_wgrad(dqkv, x, sp.qkv)
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class AttnBlockRes(torch.autograd.Function):
    """(x, res) -> res + AttnBlock(x): hidden_states = residual + self_attn(...) (qwen2/modeling_qwen2.py:757-763) with the add
    in the o projection's GEMM epilogue (bf16(res + bf16(o W^T)): the roundings of the two-step form).  The residual's gradient
    is the output's."""

    @staticmethod
    def forward(ctx, x, res, spec, *params):
        return AttnBlock._fwd(ctx, x, spec, res)

    @staticmethod
    def backward(ctx, dout):
        g = AttnBlock.backward(ctx, dout)                 # (dx, None [spec], None per parameter ...) sized by needs_input_grad
        return (g[0], dout) + tuple(g[2:])


def res_fusable(rows, fw, res):
    """Can the projection `fw` of a [rows, K] operand take the residual `res` in its epilogue?  (bias-free weight, a shape the
    4-wave 256-tile kernel runs: kernels.gemm_res_fusable.)"""
    return res is not None and fw.b is None and K.gemm_res_fusable(rows, fw.w, res)


# ------------------------------------------------------------------------------------------ dense SwiGLU MLP
class MLPBlock(torch.autograd.Function):
    """down(silu(gate(x)) * up(x)) (qwen2/modeling_qwen2.py:175-187): gate/up is ONE GEMM against the fused [2I, H]
    weight with SwiGLU in its epilogue; the [T, 2I] pre-activations are only written when a backward will read them."""

    @staticmethod
    def forward(ctx, x, spec, *params):
        return MLPBlock._fwd(ctx, x, spec, None)

    @staticmethod
    def _fwd(ctx, x, spec, res):
        need = _need(ctx)
        if spec.gu.b is None:
            act, gu = K.gemm_swiglu(x, spec.gu.w, want_gu=need)
        else:
            gu = linear_fwd(x, spec.gu)
            I = spec.gu.w.shape[0] // 2
            act = K.swiglu_fwd(gu[:, :I], gu[:, I:])
        out = linear_fwd(act, spec.down) if res is None else K.gemm_nt_res(act, spec.down.w, res)
        ctx.spec = spec
        if need:
            ctx.save_for_backward(x, gu, act if spec.down.requires_grad else None)
            for fw in (spec.gu, spec.down):
                if fw.requires_grad:
                    fw.note_use()
        return out

    @staticmethod
    def backward(ctx, dout):
        sp = ctx.spec
        x, gu, act = ctx.saved_tensors
        I = sp.gu.w.shape[0] // 2
        dout = dout.contiguous()
        if sp.down.requires_grad:
            linear_wgrad(dout, act, sp.down)
        if I % 16 == 0:      # down-projection dgrad with the SwiGLU backward in its epilogue: d(act) is never written
            dgu = K.gemm_swiglu_bwd(dout, sp.down.transposed(), gu, K=sp.down.w.shape[0])
        else:
            dact = linear_dgrad(dout, sp.down)
            dgu = torch.empty_like(gu)
            K.swiglu_bwd(dact, gu[:, :I], gu[:, I:], dgu[:, :I], dgu[:, I:])
        dx = linear_dgrad(dgu, sp.gu)
        if sp.gu.requires_grad:
            linear_wgrad(dgu, x, sp.gu)
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class EmptyExpertPass(torch.autograd.Function):
    """The local expert of a rank that received NO rows in an expert-parallel exchange (`MoE._forward_expert_parallel`, one
    local expert): nothing to compute, but the block's place in the backward must stay — `MLPBlock.backward` reports
    `down.grad_done()` then `gu.grad_done()`, which is where `engine.DataParallel._on_ready` issues the spans' asynchronous
    reduce-scatter / all-reduce over the expert-data-parallel group.  Skipping the block on the starved rank would defer its
    spans to `finish()` while its peers send them mid-backward: the collectives of one communicator would be issued in a
    different order on different ranks (a hang, or sums of the wrong spans).  This pass fires the same two hooks at the same
    point; the spans keep their zeros."""

    @staticmethod
    def forward(ctx, x, spec, *params):
        ctx.spec = spec
        ctx.noted = []
        if _need(ctx):
            for fw in (spec.gu, spec.down):
                if fw.requires_grad:
                    fw.note_use()
                    ctx.noted.append(fw)
        return x.new_empty((0, spec.down.w.shape[-2]))

    @staticmethod
    def backward(ctx, dout):
        sp = ctx.spec
        for fw in (sp.down, sp.gu):              # MLPBlock.backward's order
            if any(fw is f for f in ctx.noted):
                fw.grad_done()
        return (dout.new_empty((0, sp.gu.w.shape[-1])), None) + (None,) * (len(ctx.needs_input_grad) - 2)


class MLPBlockRes(torch.autograd.Function):
    """(x, res) -> res + MLPBlock(x): hidden_states = residual + mlp(...) (qwen2/modeling_qwen2.py:765-775) with the add in the down
    projection's GEMM epilogue."""

    @staticmethod
    def forward(ctx, x, res, spec, *params):
        return MLPBlock._fwd(ctx, x, spec, res)

    @staticmethod
    def backward(ctx, dout):
        g = MLPBlock.backward(ctx, dout)
        return (g[0], dout) + tuple(g[2:])


# ------------------------------------------------------------------------------------------ projector
def _drop_cls_idx(B, P1, device):
    return (torch.arange(B, device=device)[:, None] * P1 + torch.arange(1, P1, device=device)[None]).reshape(-1).to(torch.int32)


class ProjectorBlock(torch.autograd.Function):
    """mlp2x_gelu: Linear + exact GELU + Linear (multimodal_projector/builder.py:57-61,148-149) applied to
    the tower output [B, 1+P, Dv] with the CLS row of every image skipped by the batched-GEMM stride
    (feature_select 'patch', clip_encoder.py:36-38).  Returns [B*P, H]."""

    @staticmethod
    def forward(ctx, feats, spec, *params):
        B, P1, Dv = feats.shape
        P = P1 - 1
        H = spec.fc1.w.shape[0]
        pre = torch.empty((B * P, H), device=feats.device, dtype=BF16)
        first_patch = feats.reshape(B * P1, Dv)[1:]          # start one row in; batch stride spans the CLS row
        K.gemm_nt(first_patch, spec.fc1.w, bias=spec.fc1.b, out=pre, M=P, N=H, K=Dv, lda=Dv, ldb=Dv, ldc=H, batch=B,
                  strides=(P1 * Dv, 0, P * H))
        mid = K.gelu_fwd(pre)
        out = linear_fwd(mid, spec.fc2)
        ctx.spec = spec
        if _need(ctx):
            ctx.save_for_backward(feats, pre)
            for fw in (spec.fc1, spec.fc2):
                if fw.requires_grad:
                    fw.note_use()
        return out

    @staticmethod
    def backward(ctx, dout):
        sp = ctx.spec
        feats, pre = ctx.saved_tensors
        B, P1, Dv = feats.shape
        dout = dout.contiguous()
        dmid = linear_dgrad(dout, sp.fc2)
        if sp.fc2.requires_grad:
            linear_wgrad(dout, K.gelu_fwd(pre), sp.fc2)
        dpre = K.gelu_bwd(dmid, pre)
        if sp.fc1.requires_grad:
            xin = K.gather_rows(feats.reshape(B * P1, Dv), None, _drop_cls_idx(B, P1, feats.device), Dv)
            linear_wgrad(dpre, xin, sp.fc1)
        # the vision tower is frozen and runs under no_grad (clip_encoder.py:31,45): nothing flows past here
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


# ------------------------------------------------------------------------------------------ splice
class SpliceEmbed(torch.autograd.Function):
    """inputs_embeds[B*S', H] = rows gathered from the embedding table and the projector output according to an index
    map (llava_arch.py:236-318).  Backward routes image rows to the projector and, when the table is trainable, scatter-adds
    the text rows into its fp32 main_grad.  img_feats / inv_idx None: plain embedding lookup (text-only batch)."""

    @staticmethod
    def forward(ctx, img_feats, embed_w, idx, inv_idx):
        H = embed_w.shape[1]
        out = K.gather_rows(embed_w, img_feats, idx, H)
        ctx.save_for_backward(inv_idx, idx)
        ctx.H, ctx.embed_w = H, embed_w
        return out

    @staticmethod
    def backward(ctx, dout):
        inv_idx, idx = ctx.saved_tensors
        dout = dout.contiguous()
        ew = ctx.embed_w
        if ew.requires_grad:                   # trainable embedding table (not in the distillation shells)
            if getattr(ew, "main_grad", None) is None:
                ew.main_grad = torch.zeros(ew.shape, device=ew.device, dtype=torch.float32)
            K.embed_wgrad(dout, idx, ew.main_grad)
        dfeats = K.gather_rows(dout, None, inv_idx, ctx.H) if (inv_idx is not None and ctx.needs_input_grad[0]) else None
        return dfeats, None, None, None


# ------------------------------------------------------------------------------------------ MoE
class MoEBlock(torch.autograd.Function):
    """Sparse top-k MoE FFN with DeepSpeed-0.9.5 capacity semantics (deepspeed.moe.sharded_moe;
    reference call site llava_qwen2_moe.py:536-546, result :161-167): fp32 router -> gating kernels ->
    row gather into [E, C, H] capacity slabs -> grouped SwiGLU GEMMs (rows past each expert's live
    count skipped) -> weighted combine.  Returns (out[T,H], l_aux[1], exp_counts[E])."""

    @staticmethod
    def forward(ctx, x, spec, noise, *params):
        T, H = x.shape
        E, C, k = spec.E, spec.capacity(T), spec.k
        logits = K.moe_router_fwd(x, spec.wg.data)
        st = K.moe_gate(logits, k, C, noise, seed=getattr(spec, "seed", None), offset=getattr(spec, "offset", 0))
        rows = st.slots_used
        disp = K.gather_rows(x, None, st.slot_token, H)                       # [E*C, H], zero rows on empty slots
        I = spec.gu.w.shape[1] // 2
        need = _need(ctx)
        # grouped gate/up GEMM with SwiGLU in the epilogue (dead rows up to the next multiple of 8 are zeroed in act)
        act, gu = K.gemm_swiglu(disp.view(E, C, H), spec.gu.w, want_gu=need, m_valid=rows)
        y = torch.empty((E, C, H), device=x.device, dtype=BF16)
        K.gemm_nt(act, spec.down.w, out=y, m_valid=rows)
        out = K.moe_combine_fwd(y.view(E * C, H), st, H)
        ctx.spec, ctx.st = spec, st
        spec.last_state = st
        if need:
            ctx.save_for_backward(x, disp, gu, y, act if spec.down.requires_grad else None)
            for fw in (spec.gu, spec.down):
                if fw.requires_grad:
                    fw.note_use()
        l_aux, counts = st.l_aux.clone(), st.exp_counts
        ctx.mark_non_differentiable(counts)
        return out, l_aux, counts

    @staticmethod
    def backward(ctx, dout, dlaux, _dc):
        sp, st = ctx.spec, ctx.st
        x, disp, gu, y, act = ctx.saved_tensors
        T, H = x.shape
        E, C = st.E, st.C
        I = sp.gu.w.shape[1] // 2
        rows = st.slots_used
        if dout is None:
            dout = torch.zeros_like(x)
        dy, dw1, dw2 = K.moe_combine_bwd(dout.contiguous(), y.view(E * C, H), st, H)   # dy: zero rows on empty slots
        # reduction length of the experts' weight gradients: the routed rows rounded UP to whole 64-row K tiles (the transposes
        # zero-fill their columns to that boundary), so that the GEMM's K loop is the hand-placed one for every expert
        kv = torch.clamp((rows + 63) & -64, max=C)
        tn = wgrad_tn()         # reduction-major operands: the live rows are cut exactly by the kernel's descriptors, no copies
        if sp.down.requires_grad:
            if tn:
                K.gemm_tn(dy.view(E, C, H), act, out=sp.down.grad_buffer(), out_f32=True, accumulate=True, k_valid=rows)
            else:
                K.gemm_nt(K.transpose(dy.view(E, C, H), r_valid=rows), K.transpose(act, r_valid=rows), out=sp.down.grad_buffer(),
                          out_f32=True, accumulate=True, k_valid=kv)
            sp.down.grad_done()
        if I % 16 == 0:      # grouped down dgrad + SwiGLU backward in one launch (dead rows: zeroed up to the next 8)
            dgu = K.gemm_swiglu_bwd(dy.view(E, C, H), sp.down.transposed(), gu, m_valid=rows, K=H)
        else:
            dact = torch.empty((E, C, I), device=x.device, dtype=BF16)
            K.gemm_nt(dy.view(E, C, H), sp.down.transposed(), out=dact, m_valid=rows)
            gu2 = gu.view(E * C, 2 * I)
            dgu = torch.empty_like(gu)
            dgu2 = dgu.view(E * C, 2 * I)
            K.swiglu_bwd(dact.view(E * C, I), gu2[:, :I], gu2[:, I:], dgu2[:, :I], dgu2[:, I:], seg_rows=C, seg_valid=rows)
        d_in = torch.empty((E, C, H), device=x.device, dtype=BF16)
        K.gemm_nt(dgu, sp.gu.transposed(), out=d_in, m_valid=rows)
        if sp.gu.requires_grad:
            if tn:
                K.gemm_tn(dgu, disp.view(E, C, H), out=sp.gu.grad_buffer(), out_f32=True, accumulate=True, k_valid=rows)
            else:
                K.gemm_nt(K.transpose(dgu, r_valid=rows), K.transpose(disp.view(E, C, H), r_valid=rows), out=sp.gu.grad_buffer(),
                          out_f32=True, accumulate=True, k_valid=kv)
            sp.gu.grad_done()
        dlogits = K.moe_gate_bwd(st, dw1, dw2, dlaux.contiguous().float() if dlaux is not None else None)
        if sp.wg.requires_grad:
            if getattr(sp.wg, "main_grad", None) is None:
                sp.wg.main_grad = torch.zeros(sp.wg.shape, device=x.device, dtype=torch.float32)
            K.moe_router_wgrad(x, dlogits, sp.wg.main_grad, True)
        dx = K.moe_dispatch_bwd(d_in.view(E * C, H), st, dlogits, sp.wg.data, H)
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


class ResidualMix(torch.autograd.Function):
    """Residual-MoE output mix (deepspeed.moe.layer.MoE.forward with use_residual): coef = softmax(coefficient(x)),
    out = moe_out * coef[..., 0:1] + mlp_out * coef[..., 1:].  The 2-wide `coefficient` head runs on the router kernels
    (fp32 dot products, E = 2); its parameters take fp32 main_grads like the router's."""

    @staticmethod
    def forward(ctx, moe_out, mlp_out, x, coef, *params):
        w32 = coef.weight.detach().float().contiguous()
        b32 = (coef.bias.detach().float() if coef.bias is not None else torch.zeros(2, device=x.device)).contiguous()
        clog = K.moe_router_fwd(x, w32)
        out, p = K.residual_mix_fwd(moe_out.contiguous(), mlp_out.contiguous(), clog, b32)
        ctx.coef = coef
        ctx.save_for_backward(moe_out, mlp_out, x, p, w32)
        return out

    @staticmethod
    def backward(ctx, dout):
        moe_out, mlp_out, x, p, w32 = ctx.saved_tensors
        coef = ctx.coef
        d_moe, d_mlp, dc = K.residual_mix_bwd(dout.contiguous(), moe_out, mlp_out, p)
        if coef.weight.requires_grad:
            if getattr(coef.weight, "main_grad", None) is None:
                coef.weight.main_grad = torch.zeros(coef.weight.shape, device=x.device, dtype=torch.float32)
            K.moe_router_wgrad(x, dc, coef.weight.main_grad, True)
        if coef.bias is not None and coef.bias.requires_grad:
            if getattr(coef.bias, "main_grad", None) is None:
                coef.bias.main_grad = torch.zeros(2, device=x.device, dtype=torch.float32)
            coef.bias.main_grad.add_(dc.sum(0))                 # [T, 2] -> [2]: scalar-sized glue
        dx = K.small_linear_dgrad(dc, w32)
        return (d_moe, d_mlp, dx, None) + (None,) * (len(ctx.needs_input_grad) - 4)


# ------------------------------------------------------------------------------------------ loss head
class DistillHead(torch.autograd.Function):
    """Fused lm_head + distillation / LM / sequence-logp losses on the rows that carry loss only.

    hidden[T,H] -> gather rows -> lm_head GEMM -> ONE-pass row kernel -> per-segment sums.
    Returns (kd_sum[nseg], kd_cnt[nseg], ce_sum[nseg], ce_cnt[nseg]) so callers form
      align = -kd_sum/kd_cnt                                  (align_trainer.py:526)
      lm    =  ce_sum/ce_cnt                                  (llava_qwen2_moe.py:411-421)
      logp  = -ce_sum per sample                              (dpo_trainer.py:483-495)
    with scalar torch math.  Never materialises [T, V] logits."""

    @staticmethod
    def forward(ctx, hidden, head, plan, teacher_logits, *params):
        H = hidden.shape[1]
        rows = hidden if getattr(plan, "pregathered", False) else K.gather_rows(hidden, None, plan.row_idx, H)   # [R, H]
        logits = linear_fwd(rows, head)                                         # [R, Vs] bf16
        Vs = head.w.shape[0]
        Va = plan.align_vocab if teacher_logits is not None else Vs
        stats = K.rowloss_fwd(logits, Vs, teacher_logits, Va, plan.ce_label)
        kd_sum, kd_cnt = K.segment_wsum(stats, 3, plan.kd_w, plan.seg_off)
        ce_sum, ce_cnt = K.segment_wsum(stats, 4, plan.ce_w, plan.seg_off)
        ctx.head, ctx.plan, ctx.Va = head, plan, Va
        if _need(ctx):
            ctx.save_for_backward(logits, stats, teacher_logits, rows if head.requires_grad else None)
            if head.requires_grad:
                head.note_use()
        ctx.mark_non_differentiable(kd_cnt, ce_cnt)
        return kd_sum, kd_cnt, ce_sum, ce_cnt

    @staticmethod
    def backward(ctx, g_kd, _g1, g_ce, _g2):
        logits, stats, t_logits, rows = ctx.saved_tensors
        plan, head = ctx.plan, ctx.head
        Vs = head.w.shape[0]
        nseg = plan.seg_off.numel() - 1
        zero = torch.zeros(nseg, device=logits.device, dtype=torch.float32)
        # rowloss_bwd computes ds = ckd*(q - p): that is -ckd * d(x_kd)/ds, hence the sign flip on g_kd
        kd_scale = (-g_kd).contiguous().float() if g_kd is not None else zero
        ce_scale = g_ce.contiguous().float() if g_ce is not None else zero
        K.rowloss_bwd(logits, Vs, t_logits, ctx.Va, plan.ce_label, stats, plan.kd_w, plan.ce_w, plan.seg_id,
                      kd_scale, ce_scale)                                       # in place over the logits
        d_rows = linear_dgrad(logits, head)
        if head.requires_grad:
            linear_wgrad(logits, rows, head)
        if getattr(plan, "pregathered", False):
            dh = d_rows                                                         # hidden arrived as [R, H]
        else:
            dh = K.gather_rows(d_rows, None, plan.inv_row_idx, d_rows.shape[1])   # scatter back, zeros elsewhere
        return (dh, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)


# ------------------------------------------------------------------------------------------ plain linear
class Linear(torch.autograd.Function):
    """y = x @ W^T + b as one GEMM launch (used for the full-vocabulary `outputs.logits` path)."""

    @staticmethod
    def forward(ctx, x, fw, *params):
        y = linear_fwd(x, fw)
        ctx.fw = fw
        if _need(ctx):
            ctx.save_for_backward(x if fw.requires_grad else None)
            if fw.requires_grad:
                fw.note_use()
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = linear_dgrad(dy, ctx.fw)
        if ctx.fw.requires_grad:
            linear_wgrad(dy, x, ctx.fw)
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


# ------------------------------------------------------------------------------------------ MoE, expert-parallel form
# The fused MoEBlock above is the ep_size == 1 fast path.  With experts sharded over an expert-parallel
# group the layer is the same math cut at the two all-to-all seams of DeepSpeed's MOELayer.forward
# (einsum dispatch -> all_to_all -> local experts -> all_to_all -> einsum combine):
#     MoERoute  : x -> (capacity slabs [E*C, H], w1, w2, l_aux, exp_counts)        [local tokens, all E experts]
#     AllToAll  : [ep, E_local*C, H] slabs exchanged over the EP group (RCCL all_to_all_single; xGMI peers
#                 each receive their [E_local*C, H] slab directly); backward is the reverse exchange
#     ExpertFFN : grouped SwiGLU GEMMs of the LOCAL experts over the ep source ranks' slabs
#     MoECombine: weighted un-permute back to token order
class MoERoute(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, spec, noise, *params):
        T, H = x.shape
        C = spec.capacity(T)
        logits = K.moe_router_fwd(x, spec.wg.data)
        st = K.moe_gate(logits, spec.k, C, noise, seed=getattr(spec, "seed", None), offset=getattr(spec, "offset", 0))
        disp = K.gather_rows(x, None, st.slot_token, H)
        ctx.spec, ctx.st, ctx.H = spec, st, H
        spec.last_state = st
        ctx.save_for_backward(x)
        w2 = st.w2 if st.k == 2 else st.w1.new_zeros(1)
        l_aux, counts = st.l_aux.clone(), st.exp_counts
        ctx.mark_non_differentiable(counts)
        return disp, st.w1, w2, l_aux, counts

    @staticmethod
    def backward(ctx, d_disp, dw1, dw2, dlaux, _dc):
        sp, st = ctx.spec, ctx.st
        (x,) = ctx.saved_tensors
        z = torch.zeros(st.T, device=x.device, dtype=torch.float32)
        dlogits = K.moe_gate_bwd(st, dw1.contiguous() if dw1 is not None else z,
                                 (dw2.contiguous() if dw2 is not None else z) if st.k == 2 else None,
                                 dlaux.contiguous().float() if dlaux is not None else None)
        if sp.wg.requires_grad:
            if getattr(sp.wg, "main_grad", None) is None:
                sp.wg.main_grad = torch.zeros(sp.wg.shape, device=x.device, dtype=torch.float32)
            K.moe_router_wgrad(x, dlogits, sp.wg.main_grad, True)
        if d_disp is None:
            d_disp = torch.zeros((st.E * st.C, ctx.H), device=x.device, dtype=BF16)
        dx = K.moe_dispatch_bwd(d_disp.contiguous(), st, dlogits, sp.wg.data, ctx.H)
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


class MoECombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, w1, w2, st):
        H = y.shape[-1]
        out = K.moe_combine_fwd(y, st, H)
        ctx.st, ctx.H = st, H
        ctx.save_for_backward(y)
        return out

    @staticmethod
    def backward(ctx, dout):
        (y,) = ctx.saved_tensors
        dy, dw1, dw2 = K.moe_combine_bwd(dout.contiguous(), y, ctx.st, ctx.H)
        return dy, dw1, dw2, None


class AllToAll(torch.autograd.Function):
    """Equal-split all_to_all_single over `group` (None: identity, used by the single-GPU test)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        if group is None:
            return x
        import torch.distributed as dist
        from .engine import comm_count
        out = torch.empty_like(x)
        comm_count("all_to_all", x)
        dist.all_to_all_single(out, x.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.group is None:
            return g, None
        import torch.distributed as dist
        from .engine import comm_count
        out = torch.empty_like(g)
        comm_count("all_to_all", g)
        dist.all_to_all_single(out, g.contiguous(), group=ctx.group)
        return out, None


def _native_world_comm(group, x):
    """LMOD_DP_NATIVE=1 and an expert-parallel group that IS the world (config 5: 8 experts on 8 ranks): the process's ONE C-ABI
    communicator (llavamod.comm.shared_world_comm — the same object the engine's gradient exchange uses, created in
    DataParallel.attach) instead of torch.distributed; None otherwise."""
    import torch.distributed as dist
    if os.environ.get("LMOD_DP_NATIVE") != "1" or not x.is_cuda or x.dtype != BF16:
        return None
    if group is not dist.group.WORLD and dist.get_world_size(group) != dist.get_world_size():
        return None
    from . import comm
    return comm.shared_world_comm()


class AllToAllRows(torch.autograd.Function):
    """Unequal-split all_to_all_single of PACKED live rows [L, H]: rank d receives `in_splits[d]` rows of x and this rank
    receives `out_splits[s]` rows from rank s (host ints).  Backward is the reverse exchange with the splits swapped.
    group None: identity (single-GPU test of the decomposed path)."""

    @staticmethod
    def forward(ctx, x, in_splits, out_splits, group):
        ctx.group, ctx.in_splits, ctx.out_splits = group, list(in_splits), list(out_splits)
        if group is None:
            return x
        import torch.distributed as dist
        from .engine import comm_count
        x = x.contiguous()
        comm_count("all_to_all", x)
        nat = _native_world_comm(group, x)
        if nat is not None:                                      # the C-ABI exchange (lmod_moe_all_to_all: grouped send / recv of the live rows)
            return nat.moe_all_to_all(x, ctx.in_splits, ctx.out_splits)
        out = torch.empty((sum(out_splits), x.shape[1]), device=x.device, dtype=x.dtype)
        dist.all_to_all_single(out, x, output_split_sizes=ctx.out_splits, input_split_sizes=ctx.in_splits, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.group is None:
            return g, None, None, None
        import torch.distributed as dist
        from .engine import comm_count
        g = g.contiguous()
        comm_count("all_to_all", g)
        nat = _native_world_comm(ctx.group, g)
        if nat is not None:
            return nat.moe_all_to_all(g, ctx.out_splits, ctx.in_splits), None, None, None
        out = torch.empty((sum(ctx.in_splits), g.shape[1]), device=g.device, dtype=g.dtype)
        dist.all_to_all_single(out, g, output_split_sizes=ctx.in_splits, input_split_sizes=ctx.out_splits, group=ctx.group)
        return out, None, None, None


class _A2AStart(torch.autograd.Function):
    """First half of an ASYNCHRONOUS unequal-split all-to-all of packed rows: the exchange is handed to the backend (RCCL runs it on its
    own stream, ordered after this stream's work so far) and the not-yet-complete output is returned; `_A2AWait` completes it.  The
    handle travels in `box`.  Backward: waits for the reverse exchange that `_A2AWait.backward` started."""

    @staticmethod
    def forward(ctx, x, in_splits, out_splits, group, box):
        import torch.distributed as dist
        from .engine import comm_count
        ctx.box = box
        box["splits"] = (list(in_splits), list(out_splits), group)
        x = x.contiguous()
        out = torch.empty((sum(out_splits), x.shape[1]), device=x.device, dtype=x.dtype)
        comm_count("all_to_all", x)
        box["fwd"] = dist.all_to_all_single(out, x, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits), group=group,
                                            async_op=True)
        return out

    @staticmethod
    def backward(ctx, g):
        w = ctx.box.pop("bwd", None)
        if w is not None:
            w.wait()
        return ctx.box.pop("bwd_out"), None, None, None, None


class _A2AWait(torch.autograd.Function):
    """Second half: this stream waits for the exchange `_A2AStart` launched.  Backward: launches the reverse exchange of the incoming
    gradient asynchronously (it runs under whatever backward work autograd schedules next) and hands the pending buffer on."""

    @staticmethod
    def forward(ctx, out, box):
        ctx.box = box
        w = box.pop("fwd", None)
        if w is not None:
            w.wait()
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        from .engine import comm_count
        in_splits, out_splits, group = ctx.box["splits"]
        g = g.contiguous()
        gout = torch.empty((sum(in_splits), g.shape[1]), device=g.device, dtype=g.dtype)
        comm_count("all_to_all", g)
        ctx.box["bwd"] = dist.all_to_all_single(gout, g, output_split_sizes=in_splits, input_split_sizes=out_splits, group=group,
                                                async_op=True)
        ctx.box["bwd_out"] = gout
        return g, None               # (a token of the right shape: `_A2AStart.backward` returns the exchanged buffer once it has landed)


class _SplitRows(torch.autograd.Function):
    """x [L, H] -> its row chunks [starts[c], starts[c + 1]) (views); backward: the chunks' gradients concatenated (a copy — torch's
    own slice backward would zero-fill a full-size tensor per chunk and ADD them)."""

    @staticmethod
    def forward(ctx, x, starts):
        ctx.starts, ctx.H, ctx.dt, ctx.dev = list(starts), x.shape[1], x.dtype, x.device
        return tuple(x.narrow(0, starts[c], starts[c + 1] - starts[c]) for c in range(len(starts) - 1))

    @staticmethod
    def backward(ctx, *gs):
        parts = [g if g is not None else torch.zeros((ctx.starts[c + 1] - ctx.starts[c], ctx.H), device=ctx.dev, dtype=ctx.dt)
                 for c, g in enumerate(gs)]
        return torch.cat(parts, dim=0), None


def chunked_expert_exchange(packed, in_splits, out_splits, group, block_fn, nchunk):
    """The expert-parallel round trip of ONE local expert — all-to-all of the packed live rows, the expert block on what arrives,
    all-to-all back (`MoE._forward_expert_parallel`) — as a PIPELINE over `nchunk` row chunks (VERDICT r04 next #9b): the exchange of
    chunk c+1 is in flight while the expert GEMMs of chunk c run, and chunk c's results travel back under chunk c+1's GEMMs; the
    backward overlaps the same way (the asynchronous halves `_A2AStart` / `_A2AWait` put every reverse exchange under the next
    chunk's backward GEMMs).  Every (source, destination) block of rows is cut at the same relative positions on both sides —
    chunk c of a block of n rows is rows [n c / nchunk, n (c + 1) / nchunk) — so no extra count exchange is needed.  The expert
    block is row-wise, so the forward result equals the unchunked exchange bit for bit; the expert's weight gradient accumulates
    chunk by chunk.  packed [sum(in_splits), H] ordered by destination; returns the same shape and order."""
    import numpy as np
    ep = len(in_splits)
    cut = lambda n, c: (n * c) // nchunk
    # permutation packed order (dest-major) -> chunk-major: [chunk 0: dest 0 part, dest 1 part, ...][chunk 1: ...]
    off = np.concatenate([[0], np.cumsum(in_splits)]).astype(np.int64)
    perm, cin, cout = [], [], []
    for c in range(nchunk):
        cin.append([cut(in_splits[d], c + 1) - cut(in_splits[d], c) for d in range(ep)])
        cout.append([cut(out_splits[s_], c + 1) - cut(out_splits[s_], c) for s_ in range(ep)])
        for d in range(ep):
            perm.append(np.arange(off[d] + cut(in_splits[d], c), off[d] + cut(in_splits[d], c + 1), dtype=np.int32))
    perm = np.concatenate(perm) if perm else np.zeros(0, dtype=np.int32)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size, dtype=np.int32)
    dev = packed.device
    perm_t, inv_t = torch.from_numpy(perm).to(dev), torch.from_numpy(inv).to(dev)
    x = RowGather.apply(packed, perm_t, inv_t)                       # chunk-major rows
    starts = [0]
    for ci in cin:
        starts.append(starts[-1] + sum(ci))
    xs = _SplitRows.apply(x, starts)
    boxes_f, boxes_b = [dict() for _ in range(nchunk)], [dict() for _ in range(nchunk)]
    pend = [None] * nchunk
    pend[0] = _A2AStart.apply(xs[0], cin[0], cout[0], group, boxes_f[0])
    backs = []
    for c in range(nchunk):
        if c + 1 < nchunk:                                           # chunk c+1 leaves while chunk c is computed
            pend[c + 1] = _A2AStart.apply(xs[c + 1], cin[c + 1], cout[c + 1], group, boxes_f[c + 1])
        rows = _A2AWait.apply(pend[c], boxes_f[c])
        y = block_fn(rows)
        backs.append(_A2AStart.apply(y, cout[c], cin[c], group, boxes_b[c]))      # travels back under chunk c+1's GEMMs
    outs = [_A2AWait.apply(backs[c], boxes_b[c]) for c in range(nchunk)]
    y_all = torch.cat(outs, dim=0) if nchunk > 1 else outs[0]       # (a copy of [rows, H] bf16, no arithmetic)
    return RowGather.apply(y_all, inv_t, perm_t)                     # back to the packed (destination-major) order


def ep_live_row_plan(used, recv, C, device):
    """Index maps of the live-row expert-parallel exchange, from HOST counts.
      used [ep, El]: live slots of MY capacity slabs, by destination rank and its local expert (= st.slots_used)
      recv [ep, El]: live rows I will receive, by source rank and my local expert (the peers' `used` rows for me)
    Slab layouts are [ep, El, C] (x H).  Packed order on the wire is (rank, local expert, slot).  Returns int32 device
    maps: send_idx [Ls] slab position of every live row I send, send_inv [ep*El*C] its packed position or -1; recv_slab
    [ep*El*C] packed position filling each slab position of the receive buffer or -1, recv_inv [Lr] the inverse; and the
    split sizes in rows."""
    import numpy as np
    used, recv = np.asarray(used, dtype=np.int64), np.asarray(recv, dtype=np.int64)
    ep, El = used.shape

    def maps(cnt):
        n = int(cnt.sum())
        idx = np.empty(n, dtype=np.int32)
        inv = np.full(ep * El * C, -1, dtype=np.int32)
        o = 0
        for r in range(ep):
            for le in range(El):
                c = int(cnt[r, le])
                base = (r * El + le) * C
                idx[o:o + c] = base + np.arange(c, dtype=np.int32)
                inv[base:base + c] = o + np.arange(c, dtype=np.int32)
                o += c
        return idx, inv
    send_idx, send_inv = maps(used)
    recv_inv, recv_slab = maps(recv)
    t = lambda a: torch.from_numpy(a).to(device)
    from types import SimpleNamespace
    return SimpleNamespace(send_idx=t(send_idx), send_inv=t(send_inv), recv_slab=t(recv_slab), recv_inv=t(recv_inv),
                           in_splits=[int(v) for v in used.sum(1)], out_splits=[int(v) for v in recv.sum(1)])


class ExpertFFN(torch.autograd.Function):
    """x: [ep, E_local, C, H] (slabs received from the ep source ranks) -> same shape.  rows: int32
    [ep, E_local] live rows per slab (from the senders' slots_used) or None (compute every row)."""

    @staticmethod
    def forward(ctx, x, spec, rows, *params):
        ep, El, C, H = x.shape
        I = spec.gu.w.shape[-2] // 2
        need = _need(ctx)
        gu = torch.empty((ep, El, C, 2 * I), device=x.device, dtype=BF16) if need else None
        act = torch.zeros((ep, El, C, I), device=x.device, dtype=BF16)    # dead rows stay zero (wgrad reads whole slabs)
        for le in range(El):          # one grouped launch per local expert: batch = source ranks, shared weights
            mv = rows[:, le].contiguous() if rows is not None else None
            wgu = spec.gu.w[le] if spec.gu.stacked else spec.gu.w
            K.gemm_swiglu(x[:, le], wgu, act=act[:, le], gu=gu[:, le] if need else None, m_valid=mv)
        y = torch.empty_like(x)
        for le in range(El):
            mv = rows[:, le].contiguous() if rows is not None else None
            wd = spec.down.w[le] if spec.down.stacked else spec.down.w
            K.gemm_nt(act[:, le], wd, out=y[:, le], M=C, N=H, K=I, lda=I, ldb=I, ldc=H, batch=ep,
                      strides=(El * C * I, 0, El * C * H), m_valid=mv)
        # rows past a slab's live count are not computed: the receiving combine never reads them (empty slots)
        ctx.spec, ctx.shape, ctx.I = spec, (ep, El, C, H), I
        ctx.save_for_backward(x, gu, rows, act if (need and spec.down.requires_grad) else None)
        if need:
            for fw in (spec.gu, spec.down):
                if fw.requires_grad:
                    fw.note_use()
        return y

    @staticmethod
    def backward(ctx, dy):
        sp = ctx.spec
        x, gu, rows, act = ctx.saved_tensors
        ep, El, C, H = ctx.shape
        I = ctx.I
        dy = dy.contiguous()
        gu_t, down_t = sp.gu.transposed(), sp.down.transposed()
        fused_bwd = (I % 16 == 0)          # down-projection dgrad + SwiGLU backward in ONE grouped launch per local expert (as MoEBlock)
        dact = None if fused_bwd else torch.empty((ep, El, C, I), device=x.device, dtype=BF16)
        dgu = torch.empty_like(gu)
        for le in range(El):
            mv = rows[:, le].contiguous() if rows is not None else None
            dt = down_t[le] if sp.down.stacked else down_t
            if fused_bwd:                   # batch = source ranks (shared weights); dead rows up to the next multiple of 8 are zeroed
                K.gemm_swiglu_bwd(dy[:, le], dt, gu[:, le], out=dgu[:, le], m_valid=mv, K=H)
            else:
                K.gemm_nt(dy[:, le], dt, out=dact[:, le], M=C, N=I, K=H, lda=H, ldb=dt.stride(0), ldc=I, batch=ep,
                          strides=(El * C * H, 0, El * C * I), m_valid=mv)
        gu2 = gu.view(ep * El * C, 2 * I)
        rflat = rows.reshape(-1).contiguous() if rows is not None else None
        if sp.down.requires_grad:
            g = sp.down.grad_buffer()
            tn = wgrad_tn()
            if El == 1:     # slabs of all source ranks are contiguous: ONE wgrad GEMM over K = ep*C (dead rows are zero)
                if tn:
                    K.gemm_tn(dy.view(ep * C, H), act.view(ep * C, I), out=(g[0] if sp.down.stacked else g), out_f32=True, accumulate=True)
                else:
                    K.gemm_nt(K.transpose(dy.view(ep * C, H)), K.transpose(act.view(ep * C, I)),
                              out=(g[0] if sp.down.stacked else g), out_f32=True, accumulate=True)
            else:
                for le in range(El):
                    for src in range(ep):
                        kv = rows[src, le:le + 1].contiguous() if rows is not None else None
                        if tn:
                            K.gemm_tn(dy[src, le], act[src, le], out=(g[le] if sp.down.stacked else g), out_f32=True, accumulate=True,
                                      k_valid=kv)
                        else:
                            K.gemm_nt(K.transpose(dy[src, le]), K.transpose(act[src, le]),
                                      out=(g[le] if sp.down.stacked else g), out_f32=True, accumulate=True, k_valid=kv)
            sp.down.grad_done()
        if not fused_bwd:
            dgu2 = dgu.view(ep * El * C, 2 * I)
            K.swiglu_bwd(dact.view(ep * El * C, I), gu2[:, :I], gu2[:, I:], dgu2[:, :I], dgu2[:, I:], seg_rows=C, seg_valid=rflat)
        dx = torch.empty_like(x)
        for le in range(El):
            mv = rows[:, le].contiguous() if rows is not None else None
            gt = gu_t[le] if sp.gu.stacked else gu_t
            K.gemm_nt(dgu[:, le], gt, out=dx[:, le], M=C, N=H, K=2 * I, lda=2 * I, ldb=gt.stride(0), ldc=H, batch=ep,
                      strides=(El * C * 2 * I, 0, El * C * H), m_valid=mv)
        if sp.gu.requires_grad:
            g = sp.gu.grad_buffer()
            tn = wgrad_tn()
            if El == 1:
                if tn:
                    K.gemm_tn(dgu.view(ep * C, 2 * I), x.view(ep * C, H), out=(g[0] if sp.gu.stacked else g), out_f32=True, accumulate=True)
                else:
                    K.gemm_nt(K.transpose(dgu.view(ep * C, 2 * I)), K.transpose(x.view(ep * C, H)),
                              out=(g[0] if sp.gu.stacked else g), out_f32=True, accumulate=True)
            else:
                for le in range(El):
                    for src in range(ep):
                        kv = rows[src, le:le + 1].contiguous() if rows is not None else None
                        if tn:
                            K.gemm_tn(dgu[src, le], x[src, le], out=(g[le] if sp.gu.stacked else g), out_f32=True, accumulate=True,
                                      k_valid=kv)
                        else:
                            K.gemm_nt(K.transpose(dgu[src, le]), K.transpose(x[src, le]),
                                      out=(g[le] if sp.gu.stacked else g), out_f32=True, accumulate=True, k_valid=kv)
            sp.gu.grad_done()
        # dx rows past a slab's live count are not computed; the sender's dispatch backward reads live slots only
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)
