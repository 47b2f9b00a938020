"""AlignTrainer — mimic distillation step (reference train/align_trainer.py:180; compute_loss :530-594,
get_p :455-477, get_logp :479-501, compute_align_loss :503-528).

The reference materialises teacher probabilities and student log-probabilities as fp32 [B,S',151936]
tensors and makes >= 6 passes over them.  `compute_loss` here produces the SAME scalars
(loss, loss/align, loss/moe_balance, loss/lm) from a fused path: lm_head GEMMs only on the rows that
carry loss, one row kernel for softmax(t) x log_softmax(s) + shifted CE, segment sums; backward
writes d(logits) in place and runs the dgrad GEMM.  This class is the loss step only — the HF
Trainer / DeepSpeed control plane around it is out of scope (llavamod.engine drives the step).
"""
from collections import defaultdict
from types import SimpleNamespace

import copy

import torch

from .. import kernels as K
from .. import ops
from ..constants import ALIGN_VOCAB
from ..model.language_model.llava_qwen2 import build_loss_plan


class AlignTrainer:
    def __init__(self, model, ref_model, args=None, loss_type="only_kd", moe_loss_enable=True, label_pad_token_id=-100,
                 align_vocab=ALIGN_VOCAB):
        self.model, self.ref_model = model, ref_model
        self.args = args if args is not None else SimpleNamespace(moe_enable=True, distill_all_tokens=False)
        self.loss_type = getattr(self.args, "loss_type", loss_type)
        self.moe_loss_enable = getattr(self.args, "moe_loss_enable", moe_loss_enable)
        self.label_pad_token_id = label_pad_token_id
        self.align_vocab = align_vocab
        self._stored_metrics = defaultdict(lambda: defaultdict(list))
        if ref_model is not None:
            ref_model.eval()                                   # align_trainer.py:450
            for p in ref_model.parameters():
                p.requires_grad_(False)

    # ---- fused step -------------------------------------------------------------------------------
    def _plan(self, info, device, kd=True, ce=True):
        Va = min(self.align_vocab, self.model.vocab_size, self.ref_model.vocab_size)
        return build_loss_plan(info.labels_np, info.lens_np, kd_rows=kd, ce_rows=ce,
                               distill_all_tokens=getattr(self.args, "distill_all_tokens", False), align_vocab=Va,
                               device=device)

    def _dev(self):
        return next(self.ref_model.parameters()).device

    # ---- teacher pass --------------------------------------------------------------------------------
    def _teacher_pass(self, batch):
        """Frozen teacher forward (:556-560) down to what the loss consumes: the loss plan and the teacher's logits
        on the loss rows ([R, Vt] bf16) — its last layer and lm_head run on those rows alone."""
        with torch.no_grad():
            t_rows, _, t_info = self.ref_model.forward_hidden(**batch, plan_fn=lambda info: self._plan(info, self._dev()))
            t_logits = ops.linear_fwd(t_rows, self.ref_model.head())
        feats = getattr(self.ref_model, "_last_tower_feats", None) if self._towers_identical() else None
        return SimpleNamespace(plan=t_info.plan, logits=t_logits, event=None, tower_feats=feats)

    def _towers_identical(self):
        """Student and teacher normally load the SAME frozen CLIP checkpoint (`--image_tower` is one flag in the
        reference's shells); when their towers are frozen and bit-identical the features are computed once per batch
        and shared (the reference runs the tower twice).  Checked once, on the actual weights."""
        epoch = (getattr(self.model, "_weights_epoch", 0), getattr(self.ref_model, "_weights_epoch", 0))
        if getattr(self, "_tower_epoch", None) != epoch:         # a checkpoint was loaded since the last check
            self._tower_shared, self._tower_epoch = None, epoch
        if getattr(self, "_tower_shared", None) is None:
            ok = False
            ts, tt = self.model.get_image_tower(), self.ref_model.get_image_tower()
            if ts is not None and tt is not None and getattr(self, "share_image_tower", True):
                a, b = ts.state_dict(), tt.state_dict()
                ok = (a.keys() == b.keys() and not any(p.requires_grad for p in ts.parameters())
                      and all(a[k].shape == b[k].shape and torch.equal(a[k], b[k]) for k in a))
            self._tower_shared = ok
        return self._tower_shared

    @staticmethod
    def _batch_of(inputs):
        return dict(input_ids=inputs["input_ids"], attention_mask=inputs.get("attention_mask"),
                    labels=inputs.get("labels"), images=inputs.get("images"))

    def prefetch_teacher(self, inputs):
        """Run the teacher pass for a FUTURE batch on a side stream and return a handle for
        `compute_loss(..., teacher=handle)`.  The teacher is frozen, so its forward for batch i+1 does not depend on
        the student's update i: issued before the student's step, its HBM-bound row kernels execute under the
        student's MFMA-bound GEMMs and vice versa (same work per step, shorter wall clock)."""
        if getattr(self, "_tstream", None) is None:
            self._tstream = torch.cuda.Stream(device=self._dev())
        with torch.cuda.stream(self._tstream):
            h = self._teacher_pass(self._batch_of(inputs))
            h.event = torch.cuda.Event()
            h.event.record(self._tstream)
        return h

    def compute_loss(self, model, inputs, return_outputs=False, teacher=None):
        assert self.ref_model is not None, "ref model can not be none!"
        batch = self._batch_of(inputs)
        if teacher is None:
            teacher = self._teacher_pass(batch)
        elif teacher.event is not None:                       # produced on the side stream: order and pin its memory
            cur = torch.cuda.current_stream()
            cur.wait_event(teacher.event)
            for t in [teacher.logits, teacher.tower_feats] + [v for v in vars(teacher.plan).values() if torch.is_tensor(v)]:
                if t is not None:
                    t.record_stream(cur)
        teacher_plan, t_logits = teacher.plan, teacher.logits
        if teacher.tower_feats is not None:
            model._shared_tower_feats = teacher.tower_feats          # consumed by the student's encode_images below
        # same inputs => same spliced labels => same loss rows: the student's last (dense) layer is trimmed the same way.
        # The teacher's plan is only reused if the student's splice really produced the same layout: a different patch
        # count or tokenizer_model_max_length would silently index the wrong rows (the reference raises on the logits'
        # shape mismatch at this point, align_trainer.py:470-471).
        def student_plan(info):
            tp = teacher_plan
            same = tp.shape == (info.B, info.S)
            if same and tp.labels_np is not None and not torch.is_tensor(info.labels_np):
                same = bool((tp.labels_np == info.labels_np).all())
            elif same and torch.is_tensor(info.labels_np):          # device-built plans: compare on the device
                same = bool(torch.equal(getattr(tp, "labels_dev", info.labels_np), info.labels_np))
            if not same:
                raise ValueError("Logits (batch and sequence length dim) and labels must have the same shape: the student's and "
                                 f"the teacher's spliced sequences differ ({(info.B, info.S)} vs {tp.shape})")
            return copy.copy(tp)
        s_hidden, moe_list, s_info = model.forward_hidden(**batch, plan_fn=student_plan)
        plan = s_info.plan
        kd_sum, kd_cnt, ce_sum, ce_cnt = ops.DistillHead.apply(s_hidden, model.head(), plan, t_logits,
                                                               *model._head_trainable())
        align_loss = -(kd_sum.sum() / kd_cnt.sum())                            # :526
        policy_sft_loss = ce_sum.sum() / ce_cnt.sum()                          # CrossEntropyLoss(), shifted
        moe_all = model.moe_loss_from_list(moe_list) if hasattr(model, "moe_loss_from_list") else None
        if moe_all is not None:
            policy_sft_loss = policy_sft_loss + moe_all                        # outputs.loss already carries it
        policy_moe_loss = moe_all if (getattr(self.args, "moe_enable", True) and self.moe_loss_enable) else None
        losses = align_loss if self.loss_type == "only_kd" else align_loss + policy_sft_loss   # :570-573
        if policy_moe_loss is not None:
            # `if policy_moe_loss:` (:575) is tensor truthiness: a balance loss of exactly 0 is logged as -1 and not added.
            # Adding 0 changes nothing, so only the LOGGED value needs the test — on the device, without a host sync
            # (a bool() here stalled the teacher-prefetch pipeline every step).
            losses = losses + policy_moe_loss
            moe_loss = torch.where(policy_moe_loss.detach() != 0, policy_moe_loss.detach(), torch.full_like(align_loss, -1.0))
        else:
            moe_loss = torch.full_like(align_loss, -1.0)
        outputs = {"loss": losses.mean(), "loss/align": align_loss.mean(), "loss/moe_balance": moe_loss.mean(),
                   "loss/lm": policy_sft_loss.mean()}
        self.store_metrics({k: v.detach() for k, v in outputs.items()}, train_eval="train")
        return (losses.mean(), outputs) if return_outputs else losses.mean()

    def training_step(self, model, inputs, teacher=None):
        loss = self.compute_loss(model, inputs, teacher=teacher)
        loss.backward()
        return loss.detach()

    def store_metrics(self, metrics, train_eval="train"):
        for k, v in metrics.items():
            self._stored_metrics[train_eval][k].append(v)

    def log(self, logs):
        """The reference's `log()` (train/align_trainer.py:600-614, dpo_trainer.py same): the stored per-step metrics are
        AVERAGED into `logs` and the store is DRAINED — without the drain it grows by a few device scalars per step for
        the whole run.  One host read-back per call (the logging interval), none per step.  Returns `logs`."""
        train_eval = "train" if "loss" in logs else "eval"
        for key, metrics in self._stored_metrics[train_eval].items():
            logs[key] = torch.stack([m.detach().float().reshape(()) for m in metrics]).mean().item()
        del self._stored_metrics[train_eval]
        return logs

    def _save_checkpoint(self, model, trial=None, metrics=None, output_dir=None):
        """train/align_trainer.py:616-636, same positional signature `(model, trial, metrics)`.  The folder is the reference's
        `<args.output_dir>/checkpoint-<global_step>` unless `output_dir` names one.  With `tune_mm_mlp_adapter` only the adapter
        is saved — `config.json` (the reference calls `config.save_pretrained`) + `mm_projector.bin` holding the `mm_projector`
        parameters under their full names — and only by the rank whose `args.local_rank` is 0 or -1, like the reference (every
        rank holds the same replicated adapter; concurrent writers of one file would race).  Otherwise the full HF-layout
        checkpoint, written by ONE rank of the whole job — HF's `args.should_save` when the args carry it, else the GLOBAL
        rank 0 (a `local_rank` gate would let every node's first rank write the same shard and index files).  With expert
        parallelism (`ep_size` > 1, config 5) a rank holds only its local experts: the save is then a collective over the
        writer's expert-parallel group — EVERY rank calls this method, the group gathers the experts, and the writer saves them
        under global indices (`checkpoint.full_state_dict`)."""
        import os
        from ..checkpoint import expert_parallel_layout, job_rank
        if output_dir is None:
            step = getattr(getattr(self, "state", None), "global_step", 0)
            output_dir = os.path.join(getattr(self.args, "output_dir", "."), f"checkpoint-{step}")
        if getattr(self.args, "tune_mm_mlp_adapter", False):
            local_rank = getattr(self.args, "local_rank", -1)
            if local_rank not in (0, -1, None):
                return None
            keys = ["mm_projector", "vision_resampler"]
            if getattr(self.args, "use_im_start_end", False):
                keys.extend(["embed_tokens", "embed_in"])
            os.makedirs(output_dir, exist_ok=True)
            model.save_config(output_dir)
            return model.save_mm_adapter(output_dir, keys_to_match=tuple(keys))
        rank = job_rank(self.args)          # torch.distributed's rank, or the launcher's (process_index / RANK) when it is not initialised
        should_save = bool(getattr(self.args, "should_save", rank == 0))
        if expert_parallel_layout(model):
            return model.save_pretrained(output_dir) or None          # collective: the writer is global rank 0
        if not should_save:
            return None
        return model.save_pretrained(output_dir, writer_rank=rank)

    # ---- materialising API of the reference (slow path, kept for drop-in parity) ------------------
    def get_p(self, model, inputs):
        """Teacher probabilities softmax(logits[:, :, :151936], fp32) — materialised like the reference."""
        outputs = model(**inputs, return_dict=True)
        logits, labels = outputs.logits, outputs.labels
        if logits.shape[:-1] != labels.shape:
            raise ValueError("Logits (batch and sequence length dim) and labels must have the same shape.")
        moe_loss = outputs.moe_loss if (getattr(self.args, "moe_enable", True) and self.moe_loss_enable
                                        and "moe_loss" in outputs) else None
        return _row_softmax(logits, self.align_vocab, log=False), outputs.loss, moe_loss

    def get_logp(self, model, inputs):
        outputs = model(**inputs, return_dict=True)
        logits, labels = outputs.logits, outputs.labels
        if logits.shape[:-1] != labels.shape:
            raise ValueError("Logits (batch and sequence length dim) and labels must have the same shape.")
        moe_loss = outputs.moe_loss if (getattr(self.args, "moe_enable", True) and self.moe_loss_enable
                                        and "moe_loss" in outputs) else None
        return _row_softmax(logits, self.align_vocab, log=True), outputs.loss, moe_loss, labels

    def compute_align_loss(self, policy_logprobs, reference_probs, labels):
        """-(sum_v p*logp masked where logp is inf, over label rows) / #label rows (:503-528) on materialised
        tensors; forward-only helper (the training path is `compute_loss`)."""
        V = policy_logprobs.shape[-1]
        lp = policy_logprobs.reshape(-1, V).contiguous()
        p = reference_probs.reshape(-1, V).contiguous()
        x = K.rowdot_masked(p, lp)
        if getattr(self.args, "distill_all_tokens", False):
            mask = torch.ones_like(labels).reshape(-1).float()
        else:
            mask = (labels != self.label_pad_token_id).reshape(-1).float()
        off = torch.tensor([0, x.numel()], dtype=torch.int32, device=x.device)
        s, w = K.segment_wsum(x.view(-1, 1), 0, mask.contiguous(), off)
        return -(s[0] / w[0])


def _row_softmax(logits, align_vocab, log):
    B, S, V = logits.shape
    Va = min(align_vocab, V)
    flat = logits.reshape(B * S, V)
    return K.row_softmax_f32(flat, Va, log).view(B, S, Va)
