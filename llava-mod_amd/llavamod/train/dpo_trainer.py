"""DPOTrainer — preference distillation step (reference train/dpo_trainer.py:180; get_logp :462-495,
dpo_loss :497-562, compute_loss :564-641).  Four forwards (student chosen / rejected, teacher chosen /
rejected under no_grad) sharing `images`; per-sequence log-prob sums come from the fused loss head
(lm_head on the shifted label rows only, full-vocab log-softmax gathered at the label, per-sample
segment sums); the [B]-vector DPO / hinge / IPO / KTO-pair loss and its gradient are one small kernel.
"""
from collections import defaultdict
from types import SimpleNamespace

import torch

from .. import kernels as K
from .. import ops
from ..model.language_model.llava_qwen2 import build_loss_plan


class _DpoLoss(torch.autograd.Function):
    """mean(dpo_loss(pc, pr, rc, rr)) with the analytic gradient from the kernel."""

    @staticmethod
    def forward(ctx, pc, pr, rc, rr, beta, label_smoothing, loss_type):
        losses, cr, rj, dpc, dpr = K.dpo_loss(pc.contiguous(), pr.contiguous(), rc.contiguous(), rr.contiguous(), beta,
                                              label_smoothing, loss_type)
        ctx.save_for_backward(dpc, dpr)
        off = torch.tensor([0, losses.numel()], dtype=torch.int32, device=losses.device)
        s, _ = K.segment_wsum(losses.view(-1, 1), 0, None, off)
        mean = s[0] / losses.numel()
        ctx.mark_non_differentiable(losses, cr, rj)
        return mean, losses, cr, rj

    @staticmethod
    def backward(ctx, g, *_):
        dpc, dpr = ctx.saved_tensors
        return g * dpc, g * dpr, None, None, None, None, None


class DPOTrainer:
    def __init__(self, model, ref_model, args=None, beta=0.1, label_smoothing=0.0, loss_type="sigmoid",
                 moe_loss_enable=True, label_pad_token_id=-100):
        self.model, self.ref_model = model, ref_model
        self.args = args if args is not None else SimpleNamespace(moe_enable=True)
        self.beta = getattr(self.args, "beta", beta)
        self.label_smoothing = getattr(self.args, "label_smoothing", label_smoothing)
        self.loss_type = getattr(self.args, "loss_type", loss_type)
        self.moe_loss_enable = getattr(self.args, "moe_loss_enable", moe_loss_enable)
        self.label_pad_token_id = label_pad_token_id
        self._stored_metrics = defaultdict(lambda: defaultdict(list))
        if ref_model is not None:
            ref_model.eval()
            for p in ref_model.parameters():
                p.requires_grad_(False)

    def get_logp(self, model, inputs, average_log_prob=False):
        """(sum_t log p(y_t) over labelled shifted positions [B], sft_loss, moe_loss) — dpo_trainer.py:462-495."""
        dev = next(model.parameters()).device
        mk = lambda info: build_loss_plan(info.labels_np, info.lens_np, kd_rows=False, ce_rows=True, device=dev)
        # only the loss rows leave the decoder (a dense last layer skips the other rows, forward and backward)
        hidden, moe_list, info = model.forward_hidden(input_ids=inputs["input_ids"], attention_mask=inputs.get("attention_mask"),
                                                      labels=inputs.get("labels"), images=inputs.get("images"), plan_fn=mk)
        plan = info.plan if info.plan is not None else mk(info)
        _, _, ce_sum, ce_cnt = ops.DistillHead.apply(hidden, model.head(), plan, None, *model._head_trainable())
        logps = -ce_sum / ce_cnt if average_log_prob else -ce_sum
        sft = ce_sum.sum() / ce_cnt.sum()
        moe_all = model.moe_loss_from_list(moe_list) if hasattr(model, "moe_loss_from_list") else None
        if moe_all is not None:
            sft = sft + moe_all
        moe = moe_all if (getattr(self.args, "moe_enable", True) and self.moe_loss_enable) else None
        return logps, sft, moe

    def dpo_loss(self, policy_chosen_logps, policy_rejected_logps, reference_chosen_logps, reference_rejected_logps,
                 reference_free=False):
        if reference_free:
            reference_chosen_logps = torch.zeros_like(policy_chosen_logps)
            reference_rejected_logps = torch.zeros_like(policy_rejected_logps)
        if self.loss_type not in K.LOSS_TYPES:
            raise ValueError(f"Unknown loss type: {self.loss_type}. Should be one of ['sigmoid', 'hinge']")
        mean, losses, cr, rj = _DpoLoss.apply(policy_chosen_logps.float(), policy_rejected_logps.float(),
                                              reference_chosen_logps.float(), reference_rejected_logps.float(),
                                              self.beta, self.label_smoothing, self.loss_type)
        self._last_mean = mean
        return losses, cr, rj

    @staticmethod
    def _sides(inputs):
        ch = dict(input_ids=inputs["chosen_input_ids"], labels=inputs["chosen_labels"],
                  attention_mask=inputs["chosen_attention_mask"])
        rj = dict(input_ids=inputs["rejected_input_ids"], labels=inputs["rejected_labels"],
                  attention_mask=inputs["rejected_attention_mask"])
        if "images" in inputs:
            ch["images"] = inputs["images"]; rj["images"] = inputs["images"]
        return ch, rj

    def _reference_pass(self, ch, rj):
        assert self.ref_model is not None, "ref model can not be none!"
        with torch.no_grad():
            rc, *_ = self.get_logp(self.ref_model, ch)
            rr, *_ = self.get_logp(self.ref_model, rj)
        return SimpleNamespace(chosen=rc, rejected=rr, event=None)

    def prefetch_reference(self, inputs):
        """Reference-model log-probabilities of a FUTURE batch on a side stream (the reference model is frozen); pass the
        handle to `compute_loss(..., reference=handle)`.  Same pipelining as `AlignTrainer.prefetch_teacher`."""
        if getattr(self, "_rstream", None) is None:
            self._rstream = torch.cuda.Stream(device=next(self.ref_model.parameters()).device)
        with torch.cuda.stream(self._rstream):
            h = self._reference_pass(*self._sides(inputs))
            h.event = torch.cuda.Event()
            h.event.record(self._rstream)
        return h

    def compute_loss(self, model, inputs, return_outputs=False, reference=None):
        ch, rj = self._sides(inputs)
        pc, pc_sft, pc_moe = self.get_logp(model, ch)
        pr, _, pr_moe = self.get_logp(model, rj)
        if reference is None:
            reference = self._reference_pass(ch, rj)
        elif reference.event is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(reference.event)
            reference.chosen.record_stream(cur); reference.rejected.record_stream(cur)
        rc, rr = reference.chosen, reference.rejected
        reward_losses, chosen_rewards, rejected_rewards = self.dpo_loss(pc, pr, rc, rr)
        reward_mean = self._last_mean                           # differentiable mean of reward_losses
        if pc_moe is not None and pr_moe is not None:
            # `if policy_chosen_moe_loss and policy_rejected_moe_loss:` (:614-616) is tensor truthiness; decided on the
            # device (a bool() here is a host sync in the middle of the step): the sum is added only if BOTH are non-zero
            both = (pc_moe.detach() != 0) & (pr_moe.detach() != 0)
            added = torch.where(both, pc_moe + pr_moe, torch.zeros_like(reward_mean))
            total = reward_mean + added                         # (reward_losses + moe).mean()
            moe_loss = torch.where(both, added.detach(), torch.full_like(reward_mean, -1.0))
        else:
            moe_loss = torch.full_like(reward_mean, -1.0)
            total = reward_mean
        acc = (chosen_rewards > rejected_rewards).float()
        outputs = {"loss": total, "loss/reward": reward_mean, "loss/moe_balance": moe_loss,
                   "loss/policy_chosen": pc_sft.detach(), "rewards/chosen": chosen_rewards.mean(),
                   "rewards/rejected": rejected_rewards.mean(), "rewards/accuracies": acc.mean(),
                   "rewards/margins": (chosen_rewards - rejected_rewards).mean(), "logps/chosen": pc.detach().mean(),
                   "logps/rejected": pr.detach().mean()}
        self.store_metrics({k: v.detach() for k, v in outputs.items()}, train_eval="train")
        return (total, outputs) if return_outputs else total

    def training_step(self, model, inputs, reference=None):
        loss = self.compute_loss(model, inputs, reference=reference)
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
