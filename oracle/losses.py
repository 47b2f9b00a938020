"""Oracle: distillation losses.  TEST INFRASTRUCTURE ONLY.

  AlignTrainer.get_p / get_logp / compute_align_loss / compute_loss   train/align_trainer.py:455-594
  DPOTrainer.get_logp / dpo_loss / compute_loss                       train/dpo_trainer.py:462-641
"""
import torch
import torch.nn.functional as F

ALIGN_VOCAB = 151936      # the hard-coded slice of align_trainer.py:473,497
LABEL_PAD = -100


def get_p(logits, align_vocab=ALIGN_VOCAB):                 # align_trainer.py:473-475
    return F.softmax(logits[:, :, :align_vocab], dim=-1, dtype=torch.float32)


def get_logp(logits, align_vocab=ALIGN_VOCAB):              # align_trainer.py:497-499
    return F.log_softmax(logits[:, :, :align_vocab], dim=-1, dtype=torch.float32)


def compute_align_loss(policy_logprobs, reference_probs, labels, distill_all_tokens=False):   # :503-528
    inf_mask = torch.isinf(policy_logprobs)
    prod = torch.masked_fill(reference_probs * policy_logprobs, inf_mask, 0)
    x = torch.sum(prod, dim=-1).view(-1)
    if distill_all_tokens:
        mask = torch.ones_like(labels).int()
    else:
        mask = (labels != LABEL_PAD).int()                  # UNshifted label mask
    return -torch.sum(x * mask.view(-1), dim=0) / torch.sum(mask.view(-1), dim=0)


def mimic_loss(student_out, teacher_out, loss_type="only_kd", moe_loss_enable=True, distill_all_tokens=False,
               align_vocab=ALIGN_VOCAB):
    """AlignTrainer.compute_loss (:530-594).  Returns (loss, dict of the 4 logged scalars)."""
    p = get_p(teacher_out.logits.detach(), align_vocab)
    logp = get_logp(student_out.logits, align_vocab)
    align = compute_align_loss(logp, p, student_out.labels, distill_all_tokens)
    sft = student_out.loss
    moe = student_out.moe_loss if moe_loss_enable else None
    losses = align if loss_type == "only_kd" else align + sft
    if moe is not None and bool(moe):                       # `if policy_moe_loss:` tensor truthiness (:575)
        moe_logged = moe
        losses = losses + moe
    else:
        moe_logged = torch.full_like(align, -1.0)
    return losses.mean(), {"loss": losses.mean(), "loss/align": align.mean(), "loss/moe_balance": moe_logged.mean(),
                           "loss/lm": sft.mean() if sft is not None else None}


def seq_logp(logits, labels, average_log_prob=False):       # dpo_trainer.py:483-495
    labels = labels[:, 1:].clone()
    logits = logits[:, :-1, :]
    mask = labels != LABEL_PAD
    labels[labels == LABEL_PAD] = 0
    per_tok = torch.gather(logits.log_softmax(-1), dim=2, index=labels.unsqueeze(2)).squeeze(2)
    if average_log_prob:
        return (per_tok * mask).sum(-1) / mask.sum(-1)
    return (per_tok * mask).sum(-1)


def dpo_loss(pc, pr, rc, rr, beta=0.1, label_smoothing=0.0, loss_type="sigmoid"):   # dpo_trainer.py:497-562
    logits = (pc - pr) - (rc - rr)
    if loss_type == "sigmoid":
        losses = -F.logsigmoid(beta * logits) * (1 - label_smoothing) - F.logsigmoid(-beta * logits) * label_smoothing
    elif loss_type == "hinge":
        losses = torch.relu(1 - beta * logits)
    elif loss_type == "ipo":
        losses = (logits - 1 / (2 * beta)) ** 2
    elif loss_type == "kto_pair":
        ckl = (pc - rc).mean().clamp(min=0)
        rkl = (pr - rr).mean().clamp(min=0)
        losses = torch.cat((1 - torch.sigmoid(beta * ((pc - rc) - rkl)), 1 - torch.sigmoid(beta * (ckl - (pr - rr)))), 0)
    else:
        raise ValueError(f"Unknown loss type: {loss_type}")
    return losses, beta * (pc - rc).detach(), beta * (pr - rr).detach()


def preference_loss(s_ch, s_rj, t_ch, t_rj, beta=0.1, label_smoothing=0.0, loss_type="sigmoid", moe_loss_enable=True):
    """DPOTrainer.compute_loss (:564-641) from the four model outputs."""
    pc, pr = seq_logp(s_ch.logits, s_ch.labels), seq_logp(s_rj.logits, s_rj.labels)
    with torch.no_grad():
        rc, rr = seq_logp(t_ch.logits, t_ch.labels), seq_logp(t_rj.logits, t_rj.labels)
    reward_losses, cr, rj = dpo_loss(pc, pr, rc, rr, beta, label_smoothing, loss_type)
    mc = s_ch.moe_loss if moe_loss_enable else None
    mr = s_rj.moe_loss if moe_loss_enable else None
    if mc is not None and mr is not None and bool(mc) and bool(mr):
        moe = mc + mr
        losses = reward_losses + moe
    else:
        moe = torch.full_like(reward_losses, -1.0)
        losses = reward_losses
    return losses.mean(), {"loss": losses.mean(), "loss/reward": reward_losses.mean(), "loss/moe_balance": moe.mean(),
                           "rewards/chosen": cr.mean(), "rewards/rejected": rj.mean(),
                           "rewards/accuracies": (cr > rj).float().mean(), "rewards/margins": (cr - rj).mean(),
                           "logps/chosen": pc.detach().mean(), "logps/rejected": pr.detach().mean()}
