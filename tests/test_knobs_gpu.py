"""The reference's training knobs that no other GPU test turns, each as one mimic step against the
oracle on the same seeded inputs, through the C-ABI:

  * `--distill_all_tokens True` (config/args.py:110; shells/train/qwen/dense2sparse_distillation.sh:30;
    train/align_trainer.py:516-520): the KD mask is ALL ONES over the spliced `[B, S']` grid, padding rows of a ragged batch
    included — their logits come from zero embeddings attending to the sample's real keys;
  * `--moe_mode first_half | second_half | dense` (language_model/llava_qwen2_moe.py:517-524): which decoder layers are
    up-cycled (`second_half` / `dense` make the LAST layer sparse, so the loss-row shortcut of a dense last layer is off);
  * more than 8 experts per layer (`--num_experts`, config/args.py:46; llava_qwen2_moe.py:529-531 takes any list).
"""
import copy
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _util as U  # noqa: E402
import test_step_parity_gpu as P  # noqa: E402
from oracle import moe as omoe  # noqa: E402


def test_distill_all_tokens_ragged_step_vs_oracle():
    """All-ones KD mask on a RAGGED batch: the padding rows (two samples are 5 and 10 tokens shorter than the longest) are KD
    rows in both the oracle and the product; loss scalars to 1e-3, every gradient within 2x its bf16 floor."""
    vc, sc, tc = P._mid_cfgs()
    batch = P._mid_batch(41, 3, 48, sc.vocab_size, vc.image_size, True)
    assert not bool(batch["attention_mask"].all())
    r = P._mimic_parity_case(vc, sc, tc, 5, batch, [None, None], "distill_all_tokens (ragged)", distill_all=True)
    # the knob is live in the product's loss plan: every row of the spliced [B, S'] grid is a KD row, pads included
    from llavamod.train.align_trainer import AlignTrainer
    hb = dict(batch, images=batch["images"].to(P.DEV).to(torch.bfloat16))
    n_kd = {}
    for flag in (True, False):
        tr = AlignTrainer(r.student, r.teacher, args=type("A", (), dict(moe_enable=True, distill_all_tokens=flag, loss_type="kd_lm",
                                                                        moe_loss_enable=True))(), align_vocab=sc.vocab_size)
        n_kd[flag] = float(tr._teacher_pass(tr._batch_of(hb)).plan.kd_w.sum())
    B, Sp = batch["input_ids"].shape[0], batch["input_ids"].shape[1] - 1 + vc.num_patches
    n_pad = int((~batch["attention_mask"]).sum())
    assert n_pad > 0 and n_kd[True] == B * Sp and n_kd[False] < n_kd[True] - n_pad


@pytest.mark.parametrize("mode,expect", [("first_half", [0, 1]), ("second_half", [2, 3]), ("dense", [0, 1, 2, 3])])
def test_moe_mode_layer_choice_steps_like_the_oracle(mode, expect):
    """`initialize_moe_modules(moe_mode=...)` with no explicit indices converts exactly the reference's layer set (the oracle
    state dict — built with that set — loads key for key), records it in `config.moe`, and the step matches the oracle."""
    vc, sc, tc = P._mid_cfgs()
    sc = copy.deepcopy(sc)
    sc.moe_layers_idx = list(expect)
    margs = U.moe_args(sc)
    margs.moe_mode, margs.moe_layers_idx = mode, None
    batch = P._mid_batch(51, 2, 40, sc.vocab_size, vc.image_size, True)
    student = P._mimic_parity_case(vc, sc, tc, 7, batch, [None] * len(expect), f"moe_mode={mode}", margs=margs).student
    assert student.config.moe["moe_mode"] == mode and student.config.moe["moe_layers_idx"] == expect
    assert student.config.moe["num_experts"] == [sc.num_experts] * len(expect)
    from llavamod.model.moe_layer import MoE
    assert [i for i, l in enumerate(student.model.layers) if isinstance(l.mlp, MoE)] == expect


def test_unknown_moe_mode_raises_like_the_reference():
    vc, sc, tc = P._mid_cfgs()
    from llavamod.model import LLaVAMoDQwen2ForCausalLM
    scfg, _ = U.hip_configs(sc, vc, moe=True)
    student = LLaVAMoDQwen2ForCausalLM(scfg, device=P.DEV)
    margs = U.moe_args(sc)
    margs.moe_mode, margs.moe_layers_idx = "every_third", None
    with pytest.raises(NotImplementedError, match="Only support"):
        student.initialize_moe_modules(margs)


@pytest.mark.parametrize("experts", [16, 12])
def test_more_than_eight_experts_step_vs_oracle(experts):
    """16 and 12 experts, top-2: routing, capacity slabs, grouped expert GEMMs, combine and the router gradient for E > 8
    against the oracle (Gumbel noise on the second pick for E = 16)."""
    vc, sc, tc = P._mid_cfgs()
    sc = copy.deepcopy(sc)
    sc.num_experts, sc.moe_layers_idx = experts, [1]
    batch = P._mid_batch(61, 3, 56, sc.vocab_size, vc.image_size, False)
    Sp = batch["input_ids"].shape[1] - 1 + vc.num_patches
    noises = [omoe.gumbel_noise((3 * Sp, experts), torch.Generator().manual_seed(77)) if experts == 16 else None]
    student = P._mimic_parity_case(vc, sc, tc, 13, batch, noises, f"{experts} experts top-2", min_agree=0.95).student
    m = student.moe_layers()[0]
    assert m.num_experts == experts and int(m.last_state.slots_used.numel()) == experts
