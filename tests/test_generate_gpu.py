"""Inference path: single-query attention kernel, greedy generation with a KV cache against cache-free
recomputation, the HF-style `use_cache` forward contract, and save_pretrained -> from_pretrained round trips of the student
(base / FineTune / Eval classes) and the teacher."""
import copy
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _util as U  # noqa: E402
from test_step_parity_gpu import small_cfgs  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


@pytest.mark.parametrize("B,nh,nkv,hd,smax", [(3, 4, 4, 128, 700), (2, 8, 2, 64, 300), (1, 2, 1, 128, 5)])
def test_attn_decode_vs_torch(B, nh, nkv, hd, smax):
    from llavamod import kernels as K
    g = torch.Generator().manual_seed(B + nh + smax)
    q = torch.randn(B, nh * hd, generator=g).to(BF).to(DEV)
    kc = torch.randn(B, smax, nkv * hd, generator=g).to(BF).to(DEV)
    vc = torch.randn(B, smax, nkv * hd, generator=g).to(BF).to(DEV)
    lens = torch.tensor([smax, max(1, smax // 3), 1][:B], dtype=torch.int32, device=DEV)
    out = K.attn_decode(q, kc, vc, lens, nh, nkv, hd, 1 / math.sqrt(hd))
    rep = nh // nkv
    for b in range(B):
        L = int(lens[b])
        kk = kc[b, :L].float().view(L, nkv, hd).repeat_interleave(rep, dim=1)       # [L, nh, hd]
        vv = vc[b, :L].float().view(L, nkv, hd).repeat_interleave(rep, dim=1)
        s = torch.einsum("hd,lhd->hl", q[b].float().view(nh, hd), kk) / math.sqrt(hd)
        ref = torch.einsum("hl,lhd->hd", torch.softmax(s, -1), vv).reshape(-1)
        err = (out[b].float() - ref).abs().max().item()
        assert err <= 2 ** -7 * ref.abs().max().item() + 1e-3, (b, err)


def _pair():
    vc, sc, tc = small_cfgs()
    ssd, tsd = U.load_golden("gpusmall_student.safetensors"), U.load_golden("gpusmall_teacher.safetensors")
    student, teacher = U.build_hip_pair(ssd, tsd, sc, tc, vc, DEV)
    for m in student.moe_layers():
        m.deterministic = True
    return student, teacher


def _prompt():
    g = U.load_golden("gpusmall_mimic.safetensors")
    b = {k.split(".")[-1]: v for k, v in g.items() if k.startswith("ragged_kdlm.batch.")}
    return dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"].bool(), images=b["images"].to(DEV).to(BF))


@pytest.mark.parametrize("which", ["teacher", "student"])
def test_generate_with_kv_cache_equals_recomputation(which):
    """Greedy tokens from the cached path == greedy tokens from re-running the full prompt+generated sequence each step.
    Right-padded ragged prompts: each sample continues from its own length.  (The MoE student in eval mode: capacity from
    eval_capacity_factor over the tokens of the call — a decode step routes B tokens, the recomputation routes them all —
    so the student is compared with drop-free capacity.)"""
    student, teacher = _pair()
    model = teacher if which == "teacher" else student
    if which == "student":
        for m in model.moe_layers():
            m.eval_capacity_factor = 64.0          # no token is ever dropped, in either formulation
    model.eval()
    pr = _prompt()
    new = model.generate(**pr, max_new_tokens=6)
    assert new.shape == (pr["input_ids"].shape[0], 6)
    # cache-free reference: append token by token at each sample's own length and re-run the whole forward
    ids, mask = pr["input_ids"].clone(), pr["attention_mask"].clone()
    B, T = ids.shape
    ids = torch.cat([ids, torch.zeros(B, 6, dtype=ids.dtype)], 1)
    mask = torch.cat([mask, torch.zeros(B, 6, dtype=torch.bool)], 1)
    lens = pr["attention_mask"].sum(1)
    ref = []
    n_img_tokens = None
    for step in range(6):
        with torch.no_grad():
            out = model(input_ids=ids, attention_mask=mask, images=pr["images"])
        n_img_tokens = model.get_image_tower().num_patches - 1   # the <image> token becomes P patch rows
        last = (mask.sum(1) + n_img_tokens - 1).to(DEV)
        tok = out.logits[torch.arange(B, device=DEV), last].argmax(-1).cpu()
        ref.append(tok)
        for b in range(B):
            p = int(lens[b]) + step
            ids[b, p] = tok[b]; mask[b, p] = True
    ref = torch.stack(ref, 1)
    assert torch.equal(new.cpu(), ref), (new.cpu(), ref)
    # HF-style contract: use_cache forward returns the cache, later calls take the last token only
    o1 = model(input_ids=pr["input_ids"], attention_mask=pr["attention_mask"], images=pr["images"], use_cache=True)
    assert o1.logits.shape[:2] == (B, 1) and o1.past_key_values is not None
    t1 = o1.logits[:, 0].argmax(-1)
    assert torch.equal(t1.cpu(), ref[:, 0])
    inp = model.prepare_inputs_for_generation(torch.cat([pr["input_ids"], t1.cpu()[:, None]], 1),
                                              past_key_values=o1.past_key_values, images=pr["images"])
    assert inp["input_ids"].shape == (B, 1) and "images" not in inp
    o2 = model(**inp)
    assert torch.equal(o2.logits[:, 0].argmax(-1).cpu(), ref[:, 1])


def test_save_pretrained_from_pretrained_roundtrip(tmp_path):
    """save_pretrained -> from_pretrained gives bit-identical logits: dense teacher, up-cycled student reloaded through the
    FineTune and Eval classes (MoE layers rebuilt from config.moe), incl. the separate mm_projector.bin file."""
    from llavamod.model import (EvalLLaVAMoDQwen2ForCausalLM, LLaVAMoDQwen2ForCausalLMFineTune, LlavaQwen2ForCausalLM)
    student, teacher = _pair()
    pr = _prompt()
    teacher.save_pretrained(str(tmp_path / "t"))
    t2 = LlavaQwen2ForCausalLM.from_pretrained(str(tmp_path / "t"), attn_implementation="flash_attention_2", device=DEV)
    teacher.eval(); t2.eval()
    with torch.no_grad():
        assert torch.equal(teacher(**pr).logits, t2(**pr).logits)
    files = student.save_pretrained(str(tmp_path / "s"), max_shard_bytes=256 << 10)
    assert "mm_projector.bin" in files and os.path.exists(tmp_path / "s" / "config.json")
    student.eval()
    with torch.no_grad():
        ref = student(**pr)
    for cls in (LLaVAMoDQwen2ForCausalLMFineTune, EvalLLaVAMoDQwen2ForCausalLM):
        s2 = cls.from_pretrained(str(tmp_path / "s"), device=DEV)
        for m in s2.moe_layers():
            m.deterministic = True
        s2.eval()
        with torch.no_grad():
            out = s2(**pr)
        assert torch.equal(out.logits, ref.logits) and float(out.moe_loss) == float(ref.moe_loss), cls.__name__
    assert not any(p.requires_grad for p in s2.parameters())           # Eval: frozen
    new = s2.generate(**pr, max_new_tokens=3)
    assert new.shape == (pr["input_ids"].shape[0], 3)


def test_sampling_generate_and_padding_contract():
    """do_sample: the temperature / top-k / top-p warpers in HF's order over the KV-cache decode loop — top_k = 1 is greedy, a seed makes
    the draw reproducible, different seeds differ, every sampled token lies inside the nucleus; and the prefill refuses a left-padded
    prompt (the cache is addressed as b*S + len - 1: right padding only, the reference's `padding_side="right"`)."""
    _, teacher = _pair()
    p = _prompt()
    greedy = teacher.generate(**p, max_new_tokens=6)
    top1 = teacher.generate(**p, max_new_tokens=6, do_sample=True, top_k=1, seed=3)
    assert torch.equal(greedy, top1)
    a = teacher.generate(**p, max_new_tokens=6, do_sample=True, temperature=1.5, top_p=0.9, seed=11)
    b = teacher.generate(**p, max_new_tokens=6, do_sample=True, temperature=1.5, top_p=0.9, seed=11)
    c = teacher.generate(**p, max_new_tokens=6, do_sample=True, temperature=1.5, top_p=0.9, seed=12)
    assert torch.equal(a, b) and not torch.equal(a, c)
    # the first sampled token lies in the top-p nucleus of the prefill distribution
    _, logits = teacher._prefill(p["input_ids"], p["attention_mask"], p["images"], 1)
    pr = torch.softmax(logits.float() / 1.5, -1)
    sv, si = torch.sort(pr, -1, descending=True)
    keep = (sv.cumsum(-1) - sv) <= 0.9
    for bi in range(a.shape[0]):
        assert int(a[bi, 0]) in set(si[bi][keep[bi]].tolist())
    # text-only prompts go into the cache as they are: left padding is refused (with images the splice strips the pads by mask
    # and re-pads on the right itself, llava_arch.py:228-230,309-318)
    ids = p["input_ids"].clamp_min(0)
    am = torch.ones_like(ids, dtype=torch.bool)
    am[1, :3] = False
    with pytest.raises(ValueError, match="RIGHT-padded"):
        teacher.generate(input_ids=ids, attention_mask=am, max_new_tokens=2)
    am2 = torch.ones_like(ids, dtype=torch.bool)
    am2[1, -3:] = False
    assert teacher.generate(input_ids=ids, attention_mask=am2, max_new_tokens=2).shape == (ids.shape[0], 2)
