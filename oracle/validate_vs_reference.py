"""Pin the oracle against the IMPORTED reference and (re)generate tests/golden/*.  Authoring container
only: needs /root/reference, which never travels to the GPU box.  TEST INFRASTRUCTURE ONLY.

    python oracle/validate_vs_reference.py            # check + write fixtures
    python oracle/validate_vs_reference.py --check    # check only

What is compared (reference symbol -> oracle symbol), all fp32 on CPU, tolerance 1e-5 unless noted:
  qwen2/modeling_qwen2.py  Qwen2RMSNorm, apply_rotary_pos_emb, Qwen2MLP, Qwen2DecoderLayer (eager),
                           Qwen2ForCausalLM (logits, shifted CE, grads)          -> oracle.decoder
  llava_arch.prepare_inputs_labels_for_multimodal (ragged right-padded batch)    -> oracle.vision.splice
  multimodal_projector.builder.build_projector (mlp2x_gelu)                      -> oracle.vision.Projector
  multimodal_encoder.clip_encoder.CLIPVisionTower over transformers.CLIPVisionModel -> oracle.vision.VisionTower
  llava_qwen2.LlavaQwen2ForCausalLM end-to-end (dense teacher)                   -> oracle.llava.LlavaOracle
  llava_qwen2_moe.{MoEQwen2DecoderLayer_forward, MoEQwen2Model_forward} patched onto the vendored
      Qwen2Model with oracle.moe.OracleMoE standing in for deepspeed.moe.layer.MoE -> oracle student
  AlignTrainer.{get_p, get_logp, compute_align_loss}, DPOTrainer.{get_logp, dpo_loss} -> oracle.losses
deepspeed.moe itself is absent (not vendored, not installed): oracle/moe.py is a restatement and
that boundary stays "parity unpinned".
"""
import argparse
import importlib
import importlib.machinery
import json
import os
import sys
import tempfile
import types
from types import SimpleNamespace

sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import transformers  # noqa: E402,F401

from oracle import losses as olosses  # noqa: E402
from oracle import moe as omoe  # noqa: E402
from oracle.decoder import CausalLM, DecoderConfig, RMSNorm, apply_rope, rope_tables  # noqa: E402
from oracle.llava import LlavaOracle, freeze_like_d2s, init_weights, mimic_step, sync_experts_from_dense  # noqa: E402
from oracle.vision import IGNORE_INDEX, IMAGE_TOKEN_INDEX, Projector, VisionConfig, VisionTower, splice  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _import_reference():
    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = m

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    for n, p in [("llavamod", ""), ("llavamod.model", "/model"), ("llavamod.model.language_model", "/model/language_model"),
                 ("llavamod.model.multimodal_encoder", "/model/multimodal_encoder"),
                 ("llavamod.model.multimodal_projector", "/model/multimodal_projector"), ("llavamod.train", "/train")]:
        ns(n, REF + "/llavamod" + p)

    class _Block(nn.Module):
        pass
    stub("timm"); stub("timm.models"); stub("timm.models.vision_transformer", Block=_Block)
    stub("deepspeed"); stub("deepspeed.moe"); stub("deepspeed.moe.layer", MoE=omoe.OracleMoE)
    import transformers.trainer as tt
    if not hasattr(tt, "ALL_LAYERNORM_LAYERS"):
        tt.ALL_LAYERNORM_LAYERS = [nn.LayerNorm]
    R = SimpleNamespace()
    R.q2 = importlib.import_module("llavamod.model.language_model.qwen2.modeling_qwen2")
    R.cfg = importlib.import_module("llavamod.model.language_model.qwen2.configuration_qwen2")
    R.arch = importlib.import_module("llavamod.model.llava_arch")
    R.moe = importlib.import_module("llavamod.model.language_model.llava_qwen2_moe")
    R.lq = importlib.import_module("llavamod.model.language_model.llava_qwen2")
    R.proj = importlib.import_module("llavamod.model.multimodal_projector.builder")
    R.clip = importlib.import_module("llavamod.model.multimodal_encoder.clip_encoder")
    R.at = importlib.import_module("llavamod.train.align_trainer")
    R.dt = importlib.import_module("llavamod.train.dpo_trainer")
    # data/dataset.py (the two collators): its import chain wants PIL and `llavamod.model`'s star exports
    # (`transformers` among them) — stubbed; the collator classes themselves are pure torch
    if "PIL" not in sys.modules:
        try:
            import PIL  # noqa: F401
        except ImportError:
            stub("PIL", Image=types.SimpleNamespace(Image=object)); stub("PIL.Image")
    sys.modules["llavamod.model"].transformers = transformers
    ns("llavamod.data", REF + "/llavamod/data")
    R.data = importlib.import_module("llavamod.data.dataset")
    return R


def maxdiff(a, b):
    return (a.float() - b.float()).abs().max().item()


def check(name, a, b, tol=1e-5):
    d = maxdiff(a, b)
    scale = max(b.float().abs().max().item(), 1e-30)
    ok = d <= tol * max(1.0, scale)
    print(f"  {'OK ' if ok else 'BAD'} {name}: max|diff|={d:.3e} (scale {scale:.3e})")
    assert ok, name
    return d


# ----------------------------------------------------------------------------------------------- configs
def tiny_cfgs():
    """Config 1 of BASELINE.json / SURVEY §8d."""
    vc = VisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=4,
                      image_size=28, patch_size=14, select_layer=-2)
    sc = DecoderConfig(vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                       num_attention_heads=4, num_key_value_heads=2, moe_layers_idx=[0], num_experts=4,
                       top_k_experts=2, capacity_factor=1.5, eval_capacity_factor=2.0, min_capacity=0,
                       router_aux_loss_coef=0.01)
    tc = DecoderConfig(vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                       num_attention_heads=4, num_key_value_heads=2)
    return vc, sc, tc


def small_gpu_cfgs():
    """Smallest shapes the MI355X kernels accept (head_dim 64): the GPU-side golden case."""
    vc = VisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=1,
                      image_size=28, patch_size=14, select_layer=-2)
    sc = DecoderConfig(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                       num_attention_heads=2, num_key_value_heads=1, moe_layers_idx=[0], num_experts=4,
                       top_k_experts=2, capacity_factor=1.5, eval_capacity_factor=2.0, min_capacity=0,
                       router_aux_loss_coef=0.01)
    tc = DecoderConfig(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                       num_attention_heads=2, num_key_value_heads=2)
    return vc, sc, tc


def tiny_batch(seed=1, B=2, T=8, vocab=512, ragged=False):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab - 12, (B, T), generator=g)
    ids[:, 2] = IMAGE_TOKEN_INDEX
    labels = ids.clone()
    labels[:, :3] = IGNORE_INDEX
    mask = torch.ones(B, T, dtype=torch.bool)
    if ragged:
        pad_id = vocab - 1
        ids[1, T - 3:] = pad_id
        labels[1, T - 3:] = IGNORE_INDEX
        mask = ids.ne(pad_id)
    images = torch.randn(B, 3, 28, 28, generator=g)
    return dict(input_ids=ids, attention_mask=mask, labels=labels, images=images)


def ref_qwen2_config(R, c: DecoderConfig, attn="eager"):
    cfg = R.cfg.Qwen2Config(vocab_size=c.vocab_size, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                            num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                            num_key_value_heads=c.num_key_value_heads, rms_norm_eps=c.rms_norm_eps,
                            rope_theta=c.rope_theta, max_position_embeddings=c.max_position_embeddings,
                            use_sliding_window=False, attention_dropout=0.0, tie_word_embeddings=False)
    cfg.pad_token_id = None
    cfg._attn_implementation = attn
    cfg.use_cache = False
    return cfg


# ----------------------------------------------------------------------------------------------- checks
def check_decoder_ops(R):
    print("[decoder ops]")
    torch.manual_seed(0)
    x = torch.randn(2, 5, 64)
    rn = R.q2.Qwen2RMSNorm(64, eps=1e-6); on = RMSNorm(64, 1e-6)
    w = 1 + 0.1 * torch.randn(64)
    rn.weight.data.copy_(w); on.weight.data.copy_(w)
    check("Qwen2RMSNorm", on(x), rn(x))
    q, k = torch.randn(2, 4, 5, 16), torch.randn(2, 2, 5, 16)
    rot = R.q2.Qwen2RotaryEmbedding(16, max_position_embeddings=64, base=10000.0)
    cos, sin = rot(q, seq_len=5)
    oc, osn = rope_tables(16, 64, 10000.0, torch.float32)
    check("rotary cos table", oc[:5], cos); check("rotary sin table", osn[:5], sin)
    pid = torch.arange(5)[None]
    rq, rk = R.q2.apply_rotary_pos_emb(q, k, cos, sin, pid)
    oq, ok = apply_rope(q, k, oc, osn, pid)
    check("apply_rotary_pos_emb q", oq, rq); check("apply_rotary_pos_emb k", ok, rk)


def check_causal_lm(R):
    print("[Qwen2ForCausalLM, dense]")
    _, _, tc = tiny_cfgs()
    ours = init_weights(CausalLM(tc), seed=0)
    for attn in ("eager", "sdpa"):
        ref = R.q2.Qwen2ForCausalLM(ref_qwen2_config(R, tc, attn))
        missing = ref.load_state_dict(ours.state_dict(), strict=False)
        assert not [m for m in missing.missing_keys if "rotary" not in m and "inv_freq" not in m], missing
        g = torch.Generator().manual_seed(3)
        ids = torch.randint(0, 512, (2, 9), generator=g)
        labels = ids.clone(); labels[:, :4] = -100
        am = torch.ones(2, 9, dtype=torch.long); am[1, 6:] = 0
        ro = ref(input_ids=ids, attention_mask=am, labels=labels, return_dict=True)
        oo = ours.forward_embeds(ours.model.embed_tokens(ids), am.bool(), labels)
        # padded query rows are unconstrained garbage in both (different mask fill) -> compare live rows
        live = am.bool()
        check(f"logits ({attn})", oo.logits[live], ro.logits[live], 2e-5)
        check(f"shifted CE loss ({attn})", oo.loss, ro.loss, 2e-5)
    ref.zero_grad(); ours.zero_grad()
    ro.loss.backward(); oo.loss.backward()
    check("grad gate_proj L0", ours.model.layers[0].mlp.gate_proj.weight.grad, ref.model.layers[0].mlp.gate_proj.weight.grad, 1e-4)
    check("grad embed", ours.model.embed_tokens.weight.grad, ref.model.embed_tokens.weight.grad, 1e-4)


class _Harness(nn.Module):
    """10-line harness around the reference's LlavaMetaForCausalLM mixin for the splice check."""
    pass


def check_splice(R):
    print("[prepare_inputs_labels_for_multimodal]")
    emb = nn.Embedding(512, 64)
    torch.manual_seed(4)
    feats = torch.randn(3, 4, 64)

    class H(R.arch.LlavaMetaForCausalLM, nn.Module):
        def __init__(self):
            nn.Module.__init__(self)
            self.config = SimpleNamespace()
            self.embed_tokens = emb
            self.device = torch.device("cpu")
        def get_model(self): return self
        def get_image_tower(self): return object()
        def get_video_tower(self): return None
        def encode_images(self, images): return feats

    h = H()
    ids = torch.randint(0, 500, (3, 10), generator=torch.Generator().manual_seed(5))
    ids[0, 2] = IMAGE_TOKEN_INDEX; ids[1, 0] = IMAGE_TOKEN_INDEX; ids[2, 6] = IMAGE_TOKEN_INDEX
    labels = ids.clone(); labels[:, :4] = IGNORE_INDEX
    am = torch.ones(3, 10, dtype=torch.bool); am[1, 7:] = False; am[2, 9:] = False
    images = [torch.zeros(3, 28, 28) for _ in range(3)]
    _, pos, ram, _, remb, rlab = h.prepare_inputs_labels_for_multimodal(ids, None, am, None, labels, images)
    oemb, oam, olab = splice(emb, feats, ids, am, labels)
    check("spliced embeds", oemb, remb); assert torch.equal(olab, rlab) and torch.equal(oam, ram.bool()) and pos is None
    print(f"  OK  labels/mask identical; S'={oemb.shape[1]}, mask sums {oam.sum(1).tolist()}")


def check_projector_and_clip(R):
    print("[mm_projector, CLIPVisionTower]")
    vc, _, _ = tiny_cfgs()
    pc = SimpleNamespace(mm_image_tower="x", mm_video_tower=None, image_projector_type="mlp2x_gelu", mm_hidden_size=32, hidden_size=64)
    rp = R.proj.build_projector(pc)
    op = init_weights(Projector(32, 64), seed=7)
    rp.load_state_dict(op.state_dict())
    x = torch.randn(2, 4, 32)
    check("projector.forward_image", op.forward_image(x), rp.forward_image(x))
    from transformers import CLIPVisionConfig, CLIPVisionModel
    hc = CLIPVisionConfig(hidden_size=vc.hidden_size, intermediate_size=vc.intermediate_size,
                          num_hidden_layers=vc.num_hidden_layers, num_attention_heads=vc.num_attention_heads,
                          image_size=vc.image_size, patch_size=vc.patch_size, hidden_act="quick_gelu",
                          layer_norm_eps=vc.layer_norm_eps)
    hf = CLIPVisionModel(hc).eval()
    ov = VisionTower(vc)
    # transformers 5.x dropped the `vision_model.` prefix that 4.37 (the reference's pin) checkpoints carry
    ov.load_state_dict({(k if k.startswith("vision_model.") else "vision_model." + k): v
                        for k, v in hf.state_dict().items() if "position_ids" not in k})
    # the reference wrapper: needs a checkpoint dir whose name contains "openai"
    tmp = tempfile.mkdtemp(prefix="openai_clip_tiny_")
    hf.save_pretrained(tmp)
    json.dump({"do_resize": False, "do_center_crop": False, "do_normalize": False, "image_processor_type": "CLIPImageProcessor"},
              open(os.path.join(tmp, "preprocessor_config.json"), "w"))
    tower = R.clip.CLIPVisionTower(tmp, SimpleNamespace(mm_vision_select_layer=-2, mm_vision_select_feature="patch"))
    img = torch.randn(2, 3, 28, 28)
    check("CLIPVisionTower.forward (hidden_states[-2][:,1:])", ov(img), tower(img), 2e-5)
    return tmp


def build_pair(seed=0, cfgs=None):
    vc, sc, tc = cfgs if cfgs is not None else tiny_cfgs()
    teacher = init_weights(LlavaOracle(tc, vc, moe=False), seed=seed + 100)
    student = sync_experts_from_dense(init_weights(LlavaOracle(sc, vc, moe=True), seed=seed))
    return student, teacher


def check_llava_dense(R, clip_dir):
    print("[LlavaQwen2ForCausalLM end-to-end, dense teacher]")
    vc, _, tc = tiny_cfgs()
    _, teacher = build_pair()
    cfg = ref_qwen2_config(R, tc, "eager")
    ref = R.lq.LlavaQwen2ForCausalLM(cfg)
    margs = SimpleNamespace(image_tower=clip_dir, video_tower=None, mm_vision_select_layer=-2, mm_vision_select_feature="patch",
                            pretrain_mm_mlp_adapter=None, image_projector_type="mlp2x_gelu", video_projector_type="linear",
                            video_global_proj=False, video_temproal_proj=False, video_spatial_proj=False)
    ref.get_model().initialize_vision_modules(margs, fsdp=None)
    sd = {}
    ref_keys = set(ref.state_dict().keys())
    for k, v in teacher.state_dict().items():
        if k.startswith("lm."):
            sd[k[3:]] = v
        elif k.startswith("image_tower."):
            kk = "model.image_tower.image_tower." + k[len("image_tower."):]
            if kk not in ref_keys:          # transformers 5.x: no `vision_model.` level
                kk = kk.replace("image_tower.vision_model.", "image_tower.")
            sd[kk] = v
        elif k.startswith("mm_projector."):
            sd["model." + k] = v
    res = ref.load_state_dict(sd, strict=False)
    assert not [m for m in res.missing_keys if "rotary" not in m and "position_ids" not in m], res.missing_keys
    assert not res.unexpected_keys, res.unexpected_keys
    for ragged in (False, True):
        b = tiny_batch(ragged=ragged)
        ro = ref(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"],
                 images=list(b["images"]), return_dict=True)
        oo = teacher(**b)
        live = (ro.labels != 12345) if not ragged else None
        if ragged:
            am = torch.zeros_like(oo.labels, dtype=torch.bool)
            lens = [int(m.sum()) - 1 + 4 for m in b["attention_mask"]]
            for i, n in enumerate(lens):
                am[i, :n] = True
            live = am
        assert torch.equal(oo.labels, ro.labels)
        check(f"teacher logits (ragged={ragged})", oo.logits[live], ro.logits[live], 2e-5)
        check(f"teacher CE (ragged={ragged})", oo.loss, ro.loss, 2e-5)
    return ref


def check_moe_patched_student(R):
    print("[MoE-patched layer/model forward with OracleMoE as deepspeed.moe.layer.MoE]")
    vc, sc, _ = tiny_cfgs()
    student, _ = build_pair()
    cfg = ref_qwen2_config(R, sc, "eager")
    ref = R.q2.Qwen2ForCausalLM(cfg)
    # replicate initialize_moe_modules :536-559 on the vendored model
    for li in sc.moe_layers_idx:
        ref.model.layers[li].mlp = omoe.OracleMoE(sc.hidden_size, ref.model.layers[li].mlp, sc.num_experts, 1, sc.top_k_experts,
                                                 sc.capacity_factor, sc.eval_capacity_factor, sc.min_capacity, False)
    for m in ref.model.layers:
        m.forward = R.moe.MoEQwen2DecoderLayer_forward(m)
    ref.model.forward = R.moe.MoEQwen2Model_forward(ref.model)
    ref.model._attn_implementation = "eager"
    sd = {k[3:]: v for k, v in student.state_dict().items() if k.startswith("lm.")}
    res = ref.load_state_dict(sd, strict=False)
    assert not [m for m in res.missing_keys if "rotary" not in m], res.missing_keys
    assert not res.unexpected_keys
    g = torch.Generator().manual_seed(2)
    noise = omoe.gumbel_noise((2 * 9, sc.num_experts), g)
    for nz in (None, noise):
        ids = torch.randint(0, 512, (2, 9), generator=torch.Generator().manual_seed(3))
        emb = student.lm.model.embed_tokens(ids)
        ref.model.layers[0].mlp.noise = nz
        student.set_gate_noise([nz])
        ref.train(); student.train()
        out = ref.model(inputs_embeds=emb, attention_mask=None, return_dict=True)
        h, ml = student.lm.model(emb, None)
        check(f"student hidden (noise={'gumbel' if nz is not None else 'none'})", h, out.last_hidden_state, 2e-5)
        check("l_aux", ml[0], out.moe_loss_list[0], 1e-6)


def check_trainer_fns(R):
    print("[AlignTrainer / DPOTrainer pure functions]")
    g = torch.Generator().manual_seed(9)
    V = 152064                        # teacher vocab of the Qwen2 shells > 151936: exercises the slice
    s_logits = torch.randn(1, 6, V, generator=g)
    t_logits = torch.randn(1, 6, V, generator=g)
    labels = torch.tensor([[-100, -100, 5, 7, -100, 9]])
    me = SimpleNamespace(args=SimpleNamespace(moe_enable=True, distill_all_tokens=False), moe_loss_enable=True, label_pad_token_id=-100)

    class M:
        def __init__(self, lg): self.lg = lg
        def __call__(self, **kw): return SimpleNamespace(logits=self.lg, labels=labels, loss=torch.tensor(1.0), moe_loss=torch.tensor(0.5))
    p, _, _ = R.at.AlignTrainer.get_p(me, M(t_logits), {})
    lp, _, _, _ = R.at.AlignTrainer.get_logp(me, M(s_logits), {})
    check("get_p", olosses.get_p(t_logits), p); check("get_logp", olosses.get_logp(s_logits), lp)
    ra = R.at.AlignTrainer.compute_align_loss(me, lp, p, labels)
    check("compute_align_loss", olosses.compute_align_loss(lp, p, labels), ra)
    # `--distill_all_tokens True` (config/args.py:110; align_trainer.py:516-520): the mask is all ones, ignored / padding rows included
    me_all = SimpleNamespace(args=SimpleNamespace(moe_enable=True, distill_all_tokens=True), moe_loss_enable=True, label_pad_token_id=-100)
    ra_all = R.at.AlignTrainer.compute_align_loss(me_all, lp, p, labels)
    check("compute_align_loss (distill_all_tokens)", olosses.compute_align_loss(lp, p, labels, distill_all_tokens=True), ra_all)
    assert abs(float(ra_all) - float(ra)) > 1e-6, "the all-tokens mask must change the loss on a batch with ignored rows"
    md = SimpleNamespace(args=SimpleNamespace(moe_enable=True), moe_loss_enable=True, label_pad_token_id=-100, beta=0.1,
                         label_smoothing=0.0, loss_type="sigmoid")
    small = torch.randn(2, 6, 512, generator=g)
    lab2 = torch.tensor([[-100, -100, 5, 7, -100, 9], [-100, 3, 4, -100, -100, -100]])

    class M2:
        def __call__(self, **kw): return SimpleNamespace(logits=small, labels=lab2, loss=None, moe_loss=None)
    rl, _, _ = R.dt.DPOTrainer.get_logp(md, M2(), {})
    check("DPOTrainer.get_logp", olosses.seq_logp(small, lab2), rl)
    pc, pr = torch.tensor([-10.0, -12.0]), torch.tensor([-11.0, -11.0])
    rc, rr = torch.tensor([-10.5, -12.5]), torch.tensor([-10.0, -12.0])
    known = {}
    for lt in ("sigmoid", "hinge", "ipo", "kto_pair"):
        md.loss_type = lt
        rl_, rcw, rrw = R.dt.DPOTrainer.dpo_loss(md, pc, pr, rc, rr)
        ol, ocw, orw = olosses.dpo_loss(pc, pr, rc, rr, 0.1, 0.0, lt)
        check(f"dpo_loss {lt}", ol, rl_); check(f"rewards {lt}", ocw, rcw)
        known[lt] = [round(float(x), 6) for x in rl_]
    return {"pc": pc.tolist(), "pr": pr.tolist(), "rc": rc.tolist(), "rr": rr.tolist(), "beta": 0.1, "losses": known}


# ----------------------------------------------------------------------------------------------- fixtures
def collate_cases():
    """Seeded per-sample dicts as the reference's datasets hand them to the collators (data/dataset.py:150-158, :400-428):
    ragged lengths, one sample longer than model_max_length, a sample with two images, a sample whose image is a bare tensor."""
    def inst(n, seed, n_img=1, bare=False):
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(0, 1000, (n,), generator=g)
        lab = ids.clone(); lab[: n // 2] = IGNORE_INDEX
        imgs = [torch.randn(3, 4, 4, generator=g) for _ in range(n_img)]
        return dict(input_ids=ids, labels=lab, image=imgs[0] if bare else imgs)
    sft = [inst(5, 1), inst(15, 2, n_img=2), inst(9, 3, bare=True), inst(12, 4)]
    dpo = []
    for k in range(3):
        c, r = inst(6 + 4 * k, 10 + k, n_img=1 + (k == 1)), inst(20 - 5 * k, 20 + k)
        dpo.append(dict(chosen_input_ids=c["input_ids"], chosen_labels=c["labels"], rejected_input_ids=r["input_ids"],
                        rejected_labels=r["labels"], image=c["image"]))
    return sft, dpo


COLLATE_TOK = dict(pad_token_id=151646, model_max_length=12)


def check_collators(R, write):
    """The IMPORTED DataCollatorForSupervisedDataset / DataCollatorForDPODataset (data/dataset.py:167-232, :434-505) on seeded
    instances; their outputs become tests/golden/collate.safetensors, which tests/test_collate_cpu.py holds the product's
    collators to (VERDICT r03 next #5a)."""
    tok = SimpleNamespace(**COLLATE_TOK)
    sft, dpo = collate_cases()
    bs = R.data.DataCollatorForSupervisedDataset(tokenizer=tok)(sft)
    bd = R.data.DataCollatorForDPODataset(tokenizer=tok)(dpo)
    assert bs["input_ids"].shape == (4, 12) and bs["attention_mask"].dtype == torch.bool and len(bs["images"]) == 5
    assert bd["chosen_input_ids"].shape[1] == 14 and bd["rejected_input_ids"].shape[1] == 20 and len(bd["images"]) == 4
    out = {}
    for tag, b in (("sft", bs), ("dpo", bd)):
        for k, v in b.items():
            if k == "images":
                out[f"{tag}.images"] = torch.stack(v)
            else:
                out[f"{tag}.{k}"] = v.to(torch.int64) if v.dtype == torch.bool else v
    print("  OK  imported collators ran: sft", tuple(bs["input_ids"].shape), "dpo", tuple(bd["chosen_input_ids"].shape),
          tuple(bd["rejected_input_ids"].shape))
    if write:
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "collate.safetensors"))
    else:
        from safetensors.torch import load_file
        old = load_file(os.path.join(GOLD, "collate.safetensors"))
        assert set(old) == set(out) and all(torch.equal(old[k], out[k]) for k in out), "committed collate fixture is stale"
        print("  OK  committed collate fixture == imported collators")


def write_golden(dpo_known, ref_teacher):
    from safetensors.torch import save_file
    os.makedirs(GOLD, exist_ok=True)
    student, teacher = build_pair()
    freeze_like_d2s(student)
    json.dump(dpo_known, open(os.path.join(GOLD, "dpo_known_answers.json"), "w"), indent=1)
    save_file({k: v.contiguous() for k, v in student.state_dict().items()}, os.path.join(GOLD, "config1_student.safetensors"))
    save_file({k: v.contiguous() for k, v in teacher.state_dict().items()}, os.path.join(GOLD, "config1_teacher.safetensors"))
    out = {}
    meta = {}
    for tag, ragged, use_noise, loss_type in (("plain", False, False, "only_kd"), ("ragged_noise_kdlm", True, True, "kd_lm")):
        b = tiny_batch(ragged=ragged)
        Sp = b["input_ids"].shape[1] - 1 + 4
        noise = omoe.gumbel_noise((2 * Sp, 4), torch.Generator().manual_seed(2)) if use_noise else None
        student.zero_grad(); student.train(); teacher.eval()
        student.set_gate_noise([noise])
        loss, logs, s_out, t_out = mimic_step(student, teacher, b, loss_type=loss_type, align_vocab=512)
        for k, v in b.items():
            out[f"{tag}.batch.{k}"] = v.to(torch.int64) if v.dtype in (torch.bool,) else v
        if noise is not None:
            out[f"{tag}.gate_noise"] = noise
        out[f"{tag}.student_logits"] = s_out.logits.detach()
        out[f"{tag}.teacher_logits"] = t_out.logits.detach()
        out[f"{tag}.labels"] = s_out.labels
        for n in ("lm.model.layers.0.mlp.deepspeed_moe.gate.wg.weight",
                  "lm.model.layers.0.mlp.deepspeed_moe.experts.deepspeed_experts.1.gate_proj.weight",
                  "lm.model.layers.1.mlp.down_proj.weight", "mm_projector.image_spatial_proj.0.weight",
                  "mm_projector.image_spatial_proj.2.bias"):
            out[f"{tag}.grad.{n}"] = dict(student.named_parameters())[n].grad.detach().clone()
        meta[tag] = {"loss_type": loss_type, "ragged": ragged, "noise": use_noise, "align_vocab": 512,
                     **{k: float(v) for k, v in logs.items() if v is not None}}
        if tag == "plain":   # the imported reference teacher must agree with the fixture it is about to pin
            ro = ref_teacher(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"],
                             images=list(b["images"]), return_dict=True)
            check("fixture teacher logits == imported reference", t_out.logits, ro.logits, 2e-5)
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "config1_mimic.safetensors"))
    json.dump(meta, open(os.path.join(GOLD, "config1_mimic.json"), "w"), indent=1)
    write_gpu_small()
    # MoE restatement golden (self-generated: parity unpinned)
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(40, 4, generator=g)
    nz = omoe.gumbel_noise((40, 4), g)
    l_aux, comb, disp, cnt = omoe.top2gating(logits, 1.0, 0, nz)
    l1, c1, d1, n1 = omoe.top1gating(logits, 1.0, 0, None)
    save_file({"logits": logits, "noise": nz, "top2.l_aux": l_aux.reshape(1), "top2.combine": comb, "top2.exp_counts": cnt,
               "top1.l_aux": l1.reshape(1), "top1.combine": c1, "top1.exp_counts": n1}, os.path.join(GOLD, "moe_gating.safetensors"))
    print("wrote", sorted(os.listdir(GOLD)))


def write_gpu_small():
    """Golden vectors at head_dim 64 for the -m gpu parity test.  Weights are rounded to bf16 first (the
    GPU path holds bf16 weights); the oracle then computes in fp32 on those values."""
    from safetensors.torch import save_file
    student, teacher = build_pair(seed=7, cfgs=small_gpu_cfgs())
    for m in (student, teacher):
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "gate.wg" not in n:
                    p.copy_(p.to(torch.bfloat16).float())
    freeze_like_d2s(student)
    save_file({k: v.contiguous() for k, v in student.state_dict().items()}, os.path.join(GOLD, "gpusmall_student.safetensors"))
    save_file({k: v.contiguous() for k, v in teacher.state_dict().items()}, os.path.join(GOLD, "gpusmall_teacher.safetensors"))
    out, meta = {}, {}
    for tag, ragged, loss_type in (("plain", False, "only_kd"), ("ragged_kdlm", True, "kd_lm")):
        b = tiny_batch(seed=5, B=2, T=12, ragged=ragged)
        b["images"] = b["images"].to(torch.bfloat16).float()
        student.zero_grad(); student.train(); teacher.eval()
        student.set_gate_noise([None])
        loss, logs, s_out, t_out = mimic_step(student, teacher, b, loss_type=loss_type, align_vocab=512)
        for k, v in b.items():
            out[f"{tag}.batch.{k}"] = v.to(torch.int64) if v.dtype == torch.bool else v
        out[f"{tag}.student_logits"] = s_out.logits.detach()
        out[f"{tag}.teacher_logits"] = t_out.logits.detach()
        out[f"{tag}.labels"] = s_out.labels
        for n, p in student.named_parameters():
            if p.grad is not None:
                out[f"{tag}.grad.{n}"] = p.grad.detach().clone()
        meta[tag] = {"loss_type": loss_type, "ragged": ragged, "align_vocab": 512,
                     **{k: float(v) for k, v in logs.items() if v is not None}}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(GOLD, "gpusmall_mimic.safetensors"))
    json.dump(meta, open(os.path.join(GOLD, "gpusmall_mimic.json"), "w"), indent=1)


def ref_greedy(ref, ids, mask, images, n_new):
    """N-token greedy continuation by the IMPORTED reference model: its own forward (llava_qwen2.py:57-130 — splice, decoder,
    lm_head) on the growing right-padded batch, argmax at every sample's last real position, no cache (the vendored 4.37-era
    class has no GenerationMixin under the installed transformers; this loop is what `generate(do_sample=False)` computes).
    Returns tokens [B, n_new] and the top-1 / top-2 logit margin of every decision."""
    B = ids.shape[0]
    seqs = [ids[i, :int(mask[i].sum())].clone() for i in range(B)]
    toks, margins = [], []
    n_img_tokens = 4                                           # 28 px / patch 14: 4 patches replace the one -200 placeholder
    with torch.no_grad():
        for _ in range(n_new):
            T = max(len(q) for q in seqs)
            bi = torch.zeros(B, T, dtype=torch.long); bm = torch.zeros(B, T, dtype=torch.bool)
            for i, q in enumerate(seqs):
                bi[i, :len(q)] = q; bm[i, :len(q)] = True
            lg = ref(input_ids=bi, attention_mask=bm, images=list(images), return_dict=True).logits
            row, mrow = [], []
            for i, q in enumerate(seqs):
                last = lg[i, len(q) - 1 + n_img_tokens - 1].float()
                top = torch.topk(last, 2)
                row.append(int(top.indices[0])); mrow.append(float(top.values[0] - top.values[1]))
                seqs[i] = torch.cat([q, top.indices[:1]])
            toks.append(row); margins.append(mrow)
    return torch.tensor(toks).t().contiguous(), torch.tensor(margins).t().contiguous()


def ref_greedy_fixture(ref, n_new=8, tries=400):
    """The prompt batch whose reference continuation has the LARGEST smallest decision margin among `tries` seeded ragged batches
    (a bf16 model can only be held token-for-token to decisions that are not near-ties; the margins are stored beside the
    tokens and the GPU test prints them).  VERDICT r03 next #5b."""
    best = None
    for seed in range(100, 100 + tries):
        b = tiny_batch(seed=seed, B=2, T=12, ragged=True)
        im = b["images"].to(torch.bfloat16).float()
        toks, mg = ref_greedy(ref, b["input_ids"], b["attention_mask"], im, n_new)
        if best is None or float(mg.min()) > best[0]:
            best = (float(mg.min()), seed, b, im, toks, mg)
    mmin, seed, b, im, toks, mg = best
    print(f"  greedy fixture: prompt seed {seed}, {n_new} new tokens per sample, smallest top-1/top-2 margin {mmin:.4f}; tokens {toks.tolist()}")
    return {"gen_input_ids": b["input_ids"], "gen_attention_mask": b["attention_mask"].to(torch.int64), "gen_images": im,
            "gen_tokens": toks, "gen_margin": mg}


def write_ref_checkpoint(R):
    """A checkpoint WRITTEN BY THE REFERENCE (VERDICT r02 missing #3): the imported `LlavaQwen2ForCausalLM` at the GPU-small
    geometry (head_dim 64) is saved by its own `save_pretrained` — config.json + model.safetensors in the reference's key
    layout (`model.mm_projector.image_spatial_proj.*`, `model.image_tower.image_tower.*`, `lm_head`, decoder layers) — next
    to the tiny CLIP directory `CLIPVisionModel.save_pretrained` wrote (the directory `--image_tower` names), the adapter-only
    `mm_projector.bin` as `safe_save_model_for_hf_trainer` writes it (train/align_train.py:623-631), and the reference's own
    logits / labels / loss on a ragged batch.  tests/ load it through the product `from_pretrained` (GPU) and through the
    name mapping alone (CPU).  Tensors + JSON only; weights are bf16-representable so the bf16 product path holds them exactly."""
    from safetensors.torch import save_file
    from transformers import CLIPVisionConfig, CLIPVisionModel
    print("[reference-written checkpoint fixture]")
    out_dir = os.path.join(GOLD, "ref_ckpt")
    clip_name = "openai_clip_tiny"                           # the reference's builder wants "openai" in the path
    os.makedirs(out_dir, exist_ok=True)
    vc, _, tc = small_gpu_cfgs()
    hc = CLIPVisionConfig(hidden_size=vc.hidden_size, intermediate_size=vc.intermediate_size,
                          num_hidden_layers=vc.num_hidden_layers, num_attention_heads=vc.num_attention_heads,
                          image_size=vc.image_size, patch_size=vc.patch_size, hidden_act="quick_gelu",
                          layer_norm_eps=vc.layer_norm_eps)
    torch.manual_seed(21)
    hf = CLIPVisionModel(hc).eval()
    with torch.no_grad():
        for n_, p_ in hf.named_parameters():
            if "norm" in n_ and n_.endswith("weight"):
                p_.fill_(1.0)
            elif "norm" in n_ and n_.endswith("bias"):
                p_.zero_()
            else:
                p_.copy_((0.05 * torch.randn(p_.shape)).to(torch.bfloat16).float())
    hf.save_pretrained(os.path.join(out_dir, clip_name))
    json.dump({"do_resize": False, "do_center_crop": False, "do_normalize": False, "image_processor_type": "CLIPImageProcessor"},
              open(os.path.join(out_dir, clip_name, "preprocessor_config.json"), "w"))
    cwd = os.getcwd()
    os.chdir(out_dir)                                        # the tower is named RELATIVE to the checkpoint: portable fixture
    try:
        cfg = ref_qwen2_config(R, tc, "eager")
        ref = R.lq.LlavaQwen2ForCausalLM(cfg)
        margs = SimpleNamespace(image_tower=clip_name, video_tower=None, mm_vision_select_layer=-2,
                                mm_vision_select_feature="patch", pretrain_mm_mlp_adapter=None,
                                image_projector_type="mlp2x_gelu", video_projector_type="linear", video_global_proj=False,
                                video_temproal_proj=False, video_spatial_proj=False)
        ref.get_model().initialize_vision_modules(margs, fsdp=None)
        g = torch.Generator().manual_seed(22)
        with torch.no_grad():
            for n_, p_ in ref.named_parameters():
                if "image_tower" in n_:
                    continue                                   # loaded by the reference from the CLIP directory
                if "norm" in n_ and n_.endswith("weight"):
                    p_.fill_(1.0)
                else:
                    p_.copy_((0.02 * torch.randn(p_.shape, generator=g)).to(torch.bfloat16).float())
        ref.eval()
        b = tiny_batch(seed=5, B=2, T=12, ragged=True)
        b["images"] = b["images"].to(torch.bfloat16).float()
        ro = ref(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"],
                 images=list(b["images"]), return_dict=True)
        for f in os.listdir("."):                              # regenerate from scratch
            if f.endswith((".safetensors", ".json", ".bin")) and os.path.isfile(f):
                os.remove(f)
        try:
            ref.config.mm_image_tower = clip_name
            ref.save_pretrained(".", safe_serialization=True)
            how = "reference model.save_pretrained()"
        except Exception as e:                                 # transformers 5.x vs the 4.37-era vendored classes
            print(f"  (save_pretrained failed: {type(e).__name__}: {str(e)[:100]} -> state_dict()/config.to_dict() dump)")
            save_file({k: v.contiguous() for k, v in ref.state_dict().items()}, "model.safetensors", metadata={"format": "pt"})
            json.dump(ref.config.to_dict(), open("config.json", "w"), indent=1, default=str)
            how = "reference state_dict() + config.to_dict()"
        # safe_save_model_for_hf_trainer(tune_mm_mlp_adapter=True): get_mm_adapter_state_maybe_zero_3(named_parameters, ['mm_projector'])
        adapter = {k: v.detach().cpu().clone() for k, v in ref.named_parameters() if "mm_projector" in k}
        torch.save(adapter, "mm_projector.bin")
        live = torch.zeros_like(ro.labels, dtype=torch.bool)
        for i, m in enumerate(b["attention_mask"]):
            live[i, :int(m.sum()) - 1 + 4] = True
        gen = ref_greedy_fixture(ref)
        save_file({"input_ids": b["input_ids"], "attention_mask": b["attention_mask"].to(torch.int64), "labels": b["labels"],
                   "images": b["images"], "ref_logits": ro.logits.detach().contiguous(), "ref_labels": ro.labels,
                   "ref_loss": ro.loss.detach().reshape(1), "live": live.to(torch.int64), **gen}, "expected.safetensors")
        json.dump({"written_by": how, "reference_class": "llavamod.model.language_model.llava_qwen2.LlavaQwen2ForCausalLM",
                   "image_tower": clip_name, "transformers": transformers.__version__,
                   "state_dict_keys": sorted(ref.state_dict().keys())}, open("MANIFEST.json", "w"), indent=1)
        print(f"  wrote {out_dir} via {how}: {sorted(os.listdir('.'))}")
    finally:
        os.chdir(cwd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(8)
    R = _import_reference()
    check_decoder_ops(R)
    check_causal_lm(R)
    check_splice(R)
    clip_dir = check_projector_and_clip(R)
    ref_teacher = check_llava_dense(R, clip_dir)
    check_moe_patched_student(R)
    known = check_trainer_fns(R)
    check_collators(R, write=not a.check)
    if not a.check:
        write_golden(known, ref_teacher)
        write_ref_checkpoint(R)
    print("ALL REFERENCE CHECKS PASSED")


if __name__ == "__main__":
    main()
