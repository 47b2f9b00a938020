"""Checkpoint I/O in the reference's key layout (host logic, no GPU): round trip through HF-style shards, the separate
projector file, dense -> up-cycled loading, key dialects."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "llava-mod_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle.decoder import DecoderConfig  # noqa: E402
from oracle.vision import VisionConfig  # noqa: E402
from tests import _util as U  # noqa: E402


def _tiny(moe, seed):
    from llavamod.model import LLaVAMoDQwen2ForCausalLM, LlavaQwen2ForCausalLM
    vc = VisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, image_size=28,
                      patch_size=14, select_layer=-2)
    dc = DecoderConfig(vocab_size=96, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                       num_key_value_heads=1, moe_layers_idx=[0], num_experts=4, top_k_experts=2, capacity_factor=1.5,
                       min_capacity=0)
    cfg, _ = U.hip_configs(dc, vc, moe=moe)
    cfg.init_seed = seed
    model = (LLaVAMoDQwen2ForCausalLM if moe else LlavaQwen2ForCausalLM)(cfg, device="cpu")
    return model, dc


def test_roundtrip_sharded_and_projector_file(tmp_path):
    from llavamod.checkpoint import INDEX, load_checkpoint, read_state, save_checkpoint
    a, dc = _tiny(True, 1)
    a.initialize_moe_modules(U.moe_args(dc))
    with torch.no_grad():                               # make the experts differ from each other
        for i, (n, p) in enumerate(a.named_parameters()):
            p.add_(0.01 * (i % 7))
    files = save_checkpoint(a, str(tmp_path), max_shard_bytes=64 << 10)
    assert INDEX in files and "mm_projector.bin" in files and sum(f.endswith(".safetensors") for f in files) > 1
    keys = set(read_state(str(tmp_path)))
    assert "model.layers.0.mlp.deepspeed_moe.gate.wg.weight" in keys
    assert "model.layers.0.mlp.deepspeed_moe.experts.deepspeed_experts.3.down_proj.weight" in keys
    assert "model.layers.1.mlp.up_proj.weight" in keys and "lm_head.weight" in keys
    assert any(k.startswith("model.image_tower.image_tower.vision_model.") for k in keys)
    proj = torch.load(os.path.join(tmp_path, "mm_projector.bin"))
    assert set(proj) == {k for k in keys if "mm_projector" in k} and len(proj) == 4
    b, _ = _tiny(True, 2)
    b.initialize_moe_modules(U.moe_args(dc))
    assert load_checkpoint(b, str(tmp_path)) == ([], [])
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    # fused storage stays coherent: the q/k/v parameters are still views of one buffer after loading
    att = b.get_model().layers[0].self_attn
    att._qkv.ensure()
    assert att.q_proj.weight.data_ptr() == att._qkv.w.data_ptr()


def test_dense_checkpoint_upcycles_and_key_dialects(tmp_path):
    from safetensors.torch import save_file
    from llavamod.checkpoint import load_checkpoint, save_checkpoint
    dense, dc = _tiny(True, 3)                          # same class BEFORE initialize_moe_modules: dense FFN keys
    save_checkpoint(dense, str(tmp_path / "dense"), projector_file=False)
    moe, _ = _tiny(True, 4)
    moe.initialize_moe_modules(U.moe_args(dc))
    missing, unexpected = load_checkpoint(moe, str(tmp_path / "dense"), strict=False)
    assert missing == ["model.layers.0.mlp.deepspeed_moe.gate.wg.weight"] and unexpected == []
    ex = moe.get_model().layers[0].mlp.deepspeed_moe.experts.deepspeed_experts
    ref = dense.get_model().layers[0].mlp
    for e in ex:
        assert torch.equal(e.gate_proj.weight, ref.gate_proj.weight) and torch.equal(e.down_proj.weight, ref.down_proj.weight)
    # transformers-5.x CLIP keys (no `vision_model.` level) and DeepSpeed's `module.` prefix are accepted
    sd = {("module." + k).replace("image_tower.image_tower.vision_model.", "image_tower.image_tower."): v.contiguous()
          for k, v in dense.state_dict().items()}
    save_file(sd, str(tmp_path / "dialect.safetensors"))
    other, _ = _tiny(True, 5)
    assert load_checkpoint(other, str(tmp_path / "dialect.safetensors")) == ([], [])
    assert torch.equal(other.lm_head.weight, dense.lm_head.weight)
    with pytest.raises(KeyError):
        load_checkpoint(moe, str(tmp_path / "dense"), strict=True)
    bad = {k: (v[:1].contiguous() if k == "lm_head.weight" else v.contiguous()) for k, v in dense.state_dict().items()}
    save_file(bad, str(tmp_path / "bad.safetensors"))
    with pytest.raises(ValueError):
        load_checkpoint(other, str(tmp_path / "bad.safetensors"))


# ---- the fixture WRITTEN BY THE REFERENCE (oracle/validate_vs_reference.py::write_ref_checkpoint) ------------------------
REF_CKPT = os.path.join(ROOT, "tests", "golden", "ref_ckpt")


def test_from_pretrained_reads_the_reference_written_checkpoint():
    """`from_pretrained` on a directory the imported reference `LlavaQwen2ForCausalLM` wrote (its state_dict() + its config,
    the CLIP directory `CLIPVisionModel.save_pretrained` wrote beside it): every decoder / projector tensor arrives under the
    reference's name, the tower comes from ITS directory (clip_encoder.py:24-33), nothing is missing or left over."""
    import json
    from safetensors.torch import load_file
    from llavamod.model import LlavaQwen2ForCausalLM
    man = json.load(open(os.path.join(REF_CKPT, "MANIFEST.json")))
    assert man["reference_class"].endswith("LlavaQwen2ForCausalLM")
    model = LlavaQwen2ForCausalLM.from_pretrained(REF_CKPT, attn_implementation="sdpa", torch_dtype=torch.bfloat16, device="cpu")
    tower = model.get_image_tower()
    assert tower.is_loaded and tower.weights_source.endswith(os.path.join("openai_clip_tiny", "model.safetensors"))
    assert (tower.hidden_size, tower.num_patches, tower.config.num_attention_heads) == (64, 4, 1)
    ref = load_file(os.path.join(REF_CKPT, "model.safetensors"))
    clip = load_file(os.path.join(REF_CKPT, "openai_clip_tiny", "model.safetensors"))
    own = model.state_dict()
    seen = set()
    for k, v in ref.items():
        if k.endswith("inv_freq") or k.endswith("position_ids"):
            continue
        kk = k if "image_tower" not in k else k.replace("image_tower.image_tower.", "image_tower.image_tower.vision_model.") \
            if "vision_model." not in k else k
        assert kk in own, k                                  # every reference name exists in this module tree
        assert torch.equal(own[kk].float(), v.float()), k     # bf16-representable values: exact
        seen.add(kk)
    assert seen == set(own), sorted(set(own) - seen)[:5]
    # the tower tensors equal the CLIP directory's (the reference's state dict holds the same values: it loaded them from there)
    for k, v in clip.items():
        if k.endswith("position_ids"):
            continue
        kk = "model.image_tower.image_tower." + (k if k.startswith("vision_model.") else "vision_model." + k)
        assert torch.equal(own[kk].float(), v.float()), k
    assert not any(p.requires_grad for p in tower.parameters())


def test_adapter_only_save_matches_the_reference_file(tmp_path):
    """`save_mm_adapter` writes what the reference's `safe_save_model_for_hf_trainer(tune_mm_mlp_adapter)` wrote into the
    fixture (same key set, same tensors), and `initialize_vision_modules(pretrain_mm_mlp_adapter=...)` reads it back."""
    from types import SimpleNamespace
    from llavamod.model import LlavaQwen2ForCausalLM
    model = LlavaQwen2ForCausalLM.from_pretrained(REF_CKPT, device="cpu")
    path = model.save_mm_adapter(str(tmp_path))
    mine, ref = torch.load(path), torch.load(os.path.join(REF_CKPT, "mm_projector.bin"))
    assert set(mine) == set(ref) == {"model.mm_projector.image_spatial_proj.0.weight", "model.mm_projector.image_spatial_proj.0.bias",
                                    "model.mm_projector.image_spatial_proj.2.weight", "model.mm_projector.image_spatial_proj.2.bias"}
    for k in ref:
        assert torch.equal(mine[k].float(), ref[k].float()), k
    with torch.no_grad():
        for p in model.get_model().mm_projector.parameters():
            p.zero_()
    margs = SimpleNamespace(image_tower=os.path.join(REF_CKPT, "openai_clip_tiny"), mm_vision_select_layer=-2,
                            mm_vision_select_feature="patch", image_projector_type="mlp2x_gelu",
                            pretrain_mm_mlp_adapter=os.path.join(REF_CKPT, "mm_projector.bin"))
    model.get_model().initialize_vision_modules(margs)
    sd = model.state_dict()
    for k in ref:
        assert torch.equal(sd[k].float(), ref[k].float()), k
    assert all(p.requires_grad for p in model.get_model().mm_projector.parameters())


def test_trainer_save_checkpoint_reference_signature_rank_gate_and_config(tmp_path):
    """`AlignTrainer._save_checkpoint(model, trial, metrics)` (train/align_trainer.py:616-636): the folder is
    `<output_dir>/checkpoint-<global_step>`, the adapter-only form writes `config.json` + `mm_projector.bin`, and only the rank
    with local_rank 0 / -1 writes."""
    from types import SimpleNamespace
    from llavamod.model import LlavaQwen2ForCausalLM
    from llavamod.train.align_trainer import AlignTrainer
    model = LlavaQwen2ForCausalLM.from_pretrained(REF_CKPT, device="cpu")
    tr = AlignTrainer.__new__(AlignTrainer)
    tr.state = SimpleNamespace(global_step=7)
    tr.args = SimpleNamespace(tune_mm_mlp_adapter=True, output_dir=str(tmp_path), local_rank=1)
    assert tr._save_checkpoint(model, None) is None and not os.path.exists(tmp_path / "checkpoint-7")
    tr.args.local_rank = 0
    path = tr._save_checkpoint(model, None, metrics={"loss": 1.0})
    assert path == str(tmp_path / "checkpoint-7" / "mm_projector.bin")
    assert sorted(os.listdir(tmp_path / "checkpoint-7")) == ["config.json", "mm_projector.bin"]
    import json
    cfg = json.load(open(tmp_path / "checkpoint-7" / "config.json"))
    assert cfg["architectures"] == ["LlavaQwen2ForCausalLM"] and cfg["hidden_size"] == model.config.hidden_size
    tr.args = SimpleNamespace(tune_mm_mlp_adapter=False, output_dir=str(tmp_path), local_rank=-1)
    tr._save_checkpoint(model, None, output_dir=str(tmp_path / "full"))
    again = LlavaQwen2ForCausalLM.from_pretrained(str(tmp_path / "full"), device="cpu")
    a, b = model.state_dict(), again.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)


def test_named_tower_never_random_initialises(tmp_path):
    """A tower given by NAME loads from a local directory or from the main checkpoint's own tensors — otherwise it raises
    (VERDICT r02 missing #2: `CLIPVisionModel.from_pretrained(self.image_tower_name)`, clip_encoder.py:24-33)."""
    import json
    import shutil
    from types import SimpleNamespace
    from llavamod.model import LlavaQwen2ForCausalLM
    from llavamod.model.multimodal_encoder.builder import build_image_tower
    args = SimpleNamespace(image_tower="openai/clip-vit-large-patch14-336", mm_vision_select_layer=-2)
    with pytest.raises(FileNotFoundError):
        build_image_tower(args, device="cpu")
    lazy = build_image_tower(args, device="cpu", delay_load=True)       # config only, like the reference's delay_load
    assert not lazy.is_loaded and lazy.num_patches == 576 and lazy.hidden_size == 1024
    with pytest.raises(RuntimeError):
        lazy(torch.zeros(1, 3, 336, 336))
    # (1) hub name in config.json, no tower tensors in the checkpoint: from_pretrained raises
    work = tmp_path / "hub_no_keys"
    shutil.copytree(REF_CKPT, work)
    shutil.rmtree(work / "openai_clip_tiny")
    cfg = json.load(open(work / "config.json"))
    cfg["mm_image_tower"] = "openai/clip-vit-base-patch32"
    json.dump(cfg, open(work / "config.json", "w"))
    from safetensors.torch import load_file, save_file
    sd = load_file(str(work / "model.safetensors"))
    save_file({k: v for k, v in sd.items() if "image_tower" not in k}, str(work / "model.safetensors"))
    with pytest.raises(FileNotFoundError):
        LlavaQwen2ForCausalLM.from_pretrained(str(work), device="cpu")
    # (2) unknown hub name, but the checkpoint carries `model.image_tower.*`: geometry from the shapes, weights from the file
    cfg["mm_image_tower"] = "someone/openai-clip-like"
    json.dump(cfg, open(work / "config.json", "w"))
    save_file(sd, str(work / "model.safetensors"))
    m = LlavaQwen2ForCausalLM.from_pretrained(str(work), device="cpu")
    t = m.get_image_tower()
    assert t.is_loaded and "model.image_tower.image_tower." in t.weights_source
    assert (t.hidden_size, t.config.num_hidden_layers, t.config.image_size, t.config.patch_size) == (64, 3, 28, 14)
    ref = LlavaQwen2ForCausalLM.from_pretrained(REF_CKPT, device="cpu")
    for (ka, va), (kb, vb) in zip(m.state_dict().items(), ref.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka


def _ep_save_worker(rank, world, port, out_dir, q):
    """2 ranks = one expert-parallel group of size 2 over a 4-expert layer: each rank holds 2 LOCAL experts."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LMOD_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from llavamod.checkpoint import load_checkpoint, read_state
    from llavamod.engine import init_distributed
    from llavamod.train.align_trainer import AlignTrainer
    init_distributed()
    model, dc = _tiny(True, 1)
    margs = U.moe_args(dc)
    margs.ep_size = 2
    model.initialize_moe_modules(margs)
    ex = model.get_model().layers[0].mlp.deepspeed_moe.experts.deepspeed_experts
    assert len(ex) == 2
    with torch.no_grad():                               # local expert i of rank r is GLOBAL expert 2 r + i: tag it
        for i, e in enumerate(ex):
            for p in e.parameters():
                p.fill_(float(2 * rank + i + 1))
    tr = AlignTrainer(model, None, args=type("A", (), dict(output_dir=out_dir, local_rank=rank))())
    files = tr._save_checkpoint(model, None, output_dir=os.path.join(out_dir, "ep"))     # collective: every rank calls it
    dist.barrier()
    assert bool(files) == (rank == 0)
    sd = read_state(os.path.join(out_dir, "ep"))
    for g in range(4):                                  # ONE file set holding all four experts under global indices
        t = sd[f"model.layers.0.mlp.deepspeed_moe.experts.deepspeed_experts.{g}.up_proj.weight"]
        assert bool((t.float() == g + 1).all()), g
    assert os.path.exists(os.path.join(out_dir, "ep", "config.json"))
    assert not [f for f in os.listdir(os.path.join(out_dir, "ep")) if ".tmp" in f]
    # the expert-parallel model reads its own local experts back from the global names
    with torch.no_grad():
        for e in ex:
            for p in e.parameters():
                p.zero_()
    assert load_checkpoint(model, os.path.join(out_dir, "ep")) == ([], [])
    for i, e in enumerate(ex):
        assert all(bool((p.float() == 2 * rank + i + 1).all()) for p in e.parameters())
    # and an ep_size = 1 model (what the FineTune / Eval classes build) loads the same files
    if rank == 0:
        full, dc1 = _tiny(True, 3)
        full.initialize_moe_modules(U.moe_args(dc1))
        assert load_checkpoint(full, os.path.join(out_dir, "ep")) == ([], [])
        ex1 = full.get_model().layers[0].mlp.deepspeed_moe.experts.deepspeed_experts
        assert [float(e.down_proj.weight.float().mean()) for e in ex1] == [1.0, 2.0, 3.0, 4.0]
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_expert_parallel_checkpoint_gathers_experts_under_global_names(tmp_path):
    """with ep_size > 1 a single writer used to save only ITS local experts (under local indices) and silently drop
    the others.  The save is now a collective over the writer's expert-parallel group; the files hold every expert under its
    global index, written through temporary names."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ep_save_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, "ok"), (1, "ok")]
