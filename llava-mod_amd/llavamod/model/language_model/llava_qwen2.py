"""LlavaQwen2ForCausalLM — the dense teacher API (reference language_model/llava_qwen2.py:31-130, which wraps
the vendored Qwen2ForCausalLM, qwen2/modeling_qwen2.py:1129-1190).  Same constructor-from-config,
`get_model()`, `forward(input_ids, attention_mask, …, labels, images, return_dict)` and output fields
(`loss`, `logits` fp32 [B,S',V], `labels` after the image splice); execution is on the HIP kernels.
`LlavaQwen1_5ForCausalLM` (llava_qwen1_5.py) is the same class: the two files differ in names only.
"""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from ... import kernels as K
from ... import ops
from ...constants import IGNORE_INDEX
from ...ops import FusedWeight
from ..llava_arch import LlavaMetaForCausalLM, LlavaMetaModel
from ..utils import CausalLMOutputWithPast
from .qwen2_hip import KVCache, Qwen2Config, Qwen2Model, _Linear, init_normal_

BF16 = torch.bfloat16


class LlavaQwen2Config(Qwen2Config):
    model_type = "llava_qwen2"


class LlavaQwen2Model(LlavaMetaModel, Qwen2Model):
    config_class = LlavaQwen2Config

    def __init__(self, config, device="cuda"):
        Qwen2Model.__init__(self, config, device)
        self._init_vision(config, device)


def build_loss_plan(labels_np, lens_np, kd_rows=True, ce_rows=True, distill_all_tokens=False, align_vocab=None,
                    device="cuda"):
    """Which rows of the flattened [B*S'] hidden states carry loss, in sample-major order.
      KD row t  : labels[b,t] != -100          (UNshifted mask, align_trainer.py:522; or every row incl. pads
                                               when distill_all_tokens, :516-520)
      CE row t  : labels[b,t+1] != -100        (shifted LM / DPO rows, llava_qwen2_moe.py:413-421, dpo_trainer.py:483-485)
    """
    if torch.is_tensor(labels_np):
        return build_loss_plan_device(labels_np, kd_rows, ce_rows, distill_all_tokens, align_vocab)
    B, S = labels_np.shape
    valid = labels_np != IGNORE_INDEX
    kd = (np.ones_like(valid) if distill_all_tokens else valid) if kd_rows else np.zeros_like(valid)
    ce = np.zeros_like(valid)
    if ce_rows:
        ce[:, :-1] = valid[:, 1:]
    ce_label = np.full((B, S), -1, dtype=np.int32)
    ce_label[:, :-1] = np.where(valid[:, 1:], labels_np[:, 1:], -1)
    need = kd | ce
    bi, ti = np.nonzero(need)                                   # row-major => sample-major order
    rows = (bi * S + ti).astype(np.int32)
    seg_off = np.zeros(B + 1, dtype=np.int32)
    np.cumsum(np.bincount(bi, minlength=B), out=seg_off[1:])
    inv = np.full(B * S, -1, dtype=np.int32)
    inv[rows] = np.arange(len(rows), dtype=np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return SimpleNamespace(R=len(rows), row_idx=t(rows), inv_row_idx=t(inv), kd_w=t(kd[bi, ti].astype(np.float32)),
                           ce_w=t(ce[bi, ti].astype(np.float32)), ce_label=t(ce_label[bi, ti]),
                           seg_off=t(seg_off), seg_id=t(bi.astype(np.int32)), align_vocab=align_vocab, shape=(B, S),
                           labels_np=labels_np,
                           n_kd=int(kd.sum()), n_ce=int(ce.sum()))


def build_loss_plan_device(labels, kd_rows=True, ce_rows=True, distill_all_tokens=False, align_vocab=None):
    """`build_loss_plan` on the device (csrc/splice.hip) from the spliced labels [B, S'] (device int64): two launches and
    one B-int read-back (R, the number of loss rows, sizes the outputs)."""
    from ..._hip import call, ptr
    dev = labels.device
    B, S = labels.shape
    lb = labels.to(torch.int64).contiguous()
    counts = torch.empty(B, device=dev, dtype=torch.int32)
    flags = (int(bool(kd_rows)), int(bool(ce_rows)), int(bool(distill_all_tokens)))
    call("lmod_lossplan_count", ptr(lb), B, S, *flags, ptr(counts))
    R = int(counts.sum().item())
    i32 = dict(device=dev, dtype=torch.int32)
    row_idx, inv = torch.empty(R, **i32), torch.empty(B * S, **i32)
    kd_w, ce_w = torch.empty(R, device=dev), torch.empty(R, device=dev)
    ce_label, seg_off, seg_id = torch.empty(R, **i32), torch.empty(B + 1, **i32), torch.empty(R, **i32)
    call("lmod_lossplan_fill", ptr(lb), B, S, *flags, ptr(counts), ptr(row_idx), ptr(inv), ptr(kd_w), ptr(ce_w), ptr(ce_label),
         ptr(seg_off), ptr(seg_id))
    return SimpleNamespace(R=R, row_idx=row_idx, inv_row_idx=inv, kd_w=kd_w, ce_w=ce_w, ce_label=ce_label, seg_off=seg_off,
                           seg_id=seg_id, align_vocab=align_vocab, shape=(B, S), labels_np=None, labels_dev=lb,
                           n_kd=None, n_ce=None)


def _inverse_map(packed):
    """packed row -> padded position (int32 [T]), cached on the plan."""
    if getattr(packed, "from_packed", None) is None:
        keep = np.nonzero(packed.to_packed_np >= 0)[0].astype(np.int32)
        packed.from_packed = torch.from_numpy(keep).to(packed.to_packed.device)
    return packed.from_packed


class _CausalLMBase(nn.Module, LlavaMetaForCausalLM):
    """Shared forward machinery of the dense teacher and the MoE student."""

    def _setup_head(self, config, device):
        self.vocab_size = config.vocab_size
        self.lm_head = _Linear(config.hidden_size, config.vocab_size, False, device)
        self._head = FusedWeight([[self.lm_head.weight]])

    def get_model(self):
        return self.model

    def head(self):
        return self._head.ensure()

    # ---- internal fast path -------------------------------------------------------------------
    def forward_hidden(self, input_ids=None, attention_mask=None, labels=None, images=None, inputs_embeds=None,
                       plan_fn=None):
        """Splice + decoder.  Returns (hidden [B*S', H], moe_loss_list, info) — no logits.
        plan_fn(info) -> loss plan: the decoder then returns just the plan's rows ([R, H], `info.plan.pregathered`),
        letting a dense last layer skip (forward and backward) the rows nobody reads."""
        if inputs_embeds is None:
            _, _, attention_mask, _, inputs_embeds, labels = self.prepare_inputs_labels_for_multimodal(
                input_ids, None, attention_mask, None, labels, images)
            plan = self._plan
            if inputs_embeds is None:                          # text-only batch
                dev = self.model.embed_tokens.weight.device
                B, S = input_ids.shape
                idx = input_ids.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
                inputs_embeds = ops.SpliceEmbed.apply(None, self.model.embed_tokens.weight, idx, None).view(B, S, -1)
        else:
            plan = None
        B, S, H = inputs_embeds.shape
        seqlens = None
        packed = plan is not None and getattr(plan, "cu", None) is not None      # unpadded (varlen) execution
        if packed:
            B, S = plan.B, plan.S                              # embeds arrive as packed rows [1, T, H]
        if plan is not None:
            seqlens = plan.seqlens
        elif attention_mask is not None:
            lens = attention_mask.to(torch.int32).sum(1)
            if bool((lens != S).any()):
                seqlens = lens.to(device=inputs_embeds.device, dtype=torch.int32).contiguous()
        if labels is not None:
            if plan is not None:
                labels_np = plan.labels_np if plan.labels_np is not None else labels      # device-built plan: stay on device
            else:
                labels_np = labels if labels.is_cuda else labels.detach().cpu().numpy()
        else:
            labels_np = None
        lens_np = plan.lens_np if plan is not None else None
        info = SimpleNamespace(B=B, S=S, labels=labels, labels_np=labels_np, lens_np=lens_np,
                               attention_mask=attention_mask, plan=None)
        out_rows = inv_rows = None
        if plan_fn is not None and labels_np is not None:
            info.plan = plan_fn(info)
            if packed:
                from ..llava_arch import pack_loss_plan
                info.plan = pack_loss_plan(info.plan, plan)
            elif getattr(info.plan, "is_packed", False):
                # e.g. a teacher that ran unpadded handing its plan to a student that runs padded: the row indices would
                # address packed rows in a padded buffer
                raise ValueError("loss plan is in packed (unpadded) coordinates but this model executes padded: set `unpad` "
                                 "identically on the teacher and the student")
            info.plan.pregathered = True
            out_rows, inv_rows = info.plan.row_idx, info.plan.inv_row_idx
        if packed:
            info.packed = plan
            hidden, moe_list = self.model(inputs_embeds.reshape(-1, H), B, S, None, out_rows=out_rows, inv_rows=inv_rows,
                                          cu=plan.cu, pos=plan.pos)
        else:
            hidden, moe_list = self.model(inputs_embeds.reshape(B * S, H), B, S, seqlens, out_rows=out_rows,
                                          inv_rows=inv_rows)
        return hidden, moe_list, info

    # ---- generation (KV cache) -------------------------------------------------------------------------------------
    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, inputs_embeds=None, attention_mask=None, **kwargs):
        """HF contract of the reference (llava_qwen2_moe.py:453-473): after the first step only the last token is fed and the
        images are dropped; `images` rides along on the first step."""
        images = kwargs.pop("images", None)
        if past_key_values is not None:
            input_ids = input_ids[:, -1:]
        model_inputs = {"inputs_embeds": inputs_embeds} if (inputs_embeds is not None and past_key_values is None) \
            else {"input_ids": input_ids}
        model_inputs.update({"past_key_values": past_key_values, "use_cache": kwargs.get("use_cache", True),
                             "attention_mask": attention_mask})
        if images is not None and past_key_values is None:
            model_inputs["images"] = images
        return model_inputs

    @torch.no_grad()
    def _prefill(self, input_ids, attention_mask, images, max_new_tokens):
        """Splice + full forward storing K/V; returns (cache, bf16 logits [B, V] of each sample's LAST valid position)."""
        unpad, self.unpad = getattr(self, "unpad", False), False       # the KV cache is laid out per padded sample
        try:
            _, _, am, _, embeds, _ = self.prepare_inputs_labels_for_multimodal(input_ids, None, attention_mask, None, None, images)
        finally:
            self.unpad = unpad
        plan = self._plan
        dev = self.model.embed_tokens.weight.device
        if embeds is None:
            B, S = input_ids.shape
            idx = input_ids.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
            embeds = K.gather_rows(self.model.embed_tokens.weight, None, idx, self.model.embed_tokens.weight.shape[1]).view(B, S, -1)
            am = attention_mask
        B, S, H = embeds.shape
        if embeds is not None and plan is not None:
            # spliced batch: the plan knows every sample's real length (also when no attention_mask was given but the
            # samples splice to different lengths)
            lens = torch.from_numpy(plan.lens_np.astype("int32")).to(dev)
        elif am is not None:
            amd = am.to(device=dev).to(torch.int32)
            lens = amd.sum(1).to(torch.int32)
            # the cache and the last-token row are addressed as b*S + len-1: RIGHT padding only (the reference's
            # `padding_side="right"`, align_train.py:366).  A left-padded mask would silently read the wrong rows.
            if bool((amd[:, 1:] > amd[:, :-1]).any()):
                raise ValueError("generate/use_cache expects RIGHT-padded prompts (attention_mask must be non-increasing along "
                                 "the sequence); left padding is not supported on this path")
        else:
            lens = torch.full((B,), S, device=dev, dtype=torch.int32)
        seqlens = lens.contiguous() if bool((lens != S).any()) else None
        cfg = self.config
        cache = KVCache(cfg.num_hidden_layers, B, S + max_new_tokens, cfg.num_key_value_heads * cfg.head_dim, dev)
        rows = (torch.arange(B, device=dev, dtype=torch.int32) * S + lens - 1).to(torch.int32).contiguous()
        inv = torch.full((B * S,), -1, device=dev, dtype=torch.int32)
        inv[rows.long()] = torch.arange(B, device=dev, dtype=torch.int32)
        hidden, _ = self.model(embeds.reshape(B * S, H), B, S, seqlens, out_rows=rows, inv_rows=inv, cache=cache)
        cache.lens.copy_(lens)
        return cache, ops.linear_fwd(hidden, self.head())

    generation_budget = 512            # cache positions reserved beyond the prompt by a `use_cache=True` forward

    @torch.no_grad()
    def _cached_forward(self, input_ids, attention_mask, images, past, out_cls):
        """`forward(..., use_cache=True / past_key_values=...)` as HF's generation loop drives it: the first call prefills
        and returns the cache, later calls feed the last token.  logits: [B, 1, V] fp32 (the last position only)."""
        if past is None:
            cache, logits = self._prefill(input_ids, attention_mask, images, self.generation_budget)
        else:
            cache = past
            if int(cache.lens.max()) >= cache.smax:
                raise ValueError("KV cache exhausted: raise `generation_budget` before the prefill")
            dev = cache.lens.device
            logits = self._decode_step(input_ids[:, -1].to(device=dev, dtype=torch.int32), cache)
        return out_cls(loss=None, logits=logits.float().unsqueeze(1), past_key_values=cache)

    @torch.no_grad()
    def _decode_step(self, tokens, cache):
        """tokens [B] (device int32) -> bf16 logits [B, V] of the next position."""
        emb = K.gather_rows(self.model.embed_tokens.weight, None, tokens.contiguous(), self.model.embed_tokens.weight.shape[1])
        return ops.linear_fwd(self.model.forward_decode(emb, cache), self.head())

    @staticmethod
    def _sample(logits, temperature, top_k, top_p, gen):
        """Next token per row from bf16 logits [B, V] — HF's sampling warpers (temperature, top-k, nucleus) in that order, then
        one multinomial draw.  O(B x V) host-glue torch math on B rows (B = number of sequences being generated)."""
        x = logits.float() / max(float(temperature), 1e-6)
        if top_k is not None and 0 < top_k < x.shape[-1]:
            kth = torch.topk(x, top_k, dim=-1).values[:, -1:]
            x = x.masked_fill(x < kth, float("-inf"))
        if top_p is not None and top_p < 1.0:
            sv, si = torch.sort(x, dim=-1, descending=True)
            cp = torch.softmax(sv, dim=-1).cumsum(-1)
            drop = cp - torch.softmax(sv, dim=-1) > top_p          # keep the smallest prefix whose mass reaches top_p
            sv = sv.masked_fill(drop, float("-inf"))
            x = torch.full_like(x, float("-inf")).scatter(-1, si, sv)
        return torch.multinomial(torch.softmax(x, dim=-1), 1, generator=gen).squeeze(-1).to(torch.int32)

    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, images=None, max_new_tokens=16, eos_token_id=None,
                 pad_token_id=None, do_sample=False, temperature=1.0, top_k=None, top_p=None, seed=None, **kwargs):
        """Generation with a KV cache (the reference's eval path calls HF `generate` on the Eval model,
        llava_qwen2_moe.py:629-681; its serving code passes do_sample / temperature / top_p).  Greedy by default; do_sample=True
        draws from the temperature / top-k / top-p filtered distribution (`seed` makes the draw reproducible).  Right-padded
        prompts: every sample continues from its own length.  Returns the NEW tokens [B, <= max_new_tokens] (int64; positions
        after a sample's EOS hold pad_token_id)."""
        was_training = self.training
        self.eval()
        cache, logits = self._prefill(input_ids, attention_mask, images, max_new_tokens)
        B = logits.shape[0]
        pad = eos_token_id if pad_token_id is None else pad_token_id
        done = torch.zeros(B, dtype=torch.bool, device=logits.device)
        gen = None
        if do_sample:
            gen = torch.Generator(device=logits.device)
            # no `seed`: DRAW one from the global generator (deterministic under torch.manual_seed) — never reseed it: torch.seed()
            # would change torch.initial_seed(), from which the MoE layers derive their gating-noise stream, so a sampling
            # generate() between two training steps would silently fork the training run
            gen.manual_seed(int(seed) if seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        out = []
        for step in range(max_new_tokens):
            nxt = self._sample(logits, temperature, top_k, top_p, gen) if do_sample else K.row_argmax(logits)
            tok = nxt.long()
            if eos_token_id is not None:
                tok = torch.where(done, torch.full_like(tok, pad if pad is not None else 0), tok)
                done = done | (tok == eos_token_id)
            out.append(tok)
            if step + 1 == max_new_tokens or (eos_token_id is not None and bool(done.all())):
                break
            logits = self._decode_step(nxt, cache)
        self.train(was_training)
        return torch.stack(out, 1)

    # ---- checkpoints in the reference's layout -----------------------------------------------------------------------
    def save_config(self, save_directory):
        """`config.json` alone (the reference: `model.config.save_pretrained(output_dir)`, align_trainer.py:631)."""
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        cfg = {}
        for k, v in vars(self.config).items():
            if k == "mm_image_tower" and not isinstance(v, (str, type(None))):
                cfg[k] = {"__clip_vision_config__": {a: b for a, b in vars(v).items()}}
            elif isinstance(v, (str, int, float, bool, type(None), list, dict)):
                cfg[k] = v
        cfg["model_type"] = getattr(self.config, "model_type", None)
        cfg["architectures"] = [type(self).__name__]
        tmp = os.path.join(save_directory, f"config.json.tmp{os.getpid()}")
        with open(tmp, "w") as f:                              # temp file + rename: a crash never leaves half a config
            json.dump(cfg, f, indent=1)
        os.replace(tmp, os.path.join(save_directory, "config.json"))

    def save_pretrained(self, save_directory, max_shard_bytes=5 << 30, writer_rank=0):
        """config.json + HF-layout safetensors shards (+ mm_projector.bin), loadable by `from_pretrained`.  One rank writes
        (`writer_rank`, a global rank; without torch.distributed the caller is rank 0); an expert-parallel model is saved
        collectively with its experts gathered under global indices (`checkpoint.save_checkpoint`)."""
        from ...checkpoint import _dist_rank_world, save_checkpoint
        files = save_checkpoint(self, save_directory, max_shard_bytes=max_shard_bytes, writer_rank=writer_rank)
        rank = _dist_rank_world()[0]
        if rank == writer_rank:
            self.save_config(save_directory)
        else:                                   # not an error (the collective save calls this on every rank), but never silent
            import logging
            logging.getLogger("llavamod").info("save_pretrained(%s): rank %d is not the writer (rank %d) and wrote nothing",
                                               save_directory, rank, writer_rank)
        return files

    def save_mm_adapter(self, output_dir, keys_to_match=("mm_projector",)):
        """Adapter-only save of the reference's trainers (train/align_trainer.py:616-636 `_save_checkpoint` with
        tune_mm_mlp_adapter; train/align_train.py:623-631 `safe_save_model_for_hf_trainer`): ONLY the parameters whose name
        contains one of `keys_to_match`, under their full names, torch-pickled as `mm_projector.bin` — the file
        `--pretrain_mm_mlp_adapter` reads back (llava_arch.py:122-128)."""
        import os
        os.makedirs(output_dir, exist_ok=True)
        sd = {k: v.detach().cpu().clone() for k, v in self.state_dict().items() if any(m in k for m in keys_to_match)}
        if not sd:
            raise ValueError(f"no parameter matches {keys_to_match}")
        path = os.path.join(output_dir, "mm_projector.bin")
        tmp = f"{path}.tmp{os.getpid()}"                      # per-process temp name: two writers never share a half-written file
        torch.save(sd, tmp)
        os.replace(tmp, path)
        return path

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, attn_implementation=None, torch_dtype=None,
                        device="cuda", strict=True, config=None, cache_dir=None, **kwargs):
        """`X.from_pretrained(path, attn_implementation=...)` of the reference's entry scripts (train/align_train.py:133-139):
        read config.json, build the architecture, load the weights by name.  attn_implementation is recorded only (there is
        one attention implementation here: the HIP flash kernels); torch_dtype must be bf16 (the compute dtype).

        Image tower: a tower NAMED in the config (`mm_image_tower: "<dir or hub name>"`) takes its weights from that
        directory (also looked up relative to the checkpoint), as `CLIPVisionModel.from_pretrained` does in the reference
        (clip_encoder.py:24-33) — `model.image_tower.*` tensors of the main checkpoint are then ignored, as the reference
        ignores them (its tower is built with delay_load, llava_arch.py:31).  If the name is not a local directory (a hub
        name; no network here) the main checkpoint's own `model.image_tower.*` tensors are used; if it has none, this raises."""
        import json
        import os
        from ...checkpoint import load_checkpoint, read_state
        from ..multimodal_encoder.clip_encoder import KNOWN_GEOMETRY, CLIPVisionConfig
        if torch_dtype not in (None, torch.bfloat16):
            raise ValueError("this path computes in bf16: torch_dtype must be torch.bfloat16 (or None)")
        path = pretrained_model_name_or_path
        if config is None:
            raw = json.load(open(os.path.join(path, "config.json")))
            raw.update(kwargs)
            moe = raw.pop("moe", None)
            for drop in ("architectures", "model_type", "transformers_version", "torch_dtype", "dtype", "lora"):
                raw.pop(drop, None)
            tower = raw.get("mm_image_tower")
            if isinstance(tower, dict) and "__clip_vision_config__" in tower:
                raw["mm_image_tower"] = CLIPVisionConfig(**tower["__clip_vision_config__"])
            config = cls.config_class(**raw)
            if moe is not None:
                config.moe = moe
        config._attn_implementation = attn_implementation
        config._name_or_path = path
        model = cls(config, device=device)
        tower, src, ignore = model.get_image_tower(), None, ()
        if tower is not None and not tower.is_loaded:
            try:
                tower.load_model(search=(path,))
                ignore = ("model.image_tower.",)
            except FileNotFoundError as e:
                src = read_state(path)
                pre = "model.image_tower.image_tower."
                tsd = {k[len(pre):]: v for k, v in src.items() if k.startswith(pre)}
                if not tsd:
                    raise FileNotFoundError(f"{e}; and {path} holds no `{pre}*` tensors either") from None
                heads = KNOWN_GEOMETRY.get(tower.image_tower_name, {}).get("num_attention_heads")
                tower.load_from_state(tsd, source=f"{path} ({pre}*)", heads=heads)
        load_checkpoint(model, path, strict=strict, ignore_prefixes=ignore, state=src)
        return model

    def lm_loss_from_hidden(self, hidden, info):
        """Shifted CrossEntropyLoss() of the reference forward (mean over non-ignored), loss rows only."""
        plan = build_loss_plan(info.labels_np, info.lens_np, kd_rows=False, ce_rows=True, device=hidden.device)
        if getattr(info, "packed", None) is not None:
            from ..llava_arch import pack_loss_plan
            plan = pack_loss_plan(plan, info.packed)
        _, _, ce_sum, ce_cnt = ops.DistillHead.apply(hidden, self.head(), plan, None, *self._head_trainable())
        return ce_sum.sum() / ce_cnt.sum()

    def _head_trainable(self):
        return [self.lm_head.weight] if self.lm_head.weight.requires_grad else []

    def full_logits(self, hidden, B, S, packed=None):
        if packed is not None:                 # unpadded execution: back to the reference's padded [B, S', V] (zero rows on pads)
            hidden = ops.RowGather.apply(hidden, packed.to_packed, _inverse_map(packed))
        lg = ops.Linear.apply(hidden, self.head(), *self._head_trainable())
        return lg.view(B, S, -1).float()                        # `logits = logits.float()` (modeling_qwen2.py:1164)


class LlavaQwen2ForCausalLM(_CausalLMBase):
    config_class = LlavaQwen2Config

    def __init__(self, config, device="cuda"):
        super().__init__()
        self.config = config
        self.model = LlavaQwen2Model(config, device)
        self._setup_head(config, device)
        self._plan = None
        init_normal_(self, getattr(config, "initializer_range", 0.02), getattr(config, "init_seed", 0))   # post_init()

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                images=None, return_dict=None):
        if use_cache or past_key_values is not None:
            return self._cached_forward(input_ids, attention_mask, images, past_key_values, CausalLMOutputWithPast)
        hidden, _, info = self.forward_hidden(input_ids, attention_mask, labels, images, inputs_embeds)
        logits = self.full_logits(hidden, info.B, info.S, getattr(info, "packed", None))
        loss = self.lm_loss_from_hidden(hidden, info) if info.labels is not None else None
        return CausalLMOutputWithPast(loss=loss, logits=logits, labels=info.labels)


LlavaQwen1_5Config = LlavaQwen2Config
LlavaQwen1_5ForCausalLM = LlavaQwen2ForCausalLM
