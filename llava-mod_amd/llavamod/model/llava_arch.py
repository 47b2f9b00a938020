"""Multimodal glue: LlavaMetaModel (owns image_tower + mm_projector) and LlavaMetaForCausalLM
(encode_images, prepare_inputs_labels_for_multimodal) — reference llavamod/model/llava_arch.py:27-128,
143-148,155-334.

The reference splices image features into the token embeddings with per-sample python loops of
cat/split/pad on device tensors (and a `.tolist()` sync, :247).  Here the same layout is expressed
once, on the host, as an int32 index map over the flattened [B*S'] output rows
    idx >= 0  -> row of the embedding table (text token id)
    idx <= -2 -> row -(idx+2) of the projector output (image patch)
    idx == -1 -> zero row (right padding)
and executed by one gather kernel (lmod_gather_rows); its inverse map routes gradients back to the
projector.  Labels / attention mask / lengths come out exactly as the reference builds them.
"""
from types import SimpleNamespace

import numpy as np
import torch

from .. import ops
from ..constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX
from .multimodal_encoder.builder import build_image_tower
from .multimodal_projector.builder import build_projector


class LlavaMetaModel:
    """Mixin for the decoder `model` object: owns `image_tower` and `mm_projector` (llava_arch.py:27-128)."""

    def _init_vision(self, config, device):
        tower = getattr(config, "mm_image_tower", None)
        if tower is not None:
            # llava_arch.py:31: a tower NAMED in the config is built with delay_load=True (config only); its weights arrive
            # with initialize_vision_modules().load_model() or from_pretrained.  A CLIPVisionConfig OBJECT (synthetic
            # benchmarks / tests) builds the random-init architecture at once.
            self.image_tower = build_image_tower(config, delay_load=isinstance(tower, str), device=device,
                                                 search=(getattr(config, "_name_or_path", None),))
            self.mm_projector = build_projector(config, device=device)

    def get_image_tower(self):
        t = getattr(self, "image_tower", None)
        return t[0] if type(t) is list else t

    def get_video_tower(self):
        return None

    def initialize_vision_modules(self, model_args, fsdp=None):
        """llava_arch.py:45-128 (image path): build the tower + projector if absent, re-enable projector grads."""
        image_tower = model_args.image_tower
        assert image_tower is not None
        dev = self.embed_tokens.weight.device
        self.config.mm_image_tower = image_tower
        if self.get_image_tower() is None:
            self.image_tower = build_image_tower(model_args, device=dev)
        else:
            self.get_image_tower().load_model()
        self.config.use_mm_proj = True
        self.config.image_projector_type = getattr(model_args, "image_projector_type", None)
        self.config.mm_hidden_size = self.get_image_tower().hidden_size
        self.config.mm_vision_select_layer = model_args.mm_vision_select_layer
        self.config.mm_vision_select_feature = getattr(model_args, "mm_vision_select_feature", "patch")
        if getattr(self, "mm_projector", None) is None:
            self.mm_projector = build_projector(self.config, device=dev)
        else:
            for p in self.mm_projector.parameters():      # :115-120 "in case it is frozen"
                p.requires_grad = True
        ckpt = getattr(model_args, "pretrain_mm_mlp_adapter", None)
        if ckpt is not None:
            w = torch.load(ckpt, map_location="cpu")
            self.mm_projector.load_state_dict({k.split("mm_projector.")[1]: v for k, v in w.items() if "mm_projector" in k})


def build_splice_plan(input_ids, attention_mask, labels, n_patches, max_length=None, device="cuda"):
    """Host-side restatement of llava_arch.py:213-334 as index maps.  All inputs may be CPU or device
    tensors ([B, T] ids / mask / labels).  Returns a SimpleNamespace of device tensors + python ints."""
    ids = input_ids.detach().cpu().numpy()
    B, T = ids.shape
    am = np.ones((B, T), dtype=bool) if attention_mask is None else attention_mask.detach().cpu().numpy().astype(bool)
    lb = np.full((B, T), IGNORE_INDEX, dtype=np.int64) if labels is None else labels.detach().cpu().numpy().astype(np.int64)
    rows_idx, rows_lab = [], []
    img_pos = []            # per image: (sample, start offset in the sample's new sequence) or None if unused
    cur = 0
    for b in range(B):
        cid, clb = ids[b][am[b]], lb[b][am[b]]
        where = np.nonzero(cid == IMAGE_TOKEN_INDEX)[0]
        if len(where) == 0:                                   # :238-246 (an image slot is still consumed)
            rows_idx.append(cid.astype(np.int64)); rows_lab.append(clb)
            img_pos.append(None); cur += 1
            continue
        parts_i, parts_l, off, prev = [], [], 0, 0
        for w in where:
            seg = cid[prev:w]
            parts_i.append(seg.astype(np.int64)); parts_l.append(clb[prev:w]); off += len(seg)
            parts_i.append(-(cur * n_patches + np.arange(n_patches, dtype=np.int64) + 2))
            parts_l.append(np.full(n_patches, IGNORE_INDEX, dtype=np.int64))
            img_pos.append((b, off)); off += n_patches; cur += 1
            prev = w + 1
        parts_i.append(cid[prev:].astype(np.int64)); parts_l.append(clb[prev:])
        rows_idx.append(np.concatenate(parts_i)); rows_lab.append(np.concatenate(parts_l))
    if max_length is not None:                                # :280-283
        rows_idx = [r[:max_length] for r in rows_idx]; rows_lab = [r[:max_length] for r in rows_lab]
    lens = np.array([len(r) for r in rows_idx], dtype=np.int32)
    S = int(lens.max())
    idx = np.full((B, S), -1, dtype=np.int32)
    new_lab = np.full((B, S), IGNORE_INDEX, dtype=np.int64)
    new_am = np.zeros((B, S), dtype=bool)
    for b in range(B):
        n = lens[b]
        idx[b, :n] = rows_idx[b]; new_lab[b, :n] = rows_lab[b]; new_am[b, :n] = True
    inv = np.full(cur * n_patches, -1, dtype=np.int32)
    for k, pos in enumerate(img_pos):
        if pos is None:
            continue
        b, off = pos
        n = max(0, min(n_patches, int(lens[b]) - off))
        inv[k * n_patches:k * n_patches + n] = b * S + off + np.arange(n, dtype=np.int32)
    ragged = bool((lens != S).any())
    return SimpleNamespace(
        B=B, S=S, n_images=cur,
        idx=torch.from_numpy(idx.reshape(-1)).to(device), inv_idx=torch.from_numpy(inv).to(device),
        labels=torch.from_numpy(new_lab).to(device) if labels is not None else None,
        labels_np=new_lab, lens_np=lens,
        attention_mask=torch.from_numpy(new_am).to(device) if attention_mask is not None else None,
        seqlens=torch.from_numpy(lens).to(device) if ragged else None)


def pack_splice_plan(plan):
    """Padded plan -> UNPADDED (varlen) plan: the samples' valid rows packed back to back (cu_seqlens), no padding rows in
    any GEMM / row kernel / MoE gate.  Adds: cu (device int32 [B+1]), T (packed rows), pos (device int32 [T], position id of
    every packed row), to_packed (device int32 [B*S], packed row of a padded position or -1); idx / inv_idx are rewritten
    in packed coordinates.  Host-built plans are packed with numpy, device-built ones with a few int-tensor ops."""
    B, S = plan.B, plan.S
    lens = plan.lens_np.astype(np.int64)
    cu_np = np.zeros(B + 1, dtype=np.int32)
    np.cumsum(lens, out=cu_np[1:])
    T = int(cu_np[-1])
    dev = plan.idx.device
    keep_np = np.concatenate([b * S + np.arange(lens[b], dtype=np.int64) for b in range(B)]) if B else np.zeros(0, np.int64)
    to_packed_np = np.full(B * S, -1, dtype=np.int32)
    to_packed_np[keep_np] = np.arange(T, dtype=np.int32)
    pos_np = (keep_np % S).astype(np.int32)
    keep = torch.from_numpy(keep_np).to(dev)
    to_packed = torch.from_numpy(to_packed_np).to(dev)
    out = SimpleNamespace(**vars(plan))
    out.idx = plan.idx[keep].contiguous()
    inv = plan.inv_idx.long()
    out.inv_idx = torch.where(inv >= 0, to_packed[inv.clamp_min(0)], torch.full_like(plan.inv_idx, -1)).contiguous()
    out.cu = torch.from_numpy(cu_np).to(dev)
    out.T, out.pos, out.to_packed, out.to_packed_np = T, torch.from_numpy(pos_np).to(dev), to_packed, to_packed_np
    out.seqlens = None                                         # no key mask: a packed sample has no padding keys
    return out


def pack_loss_plan(lp, splice):
    """Loss-row plan in padded coordinates -> packed coordinates (loss rows are never padding rows unless every token is
    distilled, which the unpadded mode does not support: pads do not exist there)."""
    if getattr(lp, "is_packed", False):                       # e.g. the teacher's plan reused for the student
        return lp
    to_packed = splice.to_packed
    rows = to_packed[lp.row_idx.long()]
    if rows.numel() and bool((rows < 0).any()):                # one read-back, on the opt-in unpadded path only
        raise ValueError("a loss row is a padding position (distill_all_tokens distils pads too): the unpadded execution "
                         "(`model.unpad`) has no such rows — run this batch padded")
    out = SimpleNamespace(**vars(lp))
    out.is_packed = True
    out.row_idx = rows.contiguous()
    inv = torch.full((splice.T,), -1, device=rows.device, dtype=torch.int32)
    inv[rows.long()] = torch.arange(rows.numel(), device=rows.device, dtype=torch.int32)
    out.inv_row_idx = inv
    return out


def build_splice_plan_device(input_ids, attention_mask, labels, n_patches, max_length=None):
    """The same plan as `build_splice_plan`, built ON THE DEVICE from device-resident [B, T] int64 ids / bool mask / int64
    labels (csrc/splice.hip): two launches and one B-int read-back (the batch's spliced length S' sizes the outputs, exactly
    as the reference's pad-to-max does).  labels_np is None: the loss plan is then built on the device too."""
    from .._hip import call, ptr
    dev = input_ids.device
    B, T = input_ids.shape
    ids = input_ids.to(torch.int64).contiguous()
    am = attention_mask.to(torch.bool).contiguous().view(torch.uint8) if attention_mask is not None else None
    lb = labels.to(torch.int64).contiguous() if labels is not None else None
    cnt = torch.empty((2, B), device=dev, dtype=torch.int32)
    call("lmod_splice_count", ptr(ids), ptr(am), B, T, n_patches, int(max_length or 0), ptr(cnt[0]), ptr(cnt[1]))
    host = cnt.cpu().numpy()                                   # the one host sync of the splice: B lengths + B image counts
    lens_np, S, n_img = host[0].astype(np.int32), int(host[0].max()), int(host[1].sum())
    idx = torch.empty(B * S, device=dev, dtype=torch.int32)
    new_lab = torch.empty((B, S), device=dev, dtype=torch.int64)
    new_am = torch.empty((B, S), device=dev, dtype=torch.uint8) if attention_mask is not None else None
    inv = torch.full((n_img * n_patches,), -1, device=dev, dtype=torch.int32)
    call("lmod_splice_fill", ptr(ids), ptr(am), ptr(lb), B, T, n_patches, S, ptr(cnt[0]), ptr(cnt[1]), ptr(idx), ptr(new_lab),
         ptr(new_am), ptr(inv))
    ragged = bool((lens_np != S).any())
    return SimpleNamespace(B=B, S=S, n_images=n_img, idx=idx, inv_idx=inv,
                           labels=new_lab if labels is not None else None, labels_np=None, lens_np=lens_np,
                           attention_mask=new_am.view(torch.bool) if new_am is not None else None,
                           seqlens=cnt[0].clone() if ragged else None)


class LlavaMetaForCausalLM:
    """Mixin for the *ForCausalLM classes (llava_arch.py:131-334)."""

    def get_image_tower(self):
        return self.get_model().get_image_tower()

    def get_video_tower(self):
        return None

    def encode_images(self, images):                          # llava_arch.py:143-148
        # `_shared_tower_feats` (set by a trainer for ONE call): tower output computed by another model whose frozen
        # tower is bit-identical (student and teacher load the same CLIP checkpoint) — the tower is not run again.
        feats = getattr(self, "_shared_tower_feats", None)
        self._shared_tower_feats = None
        if feats is None:
            feats = self.get_model().get_image_tower()(images)    # [B, 1+P, Dv], CLS skipped inside the projector
        self._last_tower_feats = feats
        return self.get_model().mm_projector.forward_image(feats)

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images):
        """Same 6-tuple as the reference (input_ids=None, position_ids=None, attention_mask, pkv,
        inputs_embeds [B,S',H], labels [B,S']); additionally stashes the splice plan on `self._plan`."""
        tower = self.get_image_tower()
        model = self.get_model()
        dev = model.embed_tokens.weight.device
        if tower is None or images is None or input_ids.shape[1] == 1:
            self._plan = None
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        imgs = torch.stack([im for im in images]) if isinstance(images, (list, tuple)) else images
        if imgs.dim() != 4:
            raise NotImplementedError("video inputs are not on the distillation path")
        P = tower.num_patches
        max_len = getattr(self.config, "tokenizer_model_max_length", None)
        if input_ids.is_cuda:                                 # device-resident batch: index build on the device
            plan = build_splice_plan_device(input_ids, attention_mask, labels, P, max_len)
        else:
            plan = build_splice_plan(input_ids, attention_mask, labels, P, max_len, dev)
        if plan.n_images != imgs.shape[0]:
            raise ValueError(f"batch consumes {plan.n_images} images but {imgs.shape[0]} were given")
        if getattr(self, "unpad", False) and plan.seqlens is not None:
            # UNPADDED (varlen) execution of a ragged batch: the decoder sees only the sum(len) real rows.  Opt-in, because
            # the reference routes its padding rows through the MoE gate too (they take capacity slots and enter l_aux);
            # here they do not exist.  Labels / mask keep the reference's padded [B, S'] shape.
            plan = pack_splice_plan(plan)
        feats = self.encode_images(imgs)                      # [n_img*P, H]
        emb = ops.SpliceEmbed.apply(feats, model.embed_tokens.weight, plan.idx, plan.inv_idx)
        self._plan = plan
        H = emb.shape[1]
        if getattr(plan, "cu", None) is not None:             # packed rows [T, H] (a 3-D view keeps the 6-tuple contract)
            return None, None, plan.attention_mask, past_key_values, emb.view(1, plan.T, H), plan.labels
        return None, None, plan.attention_mask, past_key_values, emb.view(plan.B, plan.S, H), plan.labels
