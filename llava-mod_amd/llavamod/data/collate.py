"""Batch collators with the reference's contract (data/dataset.py:167-232 SFT, :434-505 DPO):
right padding with `tokenizer.pad_token_id` / -100, `attention_mask = input_ids != pad`, SFT batches truncated to
`tokenizer.model_max_length`, and `images` = the flat list of per-sample image tensors (a sample may carry several).
What the trainers in `llavamod.train` consume is exactly what these return."""
from dataclasses import dataclass
from typing import Any, Dict, Sequence

import torch

from ..constants import IGNORE_INDEX


def _pad_right(seqs, value):
    n = max((int(s.shape[0]) for s in seqs), default=0)
    out = torch.full((len(seqs), n), value, dtype=seqs[0].dtype if seqs else torch.long)
    for i, s in enumerate(seqs):
        out[i, :s.shape[0]] = s
    return out


def _flat_images(instances):
    if "image" not in instances[0]:
        raise ValueError(f"pretrain, {instances}")          # the reference refuses text-only batches the same way
    flat = []
    for inst in instances:
        im = inst["image"]
        flat.extend(im if type(im) is list else [im])
    return flat


@dataclass
class DataCollatorForSupervisedDataset(object):
    """Collate examples for supervised fine-tuning / mimic distillation (keys: input_ids, labels, image)."""
    tokenizer: Any

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, Any]:
        pad, max_len = self.tokenizer.pad_token_id, self.tokenizer.model_max_length
        ids = _pad_right([i["input_ids"] for i in instances], pad)[:, :max_len]
        labels = _pad_right([i["labels"] for i in instances], IGNORE_INDEX)[:, :max_len]
        return dict(input_ids=ids, labels=labels, attention_mask=ids.ne(pad), images=_flat_images(instances))


@dataclass
class DataCollatorForDPODataset(object):
    """Collate examples for preference distillation (keys: chosen_*/rejected_* input_ids and labels, image); no
    truncation, like the reference."""
    tokenizer: Any

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, Any]:
        pad = self.tokenizer.pad_token_id
        batch = {}
        for side in ("chosen", "rejected"):
            ids = _pad_right([i[f"{side}_input_ids"] for i in instances], pad)
            batch[f"{side}_input_ids"] = ids
            batch[f"{side}_labels"] = _pad_right([i[f"{side}_labels"] for i in instances], IGNORE_INDEX)
            batch[f"{side}_attention_mask"] = ids.ne(pad)
        batch["images"] = _flat_images(instances)
        return batch
