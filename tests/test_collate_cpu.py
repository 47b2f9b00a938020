"""Collators reproduce the reference's batch contract (data/dataset.py:167-232, :434-505)."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch
from torch.nn.utils.rnn import pad_sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "llava-mod_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from llavamod.data import DataCollatorForDPODataset, DataCollatorForSupervisedDataset  # noqa: E402

TOK = SimpleNamespace(pad_token_id=151646, model_max_length=12)


def _inst(n, seed, n_img=1):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, 1000, (n,), generator=g)
    lab = ids.clone(); lab[: n // 2] = -100
    imgs = [torch.randn(3, 4, 4, generator=g) for _ in range(n_img)]
    return dict(input_ids=ids, labels=lab, image=imgs if n_img != 1 else imgs[0])


def test_supervised_collator_contract():
    inst = [_inst(5, 1), _inst(15, 2, n_img=2), _inst(9, 3)]
    b = DataCollatorForSupervisedDataset(TOK)(inst)
    ref_ids = pad_sequence([i["input_ids"] for i in inst], batch_first=True, padding_value=TOK.pad_token_id)[:, :12]
    ref_lab = pad_sequence([i["labels"] for i in inst], batch_first=True, padding_value=-100)[:, :12]
    assert torch.equal(b["input_ids"], ref_ids) and torch.equal(b["labels"], ref_lab)
    assert torch.equal(b["attention_mask"], ref_ids.ne(TOK.pad_token_id)) and b["attention_mask"].dtype == torch.bool
    assert len(b["images"]) == 4 and all(im.shape == (3, 4, 4) for im in b["images"])
    assert b["images"][1] is inst[1]["image"][0] and b["images"][3] is inst[2]["image"]
    with pytest.raises(ValueError):
        DataCollatorForSupervisedDataset(TOK)([dict(input_ids=torch.ones(3, dtype=torch.long), labels=torch.ones(3, dtype=torch.long))])


def test_dpo_collator_contract():
    inst = []
    for k in range(3):
        c, r = _inst(6 + 4 * k, 10 + k), _inst(20 - 5 * k, 20 + k)
        inst.append(dict(chosen_input_ids=c["input_ids"], chosen_labels=c["labels"], rejected_input_ids=r["input_ids"],
                         rejected_labels=r["labels"], image=c["image"]))
    b = DataCollatorForDPODataset(TOK)(inst)
    for side in ("chosen", "rejected"):
        ref = pad_sequence([i[f"{side}_input_ids"] for i in inst], batch_first=True, padding_value=TOK.pad_token_id)
        assert torch.equal(b[f"{side}_input_ids"], ref)                      # no truncation on the DPO path
        assert torch.equal(b[f"{side}_labels"], pad_sequence([i[f"{side}_labels"] for i in inst], batch_first=True, padding_value=-100))
        assert torch.equal(b[f"{side}_attention_mask"], ref.ne(TOK.pad_token_id))
    assert len(b["images"]) == 3
    assert set(b) == {"chosen_input_ids", "chosen_labels", "chosen_attention_mask", "rejected_input_ids", "rejected_labels",
                      "rejected_attention_mask", "images"}


def _cases():
    """The very instances `oracle/validate_vs_reference.py::collate_cases` fed to the IMPORTED reference collators (restated here:
    tests may not import /root/reference, and the fixture holds the reference's OUTPUTS)."""
    def inst(n, seed, n_img=1, bare=False):
        g = torch.Generator().manual_seed(seed)
        ids = torch.randint(0, 1000, (n,), generator=g)
        lab = ids.clone(); lab[: n // 2] = -100
        imgs = [torch.randn(3, 4, 4, generator=g) for _ in range(n_img)]
        return dict(input_ids=ids, labels=lab, image=imgs[0] if bare else imgs)
    sft = [inst(5, 1), inst(15, 2, n_img=2), inst(9, 3, bare=True), inst(12, 4)]
    dpo = []
    for k in range(3):
        c, r = inst(6 + 4 * k, 10 + k, n_img=1 + (k == 1)), inst(20 - 5 * k, 20 + k)
        dpo.append(dict(chosen_input_ids=c["input_ids"], chosen_labels=c["labels"], rejected_input_ids=r["input_ids"],
                        rejected_labels=r["labels"], image=c["image"]))
    return sft, dpo


def test_collators_equal_the_imported_reference_collators():
    """VERDICT r03 next #5a: `tests/golden/collate.safetensors` holds what the reference's own `DataCollatorForSupervisedDataset`
    / `DataCollatorForDPODataset` (data/dataset.py:167-232, :434-505, imported by oracle/validate_vs_reference.py) returned for
    these instances — every key, dtype, shape and value must agree, the flat image list included."""
    from safetensors.torch import load_file
    gold = load_file(os.path.join(ROOT, "tests", "golden", "collate.safetensors"))
    sft, dpo = _cases()
    for tag, b in (("sft", DataCollatorForSupervisedDataset(TOK)(sft)), ("dpo", DataCollatorForDPODataset(TOK)(dpo))):
        keys = {k[len(tag) + 1:] for k in gold if k.startswith(tag + ".")}
        assert set(b) == keys, (tag, set(b), keys)
        for k, v in b.items():
            ref = gold[f"{tag}.{k}"]
            if k == "images":
                assert len(v) == ref.shape[0] and all(torch.equal(a, r) for a, r in zip(v, ref))
            elif k.endswith("attention_mask"):
                assert v.dtype == torch.bool and torch.equal(v, ref.bool())
            else:
                assert v.dtype == ref.dtype and torch.equal(v, ref), (tag, k)


def test_attention_backward_ds_spill_envelope_is_host_checkable(monkeypatch):
    """kernels.attn_bwd_ds_fusable mirrors the library's envelope for the 5-matmul backward (csrc/attn.hip): head dim 128, dense layout,
    whole 256-row blocks, a bounded workspace, and the LMOD_ATTN_DS=0 switch — pure host logic, no device."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-mod_amd"))
    from llavamod import kernels as K
    monkeypatch.delenv("LMOD_ATTN_DS", raising=False)
    assert K.attn_bwd_ds_fusable(16, 2048, 16, 128)
    assert K.attn_bwd_ds_fusable(4, 8192, 16, 128)                       # exactly 8 GiB of workspace
    assert not K.attn_bwd_ds_fusable(8, 8192, 16, 128)                   # 16 GiB: over the bound
    assert not K.attn_bwd_ds_fusable(16, 2048, 14, 64)                   # hd 64 keeps the two-kernel form
    assert not K.attn_bwd_ds_fusable(16, 2000, 16, 128)                  # not whole 256-row blocks
    assert not K.attn_bwd_ds_fusable(16, 2048, 16, 128, cu=object())     # packed (varlen) layout
    monkeypatch.setenv("LMOD_ATTN_DS", "0")
    assert not K.attn_bwd_ds_fusable(16, 2048, 16, 128)
