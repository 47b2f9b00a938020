"""CPU oracle for the LLaVA-MoD distillation step — TEST INFRASTRUCTURE ONLY.

A plain-PyTorch fp32 restatement of the reference's algorithm for the hot path,
each function citing the reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this package, and only as the checker / reported
baseline — never as something the product path (`llava-mod_amd/`) calls.

Parity pinning:
  * decoder ops, dense LLaVA forward, splice, projector, CLIP tower wrapper, AlignTrainer /
    DPOTrainer loss functions: PINNED against the imported reference in the authoring container
    by `oracle/validate_vs_reference.py`, which also wrote the golden vectors in `tests/golden/`.
  * deepspeed.moe (DeepSpeed 0.9.5, reference requirements.txt:9): the package is neither vendored
    in the reference nor installed here -> `oracle/moe.py` restates its published algorithm;
    PARITY UNPINNED at that boundary (property tests + self-generated golden vectors only).
"""
