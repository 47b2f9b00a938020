"""MI355X-native LLaVA-MoD distillation step behind the reference's `llavamod` API.

Only the hot path of SURVEY.md §8 lives here: `llavamod.model` (LlavaQwen2ForCausalLM teacher,
LLaVAMoDQwen2ForCausalLM student, MoE layer, CLIP tower, projector, splice) and `llavamod.train`
(AlignTrainer / DPOTrainer loss steps), all executing on the hand-written HIP kernels in
`../csrc` through the C ABI declared in `include/lmod_hip.h`.
"""
