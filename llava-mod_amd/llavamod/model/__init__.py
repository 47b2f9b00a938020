"""`llavamod.model` — the reference's model API surface for the distillation hot path
(reference llavamod/model/__init__.py exports the same names for the Qwen family)."""
from .language_model.llava_qwen2 import (LlavaQwen1_5Config, LlavaQwen1_5ForCausalLM, LlavaQwen2Config,
                                         LlavaQwen2ForCausalLM)
from .language_model.llava_qwen2_moe import (EvalLLaVAMoDQwen1_5ForCausalLM, EvalLLaVAMoDQwen2ForCausalLM,
                                             LLaVAMoDQwen1_5Config, LLaVAMoDQwen1_5ForCausalLM,
                                             LLaVAMoDQwen1_5ForCausalLMFineTune, LLaVAMoDQwen2Config,
                                             LLaVAMoDQwen2ForCausalLM, LLaVAMoDQwen2ForCausalLMFineTune)
from .moe_layer import MoE
from .multimodal_encoder.clip_encoder import CLIPVisionConfig, CLIPVisionTower
