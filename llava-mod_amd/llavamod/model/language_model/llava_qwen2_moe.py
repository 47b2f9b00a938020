"""LLaVAMoDQwen2ForCausalLM — the sparse-MoE student API (reference language_model/llava_qwen2_moe.py):
config with the `.moe` dict (:48-81), MoE-aware layer/model forward (:112-181,184-339; here the decoder
natively understands 3-tuple MLP returns), `forward` returning loss (+= moe_loss) / moe_loss / logits /
labels / moe_loss_list (:357-451), `initialize_moe_modules` up-cycling the dense FFNs into experts
(:475-561), the `…FineTune` variant that builds the MoE layers from a saved config (:564-626) and the
`Eval…` variant (:629-681).  `LLaVAMoDQwen1_5*` (llava_qwen1_5_moe.py) are the same classes.
"""
import torch

from ..moe_layer import MoE
from ..utils import MoECausalLMOutputWithPast
from .llava_qwen2 import LlavaQwen2Model, _CausalLMBase
from .qwen2_hip import Qwen2Config, init_normal_


class LLaVAMoDQwen2Config(Qwen2Config):
    model_type = "moe_llava_qwen2"

    def __init__(self, moe_enable=True, moe_mode="sparse", moe_layers_idx=None, ep_size=1, top_k_experts=2,
                 capacity_factor=1., eval_capacity_factor=1., min_capacity=4, use_residual=False,
                 router_aux_loss_coef=0.01, **kwargs):
        self.moe = dict(moe_enable=moe_enable, moe_mode=moe_mode, moe_layers_idx=moe_layers_idx, ep_size=ep_size,
                        top_k_experts=top_k_experts, capacity_factor=capacity_factor,
                        eval_capacity_factor=eval_capacity_factor, min_capacity=min_capacity,
                        use_residual=use_residual, router_aux_loss_coef=router_aux_loss_coef, train_modules=[])
        self.lora = {}
        super().__init__(**kwargs)


class LLaVAMoDQwen2Model(LlavaQwen2Model):
    config_class = LLaVAMoDQwen2Config


class LLaVAMoDQwen2ForCausalLM(_CausalLMBase):
    config_class = LLaVAMoDQwen2Config

    def __init__(self, config, device="cuda"):
        super().__init__()
        self.config = config
        self.model = LLaVAMoDQwen2Model(config, device)
        self._setup_head(config, device)
        self.router_aux_loss_coef = config.moe.get("router_aux_loss_coef", 0.01) if hasattr(config, "moe") else 0.01
        self._plan = None
        init_normal_(self, getattr(config, "initializer_range", 0.02), getattr(config, "init_seed", 0))   # post_init()

    def moe_layers(self):
        return [l.mlp for l in self.model.layers if isinstance(l.mlp, MoE)]

    def moe_loss_from_list(self, moe_list):
        """`moe_loss = router_aux_loss_coef * sum(moe_losses)` (:423-431); None when there is no MoE layer."""
        ls = [m for m in moe_list if m is not None]
        if len(ls) == 0:
            return None
        return self.router_aux_loss_coef * sum(ls)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                images=None, return_dict=None):
        if use_cache or past_key_values is not None:
            return self._cached_forward(input_ids, attention_mask, images, past_key_values, MoECausalLMOutputWithPast)
        hidden, moe_list, info = self.forward_hidden(input_ids, attention_mask, labels, images, inputs_embeds)
        logits = self.full_logits(hidden, info.B, info.S, getattr(info, "packed", None))
        loss = self.lm_loss_from_hidden(hidden, info) if info.labels is not None else None
        moe_loss = self.moe_loss_from_list(moe_list)
        if moe_loss is not None and loss is not None:
            loss = loss + moe_loss                               # `loss += moe_loss` (:432-434)
        return MoECausalLMOutputWithPast(loss=loss, moe_loss=moe_loss, logits=logits, labels=info.labels,
                                         moe_loss_list=moe_list)

    # ---- up-cycling ---------------------------------------------------------------------------
    def _build_moe_layers(self, moe_layers_idx, num_experts_list, ep_size, k, cf, ecf, min_cap, use_residual):
        for n_exp, li in zip(num_experts_list, moe_layers_idx):
            dense = self.model.layers[li].mlp
            self.model.layers[li].mlp = MoE(self.config.hidden_size, expert=dense, num_experts=n_exp, ep_size=ep_size,
                                            k=k, capacity_factor=cf, eval_capacity_factor=ecf, min_capacity=min_cap,
                                            use_residual=use_residual)
            self.model.layers[li].mlp._layer_id = li + 1         # noise stream keyed by position in the model

    def initialize_moe_modules(self, model_args):
        """llava_qwen2_moe.py:475-561: record the knobs, freeze by substring BEFORE conversion, choose the
        layers, replace each chosen FFN by `MoE(expert=<that FFN>)` (experts start as identical copies)."""
        m = self.config.moe
        m["moe_enable"] = model_args.moe_enable
        m["train_modules"] = model_args.train_modules
        m["moe_mode"] = model_args.moe_mode
        m["moe_layers_idx"] = model_args.moe_layers_idx
        m["ep_size"] = model_args.ep_size
        m["top_k_experts"] = model_args.top_k_experts
        m["capacity_factor"] = model_args.capacity_factor
        m["eval_capacity_factor"] = model_args.eval_capacity_factor
        m["min_capacity"] = model_args.min_capacity
        m["use_residual"] = model_args.use_residual
        m["router_aux_loss_coef"] = self.router_aux_loss_coef = model_args.router_aux_loss_coef
        if m["train_modules"] is not None and len(m["train_modules"]) > 0:
            for n, p in self.named_parameters():
                if not any(name in n for name in m["train_modules"]):
                    p.requires_grad = False
        L = self.config.num_hidden_layers
        idx = model_args.moe_layers_idx
        if idx is not None:
            model_args.moe_mode = "custom"
            assert len(idx) <= L and max(idx) < L and min(idx) >= 0
        else:
            mode = model_args.moe_mode
            if mode == "first_half":
                idx = list(range(0, L // 2))
            elif mode == "second_half":
                idx = list(range(L // 2, L))
            elif mode == "sparse":
                idx = list(range(L))[::2]
            elif mode == "dense":
                idx = list(range(L))
            else:
                raise NotImplementedError(
                    f'Only support ["first_half", "second_half", "sparse", "dense"], but found {mode}')
        m["moe_layers_idx"] = idx
        ne = list(model_args.num_experts)
        if len(ne) == 1:
            m["num_experts"] = ne * len(idx)
        else:
            m["num_experts"] = ne
        assert len(m["num_experts"]) == len(idx)
        self._build_moe_layers(idx, m["num_experts"], model_args.ep_size, model_args.top_k_experts,
                               model_args.capacity_factor, model_args.eval_capacity_factor, model_args.min_capacity,
                               model_args.use_residual)
        for li in idx:                                           # the reference's allclose check (:547-550)
            ex = self.model.layers[li].mlp.deepspeed_moe.experts.deepspeed_experts
            if next(ex[0].parameters()).device.type == "meta":   # shape-only construction: nothing to compare
                continue
            for e in ex[1:]:
                for (k0, v0), (k1, v1) in zip(ex[0].state_dict().items(), e.state_dict().items()):
                    assert k0 == k1 and torch.equal(v0, v1)


class LLaVAMoDQwen2ForCausalLMFineTune(LLaVAMoDQwen2ForCausalLM):
    """MoE layers are built in __init__ from the saved config.moe (:564-617); trainability is then set by
    substring (:619-626)."""

    def __init__(self, config, device="cuda"):
        super().__init__(config, device)
        m = self.config.moe
        self.router_aux_loss_coef = m["router_aux_loss_coef"]
        self._build_moe_layers(m["moe_layers_idx"], m["num_experts"], m["ep_size"], m["top_k_experts"],
                               m["capacity_factor"], m["eval_capacity_factor"], m["min_capacity"], m["use_residual"])

    def initialize_moe_modules(self, model_args):
        self.config.moe["train_modules"] = model_args.train_modules
        tm = self.config.moe["train_modules"]
        if tm is not None and len(tm) > 0:
            for n, p in self.named_parameters():
                p.requires_grad = any(name in n for name in tm)


class EvalLLaVAMoDQwen2ForCausalLM(LLaVAMoDQwen2ForCausalLMFineTune):
    """Inference variant (llava_qwen2_moe.py:629-681): MoE layers rebuilt from the saved config.moe like the FineTune class,
    every parameter frozen, eval mode (the gate then uses eval_capacity_factor); `generate` / `prepare_inputs_for_generation`
    come from the shared base (KV cache + single-query attention kernel)."""

    def __init__(self, config, device="cuda"):
        super().__init__(config, device)
        self.requires_grad_(False)
        self.eval()

    def initialize_moe_modules(self, model_args):
        raise NotImplementedError("Eval model: no training-time initialisation")


LLaVAMoDQwen1_5Config = LLaVAMoDQwen2Config
LLaVAMoDQwen1_5ForCausalLM = LLaVAMoDQwen2ForCausalLM
LLaVAMoDQwen1_5ForCausalLMFineTune = LLaVAMoDQwen2ForCausalLMFineTune
EvalLLaVAMoDQwen1_5ForCausalLM = EvalLLaVAMoDQwen2ForCausalLM
