"""Output containers with the field names the reference's trainers read (model/utils.py:120-127,
language_model/llava_qwen2_moe.py:100-109): .loss .logits .labels .moe_loss .moe_loss_list."""


class _Out(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class CausalLMOutputWithPast(_Out):
    def __init__(self, loss=None, logits=None, labels=None, past_key_values=None, hidden_states=None, attentions=None):
        super().__init__(loss=loss, logits=logits, labels=labels, past_key_values=past_key_values,
                         hidden_states=hidden_states, attentions=attentions)


class MoECausalLMOutputWithPast(_Out):
    def __init__(self, loss=None, moe_loss=None, logits=None, labels=None, past_key_values=None, hidden_states=None,
                 attentions=None, moe_loss_list=None):
        super().__init__(loss=loss, moe_loss=moe_loss, logits=logits, labels=labels, past_key_values=past_key_values,
                         hidden_states=hidden_states, attentions=attentions, moe_loss_list=moe_loss_list)
