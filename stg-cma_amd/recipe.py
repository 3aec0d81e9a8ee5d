"""Host-side mirror of the reference training recipe's parameter partition (AVE/traintest_adapt_ave29.py:38-61):
parameters train iff they belong to mlp_head or their NAME contains one of the adapter-ish substrings; everything else
(the pretrained backbone, both patch embeddings, the final norm) is frozen when freeze_base=True."""

TRAINABLE_SUBSTRINGS = ("adapter", "temporal_embedding", "ln_post", "Adapter", "my_tokens", "gate_", "ln_before",
                        "temporal_position_bias_table")
MLP_HEAD = tuple(f"mlp_head.{i}.{w}" for i in range(4) for w in ("weight", "bias"))


def is_trainable(name):
    name = name[7:] if name.startswith("module.") else name
    return name in MLP_HEAD or any(s in name for s in TRAINABLE_SUBSTRINGS)


def apply_freeze(model):
    """Set requires_grad like train(..., freeze_base=True) does; returns (adapter_params, head_params)."""
    adapt, head = [], []
    for n, p in model.named_parameters():
        p.requires_grad = is_trainable(n)
        if p.requires_grad:
            (head if (n[7:] if n.startswith("module.") else n) in MLP_HEAD else adapt).append(p)
    return adapt, head
