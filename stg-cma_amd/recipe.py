"""Host-side mirror of the reference training recipe's parameter partition (AVE/traintest_adapt_ave29.py:38-61):
parameters train iff they belong to mlp_head or their NAME contains one of the adapter-ish substrings; everything else
(the pretrained backbone, both patch embeddings, the final norm) is frozen when freeze_base=True."""

TRAINABLE_SUBSTRINGS = ("adapter", "temporal_embedding", "ln_post", "Adapter", "my_tokens", "gate_", "ln_before",
                        "temporal_position_bias_table", "avqatask_", "avstask_")   # + the AVQA / AVS loops' task-head prefixes
                                                                                   #   (traintest_adapt_avqa.py:72, _avs.py:55)
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


# ------------------------------------------------------------------------------------------------ the training step
def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """Per-iteration learning-rate table: linear warm-up, then half-cosine decay to final_value (utilities/scheduler.py:5-30;
    traintest_adapt_ave29.py:85-101 builds one table per parameter group and indexes it by global_step, :139-144)."""
    import numpy as np
    warmup_iters = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.zeros(0)
    i = np.arange(epochs * niter_per_ep - warmup_iters)
    cos = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * i / len(i)))
    table = np.concatenate((warm, cos))
    assert len(table) == epochs * niter_per_ep
    return table


def build_optimizer(model, lr, head_lr=1.0, weight_decay=5e-7, betas=(0.95, 0.999), capturable=False):
    """The reference's optimizer (traintest_adapt_ave29.py:38-69): freeze the backbone by name, Adam over two groups --
    adapters / gates / temporal tables at `lr`, the newly initialised mlp_head at lr * head_lr.  capturable=True keeps Adam's step
    counters on the device, so the whole training step can be captured in a HIP graph (capture_train_step)."""
    import torch
    adapt, head = apply_freeze(model)
    groups = [{"params": adapt, "lr": lr}] + ([{"params": head, "lr": lr * head_lr}] if head else [])   # backbones carry no mlp_head
    return torch.optim.Adam(groups, weight_decay=weight_decay, betas=betas, capturable=capturable)


def capture_train_step(step, warmup=3):
    """Capture one call of `step()` (forward + loss + zero_grad + backward + optimizer.step, the optimizer built with capturable=True)
    in a HIP graph and return (replay, static_loss): every launch of the step -- ~2 000 for Swin-B -- then costs one graph launch
    (host time 140 ms -> 2 ms per step, measured).  The library's launches go to torch's current stream, so torch.cuda.graph records
    them like ATen's; `step` must not synchronise (no .item() / float(loss)) and its inputs must be static tensors (copy new batches
    into them).  A few eager calls on a side stream first: graph capture needs every lazily built table / shadow / LDS reservation in
    place."""
    import torch
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warmup):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = step()
    return g.replay, static_loss


def train_step(model, optimizer, loss_fn, a, v, labels, mode, lr_tables=None, global_step=0):
    """One iteration of the reference loop (traintest_adapt_ave29.py:136-164): per-group LR from the cosine tables, labels
    'b t c -> (b t) c', forward, loss on float class-probability targets, zero_grad, backward, step.  The reference wraps the
    forward in fp16 autocast + GradScaler; this path computes in bf16 with fp32 accumulation and fp32 logits, which needs
    neither.  Returns the loss tensor (not synchronised)."""
    if lr_tables is not None:
        for idx, group in enumerate(optimizer.param_groups):
            group["lr"] = float(lr_tables[min(idx, len(lr_tables) - 1)][global_step])
    if labels.dim() == 3:
        labels = labels.reshape(-1, labels.shape[-1])
    loss = loss_fn(model(a, v, mode), labels)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss


# ------------------------------------------------------------------------------------------------ checkpoint tail
def wa_model(exp_dir, start_epoch, end_epoch):
    """Average the per-epoch checkpoints exp_dir/models/audio_model.{epoch}.pth, start_epoch..end_epoch inclusive, key by key
    (AVE/run_adapt_ave29.py:203-214; "not an ensemble": one state_dict comes out).  Exactly the reference's arithmetic:
    running sum in the dtype of the first checkpoint's tensors, then true division by float(count) -- so the integer buffers
    (relative_position_index, t_relative_coords) come back as float tensors, which load_state_dict(strict=True) copies into the
    int64 buffers unchanged (:226).  Keys keep their `module.` prefix (the loop saves the DataParallel-wrapped model, :226-229)."""
    import torch
    sdA = torch.load(exp_dir + '/models/audio_model.' + str(start_epoch) + '.pth', map_location='cpu')
    model_cnt = 1
    for epoch in range(start_epoch + 1, end_epoch + 1):
        sdB = torch.load(exp_dir + '/models/audio_model.' + str(epoch) + '.pth', map_location='cpu')
        for key in sdA:
            sdA[key] = sdA[key] + sdB[key]
        model_cnt += 1
    print('wa {:d} models from {:d} to {:d}'.format(model_cnt, start_epoch, end_epoch))
    for key in sdA:
        sdA[key] = sdA[key] / float(model_cnt)
    return sdA


def save_epoch_checkpoint(model, exp_dir, epoch):
    """What the reference loop writes every epoch (AVE/traintest_adapt_ave29.py:226-229): the state_dict of the wrapped model
    (keys prefixed `module.`) as exp_dir/models/audio_model.{epoch}.pth."""
    import os
    import torch
    os.makedirs(exp_dir + '/models', exist_ok=True)
    sd = model.state_dict()
    if not any(k.startswith('module.') for k in sd):
        sd = {'module.' + k: v for k, v in sd.items()}
    torch.save(sd, "%s/models/audio_model.%d.pth" % (exp_dir, epoch))
