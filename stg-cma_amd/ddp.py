"""Data parallelism for the adapter-tuning recipe: one process per MI355X, RCCL over xGMI.

What is exchanged.  Only the trainable gradients (adapters, gates, temporal bias tables, head): 5.60 M fp32 values = 22.4 MB
for Swin-B (19 M for Swin-L).  The frozen 87 M backbone parameters are never communicated -- the reference's
nn.DataParallel re-broadcasts all 92 M parameters every step (AVE/traintest_adapt_ave29.py:32-35) and is replaced, not
reproduced.

How.  ops.SwinModelFn.backward writes every trainable gradient into ONE flat fp32 arena (ops.GradArena); when a GradSync
is attached to the model the arena is averaged across ranks with a single all-reduce issued from inside the autograd node,
before the per-parameter views are handed back to autograd.  So `loss.backward()` in an unmodified training loop already
leaves averaged gradients in `p.grad` (GradScaler's inf-check then sees identical values on every rank).  On the
fully-connected 8-GPU xGMI mesh the payload is ~2.8 MB per link per phase: tens of microseconds against a >= 100 ms step, so
one un-bucketed collective at the end of backward is the right shape; there is nothing to overlap it with that would matter.

Sharding.  Clips are independent (LayerNorm only on the hot path, no cross-sample statistic), so each rank draws its own
B clips: weak scaling, no data-path collective besides the gradient average.
"""
import os

import torch
import torch.distributed as dist


class GradSync:
    """Averages the trainable gradients of one backward pass across the ranks of `group`.

    Two carriers, one accounting:
      * the backbone's flat fp32 arena (ops.GradArena), all-reduced in place from inside the autograd node (allreduce_), and
      * `extra`: trainable parameters whose gradients autograd accumulates outside that node -- the AVS dense decoder
        (`avstask_*`, ops_dec Functions) and the AVQA question-answering head (`avqatask_*`, ops_head Functions).  Their
        post-accumulate hooks arm ONE end-of-backward callback that packs every such gradient into a single flat bucket,
        all-reduces it and writes the averages back (a parameter that received no gradient on this rank contributes zeros, so
        every rank sends the same bucket).
    `last_numel` = gradient elements exchanged by the most recent backward pass (arena + bucket, without alignment padding);
    a model is fully covered when it equals the number of trainable elements."""

    def __init__(self, group=None):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("GradSync needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.calls = 0
        self.last_numel = 0
        self.extra = []
        self._hooked = set()
        self._acc = 0
        self._armed = -1                       # id of the backward pass whose end-of-pass callback is queued
        # deferred mode (recipe.capture_train_step_ddp): backward only RECORDS what has to be averaged -- the arena, and the packed
        # bucket of the task heads -- so that forward + backward can live in one HIP graph, the collectives run eagerly between two
        # graph launches (flush) and the bucket is written back to the .grad tensors inside the second graph (scatter_back)
        self.defer = False
        self._pending = []
        self._bucket = None
        self.skip_bucket = False               # deferred mode inside recipe.capture_train_step_mb: the JOIN graph packs the bucket, once, from the summed gradients

    # ---------------------------------------------------------------- the collective
    def _average(self, flat):
        if self.world == 1 or flat.numel() == 0:
            return flat
        if self.backend == "nccl":            # RCCL on ROCm: average inside the collective, no extra kernel
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:                                  # gloo (CPU tests): SUM then scale
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)
        return flat

    def allreduce_(self, flat, n_real=None):
        """Average `flat` in place.  n_real: gradient elements in it when it carries alignment padding."""
        self.calls += 1
        self._acc += flat.numel() if n_real is None else n_real
        if self.defer:
            if not any(f is flat for f in self._pending):
                self._pending.append(flat)
        else:
            self._average(flat)
        if len(self._hooked) != len(self.extra):
            self.rewatch()
        self._arm()
        return flat

    # ---------------------------------------------------------------- end-of-backward bookkeeping + the extra bucket
    def _arm(self, *_):
        gid = torch._C._current_graph_task_id()
        if gid == -1:                          # not inside a backward pass (a direct allreduce_ call): account for it now
            self.last_numel, self._acc = self._acc, 0
        elif gid != self._armed:               # first event of this backward pass (a pass that raised leaves a stale id behind)
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
            self._armed = gid

    def _finish(self):
        self._armed = -1
        ps = [p for p in self.extra if p.requires_grad]
        if ps:
            if self.defer and self.skip_bucket:            # capture_train_step_mb: the join graph packs the bucket from the SUMMED gradients
                self._bucket = None
                self._acc += sum(p.numel() for p in ps)
                self.last_numel, self._acc = self._acc, 0
                return
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in ps])
            if self.defer:                     # packed inside the capture; averaged by flush(), written back by scatter_back()
                self._bucket = (flat, ps)
                self._acc += flat.numel()
                self.last_numel, self._acc = self._acc, 0
                return
            self._average(flat)
            off = 0
            for p in ps:
                seg = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
                if p.grad is None:
                    p.grad = seg.to(p.dtype).clone()
                else:
                    p.grad.copy_(seg)
            self._acc += flat.numel()
        self.last_numel, self._acc = self._acc, 0

    def flush(self):
        """Deferred mode: run the collectives the last backward pass recorded (eagerly, on the current stream)."""
        for f in self._pending:
            self._average(f)
        if self._bucket is not None:
            self._average(self._bucket[0])

    def scatter_back(self):
        """Deferred mode: copy the averaged task-head bucket back into the parameters' .grad tensors (capturable: copies only)."""
        if self._bucket is None:
            return
        flat, ps = self._bucket
        off = 0
        for p in ps:
            seg = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
            if p.grad is None:
                p.grad = seg.to(p.dtype).clone()
            else:
                p.grad.copy_(seg)

    def watch(self, params):
        """Route parameters whose gradients do not pass through the backbone arena into the end-of-backward bucket.  Frozen
        parameters (requires_grad=False: an ablation or a linear probe that freezes part of a task head before attach) are
        recorded without a hook -- autograd refuses a hook on a tensor that needs no gradient -- and picked up by rewatch()."""
        for p in params:
            if not any(p is q for q in self.extra):
                self.extra.append(p)
        self.rewatch()

    def rewatch(self):
        """Hook every recorded parameter that requires a gradient now and has no hook yet.  Runs at every arena all-reduce (each
        backward pass of a model with trainable backbone tensors), so a parameter unfrozen after attach() is synchronised from
        the NEXT backward pass on; call it by hand after unfreezing when the backbone itself is entirely frozen."""
        for p in self.extra:
            if p.requires_grad and id(p) not in self._hooked:
                p.register_post_accumulate_grad_hook(self._arm)
                self._hooked.add(id(p))


def attach(model, group=None):
    """Enable gradient averaging for a stg-cma_amd model: the backbone arena inside its autograd node plus -- for the AVS / AVQA
    mirrors -- every task-head parameter (`avstask_*` / `avqatask_*`, AVS/traintest_adapt_avs.py:55, AVQA/traintest_adapt_avqa.py:72)
    through the end-of-backward bucket.  Call once after construction, before or after the loop's freeze: frozen task-head
    tensors are recorded and hooked when they are unfrozen (GradSync.rewatch).

    BatchNorm of the AVS decoder (TPAVI `W_z`, AVS/model/TPAVI.py:57-61): batch statistics stay PER RANK, which is what the
    reference's nn.DataParallel does (each replica normalises its own chunk, AVS/traintest_adapt_avs.py:35-38); its affine
    parameters are `avstask_*` tensors and are averaged like every other gradient.  Running statistics: the reference keeps
    replica 0's; `broadcast_buffers` copies rank 0's to every rank (call it before validation / checkpointing)."""
    sync = GradSync(group)
    plan = model._plan()
    plan.ddp = sync
    if hasattr(model, "_flat_tensors"):
        inside = set(model._flat_tensors()[0])
        sync.watch(p for n, p in model.named_parameters() if n not in inside)
    return sync


def init_from_env(backend=None):
    """torchrun-style bootstrap: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.  Returns (rank, local_rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def broadcast_parameters(model, src=0, group=None):
    """Make every rank start from rank `src`'s trainable values (frozen weights come from the same checkpoint / seed)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():                       # an in-place write on the parameter itself: bumps its version, so the bf16
        for p in model.parameters():            # weight shadows (ops.shadow, keyed on the version) are re-cast
            if p.requires_grad:
                dist.broadcast(p, src=src, group=group)


def broadcast_buffers(model, src=0, group=None):
    """Copy rank `src`'s floating-point buffers (the AVS decoder's BatchNorm running statistics) to every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for b in model.buffers():
            if b.is_floating_point():
                dist.broadcast(b, src=src, group=group)
