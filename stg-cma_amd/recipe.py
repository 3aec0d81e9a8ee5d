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


def build_optimizer(model, lr, head_lr=1.0, weight_decay=5e-7, betas=(0.95, 0.999), capturable=False, fused=None):
    """The reference's optimizer (traintest_adapt_ave29.py:38-69): freeze the backbone by name, Adam over two groups --
    adapters / gates / temporal tables at `lr`, the newly initialised mlp_head at lr * head_lr.
    fused (default: whenever the parameters live on the GPU): FusedAdam below -- torch.optim.Adam's arithmetic as ONE launch of
    stg_adam_multi, nothing of a step on the host, so the whole training step can be captured in a HIP graph (capture_train_step).
    fused=False: torch.optim.Adam itself (capturable=True keeps its step counters on the device for graph capture)."""
    import torch
    adapt, head = apply_freeze(model)
    groups = [{"params": adapt, "lr": lr}] + ([{"params": head, "lr": lr * head_lr}] if head else [])   # backbones carry no mlp_head
    if fused is None:
        fused = all(p.is_cuda for g in groups for p in g["params"])
    if fused:
        return FusedAdam(groups, weight_decay=weight_decay, betas=betas)
    return torch.optim.Adam(groups, weight_decay=weight_decay, betas=betas, capturable=capturable)


def _fused_adam_class():
    import torch
    from . import kernels as K

    class FusedAdam(torch.optim.Optimizer):
        """torch.optim.Adam (L2 weight decay, no amsgrad) on the HIP path: every trainable tensor updated by one launch of
        stg_adam_multi (+ one single-thread launch that advances the step counters and bias corrections on the device).
        param_groups carry the usual keys (lr, betas, eps, weight_decay): a loop that writes group["lr"] every iteration
        (traintest_adapt_ave29.py:139-144) works unchanged -- changed values reach the device as fill kernels, no synchronisation.
        state[p] = {step, exp_avg, exp_avg_sq} like torch's, so state_dict() / load_state_dict() interchange with torch.optim.Adam.
        Step counters are per parameter, on the device.  Under HIP-graph capture nothing is written from the host: set
        `lr_tensor(group)` (a device scalar) between replays instead."""

        def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
            super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
            ps = [p for g in self.param_groups for p in g["params"]]
            if not ps or not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps):
                raise RuntimeError("FusedAdam: contiguous fp32 parameters on the GPU only (use torch.optim.Adam elsewhere)")
            dev = ps[0].device
            total = sum((p.numel() + 3) // 4 * 4 for p in ps)
            self._m = torch.zeros(total, dtype=torch.float32, device=dev)
            self._v = torch.zeros(total, dtype=torch.float32, device=dev)
            self._hyper = torch.zeros((len(self.param_groups), 8), dtype=torch.float64, device=dev)
            self._st = torch.zeros((len(ps), 4), dtype=torch.float32, device=dev)      # per tensor: step, bc1, sqrt(bc2), -
            self._sent = [None] * len(self.param_groups)
            off = 0
            self._idx = {}
            for gi, g in enumerate(self.param_groups):
                for p in g["params"]:
                    n = p.numel()
                    self._idx[p] = len(self._idx)
                    self.state[p] = {"step": self._st[self._idx[p], 0], "exp_avg": self._m[off:off + n].view(p.shape),
                                     "exp_avg_sq": self._v[off:off + n].view(p.shape)}
                    off += (n + 3) // 4 * 4
            self._sig, self._desc, self._n, self._max, self._tables = None, None, 0, 0, {}
            # pinned staging for the descriptor tables, allocated here (not under graph capture): 4 slots recycled by eager address sets,
            # 4 that tables built during a capture keep for good (a captured copy node re-reads its slot on every replay)
            self._slot_bytes = (len(ps) * 56 + 63) // 64 * 64
            self._pinned = torch.empty(8 * self._slot_bytes, dtype=torch.uint8).pin_memory()
            self._eager_slot, self._graph_slots = 0, 0

        def lr_tensor(self, group=0):
            return self._hyper[group, 0]

        def _push_hyper(self):
            for gi, g in enumerate(self.param_groups):
                vals = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))
                if g.get("amsgrad") or g.get("maximize"):
                    raise RuntimeError("FusedAdam: amsgrad / maximize are not implemented")
                old = self._sent[gi]
                if old == vals:
                    continue
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("FusedAdam: a hyper-parameter changed during graph capture; write lr_tensor(group) between replays")
                for k, x in enumerate(vals):
                    if old is None or old[k] != x:
                        self._hyper[gi, k].fill_(x)
                self._sent[gi] = vals

        def _table(self):
            live = [(gi, p) for gi, g in enumerate(self.param_groups) for p in g["params"] if p.grad is not None]
            sig = tuple((p.data_ptr(), p.grad.data_ptr()) for _, p in live)
            if sig != self._sig:
                # one table per (parameter, gradient) address set, kept: eager steps and a captured graph (whose gradients live in the
                # graph's own memory pool) each keep theirs; the table goes up from pinned memory, which capture allows
                if sig not in self._tables:
                    for _, p in live:
                        if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.is_sparse:
                            raise RuntimeError("FusedAdam: dense contiguous fp32 gradients only")
                    ent = [(p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(),
                            self._st[self._idx[p]].data_ptr(), p.numel(), gi) for gi, p in live]
                    cap = torch.cuda.is_current_stream_capturing()
                    if cap:
                        if self._graph_slots >= 4:
                            raise RuntimeError("FusedAdam: more than 4 captured address sets")
                        slot = 4 + self._graph_slots
                        self._graph_slots += 1
                    else:
                        slot = self._eager_slot
                        self._eager_slot = (self._eager_slot + 1) % 4
                        for k in [k for k, t in self._tables.items() if t[3] == slot]:
                            torch.cuda.current_stream().synchronize()      # its upload may still be in flight: rare (addresses moved 4 times)
                            self._tables.pop(k)
                    host = self._pinned[slot * self._slot_bytes:(slot + 1) * self._slot_bytes]
                    self._tables[sig] = (K.adam_desc_table(ent, self._m.device, host) if ent else (None, None), len(ent), max([e[5] for e in ent], default=0), slot)
                (self._desc, _), self._n, self._max, _ = self._tables[sig]
                self._sig = sig
            return self._desc

        @torch.no_grad()
        def step(self, closure=None):
            loss = None
            if closure is not None:
                with torch.enable_grad():
                    loss = closure()
            self._push_hyper()
            desc = self._table()
            if desc is not None:
                with torch.cuda.device(self._m.device):
                    K.adam_multi(desc, self._n, self._max, self._hyper)
                # an in-place write outside autograd's sight: bump the version counters the bf16 weight shadows are keyed on (ops.shadow)
                torch.autograd.graph.increment_version([p for g in self.param_groups for p in g["params"] if p.grad is not None])
            return loss

        def load_state_dict(self, state_dict):
            mine = {p: (st["exp_avg"], st["exp_avg_sq"]) for p, st in self.state.items()}
            super().load_state_dict(state_dict)
            self._sent = [None] * len(self.param_groups)
            for gi, g in enumerate(self.param_groups):
                for p in g["params"]:
                    st = self.state.get(p, {})
                    m, v = mine[p]
                    if "exp_avg" in st:
                        if st["exp_avg"] is not m:
                            m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                    else:                                   # not covered by the checkpoint: torch.optim.Adam starts it from scratch
                        m.zero_(); v.zero_()
                    if "step" in st:
                        if st["step"] is not self._st[self._idx[p], 0]:
                            self._st[self._idx[p], 0].fill_(float(st["step"]))
                    else:
                        self._st[self._idx[p]].zero_()
                    self.state[p] = {"step": self._st[self._idx[p], 0], "exp_avg": m, "exp_avg_sq": v}
            self._sig = None

    return FusedAdam


class _Lazy:
    def __init__(self):
        self._cls = None

    def __call__(self, *a, **kw):
        if self._cls is None:
            self._cls = _fused_adam_class()
        return self._cls(*a, **kw)


FusedAdam = _Lazy()


def capture_train_step(step, warmup=3, params=None):
    """Capture one call of `step()` (forward + loss + zero_grad + backward + optimizer.step; the optimizer is build_optimizer's
    FusedAdam, or torch.optim.Adam built with capturable=True) in a HIP graph and return (replay, static_loss): every launch of the
    step -- ~1 500 for Swin-B -- then costs one graph launch (host time of ONE replay from an idle stream: 0.4 ms instead of ~95;
    back-to-back replays block on the launch queue once a few are in flight, so a loop's host time per replay tends to the
    GPU time).  The library's launches go to torch's current stream, so torch.cuda.graph records them like ATen's; `step` must
    not synchronise (no .item() / float(loss)) and its inputs must be static tensors (copy new batches into them).
    `warmup` >= 1 eager calls run on a side stream first: capture needs every lazily built table / shadow / LDS reservation in
    place, and a first call would leave the one-off casts of the frozen weights OUT of the graph's steady state.
    `params`: the tensors the captured optimizer updates (default: those of the optimizers / modules `step` closes over that hold a .grad).  replay() bumps
    their version counters after each launch -- FusedAdam.step's own bump is host code, which a replay does not run -- so an
    eager forward between replays re-casts its bf16 weight shadows (ops.shadow keys on the version) instead of reusing the
    ones written inside the last replay's forward."""
    import torch
    if warmup < 1:
        raise ValueError("capture_train_step: warmup must be >= 1 (lazily built tables and shadows must exist before the capture)")
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
    if params is None:                   # the optimizers / modules `step` closes over
        params = []
        for cell in (getattr(step, "__closure__", None) or ()):
            o = cell.cell_contents
            if isinstance(o, torch.optim.Optimizer):
                params += [p for grp in o.param_groups for p in grp["params"]]
            elif isinstance(o, torch.nn.Module):
                params += [p for p in o.parameters() if p.requires_grad]
    params = list({id(p): p for p in params if p.grad is not None}.values())

    def replay():
        g.replay()
        if params:
            torch.autograd.graph.increment_version(params)

    replay.graph = g
    return replay, static_loss


def capture_train_step_ddp(fwd_bwd, optimizer, sync, warmup=1, collective_in_graph=False):
    """The N > 1 form of capture_train_step: `fwd_bwd()` = forward + loss + zero_grad + backward (returns the loss), `optimizer` =
    build_optimizer's FusedAdam, `sync` = ddp.attach(model).  Returns (replay, static_loss, how):
      how = "two_graphs" (default): graph 1 = forward + backward with the GradSync in deferred mode (the backward records the arena
            and packs the task-head bucket instead of averaging them), the collectives run EAGERLY between the graph launches
            (one all-reduce of the flat arena, one of the bucket when the model has task heads), graph 2 = bucket write-back + Adam.
            Per step the host issues two graph launches and one or two collectives instead of ~1 200 kernel launches.
      how = "one_graph" (collective_in_graph=True, RCCL only): the all-reduce is captured with the rest.  Not the default: this
            repository's build boxes have one GPU, so a captured RCCL collective was never executed here.
    Captures run in thread-local error mode: the process group's watchdog thread polls events while the capture is open.
    Raises whatever the capture raises; the caller falls back to eager steps in the same process."""
    import torch
    if warmup < 1:
        raise ValueError("capture_train_step_ddp: warmup must be >= 1")

    def eager():
        loss = fwd_bwd()
        optimizer.step()
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warmup):
            eager()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    params = [p for grp in optimizer.param_groups for p in grp["params"]]
    if collective_in_graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            static_loss = eager()
        graphs, how = (g,), "one_graph"

        def replay():
            g.replay()
            torch.autograd.graph.increment_version([p for p in params if p.grad is not None])
        replay.release = lambda: None           # nothing to undo: the GradSync was never deferred (every replay form has .release())
    else:
        sync.defer, sync._pending, sync._bucket = True, [], None
        try:
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                static_loss = fwd_bwd()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
                sync.scatter_back()
                optimizer.step()
        except BaseException:
            sync.defer, sync._pending, sync._bucket = False, [], None
            raise
        graphs, how = (g1, g2), "two_graphs"

        def replay():
            g1.replay()
            sync.flush()
            g2.replay()
            torch.autograd.graph.increment_version([p for p in params if p.grad is not None])

        def release():                       # back to eager steps: the GradSync averages inside backward again
            sync.defer, sync._pending, sync._bucket = False, [], None
        replay.release = release
    replay.graphs = graphs
    return replay, static_loss, how


def _concurrent_streams(n):
    """n side streams that demonstrably run BESIDE torch's current stream (and beside each other).  HIP multiplexes a process's streams
    onto a few hardware queues (4 by default): two streams on one queue serialise, and the micro-batch step would then cost MORE than the
    plain one (two half-batch chains back to back: 132 vs 123 ms).  Which queue a pooled torch stream lands on depends on how many streams
    the process has touched before, so it is measured: a ~1 ms spin kernel on the current stream and on the candidate must take ~1 ms
    together, not ~2.  Returns (streams, all_overlap)."""
    import time
    import torch
    cur = torch.cuda.current_stream()
    ticks = 2_500_000                                    # ~1 ms of torch.cuda._sleep on MI355X (measured 0.42 ms per 1e6)

    def run(ss):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s_ in ss:
            with torch.cuda.stream(s_):
                torch.cuda._sleep(ticks)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run([cur])
    base = min(run([cur]) for _ in range(2))
    chosen = []
    cands = [torch.cuda.Stream() for _ in range(8)] + [torch.cuda.Stream(priority=-1) for _ in range(4)]
    for c in cands:
        if len(chosen) == n:
            break
        group = [cur] + chosen + [c]
        if min(run(group) for _ in range(2)) < 1.45 * base:       # all of them in the time of one
            chosen.append(c)
    ok = len(chosen) == n
    while len(chosen) < n:
        chosen.append(cands[len(chosen)])
    return chosen, ok


def capture_train_step_mb(fwd_loss, tensors, optimizer, splits=2, sync=None, warmup=1, require_overlap=True):
    """One training step as `splits` MICRO-BATCHES that run CONCURRENTLY: each micro-batch's forward + backward is its own HIP graph,
    replayed on its own stream, then one join graph sums the gradients and runs the optimizer (round 4).

    Why: every launch of the step carries a ramp-up and a tail in which most of the chip idles (~10 us per HBM-bound launch,
    tools/size_sweep.py, x ~1 200 dependent launches per step on ONE stream).  Two independent launch chains fill each other's gaps:
    forward + backward of Swin-B at B = 32 measured 123.2 ms as one graph, 132.0 ms as two B = 16 graphs back to back, 116.3 ms as the
    same two graphs on two streams (tools/two_stream_try.py; four streams of B = 8: 128.6).  Same arithmetic as the full-batch step:
    the models have no cross-sample statistic (LayerNorm only -- do NOT use this for the AVS decoder's BatchNorm), the loss of
    micro-batch i is weighted by its share of the batch, and fp32 gradient sums are added once more in fp32.

      fwd_loss(*chunk) -> mean loss over the chunk (forward only; e.g. lambda a, v, y: loss_fn(model(a, v, "fusion"), y.flatten(0, 1)))
      tensors           static batch tensors, every one with the batch on dimension 0 (copy new batches INTO them between replays)
      optimizer         build_optimizer's FusedAdam
      sync              ddp.attach(model) at N > 1: the micro-batch graphs run with the GradSync deferred, the summed arena is
                        all-reduced eagerly between the join graph and the optimizer graph.  Round 6: models with task-head gradients
                        (sync.extra: the AVQA head) too -- the join graph packs their summed gradients into ONE flat bucket (zeros for
                        a parameter without a gradient on this rank, like GradSync's own bucket), it is all-reduced beside the arena,
                        and the optimizer graph first copies the averages back.  (The AVS model stays out for a different reason:
                        BatchNorm over the batch -- half batches are not the same function.)
      require_overlap   (default) raise RuntimeError BEFORE capturing anything when no side stream measurably runs beside the launch stream:
                        the chains would then serialise and the step would cost MORE than the plain one (132 vs 123 ms measured) -- the caller
                        falls back to capture_train_step / eager steps.  False: capture anyway; `replay.streams_overlap` says what was found.
    The caller must not keep an eager step's autograd graph alive across this call (drop the last `loss`, gc.collect()): its AccumulateGrad
    nodes are bound to the stream they ran on, and a capture on another stream then fails hard (a segmentation fault was seen in torch 2.10).
    Returns (replay, static_loss, how).  replay() must be called on the stream the tensors are produced on (it forks from and joins
    back into the current stream).  If the capture raises, the parameters' .grad are reset to None (they would point into a discarded
    graph pool) and a deferred GradSync is restored."""
    import torch
    B = tensors[0].shape[0]
    S = int(splits)
    if S < 2 or B % S != 0 or any(t.shape[0] != B for t in tensors):
        raise ValueError("capture_train_step_mb: the batch (dimension 0 of every tensor) must split evenly into >= 2 micro-batches")
    h = B // S
    chunks = [tuple(t[i * h:(i + 1) * h] for t in tensors) for i in range(S)]
    params = [p for grp in optimizer.param_groups for p in grp["params"]]
    w = 1.0 / S
    streams, overlap = _concurrent_streams(S - 1)
    if require_overlap and not overlap:
        raise RuntimeError("capture_train_step_mb: no side stream runs beside the launch stream in this process (HIP put them on one hardware "
                           "queue): the micro-batch chains would serialise; use capture_train_step or pass require_overlap=False")

    def fb(chunk, zero=True):
        loss = fwd_loss(*chunk) * w
        if zero:
            for p in params:
                p.grad = None
        loss.backward()
        return loss

    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(max(int(warmup), 1)):                 # eager steps of the same arithmetic (autograd accumulates the micro-batches)
            for i, c in enumerate(chunks):
                fb(c, zero=i == 0)
            optimizer.step()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    extra_ids = {id(p) for p in sync.extra} if sync is not None else set()
    if sync is not None:
        sync.defer, sync._pending, sync._bucket = True, [], None
        sync.skip_bucket = True                             # the micro-batch graphs do not pack the task-head bucket: the join graph does, once
    try:
        graphs, losses, grads = [], [], []
        for c in chunks:
            # a version bump per micro-batch: each graph then casts the trainable weights' bf16 shadows into an arena of ITS OWN pool
            # (ops.ShadowSet keys on the version) -- shared, one graph would rewrite them while the other reads them
            torch.autograd.graph.increment_version(params)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                loss = fb(c)
            graphs.append(g)
            losses.append(loss)
            grads.append([p.grad for p in params])
        live = [k for k in range(len(params)) if all(gr[k] is not None for gr in grads)]
        if any(gr[k] is not None for gr in grads for k in range(len(params)) if k not in live):
            raise RuntimeError("capture_train_step_mb: a parameter received a gradient in one micro-batch only")
        live_x = [k for k in live if id(params[k]) in extra_ids]          # task-head parameters (gradients outside the backbone's arena)
        live_a = [k for k in live if id(params[k]) not in extra_ids]
        flats = []
        for gr in grads:                                    # one flat arena per micro-batch (ops.GradArena)?  then the sum is one launch
            bases = {id(gr[k]._base): gr[k]._base for k in live_a if gr[k]._base is not None}
            flats.append(next(iter(bases.values())) if len(bases) == 1 and all(gr[k]._base is not None for k in live_a) else None)
        same_layout = bool(live_a) and all(f is not None and f.shape == flats[0].shape for f in flats) and all(
            gr[k].storage_offset() == grads[0][k].storage_offset() for gr in grads for k in live_a)
        gj = torch.cuda.CUDAGraph()
        ga = torch.cuda.CUDAGraph() if sync is not None else None
        bucket, bucket_ps = None, []
        with torch.cuda.graph(gj, capture_error_mode="thread_local"):
            for i in range(1, S):
                if same_layout:
                    flats[0].add_(flats[i])
                elif live_a:
                    torch._foreach_add_([grads[0][k] for k in live_a], [grads[i][k] for k in live_a])
                if live_x:
                    torch._foreach_add_([grads[0][k] for k in live_x], [grads[i][k] for k in live_x])
            static_loss = torch.stack(losses).sum()
            for k, p in enumerate(params):
                p.grad = grads[0][k]
            if sync is not None and extra_ids:                # ONE bucket of every trainable task-head tensor, in sync.extra's order on every rank
                bucket_ps = [p for p in sync.extra if p.requires_grad]
                bucket = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in bucket_ps])
            if sync is None:
                optimizer.step()
        if sync is not None:
            with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                off = 0
                for p in bucket_ps:                         # the averaged task-head gradients back into the .grad tensors the optimizer reads
                    seg = bucket[off:off + p.numel()].view(p.shape)
                    off += p.numel()
                    if p.grad is None:
                        p.grad = seg.to(p.dtype).clone()
                    else:
                        p.grad.copy_(seg)
                optimizer.step()
    except BaseException:
        if sync is not None:
            sync.defer, sync._pending, sync._bucket = False, [], None
            sync.skip_bucket = False
        for p in params:                                    # they point into the discarded graphs' pools
            p.grad = None
        raise
    red = flats[0] if same_layout else None

    def replay():
        c0 = torch.cuda.current_stream()
        for s, g in zip(streams, graphs[1:]):
            s.wait_stream(c0)
            with torch.cuda.stream(s):
                g.replay()
        graphs[0].replay()
        for s in streams:
            c0.wait_stream(s)
        gj.replay()
        if sync is not None:
            if red is not None:
                sync._average(red)
            else:
                for k in live_a:
                    sync._average(grads[0][k])
            if bucket is not None:
                sync._average(bucket)
            ga.replay()
        torch.autograd.graph.increment_version([p for p in params if p.grad is not None])

    def release():
        if sync is not None:
            sync.defer, sync._pending, sync._bucket = False, [], None
            sync.skip_bucket = False
    replay.release = release
    replay.graphs = tuple(graphs) + (gj,) + ((ga,) if ga is not None else ())
    replay.streams_overlap = overlap
    return replay, static_loss, (f"{S} micro-batch graphs on {S} streams + join" + (" + eager all-reduce + optimizer graph" if sync is not None else "")
                                 + ("" if overlap else " (WARNING: no side stream was found to run beside the launch stream)"))


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
