"""Experiment (round 4): do S micro-batch steps replayed CONCURRENTLY from S HIP graphs on S streams fill each other's kernel ramps / tails?
Every HBM-bound launch of the step carries ~10 us that does not scale with its size (tools/size_sweep.py): ~1 200 launches per step.
Measures forward + backward (no optimizer step) of the Swin-B AVE workload at B = 32:
  (a) one graph, B = 32     (b) S graphs of B / S, one stream, back to back     (c) the same graphs on S streams     (d) eager, S streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, stgcma  # noqa
dev = torch.device("cuda:0")
m = bench.build_model(torch, dev, "swin_b")
loss_fn = torch.nn.CrossEntropyLoss()
B = int(os.environ.get("B", 32))
a, v, l = bench.synth_batch(torch, B, dev, 0)
l = l.reshape(-1, l.shape[-1])


def fb(aa, vv, ll):
    loss = loss_fn(m(aa, vv, "fusion"), ll)
    for p in m.parameters():
        p.grad = None
    loss.backward()
    return loss


def capture(args, stream):
    with torch.cuda.stream(stream):
        for _ in range(2):
            fb(*args)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        loss = fb(*args)
    return g, loss


N = int(os.environ.get("ITERS", 6))


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


streams = [torch.cuda.Stream() for _ in range(4)]
g32, l32 = capture((a, v, l), streams[0])


def one():
    with torch.cuda.stream(streams[0]):
        g32.replay()


print(f"one graph B={B}: {timed(one):7.2f} ms", flush=True)
for S in [int(x) for x in os.environ.get("SPLITS", "2,4").split(",")]:
    h = B // S
    parts = [(a[i * h:(i + 1) * h].contiguous(), v[i * h:(i + 1) * h].contiguous(), l[i * h * 10:(i + 1) * h * 10].contiguous()) for i in range(S)]
    gs = [capture(parts[i], streams[i])[0] for i in range(S)]

    def seq():
        with torch.cuda.stream(streams[0]):
            for g in gs:
                g.replay()

    def par():
        for i, g in enumerate(gs):
            with torch.cuda.stream(streams[i]):
                g.replay()

    def eager():
        for i in range(S):
            with torch.cuda.stream(streams[i]):
                fb(*parts[i])

    for rnd in range(2):
        print(f"S={S} (B/S={h}) round {rnd}: one stream {timed(seq):7.2f} ms | {S} streams {timed(par):7.2f} ms | eager {S} streams {timed(eager):7.2f} ms | "
              f"one graph B={B} again {timed(one):7.2f} ms", flush=True)
    del gs
    torch.cuda.empty_cache()
