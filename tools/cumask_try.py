"""Experiment (round 4): the two micro-batch graphs on two streams with DISJOINT compute-unit masks (hipExtStreamCreateWithCUMask): does a
static split of the chip let one chain's HBM-bound kernels run beside the other chain's MFMA-bound GEMMs (a GEMM workgroup owns a whole CU,
so without masks the chains only fill each other's ramps and tails)?  Forward + backward of Swin-B, B = 32 as 2 x 16."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, stgcma  # noqa

dev = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=dev)
path = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        path = line.split()[-1]
        break
hip = ctypes.CDLL(path)
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask rc={rc}")
    return torch.cuda.ExternalStream(s.value)


m = bench.build_model(torch, dev, "swin_b")
loss_fn = torch.nn.CrossEntropyLoss()
B = 32
a, v, l = bench.synth_batch(torch, B, dev, 0)
l = l.reshape(-1, l.shape[-1])
h = B // 2
parts = [(a[i * h:(i + 1) * h].contiguous(), v[i * h:(i + 1) * h].contiguous(), l[i * h * 10:(i + 1) * h * 10].contiguous()) for i in range(2)]


def fb(aa, vv, ll):
    loss = loss_fn(m(aa, vv, "fusion"), ll)
    for p in m.parameters():
        p.grad = None
    loss.backward()
    return loss


def capture(args):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            fb(*args)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fb(*args)
    return g


gs = [capture(p) for p in parts]
N = 6


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


ALL = (1 << 256) - 1
configs = {
    "no masks (plain streams)": None,
    "full masks on both": (ALL, ALL),
    "lower 128 bits | upper 128 bits": ((1 << 128) - 1, ALL ^ ((1 << 128) - 1)),
    "even bits | odd bits": (int("01" * 128, 2), int("10" * 128, 2)),
    "bits with (i % 8) < 4 | >= 4": (sum(1 << i for i in range(256) if i % 8 < 4), sum(1 << i for i in range(256) if i % 8 >= 4)),
    "bits with (i // 8) % 2 == 0 | == 1": (sum(1 << i for i in range(256) if (i // 8) % 2 == 0), sum(1 << i for i in range(256) if (i // 8) % 2 == 1)),
    "lower 160 | upper 160 (overlap 64)": ((1 << 160) - 1, ALL ^ ((1 << 96) - 1)),
}
HALF = ((1 << 128) - 1, ALL ^ ((1 << 128) - 1))
for delay_ms in (0.0, 0.1, 0.3, 1.0, 3.0):
    for name, masks in (("full | full", (ALL, ALL)), ("lower half | upper half", HALF)):
        ss = [masked_stream(masks[0]), masked_stream(masks[1])]
        cyc = int(delay_ms * 1e-3 * 100e6)          # torch.cuda._sleep counts ~100 MHz realtime ticks? calibrated below

        def par():
            with torch.cuda.stream(ss[0]):
                gs[0].replay()
            with torch.cuda.stream(ss[1]):
                if cyc:
                    torch.cuda._sleep(cyc)
                gs[1].replay()

        print(f"offset {delay_ms:4.1f} ms, {name:24s}: two graphs concurrently {timed(par):7.2f} ms", flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(int(1e6)); e1.record(); torch.cuda.synchronize()
print("torch.cuda._sleep(1e6) =", e0.elapsed_time(e1), "ms")
configs["no masks (plain streams) again"] = None
for name, masks in configs.items():
    try:
        ss = [torch.cuda.Stream(), torch.cuda.Stream()] if masks is None else [masked_stream(masks[0]), masked_stream(masks[1])]

        def par():
            for s, g in zip(ss, gs):
                with torch.cuda.stream(s):
                    g.replay()

        def one_only():
            with torch.cuda.stream(ss[0]):
                gs[0].replay()

        print(f"{name:40s}: two graphs concurrently {timed(par):7.2f} ms | graph 0 alone on stream 0 {timed(one_only):7.2f} ms", flush=True)
    except Exception as e:
        print(f"{name}: FAILED {e!r}", flush=True)
