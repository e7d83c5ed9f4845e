"""Secondary measurements (not the contract bench line): forward and forward+backward rows/s and achieved
algorithmic GB/s for the shapes/modes of BASELINE.md, timed with HIP events, rotating input sets."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sot_amd.losses import Wasserstein1D  # noqa: E402

dev = torch.device("cuda:0")
MODES = {"p1": dict(p=1), "cutoff": dict(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True),
         "nocut": dict(p=2, square_dist=True), "p3": dict(p=3)}
CASES = [(8192, 2048, "p1"), (8192, 2048, "cutoff"), (8192, 2048, "p3"), (8192, 512, "cutoff"), (8192, 512, "p1"),
         (1024, 1025, "cutoff"), (4096, 1025, "cutoff"), (16384, 1025, "cutoff"), (4096, 257, "cutoff"), (65536, 257, "cutoff"),
         (2048, 4096, "p1"), (1024, 8192, "p1"),
         (8192, 2000, "p1"), (8192, 2000, "cutoff"), (8192, 2000, "p3"), (8192, 3000, "cutoff"), (16384, 1000, "cutoff")]   # run-time lengths
if len(sys.argv) > 1:
    CASES = [tuple(int(v) if v.isdigit() else v for v in a.split(",")) for a in sys.argv[1:]]


def timeit_graph(fn, iters):
    """GPU-side time per call: `iters` calls captured into one HIP graph and replayed (no host launch cost)."""
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def timeit(fn, iters):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


print(f"{'B':>6} {'N':>5} {'mode':>7} | {'gpu us':>7} | {'fwd us':>8} {'Mrows/s':>8} {'GB/s':>7} {'%8TB/s':>6} | {'f+b(y) us':>9} {'Mrows/s':>8} {'GB/s':>7} {'%8TB/s':>6} | {'f+b(xy) us':>10}")
for B, N, mode in CASES:
    nsets = max(2, min(8, int(600e6 / (B * N * 8)) + 1))
    g = torch.Generator(device=dev).manual_seed(1)
    sets = [(torch.rand(B, N, device=dev, generator=g), torch.rand(B, N, device=dev, generator=g)) for _ in range(nsets)]
    pos = torch.fft.rfftfreq(2 * (N - 1), 1 / 16000.0).to(dev) if N % 2 else torch.linspace(0, 1, N, device=dev)
    pos = (pos / pos.max()).float().contiguous()
    pos2 = pos.clone()
    mod = Wasserstein1D(**MODES[mode]).to(dev)
    iters = 50

    def fwd(i):
        x, y = sets[i % nsets]
        with torch.no_grad():
            return mod(x, y, x_pos=pos, y_pos=pos2)

    ys = [s[1].clone().requires_grad_(True) for s in sets]
    xs = [s[0].clone().requires_grad_(True) for s in sets]

    def fb_y(i):
        x, _ = sets[i % nsets]
        y = ys[i % nsets]
        y.grad = None
        mod(x, y, x_pos=pos, y_pos=pos2).backward()

    def fb_xy(i):
        x, y = xs[i % nsets], ys[i % nsets]
        x.grad = None
        y.grad = None
        mod(x, y, x_pos=pos, y_pos=pos2).backward()

    tf = timeit(fwd, iters)
    tg = timeit_graph(fwd, 20)
    try:
        tb = timeit(fb_y, iters)
        tbb = timeit(fb_xy, iters)
    except Exception as e:  # backward keeps two extra LDS arrays: smaller size limit
        tb = tbb = float("nan")
    bf = (8 * N + 4) * B
    bb = (12 * N + 4) * B  # fwd reads x,y; bwd reads x,y again and writes grad_y: 4(n+m) + 4(n+m) + 4m ... report fwd+bwd(y) algorithmic = 4(n+m)+4m+4
    print(f"{B:6d} {N:5d} {mode:>7} | {tg * 1e6:7.1f} | {tf * 1e6:8.1f} {B / tf / 1e6:8.2f} {bf / tf / 1e9:7.0f} {100 * bf / tf / 8e12:6.1f} | "
          f"{tb * 1e6:9.1f} {B / tb / 1e6:8.2f} {bb / tb / 1e9:7.0f} {100 * bb / tb / 8e12:6.1f} | {tbb * 1e6:10.1f}")
