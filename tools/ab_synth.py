"""Interleaved A/B of library variants (tools/ablate_libs/<name>.so) on the synthesiser: forward and backward through the C ABI,
256 clips x 16 frames x 8 partials -> 4096 samples."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for name in sys.argv[1:]:
        code = f"""
import os, sys; sys.path.insert(0, {ROOT!r})
os.environ['SOT_LIB_PATH'] = {ROOT!r} + '/tools/ablate_libs/' + {name!r} + '.so'
import torch
from sot_amd import _native as nat
nat.load(build_if_missing=False)
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
amp = torch.rand(256, 16, 8, device=dev, generator=g); f0 = 40 + 1900 * torch.rand(256, 16, 1, device=dev, generator=g)
hann = torch.hann_window(512).to(dev); ga = torch.randn(256, 4096, device=dev, generator=g)
tabs = nat.synth_tap_tables(hann, 16, 4096)
def ev(fn, n=100):
    for _ in range(20): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
audio, ws = nat.synth_forward(amp, f0, hann, 4096, 16000.0, True, for_backward=True)
print('fwd %.1f  bwd %.1f us' % (ev(lambda: nat.synth_forward(amp, f0, hann, 4096, 16000.0, True)),
                                 ev(lambda: nat.synth_backward(amp, f0, hann, 4096, 16000.0, True, ga, forward_workspace=ws, tap_tables=tabs))))
"""
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        print(f"{name:10s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
