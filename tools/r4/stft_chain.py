"""Config 5's STFT kernels at 256 clips (n_fft 2048, hop 256), one library variant per process: forward of the pair with the estimate's
spectrum, backward from the spectrum (+ overlap-add), recomputing backward: python tools/r4/stft_chain.py <variant> ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rnd in range(3):
    for name in sys.argv[1:]:
        code = f"""
import os, sys; sys.path.insert(0, {ROOT!r})
os.environ['SOT_LIB_PATH'] = {ROOT!r} + '/tools/ablate_libs/' + {name!r} + '.so'
import torch
from sot_amd import _native as nat, spectra
nat.load(build_if_missing=False)
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
def ev(fn, n=100):
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for clips in (128, 256):
    a = torch.rand(clips, 4096, device=dev, generator=g) - 0.5; b = torch.rand(clips, 4096, device=dev, generator=g) - 0.5
    win = spectra._cached_window('flattop', 2048, dev)
    ma, mb, spec = nat.stft_mag_forward_pair(a, b, win, 2048, 256, want_spec_b=True)
    gm = torch.rand(mb.shape, device=dev, generator=g)
    out.append(f"{{clips}} clips: pair+spec {{ev(lambda: nat.stft_mag_forward_pair(a, b, win, 2048, 256, want_spec_b=True)):.1f}} bwd(spec) {{ev(lambda: nat.stft_mag_backward(b, win, 2048, 256, gm, spec=spec)):.1f}}")
print(' | '.join(out))
"""
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        print(f"{name:10s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
