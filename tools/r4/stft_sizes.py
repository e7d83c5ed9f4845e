"""Forward STFT pair (n_fft 2048, hop 256, 4096-sample clips) at several batch sizes, one library variant per process:
python tools/r4/stft_sizes.py <variant> ...   (tools/ablate_libs/<variant>.so)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rnd in range(2):
    for name in sys.argv[1:]:
        code = f"""
import os, sys; sys.path.insert(0, {ROOT!r})
os.environ['SOT_LIB_PATH'] = {ROOT!r} + '/tools/ablate_libs/' + {name!r} + '.so'
import torch
from sot_amd import _native as nat, spectra
nat.load(build_if_missing=False)
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
win = spectra._cached_window('flattop', 2048, dev)
def ev(fn, n=100):
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for clips in (16, 32, 64, 128, 256, 512):
    a = torch.rand(clips, 4096, device=dev, generator=g) - 0.5; b = torch.rand(clips, 4096, device=dev, generator=g) - 0.5
    out.append(f"{{2 * clips * 16}} frames: {{ev(lambda: nat.stft_mag_forward_pair(a, b, win, 2048, 256)):.1f}}")
print(' | '.join(out))
"""
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        print(f"{name:10s}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
