#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
python -m pytest tests/test_stft_producer.py tests/test_oscillator_bank.py -x -q -m gpu -s > gpurun_out/r4k/pytest_stft.log 2>&1; echo "pytest stft rc=$?"
grep -n "one-wave backward\|passed\|failed\|Error" gpurun_out/r4k/pytest_stft.log | tail -8
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from sot_amd import _native as nat, spectra
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
def ev(fn, n=100):
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for clips in (64, 128, 256, 512):
    a = torch.rand(clips, 4096, device=dev, generator=g) - 0.5; b = torch.rand(clips, 4096, device=dev, generator=g) - 0.5
    win = spectra._cached_window('flattop', 2048, dev)
    ma, mb, spec = nat.stft_mag_forward_pair(a, b, win, 2048, 256, want_spec_b=True)
    gm = torch.rand(mb.shape, device=dev, generator=g)
    print(f"{clips} clips: pair fwd+spec {ev(lambda: nat.stft_mag_forward_pair(a, b, win, 2048, 256, want_spec_b=True)):.1f} us | bwd from spec {ev(lambda: nat.stft_mag_backward(b, win, 2048, 256, gm, spec=spec)):.1f} us | bwd recomputing {ev(lambda: nat.stft_mag_backward(b, win, 2048, 256, gm)):.1f} us")
PY
