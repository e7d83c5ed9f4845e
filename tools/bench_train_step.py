"""BASELINE config 5: 256 harmonic clips -> STFT x2 (n_fft 2048, hop 256, flattop; 16 frames) -> SOT paper-cutoff
forward + backward w.r.t. the estimate (4096 rows x 1025).  Reports steps/s and the split producer / loss."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sot_amd import spectra
from sot_amd.losses import Wasserstein1D
dev = torch.device("cuda:0")
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device=dev).manual_seed(0)
target = spectra.harmonic_batch(clips, generator=g, device=dev)
estimates = [spectra.harmonic_batch(clips, generator=g, device=dev).requires_grad_(True) for _ in range(4)]
mod = Wasserstein1D(p=2, square_dist=True, dont_normalize=True, limit_quantile_range=True).to(dev)
pos = spectra.unit_frequencies(2048, 16000.0, dev); pos2 = pos.clone()

def ev(fn, n=30):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

def full(i):
    e = estimates[i % 4]; e.grad = None
    spectra.training_step_slice(mod, target, e).backward()
def stft_only(i):
    with torch.no_grad():
        spectra.stft_magnitude(target); spectra.stft_magnitude(estimates[i % 4])
sx = spectra.stft_magnitude(target)
sys_ = [spectra.stft_magnitude(e.detach()).requires_grad_(True) for e in estimates]
def loss_only(i):
    s = sys_[i % 4]; s.grad = None
    mod(sx, s, x_pos=pos, y_pos=pos2).backward()
t_full, t_stft, t_loss = ev(full), ev(stft_only), ev(loss_only)
rows = sx.shape[0] * sx.shape[1]
print(f"config 5: {clips} clips -> {rows} rows x {sx.shape[2]}: full step {t_full:.3f} ms ({1e3 / t_full:.0f} steps/s); "
      f"STFT x2 fwd {t_stft:.3f} ms; SOT fwd+bwd on spectra {t_loss:.3f} ms ({rows / t_loss / 1e3:.1f} Mrows/s)")
print("reference CPU (survey container, 8 vCPU): 509 ms per step = 1.96 steps/s (BASELINE.md)")

# the same step replayed from a HIP graph (what a launch-bound training loop would do: no per-step host work)
def graph_time(step, n=50):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step()
    return ev(lambda i: gr.replay(), n)

e0 = estimates[0]
def static_step():
    e0.grad = None
    spectra.training_step_slice(mod, target, e0).backward()
try:
    t_graph = graph_time(static_step)
    print(f"config 5 replayed from one HIP graph: {t_graph:.3f} ms per step ({1e3 / t_graph:.0f} steps/s)")
except Exception as exc:  # noqa: BLE001
    print("graph capture failed:", repr(exc)[:300])

# with the synthesiser in front: envelopes [clips, 4096, 8] -> oscillator bank -> STFT -> SOT, backward into
# both envelopes
f0 = 40 + 900 * torch.rand(clips, 1, 1, generator=g, device=dev)
freq = (f0 * torch.arange(1, 9, device=dev).view(1, 1, 8)).expand(clips, 4096, 8).contiguous().requires_grad_(True)
amp = (0.1 + 0.1 * torch.rand(clips, 4096, 8, generator=g, device=dev)).requires_grad_(True)
def synth_step(i=0):
    freq.grad = amp.grad = None
    spectra.training_step_slice(mod, target, spectra.oscillator_bank(freq, amp, 16000)).backward()
t_syn = ev(synth_step)
print(f"envelopes -> oscillator bank -> STFT -> SOT, backward into both envelopes: {t_syn:.3f} ms per step ({1e3 / t_syn:.0f} steps/s)")
try:
    t_syn_g = graph_time(synth_step)
    print(f"  replayed from one HIP graph: {t_syn_g:.3f} ms per step ({1e3 / t_syn_g:.0f} steps/s)")
except Exception as exc:  # noqa: BLE001
    print("graph capture failed:", repr(exc)[:300])
