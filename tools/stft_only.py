"""Workload for counter collection: the STFT-magnitude forward kernel alone (256 clips x 4096 samples, n_fft 2048, hop 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sot_amd import spectra
dev = torch.device("cuda:0")
audio = spectra.harmonic_batch(256, generator=torch.Generator(device=dev).manual_seed(0), device=dev)
n_fft, hop = (int(v) for v in (sys.argv[1:3] + [2048, 256][len(sys.argv) - 1:]))
with torch.no_grad():
    for _ in range(30):
        spectra.stft_magnitude(audio, n_fft, hop)
torch.cuda.synchronize()
