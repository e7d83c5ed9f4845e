#!/bin/bash
# round 4 fuzz campaign with the final library: SOT (C ABI vs C oracle), STFT (vs torch.stft + autograd, large batches included), module (GPU vs CPU route)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
python tools/fuzz_gpu.py 420 4242 > gpurun_out/r4n/fuzz_gpu.txt 2>&1; tail -n 3 gpurun_out/r4n/fuzz_gpu.txt
python tools/fuzz_stft.py 240 4242 > gpurun_out/r4n/fuzz_stft.txt 2>&1; tail -n 3 gpurun_out/r4n/fuzz_stft.txt
python tools/fuzz_module.py 240 4242 > gpurun_out/r4n/fuzz_module.txt 2>&1; tail -n 3 gpurun_out/r4n/fuzz_module.txt
