#!/bin/bash
# round 4, second session: fuzz campaign with the final library + smoke()
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4z; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt
python tools/fuzz_gpu.py 360 9191 > $O/fuzz_gpu.txt 2>&1; tail -n 1 $O/fuzz_gpu.txt
python tools/fuzz_stft.py 200 9191 > $O/fuzz_stft.txt 2>&1; tail -n 1 $O/fuzz_stft.txt
python tools/fuzz_module.py 300 9191 > $O/fuzz_module.txt 2>&1; tail -n 1 $O/fuzz_module.txt
grep -c PGRAD $O/fuzz_module.txt
