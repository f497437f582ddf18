#!/bin/bash
# round 4, job d: ds1x1 streaming kernel + skinny head A/B, emulation diagnostic with the head's masks, full-size oracle parity
mkdir -p gpurun_out/r04d
python -m pytest tests/test_gpu_skinny.py -m gpu -x -q > gpurun_out/r04d/pytest_skinny.log 2>&1
echo "rc=$?" >> gpurun_out/r04d/pytest_skinny.log
python tools/diag_bf16_emulation.py 8 1 101 > gpurun_out/r04d/diag_bf16_emulation_b8.txt 2>&1
python tools/ab_env.py --rounds 3 --steps 100 new: old:VDQN_SKINNY=0,VDQN_DS_STREAM=0 > gpurun_out/r04d/ab_new_old.txt 2>&1
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "vs_oracle or emulating" > gpurun_out/r04d/pytest_fullsize.log 2>&1
echo "rc=$?" >> gpurun_out/r04d/pytest_fullsize.log
