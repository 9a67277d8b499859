#!/bin/bash
# round 3, GPU call G: training-step A/B (wgrad XCD order + act_bwd), gradient parity, trained-scale precision tests
mkdir -p gpurun_out/r3g
timeout 1500 python -m pytest tests/test_gpu_grad.py tests/test_gpu_trained.py -x -q -m gpu -s > gpurun_out/r3g/pytest.txt 2>&1
tail -25 gpurun_out/r3g/pytest.txt | cut -c1-400
for v in base default; do
  if [ "$v" = "default" ]; then unset GLOWHIP_LIB_PATH; else export GLOWHIP_LIB_PATH=$PWD/pytorch-glow_amd/libglowhip_$v.so; fi
  python bench.py --mode train --steps 8 --warmup 3 2>&1 | tail -1 | cut -c1-300
done > gpurun_out/r3g/train_ab.txt 2>&1
cat gpurun_out/r3g/train_ab.txt
