mkdir -p gpurun_out/r5i
python -m pytest tests/test_gpu_halo.py -x -q > gpurun_out/r5i/test.log 2>&1; tail -3 gpurun_out/r5i/test.log
for k in 16 21 16 21; do INFUR_CONV_CFG=$k python scripts/cfg_ab.py 2160 3840 101 f16 2>&1 | grep -E "frame kernels|layer3 conv2 rest|layer4 conv2 first|classifier.0" >> gpurun_out/r5i/ab.log; done; cat gpurun_out/r5i/ab.log
