mkdir -p gpurun_out/r5m
for abl in 0 16 32 0; do INFUR_H4_ABL=$abl INFUR_CONV_CFG=21 python scripts/cfg_ab.py 2160 3840 101 f16 2>&1 | grep -E "frame kernels|layer3 conv2 rest|layer4 conv2 first|classifier.0" >> gpurun_out/r5m/abl.log; done; cat gpurun_out/r5m/abl.log
