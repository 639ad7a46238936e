mkdir -p gpurun_out/r5o
INSITU_MIN_GAIN=0.015 INSITU_FRAMES=15 INSITU_CFGS=21,16 python scripts/tune_insitu.py f16 > gpurun_out/r5o/insitu.log 2>&1
cp infur_amd/conv_tune_gfx950.txt gpurun_out/r5o/
tail -12 gpurun_out/r5o/insitu.log
