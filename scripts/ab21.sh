mkdir -p gpurun_out/r5p
INFUR_CONV_CFG=21 python scripts/race_screen.py 40 f16 > gpurun_out/r5p/race_h4.log 2>&1; tail -4 gpurun_out/r5p/race_h4.log
python scripts/race_screen.py 30 f16 > gpurun_out/r5p/race_db.log 2>&1; tail -4 gpurun_out/r5p/race_db.log
python -m pytest tests -x -q -m gpu > gpurun_out/r5p/gpu_tests.log 2>&1; tail -4 gpurun_out/r5p/gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r5p/bench_s20.json 2> gpurun_out/r5p/bench_s20.err; tail -c 600 gpurun_out/r5p/bench_s20.json
