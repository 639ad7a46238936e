bash scripts/gpu_profile.sh f32 > gpurun_out/prof_f32.log 2>&1; tail -2 gpurun_out/prof_f32.log
bash scripts/gpu_profile.sh f32s > gpurun_out/prof_f32s.log 2>&1; tail -2 gpurun_out/prof_f32s.log
bash scripts/gpu_profile.sh f32x > gpurun_out/prof_f32x.log 2>&1; tail -2 gpurun_out/prof_f32x.log
bash scripts/gpu_profile.sh f16hl > gpurun_out/prof_f16hl.log 2>&1; tail -2 gpurun_out/prof_f16hl.log
bash scripts/gpu_profile.sh f16 > gpurun_out/prof_f16.log 2>&1; tail -2 gpurun_out/prof_f16.log
bash scripts/gpu_profile.sh f16 4k --depth 101 --width 3840 --height 2160 --frames-per-step 2 > gpurun_out/prof_f16_4k.log 2>&1; tail -2 gpurun_out/prof_f16_4k.log
bash scripts/gpu_profile.sh i8 > gpurun_out/prof_i8.log 2>&1; tail -2 gpurun_out/prof_i8.log
# BASELINE configs[2]: the Scale kernel in front (north_star: "rocprof evidence of achieved HBM GB/s on the pre kernels")
bash scripts/gpu_profile.sh f32 scale05 --scale 0.5 > gpurun_out/prof_scale05.log 2>&1; tail -2 gpurun_out/prof_scale05.log
