# same-box A/B: Winograd threshold (min Cin) per arithmetic mode
for rep in 1 2; do
for dt in f32s f32x f16; do
for mc in 0 256; do
python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-side --no-split --dtype $dt --winograd-min-cin $mc 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$dt min_cin=$mc fps %.1f one_ctx %.1f'%(d['value'], d['config']['frames_per_s_one_context']))"
done; done; done
