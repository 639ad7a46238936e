for rep in 1 2; do
for v in 0 1; do
for dt in f32 f32s; do
INFUR_WINO_VEC=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side --no-split --dtype $dt 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); o=d['roofline']['other_kernels']
print('vec=$v $dt fps %.1f one_ctx %.1f wino_in %.3f wino_out %.3f'%(d['value'], d['config']['frames_per_s_one_context'], o['wino_input']['ms'], o['wino_output']['ms']))"
done; done; done
