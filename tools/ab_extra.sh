# A/B of whole-library builds on the secondary paths: time-domain 257-tap FIR (bench --algo 1) and bench --extra
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  LLZ_LIB=$PWD/llzlab_amd/$lib python bench.py --steps 3 --warmup 1 --no-cpu --algo 1 --extra 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
a=d['also']
print('$lib', 'td257: %.2f ms %.0f GB/s |' % (d['roofline']['kernel_ms_avg'], d['roofline']['achieved']), 'fir63: %.0f GB/s | rs13: %.2f ms %.0f GB/s | iir8: %.0f ms' % (a['fir63_64ch']['GBs'], a['resample_1to3_f32_1024ch']['ms'], a['resample_1to3_f32_1024ch']['GBs'], a['iir8_128ch']['ms']), 'parity', d['parity']['rms_abs'])"
done
