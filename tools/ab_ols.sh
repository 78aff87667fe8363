# A/B of overlap-save kernel builds on one GPU, one process per arm (same box, back to back)
# usage: ab_ols.sh lib.so[:tune=value] ...   (a library built from another tree state, llzlab_amd/<lib.so>)
show() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(d['value']), round(d['roofline']['kernel_ms_avg'],3), round(d['roofline']['frac'],4), d['parity']['rms_abs'])"; }
cd $GRAFT_REPO_ROOT
for arm in "$@"; do
  lib=${arm%%:*}; var=${arm##*:}
  [ "$var" = "$lib" ] && var=""
  LLZ_LIB=$PWD/llzlab_amd/$lib LLZ_TUNE=$var python bench.py --steps 10 --warmup 2 --no-cpu --no-also 2>/dev/null | show "$lib:$var"
done
