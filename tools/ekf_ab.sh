# A/B of engine switches on the headline bench on ONE box (run through gpurun): each argument is "ENV=.. [bench flags]"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ekf_ab
for v in "$@"; do
  set -- $v; e=$1; shift
  env $e timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 600 --warmup 100 "$@" > gpurun_out/ekf_ab/ab.json 2> gpurun_out/ekf_ab/ab.err || { tail -5 gpurun_out/ekf_ab/ab.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ekf_ab/ab.json').read().strip().split('\n')[-1]);print('$v: value', round(d['value']), 'pgemm us %.1f'%d['roofline']['launch_us'], 'frac %.3f'%d['roofline']['frac'], 'flags', d.get('factor_flags'))"
done
