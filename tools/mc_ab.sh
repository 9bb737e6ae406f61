# A/B of the batched Monte-Carlo engine's variants on ONE box (run through gpurun): each argument is "ENV=.. [bench flags]"
O=$GRAFT_REPO_ROOT/gpurun_out/mc_ab; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  set -- $v; e=$1; shift
  env $e timeout -k 10 200 python3 bench.py --workload mc --steps 400 --warmup 100 --no-cpu-baseline "$@" > $O/ab.json 2> $O/ab.err || { tail -5 $O/ab.err; exit 1; }
  grep -h stamps $O/ab.err
  python3 -c "
import json;d=json.loads(open('$O/ab.json').read().strip().split('\n')[-1]);print('$v: value', round(d['value']), 'single', round(d['single_instance']['value']), 'gain %.3f'%d['concurrency_gain'], 'pgemm us %.1f'%d['roofline']['launch_us'], 'window us %.1f'%(d['ms_per_step']*2e3), 'flags', d['factor_flags'][:2])"
done
