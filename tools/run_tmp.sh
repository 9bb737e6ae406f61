R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/final_r02
mkdir -p $O
python3 bench.py --stage-profile > $O/bench_ekf.json 2> $O/bench.err
python3 bench.py --dtype f64 --landmarks 1000 --stage-profile > $O/bench_f64_n1000.json 2>> $O/bench.err || true
python3 bench.py --workload pf --force-resample > $O/bench_pf.json 2>> $O/bench.err || true
python3 bench.py --workload pf --no-cpu-baseline > $O/bench_pf_natural.json 2>> $O/bench.err || true
python3 bench.py --workload mc --pgemm-wgs 0 --steps 400 > $O/bench_mc.json 2>> $O/bench.err || true
python3 bench.py --sequential --obs 8 --no-cpu-baseline --no-extras > $O/bench_sequential.json 2>> $O/bench.err || true
python3 bench.py --obs 64 --defer 0 --no-cpu-baseline --no-extras --stage-profile > $O/bench_m64.json 2>> $O/bench.err || true
CSLAM_PIPELINE=1 python3 bench.py --no-cpu-baseline --no-extras > $O/bench_pipelined.json 2>> $O/bench.err || true
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2>> $O/bench.err || true
for f in $O/*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); r=d.get('roofline',{})
print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,1),'us', r.get('bound'), r.get('launch_us') and round(r['launch_us'],1), r.get('frac') and round(r['frac'],3))
"; done
