R=$GRAFT_REPO_ROOT
cd $R
for cfg in "20 400" "400 400" "400 1000" "20 1000" "1000 1000"; do
set -- $cfg
python3 bench.py --dtype f64 --landmarks 1000 --warmup $1 --steps $2 --no-cpu-baseline --no-extras > gpurun_out/r2_f64_11.json 2>gpurun_out/r2_f64_11.err || tail -3 gpurun_out/r2_f64_11.err
python3 -c "
import json;d=json.load(open('gpurun_out/r2_f64_11.json'));print('f64 warmup $1 steps $2:', round(d['value']), round(d['ms_per_step']*1e3,1), 'us/step, P-GEMM', round(d['roofline']['launch_us'],1), d['roofline']['launches_timed'])"
done
