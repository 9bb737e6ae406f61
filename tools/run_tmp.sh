R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -q -m gpu -x > gpurun_out/r2_t13.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r2_t13.log
python3 bench.py --dtype f64 --landmarks 1000 --steps 300 --no-cpu-baseline --no-extras --stage-profile > gpurun_out/r2_f64_6.json 2>gpurun_out/r2_f64_6.err
python3 -c "
import json;d=json.load(open('gpurun_out/r2_f64_6.json'))
print('f64', d['value'], d['ms_per_step'], d.get('stage_us'), d['factor_flags'], d['roofline']['launch_us'])
"
for fr in "--force-resample" ""; do
python3 bench.py --workload pf --steps 300 --warmup 10 $fr --no-cpu-baseline > gpurun_out/r2_pf2.json 2>gpurun_out/r2_pf2.err || tail -5 gpurun_out/r2_pf2.err
python3 -c "
import json;d=json.load(open('gpurun_out/r2_pf2.json'))
print('pf $fr', d['value'], d['ms_per_step'], d['config']['resamples'])
"
done
