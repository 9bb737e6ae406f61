R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -q -m gpu -x > gpurun_out/r2_t16.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r2_t16.log
python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline > gpurun_out/r2_b8.json 2>gpurun_out/r2_b8.err || tail -5 gpurun_out/r2_b8.err
python3 -c "
import json;d=json.load(open('gpurun_out/r2_b8.json'))
print('headline', round(d['value']), d['ms_per_step'])
for k in ('reference_loop','reference_loop_ref_exact'): print(k, {kk:vv for kk,vv in d[k].items() if kk!='note'})
"
