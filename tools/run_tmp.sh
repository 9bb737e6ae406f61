R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -q -m gpu -x > gpurun_out/r2_t12.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r2_t12.log
CSLAM_FACTOR_STAMPS=1 python3 bench.py --dtype f64 --landmarks 1000 --steps 300 --no-cpu-baseline --no-extras --stage-profile > gpurun_out/r2_f64_5.json 2>gpurun_out/r2_f64_5.err
grep stamps gpurun_out/r2_f64_5.err | head -2
python3 -c "
import json;d=json.load(open('gpurun_out/r2_f64_5.json'))
print('f64', d['value'], d['ms_per_step'], d.get('stage_us'), d['factor_flags'], d['roofline']['launch_us'])
"
