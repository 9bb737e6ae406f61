R=$GRAFT_REPO_ROOT
cd $R
CSLAM_FACTOR_STAMPS=1 python3 bench.py --obs 64 --defer 0 --steps 100 --no-cpu-baseline --no-extras --stage-profile > gpurun_out/r2_b11.json 2> gpurun_out/r2_b11.err
grep stamps gpurun_out/r2_b11.err | head -2
python3 -c "
import json;d=json.load(open('gpurun_out/r2_b11.json'))
print('m=64 k=128', round(d['value']), d['ms_per_step'], d.get('stage_us'), d['factor_flags'])
"
python -m pytest tests/test_ekf_gpu.py -q -m gpu -x -k "batch_update" 2>&1 | tail -2
