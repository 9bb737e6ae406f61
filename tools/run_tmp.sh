R=$GRAFT_REPO_ROOT
cd $R
for cb in 4 2; do
CSLAM_F64_CB=$cb python3 bench.py --dtype f64 --landmarks 1000 --steps 400 --no-cpu-baseline --no-extras > gpurun_out/r2_f64_9.json 2>gpurun_out/r2_f64_9.err
python3 -c "
import json;d=json.load(open('gpurun_out/r2_f64_9.json'))
print('f64 cb=$cb', round(d['value']), d['ms_per_step'], d['roofline']['launch_us'])
"
done
python -m pytest tests -q -m gpu -x -k "f64 or float64 or config1 or golden or deferred or sequential" > gpurun_out/r2_t18.log 2>&1; echo "tests rc=$?"
tail -2 gpurun_out/r2_t18.log
