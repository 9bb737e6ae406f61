R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -q -m gpu -x > gpurun_out/r2_t9.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r2_t9.log
for q in 4 8 16; do for i in 2 4 8; do
GPU_MAX_HW_QUEUES=$q python3 bench.py --workload mc --steps 200 --warmup 20 --no-cpu-baseline --pgemm-wgs 0 --instances $i > gpurun_out/r2_mc3.json 2>gpurun_out/r2_mc3.err || tail -5 gpurun_out/r2_mc3.err
python3 -c "
import json;d=json.load(open('gpurun_out/r2_mc3.json'))
print('mc queues=$q inst=$i', round(d['value']), round(d['ms_per_step'],4), 'single', round(d['single_instance']['value']), 'gain', round(d['concurrency_gain'],2), round(d['roofline']['launch_us'],1))
"
done; done
