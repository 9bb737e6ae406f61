set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for sp in 0 16 64; do
  CSLAM_PGEMM_SPARE=$sp python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $R/gpurun_out/r2_b2_sp$sp.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$R/gpurun_out/r2_b2_sp$sp.json'));print('spare',$sp,d['value'],d['ms_per_step'],d['roofline']['launch_us'])"
done
CSLAM_PIPELINE=0 python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras > $R/gpurun_out/r2_b2_nopipe.json 2>/dev/null
python3 -c "import json;d=json.load(open('$R/gpurun_out/r2_b2_nopipe.json'));print('nopipe',d['value'],d['ms_per_step'],d['roofline']['launch_us'])"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2_trace1 -o run -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras > $R/gpurun_out/r2_trace1.log 2>&1
f=$(ls $R/gpurun_out/r2_trace1/*/*kernel_trace.csv $R/gpurun_out/r2_trace1/*kernel_trace.csv 2>/dev/null | head -1)
python3 $R/tools/trace_timeline.py $f 0.5 30 > $R/gpurun_out/r2_timeline1.txt
cat $R/gpurun_out/r2_timeline1.txt
