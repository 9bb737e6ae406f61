R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r2_trace3 -o run -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras > $R/gpurun_out/r2_trace3.log 2>&1
python3 $R/tools/trace_timeline.py $R/gpurun_out/r2_trace3/run_kernel_trace.csv 0.5 24 > $R/gpurun_out/r2_timeline3.txt
cat $R/gpurun_out/r2_timeline3.txt
