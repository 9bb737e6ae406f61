# Round profile recipe (run on the GPU box through gpurun): three rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy +
# GRBM clock; separate runs, --kernel-trace only), one --kernel-trace --stats run, then the bench lines kept under profiles/.
# Outputs land in gpurun_out/; tools/pmc_summary.py condenses the counter CSVs.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_g/$d -o runc -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline > $R/gpurun_out/pmc_g_$d.log 2>&1
  echo "pmc $d done"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_g -o run -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_g.log 2>&1
echo "stats done"
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_g ekf_downdate > gpurun_out/pmc_g_summary.txt 2>&1 || true
cat gpurun_out/pmc_g_summary.txt


python3 bench.py --stage-profile > gpurun_out/r01_g_bench.json 2> gpurun_out/r01_g_bench.err
python3 bench.py --stage-profile --defer 128 --no-cpu-baseline > gpurun_out/r01_g_bench_deferred128.json 2>> gpurun_out/r01_g_bench.err
python3 bench.py --stage-profile --sequential --obs 8 --no-cpu-baseline > gpurun_out/r01_g_bench_sequential.json 2>> gpurun_out/r01_g_bench.err || true
python3 bench.py --workload pf --no-cpu-baseline > gpurun_out/r01_g_bench_pf.json 2>> gpurun_out/r01_g_bench.err || true
python3 bench.py --dtype f64 --landmarks 1000 --stage-profile --no-cpu-baseline > gpurun_out/r01_g_bench_f64_n1000.json 2>> gpurun_out/r01_g_bench.err || true
tail -c 300 gpurun_out/r01_g_bench.json
