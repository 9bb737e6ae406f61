# Round profile recipe (run on the GPU box through gpurun): rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, MFMA busy + GRBM clock;
# separate runs, --kernel-trace only), --kernel-trace --stats runs, timelines, then the bench lines kept under profiles/.
# Outputs land in gpurun_out/prof_r03/; tools/pmc_summary.py condenses the counter CSVs.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$(echo $c | tr ' ' '_')
  # headline (look-ahead windows, k = 128 launches) and immediate (k = 64 launches), separate passes per counter set
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/k128_$d -o runc -- python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras > $O/pmc_k128_$d.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/k64_$d -o runc -- python3 $R/bench.py --defer 0 --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $O/pmc_k64_$d.log 2>&1
  # the batched Monte-Carlo engine's P-GEMM (8 x N = 2000 per launch)
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/mc_$d -o runc -- python3 $R/bench.py --workload mc --steps 12 --warmup 4 --no-cpu-baseline > $O/pmc_mc_$d.log 2>&1
  echo "pmc $d done"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ekf -o run -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/stats_ekf.log 2>&1
echo "stats ekf done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ekf_classic -o run -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras --no-lookahead > $O/stats_ekf_classic.log 2>&1
echo "stats ekf classic done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ekf_immediate -o run -- python3 $R/bench.py --defer 0 --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/stats_ekf_immediate.log 2>&1
echo "stats ekf immediate done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f64 -o run -- python3 $R/bench.py --dtype f64 --landmarks 1000 --steps 200 --warmup 10 --no-cpu-baseline --no-extras > $O/stats_f64.log 2>&1
echo "stats f64 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pf -o run -- python3 $R/bench.py --workload pf --steps 200 --warmup 10 --force-resample --no-cpu-baseline > $O/stats_pf.log 2>&1
echo "stats pf done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mc -o run -- python3 $R/bench.py --workload mc --steps 100 --warmup 10 --no-cpu-baseline > $O/stats_mc.log 2>&1
echo "stats mc done"
cd $R
python3 tools/pmc_summary.py $O/pmc ekf_downdate > $O/pmc_summary.txt 2>&1 || true
cat $O/pmc_summary.txt
python3 tools/trace_timeline.py $(find $O/stats_ekf -name run_kernel_trace.csv | head -1) 0.6 24 > $O/timeline_lookahead.txt 2>&1 || true
python3 tools/trace_timeline.py $(find $O/stats_ekf_classic -name run_kernel_trace.csv | head -1) 0.6 24 > $O/timeline_classic.txt 2>&1 || true
python3 tools/trace_timeline.py $(find $O/stats_f64 -name run_kernel_trace.csv | head -1) 0.6 24 > $O/timeline_f64.txt 2>&1 || true
python3 tools/trace_timeline.py $(find $O/stats_mc -name run_kernel_trace.csv | head -1) 0.3 24 > $O/timeline_mc_batch.txt 2>&1 || true
tools/probes/mfma_peak_probe > $O/mfma_peak.txt 2>&1 || true
python3 bench.py > $O/bench_ekf.json 2> $O/bench.err
python3 bench.py --no-lookahead --no-cpu-baseline --no-extras > $O/bench_ekf_classic.json 2>> $O/bench.err || true
python3 bench.py --dtype f64 --landmarks 1000 > $O/bench_f64_n1000.json 2>> $O/bench.err || true
python3 bench.py --workload pf --force-resample > $O/bench_pf.json 2>> $O/bench.err || true
python3 bench.py --workload mc > $O/bench_mc.json 2>> $O/bench.err || true
python3 bench.py --workload mc --mc-engine handles --no-cpu-baseline > $O/bench_mc_handles.json 2>> $O/bench.err || true
python3 bench.py --sequential --obs 8 --no-cpu-baseline --no-extras > $O/bench_sequential.json 2>> $O/bench.err || true
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2>> $O/bench.err || true
for f in bench_ekf bench_ekf_classic bench_f64_n1000 bench_pf bench_mc bench_mc_handles bench_sequential bench_driver_shape; do python3 -c "
import json;d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1]);print('$f', round(d['value']), round(d['ms_per_step'],5), (d.get('roofline') or {}).get('frac'))" || true; done
