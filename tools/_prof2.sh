R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d -o run -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-deferred-extra --defer 128 > $R/gpurun_out/prof_d.log 2>&1
cd $R
cut -c1-110 gpurun_out/prof_d/run_kernel_stats.csv | head -14
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_d/run_kernel_stats.csv')))
for r in rows[:12]:
    print(r['Name'].split('(')[0][-60:], r['Calls'], r['AverageNs'])
PY
