R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pf -o run -- python3 $R/bench.py --workload pf --no-cpu-baseline > $R/gpurun_out/prof_pf.log 2>&1
cd $R
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_pf/run_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    print(r['Name'].split('(')[0][-58:], r['Calls'], round(float(r['AverageNs'])/1000,2), r['Percentage'])
print('total kernel ms', tot/1e6)
PY
tail -c 700 gpurun_out/prof_pf.log
