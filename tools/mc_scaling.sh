# kernel times of the batched Monte-Carlo engine against the number of instances (run on the GPU box through gpurun)
O=$GRAFT_REPO_ROOT/gpurun_out/prof_mc_scale; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for I in 1 2 4 8 16; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/i$I -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload mc --mc-engine batch --instances $I --steps 100 --warmup 20 --no-cpu-baseline > $O/mc_$I.json 2> $O/mc_$I.err || exit 1
  echo "== I=$I"; grep -h "batch\|true>" $(find $O/i$I -name run_kernel_stats.csv) | cut -d, -f1-4 | cut -c1-140
  python3 -c "
import json;d=json.loads(open('$O/mc_$I.json').read().strip().split('\n')[-1]);print('value', round(d['value']), 'single', round(d['single_instance']['value']), 'gain', d['concurrency_gain'], 'pgemm us', d['roofline']['launch_us'])"
done
