run() { python bench.py --stage-profile --no-cpu-baseline --no-deferred-extra 2>gpurun_out/ab_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['stage_us']['downdate'],1), round(d['stage_us']['gather'],1))"; }
for m in 0 4 5 6 0 4 5 6; do CSLAM_PSYM_NT=$m run nt$m; done
