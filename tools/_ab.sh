run() { python bench.py --stage-profile --no-cpu-baseline 2>gpurun_out/ab_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['stage_us']['downdate'],1))"; }
for m in 0 1 2 3 0 2 3; do CSLAM_PSYM_NT=$m run nt$m; done
