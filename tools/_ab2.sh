run() { python bench.py --stage-profile --no-cpu-baseline --no-deferred-extra 2>gpurun_out/ab_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step']*1000,1), {k: round(v,1) for k,v in d['stage_us'].items()})"; }
for g in 8 4 2 1; do
  sed -i "s/^constexpr int kGatherObs = [0-9]*;/constexpr int kGatherObs = $g;/" conan_slam_amd/csrc/ekf_kernels.hpp
  python -m conan_slam_amd.build --force > gpurun_out/build_$g.log 2>&1 || { echo build failed; tail -5 gpurun_out/build_$g.log; exit 1; }
  run obs$g
done
