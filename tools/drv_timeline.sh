# kernel timeline of the timed region of the driver's invocation (bench.py --steps 20 --warmup 5); run through gpurun
O=$GRAFT_REPO_ROOT/gpurun_out/prof_drv; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/s -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/b.json 2> $O/b.err
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_drv/s/**/run_kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-60:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    n=r["Kernel_Name"].replace("cslam::","").replace("void ","")[:44]
    print("%-44s q%s %9.1f %9.1f %7.1f"%(n,r["Queue_Id"],(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
