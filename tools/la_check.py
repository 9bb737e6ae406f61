import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conan_slam_amd import EKF
from conan_slam_amd.synth import Workload
N=int(sys.argv[1]); steps=int(sys.argv[2])
w = Workload(N, 32, np.float32)
Zh=[];Ih=[];ctrl=[]
for t in range(steps):
    ctrl.append(w.controls(t)); Z,idf=w.observations(t); Zh.append(Z.reshape(-1,order="F")); Ih.append(idf)
dZ=torch.from_numpy(np.stack(Zh)).cuda(); dI=torch.from_numpy(np.stack(Ih)).cuda(); torch.cuda.synchronize()
res={}
for la in ("0","1"):
    os.environ["CSLAM_LOOKAHEAD"]=la
    e=EKF(N,dtype=np.float32,quirks=0,sync_mode=False); e.set_state(w.X0,w.P0); e.set_deferred(128)
    t0=time.perf_counter()
    for t in range(steps):
        v,swa=ctrl[t]; e.predict(v,swa,w.QE,w.wb,w.dt); e.update_device(dZ.data_ptr()+t*64*4,32,w.RE,dI.data_ptr()+t*32*4,batch=True)
    e.flush(); e.synchronize(); el=time.perf_counter()-t0
    X=e.get_x(); tr=e.trace(); fl=e.factor_status(); res[la]=(X,tr); print(la, "steps/s", steps/el, "flags", fl, "trace", tr, flush=True); e.close()
print("max |dX|", np.abs(res["0"][0]-res["1"][0]).max(), "dtrace", res["0"][1]-res["1"][1])
