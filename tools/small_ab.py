import sys,time,numpy as np,torch
sys.path.insert(0,".")
from rgbmanip_amd import synth,_lib
from rgbmanip_amd.adapose import AdaPoseNet, postprocess
lib=_lib.load()
for dt in ("bf16","bf16x3"):
  net=AdaPoseNet(synth.adapose_state_dict(seed=0),dtype=dt)
  for B in (1,2,8,16):
    inp=synth.adapose_inputs(B,seed=0); d={k:torch.from_numpy(v).cuda() for k,v in inp.items()}
    def f():
      o=net(d["img1"],d["choose1"],d["img2"],d["choose2"],d["P1"],d["P2"],d["depths"])
      postprocess(o["view1_nocs"],o["view1_depth"],o["view1_r"],d["choose1"],d["K1"],d["E1"])
      return o
    res={}
    for flag in (0, 1<<24, 0, 1<<24):
      lib.rgbm_debug_flags(flag)
      for _ in range(3): f()
      torch.cuda.synchronize(); lat=[]
      for _ in range(40):
        t=time.perf_counter(); o=f(); torch.cuda.synchronize(); lat.append(time.perf_counter()-t)
      res.setdefault(flag,[]).append(round(float(np.median(lat))*1e3,3))
      outs=res.setdefault(("o",flag),{k:v.clone() for k,v in o.items()})
    lib.rgbm_debug_flags(0)
    diff=max(float((res[("o",0)][k]-res[("o",1<<24)][k]).abs().max()/res[("o",0)][k].abs().max()) for k in res[("o",0)])
    print(dt,"B",B,"slim-small",res[0],"wide",res[1<<24],"max rel diff",f"{diff:.1e}")
