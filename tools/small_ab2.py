import sys,time,numpy as np,torch
sys.path.insert(0,".")
from rgbmanip_amd import synth,_lib
from rgbmanip_amd.adapose import AdaPoseNet, postprocess
lib=_lib.load()
for dt in ("bf16x3","fp32","bf16"):
  net=AdaPoseNet(synth.adapose_state_dict(seed=0),dtype=dt)
  for B in (1,2,4,8):
    inp=synth.adapose_inputs(B,seed=0); d={k:torch.from_numpy(v).cuda() for k,v in inp.items()}
    def f():
      o=net(d["img1"],d["choose1"],d["img2"],d["choose2"],d["P1"],d["P2"],d["depths"])
      postprocess(o["view1_nocs"],o["view1_depth"],o["view1_r"],d["choose1"],d["K1"],d["E1"])
      return o
    res={}
    for rows in (0, 512, 1024, 2048, 4096, 0, 512, 1024, 2048, 4096):
      lib.rgbm_set_tuning(b"ws_min_rows", rows)
      for _ in range(3): f()
      torch.cuda.synchronize(); lat=[]
      for _ in range(30):
        t=time.perf_counter(); o=f(); torch.cuda.synchronize(); lat.append(time.perf_counter()-t)
      res.setdefault(rows,[]).append(round(float(np.median(lat))*1e3,3))
    lib.rgbm_set_tuning(b"ws_min_rows", 0)
    print(dt,"B",B,res)
