"""Per-chunk host timings of estimate()'s upload pipeline (RGBM_UPLOAD_TRACE=1): how long the host waits for a staging slot and how long the
staging copies take.  usage: boundary_trace.py [chunk] [threads]"""
import os, sys, time
os.environ["RGBM_UPLOAD_TRACE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rgbmanip_amd import synth, estimator as E
from rgbmanip_amd.config import ADAPOSE_CFGS
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 32
if len(sys.argv) > 2:
    E._HOST_THREADS = int(sys.argv[2])
dev = torch.device("cuda", 0)
_, d, fr = bench.make_inputs_crop(256, dev, seed=0, keep_frames=True)
Kh, E1h, E2h = fr["K"].cpu().numpy(), fr["E1"].cpu().numpy(), fr["E2"].cpu().numpy()
r1, r2 = fr["rgb1"].cpu().numpy().astype(np.float64), fr["rgb2"].cpu().numpy().astype(np.float64)
m1, m2 = fr["mask1"].cpu().numpy().astype(np.float64), fr["mask2"].cpu().numpy().astype(np.float64)
del fr, d
est = E.AdaPoseEstimator_v5(None, dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device", hip_upload_chunk=chunk), None,
                            state_dict=synth.adapose_state_dict(seed=0), dtype="bf16")
for _ in range(3):
    t = time.perf_counter()
    est.estimate(Kh, r1, m1, E1h, r2, m2, E2h)
    print(f"call: {(time.perf_counter() - t) * 1e3:.1f} ms  (threads {E._HOST_THREADS})", file=sys.stderr)
