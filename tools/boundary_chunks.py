"""estimate() with host float64 frames at N = 256 for several hip_upload_chunk values (pinned staging = 2 slots x chunk x 2 views x
7.4 MB + masks).  usage: boundary_chunks.py [dtype]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rgbmanip_amd import synth
from rgbmanip_amd.config import ADAPOSE_CFGS
from rgbmanip_amd.estimator import AdaPoseEstimator_v5
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda", 0)
_, d, fr = bench.make_inputs_crop(256, dev, seed=0, keep_frames=True)
Kh, E1h, E2h = fr["K"].cpu().numpy(), fr["E1"].cpu().numpy(), fr["E2"].cpu().numpy()
r1, r2 = fr["rgb1"].cpu().numpy().astype(np.float64), fr["rgb2"].cpu().numpy().astype(np.float64)
m1, m2 = fr["mask1"].cpu().numpy().astype(np.float64), fr["mask2"].cpu().numpy().astype(np.float64)
del fr, d
sd = synth.adapose_state_dict(seed=0)
base = None
fdt = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
if fdt == np.float32:
    r1, r2, m1, m2 = r1.astype(np.float32), r2.astype(np.float32), m1 != 0, m2 != 0
for chunk in (64, 32, 16, 8, 16, 32, 64, 24):
    est = AdaPoseEstimator_v5(None, dict(ADAPOSE_CFGS["adapose_cabinet"], load=False, hip_prepare="device", hip_upload_chunk=chunk), None, state_dict=sd, dtype=dt)
    bb = est.estimate(Kh, r1, m1, E1h, r2, m2, E2h)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(4):
        bb = est.estimate(Kh, r1, m1, E1h, r2, m2, E2h)
    ms = (time.perf_counter() - t) / 4 * 1e3
    if base is None:
        base = bb
    print(f"{dt} {fdt.__name__} chunk {chunk}: {ms:.1f} ms per call = {256 / ms * 1e3:.0f} poses/s, pinned {sum(t.numel() * t.element_size() for s in est._pipe_pin for t in s) / 1e9:.2f} GB, max |box - chunk-64 box| {np.abs(bb - base).max():.2e}")
    del est
    torch.cuda.empty_cache()
