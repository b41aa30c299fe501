"""Time the post-processing (rgbm_adapose_postprocess_ws) alone: fast vs generic median selection, per batch size.
usage: python tools/pp_ms.py [B ...]   (inputs: one real forward of random-weight bf16 nets on crop inputs, like bench.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from rgbmanip_amd import _lib
from rgbmanip_amd.adapose import AdaPoseNet, postprocess

lib = _lib.load()
Bs = [int(a) for a in sys.argv[1:]] or [1, 8, 256]
for B in Bs:
    torch.manual_seed(0)


    # stand-alone inputs: concentrated-ratio poses (the shape a trained net gives) and spread ones (random weights)
    rng = np.random.default_rng(B)
    P, img = 1024, 224
    rows = []
    for kind in ("spread", "tight"):
        nocs = rng.uniform(-0.45, 0.45, (B, P, 3)).astype(np.float32)
        K = np.tile(np.array([[300.0, 0, 112], [0, 300.0, 112], [0, 0, 1]]), (B, 1, 1))
        E = np.tile(np.eye(4), (B, 1, 1))
        R = np.tile(np.eye(3, dtype=np.float32), (B, 1, 1))
        cam = 0.4 * nocs.astype(np.float64) + np.array([0, 0, 0.9])
        if kind == "spread":
            cam = cam + rng.normal(0, 0.05, cam.shape)
        u = np.clip(np.round(cam[..., 0] / cam[..., 2] * 300 + 112), 0, img - 1)
        v = np.clip(np.round(cam[..., 1] / cam[..., 2] * 300 + 112), 0, img - 1)
        choose = (v * img + u).astype(np.int32)
        depth = cam[..., 2].astype(np.float32)
        args = (torch.from_numpy(nocs).cuda(), torch.from_numpy(depth).cuda(), torch.from_numpy(R).cuda(), torch.from_numpy(choose).cuda(),
                torch.from_numpy(K).cuda(), torch.from_numpy(E).cuda())      # device-resident: the events see the kernels only
        for name, flags in (("fast", 0), ("generic", 1 << 25)):
            _lib.check(lib.rgbm_debug_flags(flags))
            for _ in range(3):
                out = postprocess(*args)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                out = postprocess(*args)
            e1.record()
            torch.cuda.synchronize()
            rows.append((kind, name, e0.elapsed_time(e1) / 10, float(out[1][0, 3])))
        _lib.check(lib.rgbm_debug_flags(0))
    for r in rows:
        print(f"B={B:4d} {r[0]:7s} {r[1]:8s} {r[2]:8.3f} ms per call   scale[0]={r[3]:.12g}")
