"""Times rgbm_upsample_conv3x3_final (the fused PSPNet tail) at the batch-256 shape.  usage: bench_tail.py [dtype] [V]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbmanip_amd import _lib

dt = {"bf16": _lib.BF16, "f16": _lib.F16, "bf16x3": _lib.BF16X3}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
V = int(sys.argv[2]) if len(sys.argv) > 2 else 512
h = w = 112
lib = _lib.load()
esz = 4 if dt == _lib.BF16X3 else 2
x = torch.randint(0, 255, (V * h * w * 64 * esz,), dtype=torch.uint8, device="cuda")
xr = torch.randn(V * h * w * 64, device="cuda")          # real-valued data: the matrix pipe's power (and so the clock) depends on it
if dt == _lib.BF16X3:
    hi = xr.bfloat16()
    lo = (xr - hi.float()).bfloat16()
    hi16, lo16 = hi.view(torch.int16).view(-1, 4), lo.view(torch.int16).view(-1, 4)
    x = torch.cat([hi16, lo16], dim=1).contiguous().view(torch.uint8).view(-1)
else:
    x = (xr.half() if dt == _lib.F16 else xr.bfloat16()).view(torch.uint8).view(-1)
del xr
out = torch.empty(V * 4 * h * w * 32 * esz, dtype=torch.uint8, device="cuda")
g = np.random.default_rng(0)
w3 = (g.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)
b3 = g.standard_normal(64).astype(np.float32)
wf = (g.standard_normal((32, 64)) / 8).astype(np.float32)
bfin = g.standard_normal(32).astype(np.float32)
p = lambda a: a.ctypes.data_as(_lib.C.c_void_p)
ts = []
for it in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    _lib.check(lib.rgbm_upsample_conv3x3_final(dt, _lib.ptr(x), V, h, w, p(w3), p(b3), 0.25, p(wf), p(bfin), _lib.ptr(out), 0, _lib.stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(os.environ.get("RGBM_HIP_LIB", "default"), sys.argv[1:], "ms:", " ".join(f"{t:.3f}" for t in ts), flush=True)

