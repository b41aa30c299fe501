"""Per-tile fixed cost of the depth-sweeping conv0 kernels: time of the stand-alone sweep at D = 12 / 24 / 48 depth planes (same tiles, same
tile walk), fitted as t = tiles_per_CU * (a + b * D).  a / (a + 24 b) is the share of a D = 24 tile that is fill / drain between tiles.
usage: sweep_depth_fit.py [bf16x3|bf16|fp16] [views]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from rgbmanip_amd import _lib
from gpu_util import to_channels_last, host_f32, empty_out
from test_gpu_kernels import _sweep_case
lib = _lib.load()
dt_name = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
V = int(sys.argv[2]) if len(sys.argv) > 2 else 128
DT = {"bf16x3": _lib.BF16X3, "bf16": _lib.BF16, "fp16": _lib.F16}[dt_name]
B, H, W = V // 2, 224, 224
g = torch.Generator().manual_seed(7)
w = torch.randn(8, 32, 3, 3, 3, generator=g) / np.sqrt(32 * 27)
scale = torch.rand(8, generator=g) + 0.5
shift = torch.randn(8, generator=g) * 0.1
wa, wp = host_f32(w); sa, sp = host_f32(scale); ha, hp = host_f32(shift)
res = {}
for D in (12, 24, 48):
    feat, P, dep = _sweep_case(B, D, H, W, seed=11)
    fd = to_channels_last(feat, _lib.F32 if DT == _lib.BF16X3 else DT)
    Pd, dd = P.cuda(), dep.cuda()
    hom = torch.empty(V * 12, dtype=torch.float32, device="cuda")
    out = empty_out((V, D, H, W, 8), DT)
    def run():
        _lib.check(lib.rgbm_conv0_sweep_dt(DT, _lib.ptr(fd), _lib.ptr(Pd), _lib.ptr(dd), _lib.ptr(hom), wp, sp, hp, _lib.ptr(out), V, B, D, H, W, _lib.stream_ptr()))
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    res[D] = float(np.median(ts))
    print(f"{dt_name} V={V} D={D}: {res[D]:.3f} ms", flush=True)
    del fd, out
    torch.cuda.empty_cache()
b = (res[48] - res[12]) / 36.0
a = res[24] - 24 * b
print(f"fit: per-launch fixed part {a:.3f} ms of {res[24]:.3f} ms at D = 24 ({100 * a / res[24]:.1f} %), {b * 1e3:.1f} us per plane and launch")
