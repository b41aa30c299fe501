"""Characterise the error pattern of the bf16x3 conv0 sweep against build_volume(fp32) + CPU conv3d (debug tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.nn.functional as F
from rgbmanip_amd import _lib
from gpu_util import to_channels_last, from_channels_last, host_f32, empty_out
from test_gpu_kernels import _sweep_case
lib = _lib.load()
D, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5, 20, 37)
B, V = 2, 4
g = torch.Generator().manual_seed(7)
w = torch.randn(8, 32, 3, 3, 3, generator=g) / np.sqrt(32 * 27)
scale = torch.rand(8, generator=g) + 0.5
shift = torch.randn(8, generator=g) * 0.1
wf = w * scale.view(-1, 1, 1, 1, 1)
wa, wp = host_f32(w); sa, sp = host_f32(scale); ha, hp = host_f32(shift)
feat, P, dep = _sweep_case(B, D, H, W, seed=11)
fd = to_channels_last(feat, _lib.F32)
Pd, dd = P.cuda(), dep.cuda()
hom = torch.empty(V * 12, dtype=torch.float32, device="cuda")
vol = torch.empty(V, D, H, W, 32, dtype=torch.float32, device="cuda")
_lib.check(lib.rgbm_build_volume(_lib.F32, _lib.ptr(fd), _lib.ptr(Pd), _lib.ptr(dd), _lib.ptr(hom), _lib.ptr(vol), V, B, D, H, W, _lib.stream_ptr()))
out = empty_out((V, D, H, W, 8), _lib.BF16X3)
_lib.check(lib.rgbm_conv0_sweep_dt(_lib.BF16X3, _lib.ptr(fd), _lib.ptr(Pd), _lib.ptr(dd), _lib.ptr(hom), wp, sp, hp, _lib.ptr(out), V, B, D, H, W, _lib.stream_ptr()))
torch.cuda.synchronize()
x = vol.cpu().permute(0, 4, 1, 2, 3)
ref = F.relu(F.conv3d(x, wf, None, 1, 1) + shift.view(1, -1, 1, 1, 1))
y = from_channels_last(out)
e = (y - ref).abs()
sc = ref.abs().max()
print("max rel err", float(e.max() / sc), "nan count", int(torch.isnan(y).sum()))
print("per view   ", (e.amax(dim=(1, 2, 3, 4)) / sc).numpy().round(5))
print("per channel", (e.amax(dim=(0, 2, 3, 4)) / sc).numpy().round(5))
print("per depth  ", (e.amax(dim=(0, 1, 3, 4)) / sc).numpy().round(5))
print("per row    ", (e.amax(dim=(0, 1, 2, 4)) / sc).numpy().round(4))
print("per col    ", (e.amax(dim=(0, 1, 2, 3)) / sc).numpy().round(4))
# partial references: which taps are missing?  conv with only kd = k
for kd in range(3):
    wk = torch.zeros_like(wf); wk[:, :, kd] = wf[:, :, kd]
    part = F.conv3d(x, wk, None, 1, 1)
    full = F.conv3d(x, wf, None, 1, 1) + shift.view(1, -1, 1, 1, 1)
    pre = torch.where(y > 0, y, torch.zeros_like(y))
    # is y ~ relu(full - part)?
    alt = F.relu(full - part)
    print(f"if kd={kd} taps were missing: err {float((y - alt).abs().max() / sc):.4f}")
bad = (e / sc > 1e-3)
print("bad voxels", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()
import collections
print("bad (view, ch, d, h, w) samples:", idx[:12].tolist())
print("distinct (v,d,h,w):", sorted(set((int(a), int(c), int(d_), int(e_)) for a, b, c, d_, e_ in idx.tolist()))[:40])
