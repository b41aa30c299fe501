import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.nn.functional as F
from rgbmanip_amd import _lib
from gpu_util import to_channels_last, from_channels_last, host_f32, TORCH_DT
lib = _lib.load()
C3T = {1: (8, 16, 2, False), 3: (16, 32, 2, False)}
for layer in (1, 3):
  for trial in range(3):
    dtype = 1
    N, D, H, W = 1, 6, 20, 12
    Cin, Cout, stride, tr = C3T[layer]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, D, H, W, generator=g)
    if trial == 1: x = torch.ones_like(x)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / np.sqrt(Cin * 27)
    scale = torch.ones(Cout); shift = torch.zeros(Cout)
    ref = F.relu(F.conv3d(x.bfloat16().float(), w.bfloat16().float(), None, stride, 1))
    xd = to_channels_last(x, dtype)
    out = torch.full(tuple(ref.permute(0, 2, 3, 4, 1).shape), 777.0, dtype=TORCH_DT[dtype], device="cuda")
    wa, wp = host_f32(w); sa, sp = host_f32(scale); ha, hp = host_f32(shift)
    _lib.check(lib.rgbm_conv3d_tile(layer, dtype, _lib.ptr(xd), N, D, H, W, wp, sp, hp, None, _lib.ptr(out), _lib.stream_ptr()))
    torch.cuda.synchronize()
    y = from_channels_last(out)
    bad = (~torch.isfinite(y)) | (y == 777.0) | ((y - ref).abs() > 0.05 * ref.abs().max())
    idx = bad.nonzero()
    vox = sorted(set((int(a[2]), int(a[3]), int(a[4])) for a in idx))
    print(f"layer {layer} trial {trial}: bad elems {int(bad.sum())} voxels {vox[:10]}")
    for v in vox[:2]:
        print("   got", y[0, :4, v[0], v[1], v[2]].tolist(), "ref", ref[0, :4, v[0], v[1], v[2]].tolist())
