import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, torch.nn.functional as F
from rgbmanip_amd import _lib
from gpu_util import to_channels_last, from_channels_last, host_f32, TORCH_DT
lib = _lib.load()
C3T = {0: (32, 8, 1, False), 1: (8, 16, 2, False), 2: (16, 16, 1, False), 3: (16, 32, 2, False), 5: (32, 64, 2, False), 7: (64, 32, 2, True)}
for layer in (1, 3, 5, 2, 7):
  for dtype in (0, 1):
    for (N, D, H, W) in ((1, 4, 16, 16), (1, 6, 20, 12), (2, 5, 9, 7)):
        Cin, Cout, stride, tr = C3T[layer]
        g = torch.Generator().manual_seed(1)
        x = torch.randn(N, Cin, D, H, W, generator=g)
        w = (torch.randn(Cin, Cout, 3, 3, 3, generator=g) if tr else torch.randn(Cout, Cin, 3, 3, 3, generator=g)) / np.sqrt(Cin * 27)
        scale = torch.ones(Cout); shift = torch.zeros(Cout)
        ref = F.conv_transpose3d(x, w, None, 2, 1, 1) if tr else F.conv3d(x, w, None, stride, 1)
        ref = F.relu(ref)
        xd = to_channels_last(x, dtype)
        out = torch.full(tuple(ref.permute(0, 2, 3, 4, 1).shape), float("nan"), dtype=TORCH_DT[dtype], device="cuda")
        wa, wp = host_f32(w); sa, sp = host_f32(scale); ha, hp = host_f32(shift)
        _lib.check(lib.rgbm_conv3d_tile(layer, dtype, _lib.ptr(xd), N, D, H, W, wp, sp, hp, None, _lib.ptr(out), _lib.stream_ptr()))
        torch.cuda.synchronize()
        y = from_channels_last(out)
        bad = ~torch.isfinite(y)
        err = ((y - ref).abs() * (~bad)).max().item() / ref.abs().max().item()
        idx = bad.nonzero()
        print(f"layer {layer} dt {dtype} shape {(N,D,H,W)} -> out {tuple(ref.shape)} nonfinite {int(bad.sum())} relerr(finite) {err:.2e}",
              "first bad:", idx[:3].tolist() if len(idx) else "", "chan set:", sorted(set(idx[:,1].tolist()))[:8] if len(idx) else "")
