import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, torch.nn.functional as F
from rgbmanip_amd import _lib
from gpu_util import conv_nd, rel_err, quantise
lib = _lib.load()
torch.manual_seed(0)
for (N, Cin, H, W, Cout, k) in [(2, 32, 128, 130, 256, 1), (2, 64, 128, 130, 256, 1), (2, 64, 128, 130, 256, 3)]:
    x = quantise(torch.randn(N, Cin, H, W), _lib.BF16X3)
    w = quantise(torch.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k), _lib.BF16X3)
    ref = F.conv2d(x, w, None, 1, k // 2, 1)
    for kern in (0, 1, 2):
        lib.rgbm_set_tuning(b"gemm_kernel", kern)
        y = conv_nd(_lib.BF16X3, x, w, stride=1, pad=k // 2, dil=1)
        e = (y - ref).abs()
        print(f"Cin {Cin} k{k} kernel {kern}: rel err {rel_err(y, ref):.3e}; worst channel err by ch%32: {[round(float(e[:, c::32].max()), 3) for c in range(0, 32, 4)]}; by pixel%64: {[round(float(e.flatten(2)[:, :, p::64].max()), 3) for p in range(0, 64, 8)]}", flush=True)
    # structure test: x one-hot in channel c0 -> y[:, co] = w[co, c0]
    for c0 in (0, 5, 17, 30):
        xo = torch.zeros(N, Cin, H, W); xo[:, c0] = 1.0
        lib.rgbm_set_tuning(b"gemm_kernel", 1)
        y = conv_nd(_lib.BF16X3, xo, w, stride=1, pad=k // 2, dil=1)
        got = y[0, :, H // 2, W // 2]
        want = w[:, c0].sum(dim=(1, 2)) if k > 1 else w[:, c0, 0, 0]
        # which input channel does the output look like?
        cand = [int(torch.argmin(((w.sum(dim=(2, 3)) if k > 1 else w[:, :, 0, 0]) - got[:, None]).abs().sum(0)))]
        print(f"  one-hot c0={c0}: err {float((got - want).abs().max()):.3e}; output matches weight column {cand}")
lib.rgbm_set_tuning(b"gemm_kernel", 2)
