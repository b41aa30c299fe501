import sys, os, zlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, torch.nn.functional as F
from rgbmanip_amd import _lib
from gpu_util import conv_nd
import test_gpu_kernels as tk
name = "ws128_res_pre"
case = [c for c in tk.CONV2D_CASES if c[0] == name][0]
_, N, Cin, H, W, Cout, k, stride, pad, dil, has_bias, act, res_mode = case
g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
x = (torch.randn(N, Cin, H, W, generator=g) * 2000.0).half().float()
w = (torch.randn(Cout, Cin, k, k, generator=g) * 40.0 / np.sqrt(Cin * k * k)).half().float()
b = None
res = (torch.randn(N, Cout, H, W, generator=g) * 3e4).half().float()
ref = F.conv2d(x, w, b, stride, pad, dil) + res
ref = F.relu(ref).clamp(-65504.0, 65504.0)
y = conv_nd(_lib.F16, x, w, stride=stride, pad=pad, dil=dil, bias=b, res=res, res_mode=res_mode, act=act, slope=0.25)
d = (y - ref).abs()
i = np.unravel_index(int(d.argmax()), d.shape)
print("max diff", float(d.max()), "at", i, "y", float(y[i]), "ref", float(ref[i]), "count>100:", int((d > 100).sum()))
bad = (d > 100).nonzero()
print(bad[:10])
pre = F.conv2d(x, w, b, stride, pad, dil) + res
print("pre-act values at bad:", [float(pre[tuple(t)]) for t in bad[:10]])
nanpos = torch.isnan(y).nonzero()
print("n NaN", len(nanpos), "of", y.numel())
conv = F.conv2d(x, w, b, stride, pad, dil)
for t in nanpos[:12]:
    t = tuple(int(v) for v in t)
    print(t, "conv", float(conv[t]), "res", float(res[t]), "sum", float(pre[t]))
import collections
print("channels", collections.Counter(int(t[1]) % 16 for t in nanpos).most_common(6), "x mod 16", collections.Counter(int(t[3]) % 16 for t in nanpos).most_common(6))
