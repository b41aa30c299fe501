"""Micro-benchmark of the implicit-GEMM conv at a given shape through rgbm_conv_nd, timed by the in-library event profiler.
usage: bench_conv.py N H W Cin Cout k dil [res] [flags...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from rgbmanip_amd import _lib
lib = _lib.load()
N, H, W, Cin, Cout, k, dil = [int(x) for x in sys.argv[1:8]]
res = int(sys.argv[8]) if len(sys.argv) > 8 else 0
flags = [int(x) for x in sys.argv[9:]] or [0]
g = torch.Generator().manual_seed(0)
x = torch.randn(N, H, W, Cin, generator=g).bfloat16().cuda()
w = (torch.randn(Cout, Cin, 1, k, k, generator=g) / np.sqrt(Cin * k * k)).contiguous()
bias = torch.randn(Cout, generator=g).contiguous()
out = torch.empty(N, H, W, Cout, dtype=torch.bfloat16, device="cuda")
r = torch.randn(N, H, W, Cout, generator=g).bfloat16().cuda() if res else None
fp = lambda t: t.numpy().ctypes.data_as(C.c_void_p)
pad = dil * (k // 2)
flops = 2.0 * N * H * W * Cout * Cin * k * k
first = None
for f in flags:
    lib.rgbm_debug_flags(f)
    def run():
        _lib.check(lib.rgbm_conv_nd(_lib.BF16, _lib.ptr(x), N, 1, H, W, Cin, Cin, fp(w), Cout, Cout, 1, k, k, 1, 1, 0, pad, dil, 0,
                                    fp(bias), None, None, _lib.ptr(r), 1 if res else 0, 1, 0.0, _lib.ptr(out), _lib.stream_ptr()), "conv_nd")
    out.zero_(); run(); torch.cuda.synchronize()
    if first is None:
        first = out.float().clone()
    else:
        print(f"flags {f}: max |out - out(flags {flags[0]})| = {float((out.float() - first).abs().max()):.3e}  (max |out| {float(first.abs().max()):.2f})")
    lib.rgbm_prof_start()
    for _ in range(3): run()
    torch.cuda.synchronize()
    st = (C.c_double * (4 * _lib.PROF_ROWS))(); lib.rgbm_prof_stop(st)
    st = np.array(list(st)).reshape(_lib.PROF_ROWS, 4)
    for v in range(_lib.PROF_ROWS):
        if st[v, 0] > 0:
            ms = st[v, 1] / st[v, 0]
            print(f"flags {f}: variant {v}: {ms:.4f} ms/launch  {flops / ms / 1e9:.0f} TFLOP/s   (N={N} {H}x{W} {Cin}->{Cout} k{k} d{dil} res={res})")
lib.rgbm_debug_flags(0)
