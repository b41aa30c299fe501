"""Same-process interleaved A/B of the 256 x 128 implicit-GEMM kernels (rgbm_set_tuning("gemm_kernel", k)) on the backbone's launch shapes
at batch 256, through rgbm_conv_nd, timed by the in-library event profiler.  usage: bench_gemm_ab.py [rounds] [kernels e.g. 0,1,2] [dtype]
Prints per shape: ms per launch and TFLOP/s per kernel (median over rounds), and max |out_k - out_0|."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from rgbmanip_amd import _lib
lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
kernels = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2").split(",")]      # 10 + k: kernel k with debug flag 134217728 (64-bit global addresses in the ws request waves)
dt_name = sys.argv[3] if len(sys.argv) > 3 else "bf16"
DT = {"bf16": _lib.BF16, "fp16": _lib.F16, "bf16x3": _lib.BF16X3}[dt_name]
N = int(os.environ.get("AB_VIEWS", "512"))
# (name, H, W, Cin, Cout, k, dil, res)
shapes = [("layer3 256->256 d2 res", 28, 28, 256, 256, 3, 2, 1), ("layer3 128->256 d1", 28, 28, 128, 256, 3, 1, 0),
          ("layer4 256->512 d2", 28, 28, 256, 512, 3, 2, 0), ("layer4 512->512 d4 res", 28, 28, 512, 512, 3, 4, 1),
          ("up_1 1x1 1024->2304", 28, 28, 1024, 2304, 1, 1, 0), ("psp 1x1 512->256", 28, 28, 512, 256, 1, 1, 0)]
if os.environ.get("AB_SHAPES"): shapes = [shapes[int(i)] for i in os.environ["AB_SHAPES"].split(",")]
g = torch.Generator().manual_seed(0)
fp = lambda t: t.numpy().ctypes.data_as(C.c_void_p)
def to_store(t):
    if DT == _lib.BF16: return t.bfloat16().cuda()
    if DT == _lib.F16: return t.half().cuda()
    from gpu_util import bx3_pack
    return torch.cat([bx3_pack(c.cuda()) for c in t.float().chunk(16)]).contiguous()
for name, H, W, Cin, Cout, k, dil, res in shapes:
    x = to_store(torch.randn(N, H, W, Cin, generator=g))
    w = (torch.randn(Cout, Cin, 1, k, k, generator=g) / np.sqrt(Cin * k * k)).contiguous()
    bias = torch.randn(Cout, generator=g).contiguous()
    out = torch.empty(N, H, W, Cout, dtype=x.dtype, device="cuda")
    r = to_store(torch.randn(N, H, W, Cout, generator=g)) if res else None
    pad = dil * (k // 2)
    flops = 2.0 * N * H * W * Cout * Cin * k * k
    def run():
        _lib.check(lib.rgbm_conv_nd(DT, _lib.ptr(x), N, 1, H, W, Cin, Cin, fp(w), Cout, Cout, 1, k, k, 1, 1, 0, pad, dil, 0,
                                    fp(bias), None, None, _lib.ptr(r), 1 if res else 0, 1, 0.0, _lib.ptr(out), _lib.stream_ptr()), "conv_nd")
    ms = {kk: [] for kk in kernels}
    ref = None
    diffs = {}
    for kk in kernels:
        _lib.check(lib.rgbm_set_tuning(b"gemm_kernel", kk % 10), "tuning"); lib.rgbm_debug_flags((1 << 27) if kk >= 10 else 0)
        out.zero_(); run(); torch.cuda.synchronize()
        o = out.view(torch.int32 if x.dtype in (torch.float32, torch.int32) else torch.int16).clone()
        if ref is None: ref = o
        else: diffs[kk] = int((o != ref).sum())
    for rd in range(rounds):
        for kk in kernels:
            _lib.check(lib.rgbm_set_tuning(b"gemm_kernel", kk % 10), "tuning"); lib.rgbm_debug_flags((1 << 27) if kk >= 10 else 0)
            lib.rgbm_prof_start()
            for _ in range(4): run()
            torch.cuda.synchronize()
            st = (C.c_double * (4 * _lib.PROF_ROWS))(); lib.rgbm_prof_stop(st)
            st = np.array(list(st)).reshape(_lib.PROF_ROWS, 4)
            tot = sum(st[v, 1] for v in range(_lib.PROF_ROWS))
            ms[kk].append(tot / 4)      # per call: a layer may be two launches (whole rounds of 256 x 256 tiles + a tail)
    line = f"{name:28s}"
    for kk in kernels:
        m = float(np.median(ms[kk])); line += f" | k{kk}: {m:.4f} ms {flops / m / 1e9:6.0f} TF (min {min(ms[kk]):.4f})"
    print(line + f" | elements differing from k{kernels[0]}: {diffs}", flush=True)
_lib.check(lib.rgbm_set_tuning(b"gemm_kernel", 2), "tuning"); lib.rgbm_debug_flags(0)
