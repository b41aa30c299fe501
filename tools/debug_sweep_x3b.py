"""X3_DBG=4 build: the kernel dumps what the consumers read from the LDS ring (centre tap, channels 0..7); compare with the
fp32 volume rounded to 16 significand bits.  Tells producer / LDS problems from consumer arithmetic problems."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from rgbmanip_amd import _lib
from gpu_util import to_channels_last, from_channels_last, host_f32, empty_out, bx3_round
from test_gpu_kernels import _sweep_case
lib = _lib.load()
D, H, W = (int(a) for a in sys.argv[1:4])
B, V = 2, 4
g = torch.Generator().manual_seed(7)
w = torch.randn(8, 32, 3, 3, 3, generator=g) / np.sqrt(32 * 27)
scale = torch.rand(8, generator=g) + 0.5
shift = torch.randn(8, generator=g) * 0.1
wa, wp = host_f32(w); sa, sp = host_f32(scale); ha, hp = host_f32(shift)
feat, P, dep = _sweep_case(B, D, H, W, seed=11)
fd = to_channels_last(feat, _lib.F32)
Pd, dd = P.cuda(), dep.cuda()
hom = torch.empty(V * 12, dtype=torch.float32, device="cuda")
vol = torch.empty(V, D, H, W, 32, dtype=torch.float32, device="cuda")
_lib.check(lib.rgbm_build_volume(_lib.F32, _lib.ptr(fd), _lib.ptr(Pd), _lib.ptr(dd), _lib.ptr(hom), _lib.ptr(vol), V, B, D, H, W, _lib.stream_ptr()))
tot_bad = 0
for rep in range(int(sys.argv[4]) if len(sys.argv) > 4 else 3):
    out = empty_out((V, D, H, W, 8), _lib.BF16X3)
    _lib.check(lib.rgbm_conv0_sweep_dt(_lib.BF16X3, _lib.ptr(fd), _lib.ptr(Pd), _lib.ptr(dd), _lib.ptr(hom), wp, sp, hp, _lib.ptr(out), V, B, D, H, W, _lib.stream_ptr()))
    torch.cuda.synchronize()
    y = from_channels_last(out)                                  # [V,8,D,H,W]
    ref = vol.cpu()[..., :8].permute(0, 4, 1, 2, 3)
    e = (y - ref).abs() / ref.abs().max()
    bad = e > 1e-4
    print(f"rep {rep}: max err {float(e.max()):.3e}  bad voxels {int(bad.any(dim=1).sum())} of {V*D*H*W}")
    if bad.any():
        idx = bad.any(dim=1).nonzero()
        vs, ds, hs, ws = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
        print("   views", sorted(set(vs.tolist())), "planes", sorted(set(ds.tolist()))[:12], "rows%12", sorted(set((hs % 12).tolist())), "cols%16", sorted(set((ws % 16).tolist())))
        i = idx[0]
        print("   sample got", y[i[0], :, i[1], i[2], i[3]].numpy().round(4), "\n   sample ref", ref[i[0], :, i[1], i[2], i[3]].numpy().round(4))
        # does the bad voxel equal another plane's value?
        for q in range(D):
            if torch.allclose(y[i[0], :, i[1], i[2], i[3]], bx3_round(ref[i[0], :, q, i[2], i[3]]), atol=1e-4):
                print("   == plane", q, "of the same pixel")
