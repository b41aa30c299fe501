"""Helpers shared by the -m gpu parity tests (device layout conversion, error metrics)."""
import ctypes as C

import numpy as np
import torch

from rgbmanip_amd import _lib

TORCH_DT = {_lib.F32: torch.float32, _lib.BF16: torch.bfloat16, _lib.F16: torch.float16}


def rel_err(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def to_channels_last(x, dtype, cpad=None):
    """[N,C,*spatial] fp32 (cpu) -> device [N,*spatial,Cpad] in dtype."""
    nd = x.dim()
    perm = [0] + list(range(2, nd)) + [1]
    y = x.permute(*perm).contiguous()
    C_ = y.shape[-1]
    if cpad and cpad > C_:
        y = torch.nn.functional.pad(y, (0, cpad - C_))
    return y.to("cuda", TORCH_DT[dtype]).contiguous()


def from_channels_last(y, C_=None):
    """device [N,*spatial,Cpad] -> cpu fp32 [N,C,*spatial]."""
    y = y.float().cpu()
    if C_ is not None:
        y = y[..., :C_]
    nd = y.dim()
    perm = [0, nd - 1] + list(range(1, nd - 1))
    return y.permute(*perm).contiguous()


def host_f32(a):
    if a is None:
        return None, C.c_void_p(0)
    arr = np.ascontiguousarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float32)
    return arr, C.c_void_p(arr.ctypes.data)


def conv_nd(dtype, x, w, *, stride=1, stride_d=None, pad=0, pad_d=None, dil=1, transposed=False, bias=None, bn_scale=None,
            bn_shift=None, res=None, res_mode=0, act=0, slope=0.0, cin_pad=None, cout_pad=None):
    """x [N,C,D,H,W] or [N,C,H,W] cpu fp32; w torch layout.  Returns cpu fp32 [N,Cout,...] via the HIP conv."""
    lib = _lib.load()
    is2d = x.dim() == 4
    if is2d:
        x = x.unsqueeze(2)
        w = w.unsqueeze(2)
        if res is not None:
            res = res.unsqueeze(2)
    N, Cin, D, H, W = x.shape
    if transposed:
        Cout = w.shape[1]
        KD, KH, KW = w.shape[2:]
        Do, Ho, Wo = 2 * D, 2 * H, 2 * W
    else:
        Cout = w.shape[0]
        KD, KH, KW = w.shape[2:]
        sd = stride_d if stride_d is not None else (1 if is2d else stride)
        pd = pad_d if pad_d is not None else (0 if is2d else pad)
        Do = (D + 2 * pd - (KD - 1) - 1) // sd + 1
        Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    E = 4 if dtype == _lib.F32 else 8
    cin_pad = cin_pad or (Cin + E - 1) // E * E
    cout_pad = cout_pad or (Cout + 3) // 4 * 4
    xd = to_channels_last(x, dtype, cin_pad)
    rd = to_channels_last(res, dtype, cout_pad) if res is not None else None
    out = torch.full((N, Do, Ho, Wo, cout_pad), float("nan"), dtype=TORCH_DT[dtype], device="cuda")
    wa, wp = host_f32(w)
    ba, bp = host_f32(bias)
    sa, sp = host_f32(bn_scale)
    ha, hp = host_f32(bn_shift)
    sd = stride_d if stride_d is not None else (1 if is2d else stride)
    pd = pad_d if pad_d is not None else (0 if is2d else pad)
    torch.cuda.synchronize()
    rc = lib.rgbm_conv_nd(dtype, _lib.ptr(xd), N, D, H, W, Cin, cin_pad, wp, Cout, cout_pad, KD, KH, KW,
                          2 if transposed else sd, 2 if transposed else stride, 1 if transposed else pd,
                          1 if transposed else pad, dil, int(transposed), bp, sp, hp, _lib.ptr(rd), res_mode, act, slope,
                          _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "rgbm_conv_nd")
    torch.cuda.synchronize()
    y = from_channels_last(out, Cout)
    return y.squeeze(2) if is2d else y
