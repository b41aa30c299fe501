"""Helpers shared by the -m gpu parity tests (device layout conversion, error metrics)."""
import ctypes as C

import numpy as np
import torch

from rgbmanip_amd import _lib

# BF16X3 tensors are raw 4-byte slots (int32): 16-byte chunks of 4 channels {hi01, hi23, lo01, lo23} (include/rgbm.h)
TORCH_DT = {_lib.F32: torch.float32, _lib.BF16: torch.bfloat16, _lib.F16: torch.float16, _lib.BF16X3: torch.int32}


def bx3_round(x):
    """fp32 -> the value a BF16X3 slot holds: bf16(x) + bf16(x - bf16(x))."""
    hi = x.float().bfloat16().float()
    return hi + (x.float() - hi).bfloat16().float()


def bx3_pack(x):
    """[..., C] fp32 (C % 4 == 0) -> int32 tensor of the same shape in the split-pair chunk layout."""
    x = x.float().contiguous()
    assert x.shape[-1] % 4 == 0
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    h = hi.view(torch.int16).to(torch.int32).bitwise_and(0xffff).reshape(*x.shape[:-1], -1, 4)
    l = lo.view(torch.int16).to(torch.int32).bitwise_and(0xffff).reshape(*x.shape[:-1], -1, 4)
    w = torch.stack([h[..., 0] | (h[..., 1] << 16), h[..., 2] | (h[..., 3] << 16),
                     l[..., 0] | (l[..., 1] << 16), l[..., 2] | (l[..., 3] << 16)], dim=-1)
    return w.reshape(x.shape).to(torch.int32)


def bx3_unpack(w):
    """inverse of bx3_pack: int32 [..., C] -> fp32 values hi + lo."""
    w = w.to(torch.int64).bitwise_and(0xffffffff).reshape(*w.shape[:-1], -1, 4)

    def f32_bits(u):    # 32-bit pattern held in an int64 -> fp32
        return (u - ((u >> 31) << 32)).to(torch.int32).view(torch.float32)

    def halves(d):      # dword -> (low half, high half), each a bf16 bit pattern widened to fp32
        return f32_bits((d & 0xffff) << 16), f32_bits(d & 0xffff0000)
    a0, a1 = halves(w[..., 0])
    a2, a3 = halves(w[..., 1])
    b0, b1 = halves(w[..., 2])
    b2, b3 = halves(w[..., 3])
    out = torch.stack([a0 + b0, a1 + b1, a2 + b2, a3 + b3], dim=-1)
    return out.reshape(*out.shape[:-2], -1)


def quantise(x, dtype):
    """fp32 -> the fp32 value the storage type keeps (identity for F32)."""
    if dtype == _lib.F32:
        return x
    if dtype == _lib.BF16X3:
        return bx3_round(x)
    return x.to(TORCH_DT[dtype]).float()


def empty_out(shape, dtype, fill=float("nan")):
    """device output tensor in the storage type; float types are pre-filled (NaN by default) to catch unwritten elements."""
    if dtype == _lib.BF16X3:
        return torch.full(tuple(shape), 0x7fc07fc0, dtype=torch.int32, device="cuda")      # NaN hi halves
    return torch.full(tuple(shape), fill, dtype=TORCH_DT[dtype], device="cuda")


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
    if dtype == _lib.BF16X3:
        return bx3_pack(y).to("cuda").contiguous()
    return y.to("cuda", TORCH_DT[dtype]).contiguous()


def from_channels_last(y, C_=None):
    """device [N,*spatial,Cpad] -> cpu fp32 [N,C,*spatial]."""
    y = bx3_unpack(y.cpu()) if y.dtype == torch.int32 else y.float().cpu()
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
    E = 4 if dtype in (_lib.F32, _lib.BF16X3) else 8
    cin_pad = cin_pad or (Cin + E - 1) // E * E
    cout_pad = cout_pad or (Cout + 3) // 4 * 4
    xd = to_channels_last(x, dtype, cin_pad)
    rd = to_channels_last(res, dtype, cout_pad) if res is not None else None
    out = empty_out((N, Do, Ho, Wo, cout_pad), dtype)
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
