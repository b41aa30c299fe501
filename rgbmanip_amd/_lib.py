"""ctypes binding of librgbm_hip.so (the C ABI declared in include/rgbm.h).

The library is built in-tree by `rgbmanip_amd/csrc/build.sh` (hipcc, gfx950).  There is no CPU
fallback: if the shared object is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RGBM_HIP_LIB selects another build of the same library (kernel ablation builds made by tools/abl_sweep.sh)
LIB_PATH = os.environ.get("RGBM_HIP_LIB") or os.path.join(_HERE, "librgbm_hip.so")

F32, BF16, F16, BF16X3 = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_PRELU, ACT_TANH = 0, 1, 2, 3
RES_NONE, RES_PRE_ACT, RES_POST_ACT = 0, 1, 2


# rows of rgbm_prof_stop (include/rgbm.h): (kernel name as rocprofv3 prints it, arithmetic dtype)
PROF_ROWS = 42          # == RGBM_PROF_ROWS (include/rgbm.h); load() checks it against rgbm_prof_rows()
PROF_KERNELS = [
    ("conv_igemm_glds_kernel<float, 16, 256>", "fp32"), ("conv_igemm_glds_kernel<float, 32, 256>", "fp32"),
    ("conv_igemm_glds_kernel<float, 64, 256>", "fp32"), ("conv_igemm_glds_kernel<float, 128, 128>", "fp32"),
    ("conv_igemm_glds_kernel<unsigned short, 16, 256>", "bf16"), ("conv_igemm_glds_kernel<unsigned short, 32, 256>", "bf16"),
    ("conv_igemm_glds_kernel<unsigned short, 64, 256>", "bf16"), ("conv_igemm_glds_kernel<unsigned short, 128, 128>", "bf16"),
    ("conv3d_tile_kernel<float, ...> (conv1..conv11)", "fp32"), ("conv3d_tile_kernel<unsigned short, ...> (conv1..conv11)", "bf16"),
    ("conv3d_tile_kernel<float, 32, 16, 4, 8, 8, 1, false, true> (conv0 + fused plane sweep)", "fp32"),
    ("conv3d_tile_kernel<unsigned short, 32, 16, 4, 8, 8, 1, false, true> (conv0 + fused plane sweep)", "bf16"),
    ("conv_igemm_ws_kernel<float, ...>", "fp32"), ("conv_igemm_ws_kernel<unsigned short, false, false> (128 channels x 256 pixels)", "bf16"),    # + conv_igemm_v3_kernel for non-uniform taps
    ("conv0_sweep_persistent_kernel<unsigned short> (conv0 + fused plane sweep of bf16 nets; sweep_f16 = 0 / fp16 nets: conv0_sweep_kernel)", "bf16"), ("conv_igemm_ws64_kernel", "bf16"),
    ("conv3d_tile_kernel<unsigned short, 32, 16, 6, 8, 8, 1, false, false> (conv0 on a materialised volume)", "bf16"),
    ("conv3d_tile_kernel<unsigned short, 8, 16, 2, 8, 8, 2, false, false> (conv1)", "bf16"),
    ("conv3d_tile_kernel<unsigned short, 16, 16, 4, 8, 8, 1, false, false> (conv2)", "bf16"),
    ("conv3d_tile_kernel<unsigned short, 16, 32, 2, 8, 8, 2, false, false> (conv3)", "bf16"),
    ("conv3d_tile_kernel<unsigned short, 32, 32, 2, 8, 8, 1, false, false> (conv4)", "bf16"),
    ("conv3d_tile_kernel<unsigned short, 32, 64, 1, 8, 8, 2, false, false> (conv5)", "bf16"),
    ("conv3d_tile_kernel<unsigned short, 64, 64, 1, 8, 8, 1, false, false> (conv6)", "bf16"),
    ("conv3d_tile_kernel<unsigned short, 64, 32, 1, 8, 8, 1, true, false> (conv7)", "bf16"),
    ("conv3d_tile_kernel<unsigned short, 32, 16, 2, 8, 8, 1, true, false> (conv9)", "bf16"),
    ("conv3d_tile_kernel<unsigned short, 16, 16, 4, 8, 8, 1, true, false> (conv11)", "bf16"),
] + [("conv_igemm_glds_kernel<bx3_t, ...> (all channel tiles)", "bf16x3"), ("conv3d_tile_kernel<bx3_t, ...> (conv1..conv11)", "bf16x3"),
     ("conv0 + fused plane sweep <bx3_t>", "bf16x3"), ("conv_igemm_ws_kernel<rgbm::bx3_t, false, false> (128 channels x 256 pixels)", "bf16x3"),
     ("unused (conv_igemm_w256_kernel of rounds 2-5)", "bf16"),
     ("conv_igemm_m32_kernel<unsigned short, 256, 256> (256 channels x 256 pixels, 32x32x16 MFMAs; gemm_kernel = 0: conv_igemm_ws_kernel<unsigned short, true, false>, 256 x 128)", "bf16"),
     ("unused (row-halo ws tile of rounds 2-5)", "bf16"),
     ("conv_igemm_m32_kernel<rgbm::bx3_t, 256, 256> (256 channels x 256 pixels, 32x32x16 MFMAs; gemm_kernel = 0: conv_igemm_ws_kernel<rgbm::bx3_t, true, false>, 256 x 128)", "bf16x3"),
     ("conv_igemm_ws_kernel<rgbm::bx3_t, false, false, true> (64 channels x 256 pixels, four multiply waves)", "bf16x3"),
     ("conv_igemm_ws_kernel<unsigned short, false, false, true> (64 channels x 256 pixels, four multiply waves)", "bf16"),
     ("conv_igemm_ws_kernel<float, false, false, true> (64 channels x 256 pixels, four multiply waves)", "fp32"),
     ("upconv_combine_kernel<16-bit> (PSPUpsample tap combination)", "bf16"), ("upconv_combine_kernel<4-byte> (PSPUpsample tap combination)", "bf16x3"),
     ("upconv_final_kernel (up_3 + final in one kernel; either storage width)", "bf16"),
     ("conv_igemm_m32_kernel<unsigned short, 128, 64 / 128 / 256> (128-pixel tiles: tail and small-batch launches of the 256-channel GEMM)", "bf16"),
     ("conv_igemm_m32_kernel<rgbm::bx3_t, 128, 64 / 128 / 256> (128-pixel tiles: tail and small-batch launches of the 256-channel GEMM)", "bf16x3")]
assert len(PROF_KERNELS) == PROF_ROWS


class RgbmError(RuntimeError):
    pass


class WeightDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int), ("shape", C.POINTER(C.c_int64))]


class PolicyLayout(C.Structure):
    _fields_ = [("dims", C.c_int * 5), ("log_std", C.c_int), ("w", (C.c_int * 4) * 2), ("b", (C.c_int * 4) * 2), ("total", C.c_int)]


class AdaposeOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("view1_nocs", "view2_nocs", "view1_depth", "view2_depth", "view1_r", "view2_r",
                                           "view1_t", "view2_t", "view1_s", "view2_s")]


class ControlRewardArgs(C.Structure):            # rgbm_control_reward_args
    _fields_ = [(n, C.c_void_p) for n in ("action", "cam_pose", "target", "move_success", "bbox", "avail", "gt_bbox", "pred_bbox",
                                           "pose_cur", "pose_prev", "robot_pose", "success", "reward", "terms")] + [
        ("coef", C.c_double * 14), ("proper_pos", C.c_double * 3), ("precision2", C.c_double),
        ("N", C.c_int), ("T", C.c_int), ("lda", C.c_int), ("pots", C.c_int), ("first", C.c_int), ("pad_", C.c_int)]


class SynthScene(C.Structure):                   # rgbm_synth_scene
    _fields_ = [("cam_pose", C.c_void_p), ("robot_pose", C.c_void_p), ("box", C.c_void_p),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("env0", C.c_int)]


_vp, _i, _f, _d, _sz, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_int64

# symbol -> (restype, argtypes); must list every function declared in include/rgbm.h
SIGNATURES = {
    "rgbm_version": (_i, []),
    "rgbm_last_error": (C.c_char_p, []),
    "rgbm_adapose_create": (_i, [C.POINTER(_vp), _i, C.POINTER(WeightDesc), _i, _i, _i]),
    "rgbm_adapose_destroy": (_i, [_vp]),
    "rgbm_adapose_set_chunk": (_i, [_vp, _i]),
    "rgbm_adapose_set_option": (_i, [_vp, C.c_char_p, _i]),
    "rgbm_adapose_workspace_bytes": (_i, [_vp, _i, C.POINTER(_sz)]),
    "rgbm_adapose_forward": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(AdaposeOut), _vp]),
    "rgbm_adapose_forward_ex": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(AdaposeOut), _i, _vp]),
    "rgbm_adapose_forward_graph": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(AdaposeOut), _vp, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32)]),
    "rgbm_adapose_graph_clear": (_i, [_vp]),
    "rgbm_adapose_fetch": (_i, [_vp, _i, _vp, C.c_char_p, _vp, _sz, C.POINTER(_sz), _vp]),
    "rgbm_adapose_postprocess": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rgbm_adapose_postprocess_scratch_bytes": (_i, [_i, C.POINTER(_sz)]),
    "rgbm_adapose_postprocess_ws": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rgbm_adapose_postprocess_ransac": (_i, [_i, _i, _i, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rgbm_adapose_postprocess_pnp": (_i, [_i, _i, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rgbm_gae": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "rgbm_adv_normalise": (_i, [_i64, _vp, _vp, _d, _vp]),
    "rgbm_policy_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rgbm_ppo_partial_floats": (_i, [_vp, _i, C.POINTER(_sz)]),
    "rgbm_ppo_minibatch_fwd_bwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp]),
    "rgbm_ppo_clip_adam": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _vp]),
    "rgbm_conv_nd": (_i, [_i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp,
                          _vp, _i, _i, _f, _vp, _vp]),
    "rgbm_conv3d_tile": (_i, [_i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rgbm_upsample_conv3x3": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _f, _vp, _vp, _vp]),
    "rgbm_upsample_conv3x3_final": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _i, _vp]),
    "rgbm_stem": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "rgbm_maxpool3x3s2": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rgbm_resize_bilinear_ac": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rgbm_adaptive_avgpool": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rgbm_build_volume": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rgbm_conv0_sweep": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rgbm_conv0_sweep_dt": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rgbm_conv0_sweep_f16feat": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rgbm_prepare_inputs": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rgbm_prepare_inputs_indexed": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rgbm_prepare_inputs_ex": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rgbm_projection": (_i, [_vp, _vp, _vp, _i, _vp]),
    "rgbm_mask_extent": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "rgbm_lookat_quat": (_i, [_vp, _i, _i, _vp, _vp]),
    "rgbm_control_action_to_pose": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), _i, _vp, _vp]),
    "rgbm_control_reward": (_i, [C.POINTER(ControlRewardArgs), _vp]),
    "rgbm_control_grasp_frame": (_i, [_vp, _i, _vp, _vp, _vp]),
    "rgbm_synth_camera": (_i, [C.POINTER(SynthScene), _vp, _vp, _vp, _vp]),
    "rgbm_synth_render": (_i, [C.POINTER(SynthScene), _vp, _vp, _vp, _vp]),
    "rgbm_debug_flags": (_i, [_i]),
    "rgbm_set_tuning": (_i, [C.c_char_p, _i64]),
    "rgbm_prof_rows": (_i, []),
    "rgbm_microbench_mfma_scratch_floats": (_i, [C.POINTER(C.c_int)]),
    "rgbm_microbench_mfma": (_i, [_vp, _i, _i, C.POINTER(C.c_double), _vp]),
    "rgbm_microbench_copy": (_i, [_vp, _vp, C.c_size_t, _vp]),
    "rgbm_prof_start": (_i, []),
    "rgbm_prof_select": (_i, [_i]),
    "rgbm_prof_stop": (_i, [C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RgbmError(f"{LIB_PATH} not found: build it with rgbmanip_amd/csrc/build.sh "
                        f"(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.rgbm_prof_rows() != PROF_ROWS:
        raise RgbmError(f"{LIB_PATH}: rgbm_prof_rows() = {lib.rgbm_prof_rows()}, this binding expects {PROF_ROWS} (stale build?)")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().rgbm_last_error()
        raise RgbmError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def stream_ptr(stream=None):
    """Raw hipStream_t of a torch.cuda.Stream (default: current stream)."""
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
