"""Host-side wrapper of the HIP AdaPose network (the C ABI in include/rgbm.h).

`AdaPoseNet` mirrors the call surface of the reference module `StereoPoseNet_with_depth`
(`/root/reference/models/pose_estimator/AdaPose/lib/network_v5.py:301-519`): it is built from a
state_dict with the reference's key names (optionally `module.`-prefixed, as saved by the
reference's DataParallel wrapper, `interface_v5.py:48,55-56`) and called with
`(view1_img, view1_choose, view2_img, view2_choose, view1_proj, view2_proj, depth_values)`,
returning the same 10-entry dict of tensors.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_DTYPES = {"fp32": _lib.F32, "f32": _lib.F32, "float32": _lib.F32, "bf16": _lib.BF16, "bfloat16": _lib.BF16,
           "fp16": _lib.F16, "f16": _lib.F16, "float16": _lib.F16, "bf16x3": _lib.BF16X3}


class AdaPoseNet:
    def __init__(self, state_dict, dtype: str = "fp32", device: int = 0, max_chunk_views: int | None = None,
                 cost_impl: int | None = None, sparse_tail: int | None = None, options: dict | None = None,
                 norm_mode: int | str = 0, poison_workspace: bool = False, graph: bool = False, graph_max_batch: int = 32,
                 split_streams: bool | int = False, split_min_batch: int = 128):
        self.lib = _lib.load()
        # graph: forwards of at most `graph_max_batch` poses are replayed from a hipGraph captured per batch size
        # (rgbm_adapose_forward_graph): static input / output / workspace buffers per batch size, one hipGraphLaunch instead of ~150
        # launches — the small-batch deployment path (interface_v5.py:213-227 calls the network per env; cfg/task/open_cabinet.yaml
        # ships num_envs: 8).  Larger batches run eagerly: they are bound by the kernels, not by their launches.
        self.graph = bool(graph)
        self.graph_max_batch = int(graph_max_batch)
        self._static = {}
        self._gstream = None
        self.last_graph_nodes = 0
        # debug: fill the whole workspace with 0xFF bytes (NaN in every storage type, -1 in index lists) in front of EVERY forward, so
        # that a kernel reading a tile the sparse cost regularisation skipped — or anything else a forward did not write itself —
        # cannot find a previous run's (correct) values there (tests/test_gpu_at_batch.py, bench.py's at-batch check)
        self.poison_workspace = bool(poison_workspace)
        # split_streams (opt-in): a forward of at least `split_min_batch` (even) poses runs as two half batches on two side streams, each with
        # its own workspace: the tail of every launch of one half (partial last round of workgroups, the drain of a persistent kernel) is
        # filled by the other half's kernels (-0.7 % bf16, -2.6 % bf16x3 at batch 256).  Same kernels, same per-pose arithmetic:
        # outputs bit-identical to the one-stream forward (tests/test_gpu_at_batch.py) — which needs a library without packed fp32
        # instructions (build.sh).  Intermediate taps (fetch) refer to the one-stream workspace and are refused after a split forward.
        # split_streams = n > 2 (round 6): n equal parts on n side streams.
        self.split_streams = bool(split_streams)
        self.split_parts = 2 if split_streams is True else max(int(split_streams), 2)
        self.split_min_batch = int(split_min_batch)
        self._split = None                           # ([side streams], [workspaces], batch)
        self._last_split = False
        if not torch.cuda.is_available():
            raise _lib.RgbmError("AdaPoseNet needs a HIP device (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device("cuda", device)
        self.dtype_name = dtype
        self.dtype = _DTYPES[dtype]
        keep, descs = [], []
        for k, v in state_dict.items():
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if a.dtype != np.float32:
                if a.dtype.kind != "f":
                    continue                      # num_batches_tracked (int64)
                a = a.astype(np.float32)
            a = np.ascontiguousarray(a)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape) if a.ndim else (C.c_int64 * 1)(1)
            name = k.encode()
            keep.append((a, shape, name))
            descs.append(_lib.WeightDesc(name, a.ctypes.data, a.ndim, shape))
        arr = (_lib.WeightDesc * len(descs))(*descs)
        self._h = C.c_void_p()
        # norm_mode: 0 / "eval" = BatchNorm3d with running statistics (default); 1 / "per_sample" = the reference's as-shipped
        # train-mode statistics at batch 1 (every view normalised with its own volume's mean / variance)
        self.norm_mode = {"eval": 0, "per_sample": 1}.get(norm_mode, norm_mode)
        _lib.check(self.lib.rgbm_adapose_create(C.byref(self._h), device, arr, len(descs), self.dtype, int(self.norm_mode)),
                   "rgbm_adapose_create")
        if max_chunk_views:
            _lib.check(self.lib.rgbm_adapose_set_chunk(self._h, int(max_chunk_views)), "rgbm_adapose_set_chunk")
        if cost_impl is not None:
            _lib.check(self.lib.rgbm_adapose_set_option(self._h, b"cost_impl", int(cost_impl)), "rgbm_adapose_set_option")
        if sparse_tail is not None:
            _lib.check(self.lib.rgbm_adapose_set_option(self._h, b"sparse_tail", int(sparse_tail)), "rgbm_adapose_set_option")
        self.options = {}                                 # what was set through this object (a sharing estimator reads view2_heads back)
        for key, val in (options or {}).items():          # any rgbm_adapose_set_option key (include/rgbm.h), e.g. fuse_final
            _lib.check(self.lib.rgbm_adapose_set_option(self._h, key.encode(), int(val)), "rgbm_adapose_set_option")
            self.options[key] = int(val)
        self._ws = None
        self._ws_B = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.rgbm_adapose_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def workspace_bytes(self, B: int) -> int:
        n = C.c_size_t()
        _lib.check(self.lib.rgbm_adapose_workspace_bytes(self._h, B, C.byref(n)), "rgbm_adapose_workspace_bytes")
        return n.value

    def _workspace(self, B: int):
        if self._ws is None or self._ws_B != B:
            self._ws = None
            self._ws = torch.empty(self.workspace_bytes(B) + 256, dtype=torch.uint8, device=self.device)
            self._ws_B = B
        off = (-self._ws.data_ptr()) % 256
        return self._ws.data_ptr() + off, self._ws.numel() - off

    def _prep(self, t, dtype):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _forward_graph(self, args):
        """Replay (first call per batch size: capture) the forward on static buffers; returns fresh output tensors."""
        def as_t(x):
            return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
        args = [as_t(a) for a in args]
        B = int(args[0].shape[0])
        st = self._static.get(B)
        if st is None:
            f32 = dict(dtype=torch.float32, device=self.device)
            i32 = dict(dtype=torch.int32, device=self.device)
            ins = [torch.empty(B, 3, 224, 224, **f32), torch.empty(B, 1024, **i32), torch.empty(B, 3, 224, 224, **f32), torch.empty(B, 1024, **i32),
                   torch.empty(B, 4, 4, **f32), torch.empty(B, 4, 4, **f32), torch.empty(B, 24, **f32)]
            out = {"view1_nocs": torch.empty(B, 1024, 3, **f32), "view2_nocs": torch.empty(B, 1024, 3, **f32),
                   "view1_depth": torch.empty(B, 1024, **f32), "view2_depth": torch.empty(B, 1024, **f32),
                   "view1_r": torch.empty(B, 3, 3, **f32), "view2_r": torch.empty(B, 3, 3, **f32),
                   "view1_t": torch.empty(B, 3, **f32), "view2_t": torch.empty(B, 3, **f32),
                   "view1_s": torch.empty(B, 3, **f32), "view2_s": torch.empty(B, 3, **f32)}
            ws = torch.empty(self.workspace_bytes(B) + 256, dtype=torch.uint8, device=self.device)
            st = self._static[B] = (ins, out, ws)
        ins, out, ws = st
        for dst, src in zip(ins, args):
            assert tuple(src.shape) == tuple(dst.shape), (tuple(src.shape), tuple(dst.shape))
            dst.copy_(src, non_blocking=True)              # converts dtype / uploads as needed, on the caller's stream
        if self._gstream is None:
            self._gstream = torch.cuda.Stream(device=self.device)      # capture is not allowed on the default stream
        cur = torch.cuda.current_stream(self.device)
        gs = self._gstream
        gs.wait_stream(cur)
        off = (-ws.data_ptr()) % 256
        o = _lib.AdaposeOut(*[out[n].data_ptr() for n, _ in _lib.AdaposeOut._fields_])
        nodes, cap = C.c_int32(), C.c_int32()
        if self.poison_workspace:
            with torch.cuda.stream(gs):
                ws.fill_(0xFF)
        # (img1, choose1, img2, choose2, P1, P2, depths) -> the C ABI's (img1, img2, choose1, choose2, P1, P2, depths)
        _lib.check(self.lib.rgbm_adapose_forward_graph(self._h, B, _lib.ptr(ins[0]), _lib.ptr(ins[2]), _lib.ptr(ins[1]), _lib.ptr(ins[3]),
                                                       _lib.ptr(ins[4]), _lib.ptr(ins[5]), _lib.ptr(ins[6]), C.c_void_p(ws.data_ptr() + off),
                                                       ws.numel() - off, C.byref(o), C.c_void_p(gs.cuda_stream), C.byref(nodes), C.byref(cap)),
                   "rgbm_adapose_forward_graph")
        self.last_graph_nodes = nodes.value
        cur.wait_stream(gs)
        return {k: v.clone() for k, v in out.items()}      # the static outputs are overwritten by the next replay

    def forward(self, view1_img, view1_choose, view2_img, view2_choose, view1_proj, view2_proj, depth_values,
                stop_after: int = 0, stream=None):
        if self.graph and stop_after == 0 and stream is None and len(view1_img) <= self.graph_max_batch:
            self._last_graph = True
            return self._forward_graph((view1_img, view1_choose, view2_img, view2_choose, view1_proj, view2_proj, depth_values))
        self._last_graph = False
        img1 = self._prep(view1_img, torch.float32)
        img2 = self._prep(view2_img, torch.float32)
        ch1 = self._prep(view1_choose, torch.int32)
        ch2 = self._prep(view2_choose, torch.int32)
        P1 = self._prep(view1_proj, torch.float32)
        P2 = self._prep(view2_proj, torch.float32)
        dep = self._prep(depth_values, torch.float32)
        B = img1.shape[0]
        assert img1.shape == (B, 3, 224, 224) and img2.shape == img1.shape, img1.shape
        assert ch1.shape == (B, 1024) and ch2.shape == ch1.shape
        assert P1.shape == (B, 4, 4) and P2.shape == (B, 4, 4) and dep.shape == (B, 24)
        f32 = dict(dtype=torch.float32, device=self.device)
        out = {
            "view1_nocs": torch.empty(B, 1024, 3, **f32), "view2_nocs": torch.empty(B, 1024, 3, **f32),
            "view1_depth": torch.empty(B, 1024, **f32), "view2_depth": torch.empty(B, 1024, **f32),
            "view1_r": torch.empty(B, 3, 3, **f32), "view2_r": torch.empty(B, 3, 3, **f32),
            "view1_t": torch.empty(B, 3, **f32), "view2_t": torch.empty(B, 3, **f32),
            "view1_s": torch.empty(B, 3, **f32), "view2_s": torch.empty(B, 3, **f32),
        }
        if self.split_streams and stop_after == 0 and B >= self.split_min_batch and B % self.split_parts == 0:
            self._forward_split(B, (img1, img2, ch1, ch2, P1, P2, dep), out, stream)
            return out
        self._last_split = False
        o = _lib.AdaposeOut(*[out[n].data_ptr() for n, _ in _lib.AdaposeOut._fields_])
        ws_ptr, ws_bytes = self._workspace(B)
        if self.poison_workspace:
            with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
                self._ws.fill_(0xFF)
        _lib.check(self.lib.rgbm_adapose_forward_ex(self._h, B, _lib.ptr(img1), _lib.ptr(img2), _lib.ptr(ch1), _lib.ptr(ch2),
                                                    _lib.ptr(P1), _lib.ptr(P2), _lib.ptr(dep), C.c_void_p(ws_ptr), ws_bytes,
                                                    C.byref(o), stop_after, _lib.stream_ptr(stream)), "rgbm_adapose_forward")
        self._last = (img1, img2, ch1, ch2, P1, P2, dep)     # keep inputs alive until the stream has consumed them
        return out

    def _forward_split(self, B, args, out, stream):
        n = self.split_parts
        h = B // n
        if self._split is None or self._split[2] != B:
            self._split = None
            need = self.workspace_bytes(h) + 256
            self._split = ([torch.cuda.Stream(device=self.device) for _ in range(n)],
                           [torch.empty(need, dtype=torch.uint8, device=self.device) for _ in range(n)], B)
        side, wss, _ = self._split
        cur = stream if stream is not None else torch.cuda.current_stream(self.device)
        fork = torch.cuda.Event()
        fork.record(cur)
        for i in range(n):
            si, ws = side[i], wss[i]
            si.wait_event(fork)                       # inputs (and the previous use of the outputs) are ordered on `cur`
            off = (-ws.data_ptr()) % 256
            if self.poison_workspace:
                with torch.cuda.stream(si):
                    ws.fill_(0xFF)
            sl = slice(i * h, (i + 1) * h)
            o = _lib.AdaposeOut(*[out[n][sl].data_ptr() for n, _ in _lib.AdaposeOut._fields_])
            a = [t[sl] for t in args]                  # leading-dimension slices of contiguous tensors: contiguous views
            _lib.check(self.lib.rgbm_adapose_forward_ex(self._h, h, *[_lib.ptr(t) for t in a], C.c_void_p(ws.data_ptr() + off),
                                                        ws.numel() - off, C.byref(o), 0, _lib.stream_ptr(si)), "rgbm_adapose_forward")
        for si in side:
            join = torch.cuda.Event()
            join.record(si)
            cur.wait_event(join)
        self._last = args
        self._last_split = True

    __call__ = forward

    def fetch(self, B: int, name: str, max_elems: int) -> torch.Tensor:
        """Debug/test access to a named intermediate of the last forward (fp32, flat)."""
        if self._last_split:
            raise _lib.RgbmError("fetch: the last forward ran as two half batches (split_streams); run it with split_streams=False for taps")
        if getattr(self, "_last_graph", False):
            raise _lib.RgbmError("fetch: the last forward replayed a captured graph (its intermediates live in the graph's own workspace); "
                                 "run it with graph=False for taps")
        buf = torch.empty(max_elems, dtype=torch.float32, device=self.device)
        n = C.c_size_t()
        ws_ptr, _ = self._workspace(B)
        _lib.check(self.lib.rgbm_adapose_fetch(self._h, B, C.c_void_p(ws_ptr), name.encode(), _lib.ptr(buf), max_elems,
                                               C.byref(n), _lib.stream_ptr()), "rgbm_adapose_fetch")
        return buf[: n.value]


def postprocess(view1_nocs, view1_depth, view1_r, view1_choose, K_crop, E1, img_size: int = 224, stream=None):
    """Device post-processing: returns (bbox_world [B,8,3] f64, ts [B,4] f64, valid [B] i32) CUDA tensors.

    Mirrors the tail of `AdaPoseEstimator_v5.predict` (`interface_v5.py:318-321,354-374`)."""
    lib = _lib.load()
    dev = view1_nocs.device
    B, P = view1_depth.shape
    nocs = view1_nocs.to(torch.float32).contiguous()
    depth = view1_depth.to(torch.float32).contiguous()
    r = view1_r.to(torch.float32).contiguous()
    ch = torch.as_tensor(view1_choose).to(device=dev, dtype=torch.int32).contiguous()
    K = torch.as_tensor(K_crop).to(device=dev, dtype=torch.float64).contiguous()
    E = torch.as_tensor(E1).to(device=dev, dtype=torch.float64).contiguous()
    bbox = torch.empty(B, 8, 3, dtype=torch.float64, device=dev)
    ts = torch.empty(B, 4, dtype=torch.float64, device=dev)
    valid = torch.empty(B, dtype=torch.int32, device=dev)
    # small batches: the exact-median search is sliced over several workgroups per pose (needs device scratch; bit-identical)
    nb = C.c_size_t()
    _lib.check(lib.rgbm_adapose_postprocess_scratch_bytes(B, C.byref(nb)), "rgbm_adapose_postprocess_scratch_bytes")
    scratch = torch.empty(nb.value // 8, dtype=torch.int64, device=dev) if nb.value else None
    if scratch is not None and stream is not None:
        scratch.record_stream(stream)      # allocated on the current stream, used (and dropped on return) on `stream`: the caching allocator must not reuse it early
    _lib.check(lib.rgbm_adapose_postprocess_ws(B, P, img_size, _lib.ptr(nocs), _lib.ptr(depth), _lib.ptr(r), _lib.ptr(ch),
                                               _lib.ptr(K), _lib.ptr(E), _lib.ptr(bbox), _lib.ptr(ts), _lib.ptr(valid),
                                               _lib.ptr(scratch), nb.value, _lib.stream_ptr(stream)), "rgbm_adapose_postprocess_ws")
    return bbox, ts, valid


def postprocess_ransac(view1_nocs, view1_depth, view1_choose, K_crop, E1, img_size: int = 224, seed: int = 0, stream=None):
    """Device tail of `predict` for `direct_regression: False`, `use_depth: True` (`interface_v5.py:322-339, 348-374`,
    `lib/align.py:10-104`): returns (bbox_world [B,8,3] f64, srt [B,13] f64 = scale, R, t, valid [B] i32) CUDA tensors."""
    lib = _lib.load()
    dev = view1_nocs.device
    B, P = view1_depth.shape
    nocs = view1_nocs.to(torch.float32).contiguous()
    depth = view1_depth.to(torch.float32).contiguous()
    ch = torch.as_tensor(view1_choose).to(device=dev, dtype=torch.int32).contiguous()
    K = torch.as_tensor(K_crop).to(device=dev, dtype=torch.float64).contiguous()
    E = torch.as_tensor(E1).to(device=dev, dtype=torch.float64).contiguous()
    bbox = torch.empty(B, 8, 3, dtype=torch.float64, device=dev)
    srt = torch.empty(B, 13, dtype=torch.float64, device=dev)
    valid = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.check(lib.rgbm_adapose_postprocess_ransac(B, P, img_size, int(seed) & 0xFFFFFFFF, _lib.ptr(nocs), _lib.ptr(depth),
                                                   _lib.ptr(ch), _lib.ptr(K), _lib.ptr(E), _lib.ptr(bbox), _lib.ptr(srt),
                                                   _lib.ptr(valid), _lib.stream_ptr(stream)), "rgbm_adapose_postprocess_ransac")
    return bbox, srt, valid


def postprocess_pnp(view1_nocs, view1_pts2d, view2_nocs, view2_pts2d, K, E1, E2, seed: int = 0, stream=None):
    """Device tail of `predict` for `direct_regression: False`, `use_depth: False` (`interface_v5.py:340-346`, `lib/utils.py:121-195`,
    `lib/align.py:104-115`): returns (bbox_world [B,8,3] f64, srt [B,13] f64 = scale, R, t, info [B,4] i32 = matches / RANSAC ok /
    inliers / hypotheses examined, valid [B] i32) CUDA tensors.  pts2d: pixels of the chosen points in the ORIGINAL frame."""
    lib = _lib.load()
    dev = view1_nocs.device
    B, P = view1_nocs.shape[:2]
    f32 = lambda x: torch.as_tensor(x).to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
    f64 = lambda x: torch.as_tensor(x).to(device=dev, dtype=torch.float64).contiguous()  # noqa: E731
    n1, p1, n2, p2 = f32(view1_nocs), f32(view1_pts2d), f32(view2_nocs), f32(view2_pts2d)
    Kd, E1d, E2d = f64(K), f64(E1), f64(E2)
    bbox = torch.empty(B, 8, 3, dtype=torch.float64, device=dev)
    srt = torch.empty(B, 13, dtype=torch.float64, device=dev)
    info = torch.empty(B, 4, dtype=torch.int32, device=dev)
    valid = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.check(lib.rgbm_adapose_postprocess_pnp(B, P, int(seed) & 0xFFFFFFFF, _lib.ptr(n1), _lib.ptr(p1), _lib.ptr(n2), _lib.ptr(p2),
                                                _lib.ptr(Kd), _lib.ptr(E1d), _lib.ptr(E2d), _lib.ptr(bbox), _lib.ptr(srt), _lib.ptr(info),
                                                _lib.ptr(valid), _lib.stream_ptr(stream)), "rgbm_adapose_postprocess_pnp")
    return bbox, srt, info, valid


def prepare_inputs(rgb, mask, K, img_size: int = 224, n_pts: int = 1024, seed: int = 0, want_pts2d: bool = False, stream=None,
                   frame_map=None, frame0: int = 0):
    """Batched device-side `AdaPoseEstimator_v5.prepare_model_input` (`interface_v5.py:58-170`).

    rgb [N,H,W,3] float32 in [0,1], mask [N,H,W] (0/1), K [N,3,3]: torch CUDA tensors (or anything torch.as_tensor accepts).
    With `frame_map` [N] int32, rgb / mask are a pool [M,H,W,..] (e.g. a view queue) and frame f reads entry frame_map[f]
    (negative: no view -> valid 0) — no gather of the selected frames is needed; K stays [N,3,3].
    Returns dict(img [N,3,S,S] f32, choose [N,P] i32, Kcrop [N,3,3] f64, window [N,4] i32, valid [N] i32[, pts2d])."""
    lib = _lib.load()
    dev = rgb.device if isinstance(rgb, torch.Tensor) and rgb.is_cuda else torch.device("cuda", torch.cuda.current_device())
    rgb = torch.as_tensor(rgb).to(device=dev, dtype=torch.float32).contiguous()
    mask = torch.as_tensor(mask).to(device=dev)
    if mask.dtype != torch.uint8:                       # the kernels test for non-zero, so a uint8 mask is used as it is
        mask = (mask != 0).to(torch.uint8)
    mask = mask.contiguous()
    K = torch.as_tensor(K).to(device=dev, dtype=torch.float64).contiguous()
    _, H, W, _ = rgb.shape
    N = K.shape[0]
    if frame_map is None:
        assert rgb.shape[0] == N and mask.shape[0] == N
    else:
        frame_map = torch.as_tensor(frame_map).to(device=dev, dtype=torch.int32).contiguous()
        assert frame_map.shape == (N,) and mask.shape[0] == rgb.shape[0]
    S, P = int(img_size), int(n_pts)
    img = torch.empty(N, 3, S, S, dtype=torch.float32, device=dev)
    choose = torch.empty(N, P, dtype=torch.int32, device=dev)
    pts2d = torch.empty(N, P, 2, dtype=torch.float32, device=dev) if want_pts2d else None
    Kcrop = torch.empty(N, 3, 3, dtype=torch.float64, device=dev)
    window = torch.empty(N, 4, dtype=torch.int32, device=dev)
    valid = torch.empty(N, dtype=torch.int32, device=dev)
    scratch = torch.empty(N * S * S, dtype=torch.uint8, device=dev)
    tail = (N, H, W, S, P, int(seed) & 0xFFFFFFFF, _lib.ptr(img), _lib.ptr(choose), _lib.ptr(pts2d), _lib.ptr(Kcrop), _lib.ptr(window),
            _lib.ptr(valid), _lib.ptr(scratch), _lib.stream_ptr(stream))
    if frame0:          # a piece of a larger batch: frame f hashes as frame frame0 + f of the whole batch would
        _lib.check(lib.rgbm_prepare_inputs_ex(_lib.ptr(rgb), _lib.ptr(mask), _lib.ptr(K), _lib.ptr(frame_map), int(frame0), *tail),
                   "rgbm_prepare_inputs_ex")
    elif frame_map is None:
        _lib.check(lib.rgbm_prepare_inputs(_lib.ptr(rgb), _lib.ptr(mask), _lib.ptr(K), *tail), "rgbm_prepare_inputs")
    else:
        _lib.check(lib.rgbm_prepare_inputs_indexed(_lib.ptr(rgb), _lib.ptr(mask), _lib.ptr(K), _lib.ptr(frame_map), *tail),
                   "rgbm_prepare_inputs_indexed")
    out = {"img": img, "choose": choose, "Kcrop": Kcrop, "window": window, "valid": valid}
    if want_pts2d:
        out["pts2d"] = pts2d
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# Host mirror of the device's dependency cone (csrc/prob_sparse.hip::cone_of), for reporting and capacity planning: which part of
# the plane-sweep volume the network needs for a given set of chosen pixels (option "sparse_dec"; network_v5.py:260-291, 449-455).
def needed_c0_interval(y: int, S: int = 224):
    """Index interval [lo, hi] of c0 (full resolution, one axis) that the probabilities of a pixel at coordinate y depend on."""
    def clamp(a, b, n):
        return max(a, 0), min(b, n - 1)
    tr = lambda o, n: clamp(o[0] >> 1, (o[1] + 1) >> 1, n)      # noqa: E731  ConvTranspose3d k3 s2 p1 op1
    s1 = lambda o, n: clamp(o[0] - 1, o[1] + 1, n)              # noqa: E731  Conv3d k3 p1
    s2 = lambda o, n: clamp(2 * o[0] - 1, 2 * o[1] + 1, n)      # noqa: E731  Conv3d k3 p1 stride 2
    u = lambda p, q: (min(p[0], q[0]), max(p[1], q[1]))         # noqa: E731
    u11 = clamp(y - 1, y + 1, S)
    u9 = tr(u11, S // 2)
    u7 = tr(u9, S // 4)
    c5 = s1(tr(u7, S // 8), S // 8)
    c4 = u(s2(c5, S // 4), u7)
    c2 = u(s2(s1(c4, S // 4), S // 2), u9)
    return u(s2(s1(c2, S // 2), S), u11)


def sweep_tiles_needed_fraction(choose, S: int = 224, th: int = 12, tw: int = 16) -> float:
    """Fraction of the depth-sweeping conv0's th x tw-pixel tiles inside the dependency cones of the chosen pixels; choose [V, P]."""
    ch = np.asarray(choose.cpu() if isinstance(choose, torch.Tensor) else choose).reshape(-1, np.asarray(choose.shape)[-1])
    lo = np.array([needed_c0_interval(v, S)[0] for v in range(S)])
    hi = np.array([needed_c0_interval(v, S)[1] for v in range(S)])
    nth, ntw = -(-S // th), -(-S // tw)
    total = 0
    for row in ch:
        y, x = row // S, row % S
        m = np.zeros((nth, ntw), bool)
        for ra, rb, ca, cb in set(zip(lo[y] // th, hi[y] // th, lo[x] // tw, hi[x] // tw)):
            m[ra:rb + 1, ca:cb + 1] = True
        total += int(m.sum())
    return total / (len(ch) * nth * ntw)
