"""`pose_estimator=adapose_*` plugin: AdaPoseEstimator_v5 on the HIP network.

Mirrors `/root/reference/models/pose_estimator/AdaPose/interface_v5.py:37-374` and the base class
`models/pose_estimator/base_estimator.py:5-20`: `AdaPoseEstimator_v5(env, cfg, logger)`,
`estimate(K, rgb1, mask1, E1, rgb2, mask2, E2) -> ndarray [N,8,3]`, `predict(...)`, `prepare_model_input(...)`,
never raises for bad samples (empty mask / non-finite result -> `default_bbox`, the +10 cube).

Differences that are the point of this build: `estimate` prepares all N samples, runs ONE batched network call and ONE
batched post-processing launch on the device instead of N serial B=1 calls with a host round trip each
(interface_v5.py:218-225, 259-286, 318-321), and it skips `draw_result` (a discarded debug drawing, :364).
The crop/resize/sampling step runs on the host in numpy by default (the reference's arithmetic incl. its global-RNG
subset) or, with cfg["hip_prepare"] == "device", batched on the GPU (`rgbm_prepare_inputs`; the 1024-subset
is then a seeded hash, cfg["hip_prepare_seed"]); `estimate_device` takes device-resident frames and never leaves the GPU.
"""
from __future__ import annotations

import os
import sys
import time
import warnings

import numpy as np
import torch

from . import _lib
from .adapose import AdaPoseNet, postprocess, postprocess_pnp, postprocess_ransac, prepare_inputs

DEFAULT_BBOX = np.asarray([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]],
                          dtype=np.float64) + 10.0
_MEAN = np.array([0.485, 0.456, 0.406])
_STD = np.array([0.229, 0.224, 0.225])


class BasePoseEstimator:
    def __init__(self, env, cfg: dict, logger):
        self.env = env
        self.cfg = cfg
        self.logger = logger

    def append_picture(self, pic, pose):
        pass

    def estimate(self):
        pass


def get_bbox(bbox):
    """Square crop window: side = multiple of 40 (<= 440), clamped into the 480x640 frame (lib/utils.py:10-38)."""
    y1, x1, y2, x2 = bbox
    win = min((max(y2 - y1, x2 - x1) // 40 + 1) * 40, 440)
    half = int(win / 2)
    cy, cx = (y1 + y2) // 2, (x1 + x2) // 2
    rmin, rmax, cmin, cmax = cy - half, cy + half, cx - half, cx + half
    if rmin < 0:
        rmin, rmax = 0, rmax - rmin
    if cmin < 0:
        cmin, cmax = 0, cmax - cmin
    if rmax > 480:
        rmin, rmax = rmin - (rmax - 480), 480
    if cmax > 640:
        cmin, cmax = cmin - (cmax - 640), 640
    return rmin, rmax, cmin, cmax


def _resize_nearest(img, size):
    h, w = img.shape[:2]
    ys = np.minimum((np.arange(size) * (h / size)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(size) * (w / size)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def _resize_linear(img, size):
    """OpenCV INTER_LINEAR arithmetic for float images (half-pixel centres, edge clamp, no antialias)."""
    h, w = img.shape[:2]

    def taps(n_src):
        f = (np.arange(size) + 0.5) * (n_src / size) - 0.5
        i0 = np.floor(f).astype(np.int64)
        a = (f - i0).astype(np.float32)
        a = np.where(i0 < 0, 0.0, a)
        i0 = np.maximum(i0, 0)
        a = np.where(i0 >= n_src - 1, 0.0, a).astype(np.float32)
        i0 = np.minimum(i0, n_src - 1)
        return i0, np.minimum(i0 + 1, n_src - 1), a
    y0, y1, ay = taps(h)
    x0, x1, ax = taps(w)
    img = img.astype(np.float32)
    ax = ax[None, :, None]
    ay = ay[:, None, None]
    top = img[y0][:, x0] * (1 - ax) + img[y0][:, x1] * ax
    bot = img[y1][:, x0] * (1 - ax) + img[y1][:, x1] * ax
    return top * (1 - ay) + bot * ay


def _mix32(seed, frame, idx):
    """Seeded subset hash of csrc/prepare.hip (murmur3 finaliser), uint32 arithmetic."""
    with np.errstate(over="ignore"):
        h = np.uint32(seed) ^ (np.uint32(frame) * np.uint32(0x9E3779B9)) ^ (np.asarray(idx, dtype=np.uint32) * np.uint32(0x85EBCA6B))
        h = h ^ (h >> np.uint32(16)); h = h * np.uint32(0x85EBCA6B)
        h = h ^ (h >> np.uint32(13)); h = h * np.uint32(0xC2B2AE35)
        h = h ^ (h >> np.uint32(16))
    return h.astype(np.uint32)


_HOST_THREADS = max(1, min(32, (os.cpu_count() or 1)))
_POOL = None


def _host_pool():
    """Thread pool of the host-side frame conversion (created on first use)."""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=_HOST_THREADS, thread_name_prefix="rgbm-upload")
    return _POOL


def _cast_into(dst, src):
    """dst[...] = src cast to dst's dtype (float64 / float16 frames -> float32 staging; same dtype: a plain copy).  numpy releases the GIL."""
    np.copyto(dst, src, casting="same_kind")


def _nonzero_into(dst_u8, src):
    np.not_equal(src, 0, out=dst_u8.view(np.bool_))


def _split(n, parts):
    """[lo, hi) ranges cutting n rows into at most `parts` nearly equal pieces"""
    parts = max(1, min(parts, n))
    return [(n * i // parts, n * (i + 1) // parts) for i in range(parts)]


class AdaPoseEstimator_v5(BasePoseEstimator):
    def __init__(self, env, cfg, logger, state_dict=None, dtype=None, device=0, net=None):
        """`net`: an already built `AdaPoseNet` to share (weights + workspace) instead of building one from `state_dict`."""
        super().__init__(env, cfg, logger)
        if net is not None:
            state_dict = {}
        elif state_dict is None:
            if cfg.get("load", False):
                state_dict = torch.load(cfg["checkpoint_path"], map_location="cpu")      # DataParallel keys ("module.")
            else:
                from . import synth
                state_dict = synth.adapose_state_dict(seed=0)
                if logger is not None:
                    logger.warning("AdaPoseEstimator_v5: cfg.load is False -> synthetic (seeded) weights")
        self.dtype = dtype or cfg.get("hip_dtype", "bf16x3")      # the fastest mode inside north_star's 1e-4 (fp32: 4x slower, bf16: 2.4x faster at 1e-2)
        # hip_graph (default off: measured, small batches are bound by their kernels, not by the ~100 launches): batches of at most
        # hip_graph_max_batch poses replay a captured hipGraph.
        # hip_view2_heads (default: only where the box tail reads view-2 outputs, i.e. the PnP branch): the reference network returns
        # ten outputs and `predict` builds the box from view1_nocs / view1_depth / view1_r alone (interface_v5.py:318-374), so the
        # cost volume, point heads and pose regression of the view-2 crops are skipped — the backbone still runs on both views
        self.view2_heads = bool(cfg.get("hip_view2_heads", not cfg.get("direct_regression", True) and not cfg.get("use_depth", True)))
        self.estimator = net if net is not None else AdaPoseNet(state_dict, dtype=self.dtype, device=device,
                                                                norm_mode=cfg.get("hip_norm_mode", "eval"),
                                                                graph=bool(cfg.get("hip_graph", False)),
                                                                graph_max_batch=int(cfg.get("hip_graph_max_batch", 32)),
                                                                options={**{str(k): int(v) for k, v in dict(cfg.get("hip_options", {}) or {}).items()},
                                                                         "view2_heads": int(self.view2_heads)})
        # hip_options: any rgbm_adapose_set_option key (include/rgbm.h), e.g. {"sweep_f16": 0} for a bf16 checkpoint whose 32-channel
        # feature map can exceed the f16 range (the plane sweep of bf16 nets reads it as f16 since round 5)
        if net is not None:
            # a shared network keeps ITS setting (changing it on the shared handle would drop every captured graph and change the other
            # users' outputs): report what it computes, and refuse the combination that would feed never-written view-2 outputs to the PnP tail
            net_v2 = bool(net.options.get("view2_heads", 1))
            if self.view2_heads and not net_v2:
                raise ValueError("AdaPoseEstimator_v5: this cfg needs the view-2 heads (hip_view2_heads, or the PnP tail of "
                                 "direct_regression=False / use_depth=False), but the shared net was built with view2_heads=0")
            self.view2_heads = net_v2
            # ... and the same for hip_options: a key this cfg names must already hold on the shared net (round-5 advice: silently
            # ignoring e.g. {"sweep_f16": 0} would run the f16 feature map the user opted out of)
            want = {str(k): int(v) for k, v in dict(cfg.get("hip_options", {}) or {}).items()}
            differ = {k: (v, net.options.get(k)) for k, v in want.items() if net.options.get(k) != v}
            if differ:
                raise ValueError("AdaPoseEstimator_v5: cfg.hip_options is not applied to a shared net; build the AdaPoseNet with "
                                 f"options={want} (requested vs the net's: {differ})")
        self.rng = np.random          # the reference shuffles with the global numpy RNG (interface_v5.py:129)
        # "device" (default since round 5): frames are uploaded once and cropped / resized / sub-sampled on the GPU (rgbm_prepare_inputs);
        # a mask with more than 1024 pixels keeps the 1024 smallest hash keys (hip_prepare_seed) instead of the pixels np.random.shuffle
        # would pick (interface_v5.py:126-130) — same distribution, another stream.  "host": the reference's per-frame numpy path on the
        # global numpy RNG (one host core: 46 ms per pose), for RNG-stream parity with the reference.
        self.prepare_mode = cfg.get("hip_prepare", "device")
        self.prepare_seed = int(cfg.get("hip_prepare_seed", 0))
        self._frame = 0               # hash-subset mode of the host path: index of the sample being prepared

    # ------------------------------------------------------------------ interface_v5.py:58-170
    def prepare_model_input(self, rgb, mask, intrinsic, resize_size):
        rgb = np.asarray(rgb)
        if rgb.dtype == np.uint8:           # transforms.ToTensor scales uint8 images to [0, 1] (interface_v5.py:52-54,149); floats pass as they are
            rgb = rgb.astype(np.float32) / np.float32(255.0)
        elif rgb.dtype.kind != "f":
            raise TypeError(f"prepare_model_input: rgb must be a float image in [0, 1] or uint8, got {rgb.dtype}")
        ys, xs = np.nonzero(mask)
        if len(ys) == 0:
            return None, None, None, None
        rmin, rmax, cmin, cmax = get_bbox([int(ys.min()), int(xs.min()), int(ys.max()), int(xs.max())])
        small = _resize_nearest(mask[rmin:rmax, cmin:cmax].astype(np.float32), resize_size)
        choose = small.flatten().nonzero()[0]
        if len(choose) > 1024:
            if isinstance(self.rng, tuple):             # ("hash", seed): the device path's reproducible subset, on the host
                keys = _mix32(self.rng[1], self._frame, choose).astype(np.uint64)
                choose = choose[np.sort(np.lexsort((np.arange(len(choose)), keys))[:1024])]
            else:
                keep = np.zeros(len(choose), dtype=int)
                keep[:1024] = 1
                self.rng.shuffle(keep)
                choose = choose[keep.nonzero()]
        elif len(choose) == 0:
            return None, None, None, None
        else:
            choose = np.pad(choose, (0, 1024 - len(choose)), "wrap")
        ratio = resize_size / (rmax - rmin)
        pts2d = np.stack(((choose % resize_size).astype(np.float32) / ratio + cmin,
                          (choose // resize_size).astype(np.float32) / ratio + rmin), axis=-1)
        crop = _resize_linear(rgb[rmin:rmax, cmin:cmax, :], resize_size).astype(rgb.dtype)
        view = (np.transpose(crop, (2, 0, 1)) - _MEAN.astype(rgb.dtype)[:, None, None]) / _STD.astype(rgb.dtype)[:, None, None]
        K = np.eye(3)
        K[0, 0], K[1, 1] = intrinsic[0, 0] * ratio, intrinsic[1, 1] * ratio
        K[0, 2] = (intrinsic[0, 2] - (float(cmin + cmax) / 2 - float(cmax - cmin + 1) / 2)) * ratio
        K[1, 2] = (intrinsic[1, 2] - (float(rmin + rmax) / 2 - float(rmax - rmin + 1) / 2)) * ratio
        return torch.from_numpy(np.ascontiguousarray(view)), choose, pts2d, K

    # ------------------------------------------------------------------ interface_v5.py:213-227
    def estimate(self, camera_intrinsic_batch, rgb1_batch, view1_mask_batch, view1_extrinsic_batch, rgb2_batch,
                 view2_mask_batch, view2_extrinsic_batch):
        S = self.cfg["img_size"]
        n = len(rgb1_batch)
        if self.prepare_mode == "device":
            return self._estimate_host_frames(camera_intrinsic_batch, rgb1_batch, view1_mask_batch, view1_extrinsic_batch, rgb2_batch,
                                              view2_mask_batch, view2_extrinsic_batch)
        out = np.repeat(DEFAULT_BBOX[None], n, axis=0)
        rows, img1, img2, ch1, ch2, P1, P2, K1, E1 = [], [], [], [], [], [], [], [], []
        pt1, pt2, E2, K0 = [], [], [], []                    # the PnP branch also needs the pixels, the second extrinsic and the original K
        for i in range(n):
            self._frame = i
            if isinstance(self.rng, tuple):
                self.rng = ("hash", self.prepare_seed)
            a = self.prepare_model_input(rgb1_batch[i], view1_mask_batch[i], camera_intrinsic_batch[i], S)
            if isinstance(self.rng, tuple):
                self.rng = ("hash", self.prepare_seed + 1)
            b = self.prepare_model_input(rgb2_batch[i], view2_mask_batch[i], camera_intrinsic_batch[i], S)
            if a[0] is None or b[0] is None:
                continue
            p1, p2 = np.eye(4), np.eye(4)
            p1[:3, :] = a[3] @ np.asarray(view1_extrinsic_batch[i])[:3, :]
            p2[:3, :] = b[3] @ np.asarray(view2_extrinsic_batch[i])[:3, :]
            rows.append(i)
            img1.append(a[0].float()); img2.append(b[0].float())
            ch1.append(a[1]); ch2.append(b[1])
            P1.append(p1.astype(np.float32)); P2.append(p2.astype(np.float32))
            K1.append(a[3]); E1.append(np.asarray(view1_extrinsic_batch[i], dtype=np.float64))
            pt1.append(a[2]); pt2.append(b[2]); E2.append(np.asarray(view2_extrinsic_batch[i], dtype=np.float64))
            K0.append(np.asarray(camera_intrinsic_batch[i], dtype=np.float64))
        if not rows:
            return out
        B = len(rows)
        depths = np.tile(np.arange(0.1, 0.1 * (24 - 0.5) + 0.1, 0.1, dtype=np.float32)[None], (B, 1))
        ch1 = np.stack(ch1)
        pred = self.estimator(torch.stack(img1), ch1, torch.stack(img2), np.stack(ch2), np.stack(P1), np.stack(P2), depths)
        bbox = self._bbox_tail(pred, ch1, np.stack(K1), np.stack(E1), pts2d=(np.stack(pt1), np.stack(pt2)), E2=np.stack(E2), K=np.stack(K0))
        out[np.asarray(rows)] = bbox.cpu().numpy()
        return out

    # ------------------------------------------------------------------ host frames -> HBM
    _CHUNK_BYTES = 64 << 20

    def _estimate_host_frames(self, K, rgb1, mask1, E1, rgb2, mask2, E2):
        """`estimate` with `hip_prepare: device` for host arrays (what rl_pose.py:210-218 hands over: [N,480,640,3] float64 frames,
        3.8 GB per call at N = 256).  Batches larger than `hip_upload_chunk` poses (default 32: measured best of 8 .. 128 for float64 and
        float32 frames in bf16 and bf16x3, tools/boundary_chunks.py; 1 GB of pinned staging for float64 frames) run as a three-stage pipeline over
        chunks of poses: host threads copy chunk c + 1 into pinned staging buffers while the copy engine moves chunk c to the device
        on its own stream and the kernels (dtype conversion, crop / resize / subset, network, post-processing) work on chunk c - 1.
        Poses are independent, but a chunk is a smaller batch: below ~1000 GEMM rows per launch and at launches that fit one round of the
        persistent grid the dispatcher picks other tiles (summation order), so a pose's box agrees with the unchunked call's to the
        storage type's rounding (1e-6 .. 1e-5 relative in fp32 / bf16x3), not bit for bit (include/rgbm.h, rgbm_set_tuning)."""
        n = len(rgb1)
        chunk = int(self.cfg.get("hip_upload_chunk", 32))
        on_dev = any(isinstance(x, torch.Tensor) and x.is_cuda for x in (rgb1, rgb2))
        if on_dev or n <= chunk or chunk <= 0:
            return self.estimate_device(np.asarray(K), self._upload_frames(rgb1), self._upload_masks(mask1), np.asarray(E1),
                                        self._upload_frames(rgb2), self._upload_masks(mask2), np.asarray(E2)).cpu().numpy()
        dev = self.estimator.device
        srcs = [x.numpy() if isinstance(x, torch.Tensor) else np.ascontiguousarray(np.asarray(x)) for x in (rgb1, rgb2, mask1, mask2)]
        for a in srcs[:2]:
            if a.dtype != np.uint8 and a.dtype.kind != "f":
                raise TypeError(f"estimate: rgb frames must be float images in [0, 1] or uint8, got {a.dtype}")
        Kd = torch.as_tensor(np.asarray(K)).to(dev)
        E1d, E2d = torch.as_tensor(np.asarray(E1)).to(dev), torch.as_tensor(np.asarray(E2)).to(dev)
        # float frames are converted to float32 WHILE they are copied into the pinned staging buffers (numpy's casting copy is as fast as its
        # plain copy once a pool of threads runs it: both are bound by host memory, tools/host_convert_bw.py), so float64 frames cross PCIe
        # at half their size; the value every later stage sees is the same float32(frame) the device-side conversion produced
        tdt = [torch.uint8 if a.dtype == np.uint8 else torch.float32 for a in srcs[:2]] + [torch.uint8, torch.uint8]
        shp = [tuple(a.shape[1:]) for a in srcs]
        key = ("pipe", chunk, tuple(shp), tuple(tdt))
        if getattr(self, "_pipe_key", None) != key:
            self._pipe_pin = [[torch.empty((chunk,) + shp[i], dtype=tdt[i], pin_memory=True) for i in range(4)] for _ in range(2)]
            self._pipe_np = [[t.numpy() for t in slot] for slot in self._pipe_pin]
            self._pipe_dev = [[torch.empty((chunk,) + shp[i], dtype=tdt[i], device=dev) for i in range(4)] for _ in range(2)]
            self._pipe_h2d = [torch.cuda.Event(), torch.cuda.Event()]
            self._pipe_done = [torch.cuda.Event(), torch.cuda.Event()]
            self._pipe_stream = torch.cuda.Stream(device=dev)
            self._pipe_key = key
        pool = _host_pool()
        cur = torch.cuda.current_stream(dev)
        out = torch.empty(n, 8, 3, dtype=torch.float64, device=dev)
        used = [False, False]

        def stage(slot, a, b):
            tasks = []
            for i in range(4):
                dst, src = self._pipe_np[slot][i], srcs[i]
                for lo, hi in _split(b - a, max(1, _HOST_THREADS // 2)):
                    if i < 2:
                        tasks.append((_cast_into, dst[lo:hi], src[a + lo:a + hi]))
                    elif src.dtype == np.uint8:
                        tasks.append((np.copyto, dst[lo:hi], src[a + lo:a + hi]))
                    elif src.dtype == np.bool_:
                        tasks.append((np.copyto, dst[lo:hi], src[a + lo:a + hi].view(np.uint8)))
                    else:                                  # any number type: non-zero = object, one byte per pixel crosses PCIe
                        tasks.append((_nonzero_into, dst[lo:hi], src[a + lo:a + hi]))
            list(pool.map(lambda t: t[0](t[1], t[2]), tasks))

        trace = [] if os.environ.get("RGBM_UPLOAD_TRACE") == "1" else None
        for c, a in enumerate(range(0, n, chunk)):
            b = min(a + chunk, n)
            slot = c & 1
            t0 = time.perf_counter()
            if used[slot]:
                self._pipe_h2d[slot].synchronize()         # the copy that last read this slot's pinned buffers has finished
            t1 = time.perf_counter()
            stage(slot, a, b)                              # host threads; overlaps the device's work on the previous chunks
            if trace is not None:
                trace.append((round((t1 - t0) * 1e3, 2), round((time.perf_counter() - t1) * 1e3, 2)))
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if trace is not None else None
            with torch.cuda.stream(self._pipe_stream):
                if used[slot]:
                    self._pipe_stream.wait_event(self._pipe_done[slot])      # the kernels that read this slot's device buffers are done
                if ev: ev[0].record(self._pipe_stream)
                for i in range(4):
                    self._pipe_dev[slot][i][: b - a].copy_(self._pipe_pin[slot][i][: b - a], non_blocking=True)
                self._pipe_h2d[slot].record(self._pipe_stream)
                if ev: ev[1].record(self._pipe_stream)
            cur.wait_event(self._pipe_h2d[slot])
            if ev: ev[2].record(cur)
            d = [t[: b - a] for t in self._pipe_dev[slot]]
            out[a:b] = self.estimate_device(Kd[a:b], self._upload_frames(d[0]), d[2], E1d[a:b], self._upload_frames(d[1]), d[3], E2d[a:b], frame0=a)
            self._pipe_done[slot].record(cur)
            if ev:
                ev[3].record(cur)
                trace[-1] = trace[-1] + (round((time.perf_counter() - t0) * 1e3, 2), ev)
            used[slot] = True
        res = out.cpu().numpy()
        if trace is not None:
            e00 = trace[0][3][0]
            rows = [(w, st, tot, round(e00.elapsed_time(ev[0]), 1), round(e00.elapsed_time(ev[1]), 1), round(e00.elapsed_time(ev[2]), 1), round(e00.elapsed_time(ev[3]), 1))
                    for (w, st, tot, ev) in trace]
            print("[rgbm upload trace] per chunk (host: wait for slot ms, stage ms, whole iteration ms | device clock from the first copy's start: copy start, "
                  "copy end, kernels start, kernels end):", rows, file=sys.stderr)
        return res

    def _upload_frames(self, frames):
        """[N,H,W,3] host frames (float64 / float32 in [0,1], or uint8) -> CUDA float32 [N,H,W,3] in [0,1].  Frames that already are
        CUDA tensors pass through.  A pool of host threads copies each chunk into one of two pinned staging buffers while the previous
        chunk's copy is in flight; float frames are cast to float32 by that copy (3.8 GB of float64 frames arrive per call at N = 256 and
        cross PCIe as 1.9 GB), uint8 frames cross as bytes and are scaled on the device."""
        if isinstance(frames, torch.Tensor) and frames.is_cuda:
            return frames.to(torch.float32) if frames.dtype != torch.uint8 else (frames.to(torch.float64) / 255.0).to(torch.float32)      # (device-side dtype conversion of an uploaded chunk)
        src = frames.numpy() if isinstance(frames, torch.Tensor) else np.ascontiguousarray(np.asarray(frames))
        dev = self.estimator.device
        if src.dtype != np.uint8 and src.dtype.kind != "f":
            raise TypeError(f"estimate: rgb frames must be float images in [0, 1] or uint8, got {src.dtype}")
        tdt = torch.uint8 if src.dtype == np.uint8 else torch.float32      # float frames: converted to float32 by the staging copy itself
        n = src.shape[0]
        per = max(1, int(np.prod(src.shape[1:]))) * (1 if src.dtype == np.uint8 else 4)
        rows = max(1, min(n, self._CHUNK_BYTES // per))
        key = (rows, tuple(src.shape[1:]), tdt)
        if getattr(self, "_stage_key", None) != key:
            self._stage = [torch.empty((rows,) + tuple(src.shape[1:]), dtype=tdt, pin_memory=True) for _ in range(2)]
            self._stage_np = [t.numpy() for t in self._stage]
            self._stage_dev = [torch.empty((rows,) + tuple(src.shape[1:]), dtype=tdt, device=dev) for _ in range(2)]
            self._stage_ev = [torch.cuda.Event(), torch.cuda.Event()]
            self._stage_key = key
            self._stage_used = [False, False]
        out = torch.empty(tuple(src.shape), dtype=torch.float32, device=dev)
        pool = _host_pool()
        for i, a in enumerate(range(0, n, rows)):
            b = min(a + rows, n)
            k = i & 1
            if self._stage_used[k]:
                self._stage_ev[k].synchronize()                    # the copy that last read this staging buffer has finished
            dst = self._stage_np[k]
            list(pool.map(lambda p: _cast_into(dst[p[0]:p[1]], src[a + p[0]:a + p[1]]), _split(b - a, _HOST_THREADS)))
            self._stage_dev[k][: b - a].copy_(self._stage[k][: b - a], non_blocking=True)
            self._stage_ev[k].record()
            self._stage_used[k] = True
            # stream-ordered: this conversion runs before the copy that next overwrites _stage_dev[k] (two chunks later)
            if src.dtype == np.uint8:
                # x / 255 through float64: torch's float32 division on ROCm is not correctly rounded (126 of the 256 byte values
                # differ from numpy's float32(x) / float32(255) by one ulp, tools/check_div.py); the float64 quotient rounded to
                # float32 equals the correctly rounded float32 quotient for every byte value
                out[a:b].copy_(self._stage_dev[k][: b - a].to(torch.float64) / 255.0)
            else:
                out[a:b].copy_(self._stage_dev[k][: b - a])
        return out

    def _upload_masks(self, masks):
        """[N,H,W] host masks (bool / uint8 / any number type, non-zero = object) -> CUDA uint8."""
        if isinstance(masks, torch.Tensor) and masks.is_cuda:
            return masks
        m = masks if isinstance(masks, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(masks)))
        if m.dtype == torch.bool:
            m = m.view(torch.uint8)
        elif m.dtype != torch.uint8:
            m = m.ne(0).view(torch.uint8)                              # multi-threaded on the host: 1 byte per pixel crosses PCIe
        return m.to(self.estimator.device, non_blocking=False)

    # ------------------------------------------------------------------ the same pipeline without leaving the device
    def estimate_device(self, K, rgb1, mask1, E1, rgb2, mask2, E2, frame0: int = 0):
        """`estimate` for frames that already live on the GPU (or get uploaded once): K [N,3,3], rgb [N,H,W,3] float32 in [0,1],
        mask [N,H,W], E [N,4,4] world->camera.  Returns a CUDA tensor [N,8,3] float64; samples the reference would skip
        (empty mask) or reject (non-finite box) hold `default_bbox`."""
        S = self.cfg["img_size"]
        dev = self.estimator.device
        Kd = torch.as_tensor(K).to(dev)
        wp = self._pnp_branch()
        a = prepare_inputs(torch.as_tensor(rgb1).to(dev), torch.as_tensor(mask1).to(dev), Kd, S, 1024, self.prepare_seed, want_pts2d=wp, frame0=frame0)
        b = prepare_inputs(torch.as_tensor(rgb2).to(dev), torch.as_tensor(mask2).to(dev), Kd, S, 1024, self.prepare_seed + 1, want_pts2d=wp, frame0=frame0)
        return self._estimate_prepared(a, b, E1, E2, Kd)

    def estimate_device_indexed(self, K, rgb_pool, mask_pool, E1, E2, map1, map2):
        """`estimate_device` reading the two views of sample i from entries map1[i] / map2[i] of a frame pool
        (rgb_pool [M,H,W,3] float32, mask_pool [M,H,W] uint8 — e.g. the controller's view queue) instead of from gathered
        batches; a negative entry means "no such view" (the reference hands an all-zero frame over, which is skipped).
        K [N,3,3] (both views use it, interface_v5.py:213-227), E1 / E2 [N,4,4]."""
        S = self.cfg["img_size"]
        wp = self._pnp_branch()
        a = prepare_inputs(rgb_pool, mask_pool, K, S, 1024, self.prepare_seed, frame_map=map1, want_pts2d=wp)
        b = prepare_inputs(rgb_pool, mask_pool, K, S, 1024, self.prepare_seed + 1, frame_map=map2, want_pts2d=wp)
        return self._estimate_prepared(a, b, E1, E2, K)

    def _pnp_branch(self):
        return not self.cfg.get("direct_regression", True) and not self.cfg.get("use_depth", True)

    def _estimate_prepared(self, a, b, E1, E2, K=None):
        S = self.cfg["img_size"]
        dev = self.estimator.device
        E1d = torch.as_tensor(E1).to(device=dev, dtype=torch.float64)
        E2d = torch.as_tensor(E2).to(device=dev, dtype=torch.float64)
        n = E1d.shape[0]

        def proj(Kc, E):                                  # P = K' E[:3], padded to 4x4 (interface_v5.py:264-270), fp64 -> fp32
            P = torch.empty(n, 4, 4, dtype=torch.float32, device=dev)
            _lib.check(_lib.load().rgbm_projection(_lib.ptr(Kc.contiguous()), _lib.ptr(E.contiguous()), _lib.ptr(P), n, _lib.stream_ptr()),
                       "rgbm_projection")
            return P
        # constants live on the device: a pageable host -> device copy here would block the host until the stream has drained, i.e.
        # serialise the chunk pipeline of _estimate_host_frames (measured: staging, copy and kernels ran back to back)
        consts = getattr(self, "_dev_consts", None)
        if consts is None or consts[0].device != dev:
            consts = self._dev_consts = (torch.from_numpy(DEFAULT_BBOX).to(dev),
                                         torch.from_numpy(np.arange(0.1, 0.1 * (24 - 0.5) + 0.1, 0.1, dtype=np.float32)).to(dev))
        depths = consts[1][None].expand(n, 24).contiguous()
        pred = self.estimator(a["img"], a["choose"], b["img"], b["choose"], proj(a["Kcrop"], E1d), proj(b["Kcrop"], E2d), depths)
        bbox = self._bbox_tail(pred, a["choose"], a["Kcrop"], E1d, pts2d=(a.get("pts2d"), b.get("pts2d")), E2=E2d, K=K)
        ok = ((a["valid"] != 0) & (b["valid"] != 0)).view(n, 1, 1)
        return torch.where(ok, bbox, consts[0].expand(n, 8, 3))

    def _bbox_tail(self, pred, choose, Kcrop, E1, pts2d=None, E2=None, K=None):
        """interface_v5.py:318-374: scale / translation from the regressed rotation (`direct_regression`, the shipped configs)
        or Umeyama-RANSAC between predicted NOCS and the back-projected predicted depth (`use_depth`), then the world box."""
        S = self.cfg["img_size"]
        if self.cfg.get("direct_regression", True):
            return postprocess(pred["view1_nocs"], pred["view1_depth"], pred["view1_r"], choose, Kcrop, E1, img_size=S)[0]
        if self.cfg.get("use_depth", True):
            return postprocess_ransac(pred["view1_nocs"], pred["view1_depth"], choose, Kcrop, E1, img_size=S,
                                      seed=int(self.cfg.get("hip_ransac_seed", 0)))[0]
        # use_depth False (interface_v5.py:340-346): NOCS matches of the two views -> scale -> EPnP-RANSAC + VVS on the ORIGINAL
        # intrinsics and the chosen points' pixels in the original frame
        if not getattr(self, "_pnp_warned", False):
            self._pnp_warned = True
            msg = ("AdaPoseEstimator_v5: direct_regression=False with use_depth=False runs csrc/pnp.hip, a restatement of OpenCV's "
                   "triangulatePoints / solvePnPRansac(EPNP) / solvePnPRefineVVS whose RANSAC subset stream and tie-breaks are NOT "
                   "pinned against cv2 (no OpenCV in the build image; DESIGN.md section 2): poses agree with ground truth, inlier sets "
                   "may differ from the reference's")
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
            if self.logger is not None:
                self.logger.warning(msg)
        return postprocess_pnp(pred["view1_nocs"], pts2d[0], pred["view2_nocs"], pts2d[1], K, E1, E2,
                               seed=int(self.cfg.get("hip_ransac_seed", 0)))[0]

    def predict(self, camera_intrinsic, rgb1, view1_mask, view1_extrinsic, rgb2, view2_mask, view2_extrinsic):
        return self.estimate([camera_intrinsic], [rgb1], [view1_mask], [view1_extrinsic], [rgb2], [view2_mask],
                             [view2_extrinsic])[0]
