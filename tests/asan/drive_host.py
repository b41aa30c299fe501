"""Drives the HOST side of the library under AddressSanitizer (subprocess of tests/test_asan_host.py; numpy only, no torch).

RGBM_HIP_LIB points at rgbmanip_amd/librgbm_hip_asan_host.so (tools/build_asan_host.sh): every source of csrc compiled
--cuda-host-only with -fsanitize=address and linked against tests/asan/hip_stub.cpp, where "device" memory is host memory,
copies are memcpy and kernel launches do nothing.  So this exercises, with real shapes: the weight packers of every layer in all
four storage types (ConvLayer::init, conv3d_tile_pack, conv0_sweep_pack, the split-pair uploads, UpConvLayer / UpConvFinal), the
workspace planner, the chunk loops and every launcher's host code (geometry checks, descriptors, pointer arithmetic), and the
sizes of every upload (a hipMemcpy past the end of a packed host vector or of an allocation is an ASan report)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rgbmanip_amd import _lib, synth  # noqa: E402

assert "asan_host" in os.environ.get("RGBM_HIP_LIB", ""), "run through tests/test_asan_host.py"
lib = _lib.load()
sd = synth.adapose_state_dict(seed=0, prefix="module.")


def create(dtype, norm_mode, canary=False):
    keep, descs = [], []
    for k, v in sd.items():
        a = np.asarray(v)
        if a.dtype.kind != "f":
            continue
        a = np.ascontiguousarray(a, dtype=np.float32)
        shape, nd = (C.c_int64 * max(a.ndim, 1))(*(a.shape or (1,))), a.ndim
        if canary and k.endswith("layer4.2.conv2.weight"):
            a = a.reshape(-1)[: a.size // 2].copy()                     # half the data the descriptor promises: the packer over-reads
        name = k.encode()
        keep.append((a, shape, name))
        descs.append(_lib.WeightDesc(name, a.ctypes.data, nd, shape))
    arr = (_lib.WeightDesc * len(descs))(*descs)
    h = C.c_void_p()
    _lib.check(lib.rgbm_adapose_create(C.byref(h), 0, arr, len(descs), dtype, norm_mode), "create")
    return h


def vp(a):
    return C.c_void_p(a.ctypes.data)


def forward(h, B):
    n = C.c_size_t()
    _lib.check(lib.rgbm_adapose_workspace_bytes(h, B, C.byref(n)), "workspace_bytes")
    ws = np.empty(n.value + 256, dtype=np.uint8)          # "device" memory of the stub runtime
    base = ws.ctypes.data + ((-ws.ctypes.data) % 256)
    inp = synth.adapose_inputs(B, seed=1)
    img1, img2 = (np.ascontiguousarray(inp[k], dtype=np.float32) for k in ("img1", "img2"))
    ch1, ch2 = (np.ascontiguousarray(inp[k], dtype=np.int32) for k in ("choose1", "choose2"))
    P1, P2, dep = (np.ascontiguousarray(inp[k], dtype=np.float32) for k in ("P1", "P2", "depths"))
    outs = [np.empty(s, np.float32) for s in ((B, 1024, 3), (B, 1024, 3), (B, 1024), (B, 1024), (B, 3, 3), (B, 3, 3), (B, 3), (B, 3),
                                              (B, 3), (B, 3))]
    out = _lib.AdaposeOut(*[o.ctypes.data for o in outs])
    _lib.check(lib.rgbm_adapose_forward(h, B, vp(img1), vp(img2), vp(ch1), vp(ch2), vp(P1), vp(P2), vp(dep), C.c_void_p(base), n,
                                        C.byref(out), None), "forward")
    return n.value


if "canary" in sys.argv:      # the sanitizer must see a real over-read (proves that the build and the preload are live)
    create(_lib.BF16, 0, canary=True)
    print("CANARY_NOT_CAUGHT")
    sys.exit(0)

runs = 0
for dtype, modes in ((_lib.F32, (0, 1)), (_lib.BF16, (0,)), (_lib.F16, (0,)), (_lib.BF16X3, (0, 1))):
    for nm in modes:
        h = create(dtype, nm)
        sizes = {B: forward(h, B) for B in (1, 2, 3)}
        assert sizes[1] < sizes[2] <= sizes[3]
        for B in (48, 256, 512):                              # planner only: 33-38 GB at the benched batch
            n = C.c_size_t()
            _lib.check(lib.rgbm_adapose_workspace_bytes(h, B, C.byref(n)), "workspace_bytes")
            assert n.value > sizes[3]
        _lib.check(lib.rgbm_adapose_set_chunk(h, 2), "set_chunk")      # 6 views in chunks of 2: the chunk loops
        forward(h, 3)
        if nm == 0:
            for key, vals in ((b"upconv", (0, 7, 3)), (b"fuse_final", (0, 1)), (b"sparse_tail", (0, 1)), (b"igemm_conv6", (0, 1)),
                              (b"cost_impl", (1, 2, 3))):
                for v in vals:
                    _lib.check(lib.rgbm_adapose_set_option(h, key, v), "set_option")
                    forward(h, 2)
                    runs += 1
        _lib.check(lib.rgbm_adapose_destroy(h), "destroy")
        runs += 1

# the hipGraph cache of rgbm_adapose_forward_graph: capture, replay, least-recently-used eviction past 8 entries, invalidation by an
# option change, the null-stream refusal, destroy with live graphs (the stub's graph handles are heap blocks: leaks / double frees show)
h = create(_lib.BF16, 0)
keep = []


def fwd_graph(B, slot, stream=C.c_void_p(1)):
    key = (B, slot)
    bufs = next((b for k, b in keep if k == key), None)
    if bufs is None:
        n = C.c_size_t()
        _lib.check(lib.rgbm_adapose_workspace_bytes(h, B, C.byref(n)), "workspace_bytes")
        ws = np.empty(n.value + 256, dtype=np.uint8)
        inp = synth.adapose_inputs(B, seed=1)
        arrs = [np.ascontiguousarray(inp[k], dtype=np.float32) for k in ("img1", "img2")] + \
               [np.ascontiguousarray(inp[k], dtype=np.int32) for k in ("choose1", "choose2")] + \
               [np.ascontiguousarray(inp[k], dtype=np.float32) for k in ("P1", "P2", "depths")]
        outs = [np.empty(sh, np.float32) for sh in ((B, 1024, 3), (B, 1024, 3), (B, 1024), (B, 1024), (B, 3, 3), (B, 3, 3), (B, 3), (B, 3),
                                                    (B, 3), (B, 3))]
        bufs = (ws, arrs, outs, n.value)
        keep.append((key, bufs))
    ws, arrs, outs, nbytes = bufs
    base = ws.ctypes.data + ((-ws.ctypes.data) % 256)
    out = _lib.AdaposeOut(*[o.ctypes.data for o in outs])
    nodes, cap = C.c_int32(), C.c_int32()
    rc = lib.rgbm_adapose_forward_graph(h, B, *[vp(a) for a in arrs], C.c_void_p(base), nbytes, C.byref(out), stream, C.byref(nodes),
                                        C.byref(cap))
    return rc, nodes.value, cap.value


assert fwd_graph(1, 0) == (0, 7, 1) and fwd_graph(1, 0) == (0, 7, 0)          # capture, then replay
assert fwd_graph(1, 0, stream=None)[0] != 0                                    # the null stream cannot be captured
for slot in range(1, 10):                                                      # ten distinct pointer sets: evicts the oldest two
    assert fwd_graph(2, slot)[2] == 1
assert fwd_graph(2, 9)[2] == 0 and fwd_graph(1, 0)[2] == 1                     # the newest is cached, the oldest was evicted
_lib.check(lib.rgbm_adapose_set_option(h, b"sparse_dec", 0), "set_option")     # settings changed: every graph is dropped
assert fwd_graph(2, 9)[2] == 1
_lib.check(lib.rgbm_prof_select(31), "prof_select")                            # one row only, then every row again
_lib.check(lib.rgbm_prof_select(-1), "prof_select")
assert lib.rgbm_prof_select(4096) != 0                                          # out of range: refused
_lib.check(lib.rgbm_prof_start(), "prof_start")                                # profiler on: eager, reported as -1
assert fwd_graph(2, 9)[2] == -1
stats = (C.c_double * (4 * _lib.PROF_ROWS))()
_lib.check(lib.rgbm_prof_stop(stats), "prof_stop")
_lib.check(lib.rgbm_adapose_graph_clear(h), "graph_clear")
assert fwd_graph(2, 9)[2] == 1
_lib.check(lib.rgbm_adapose_destroy(h), "destroy")                             # with a live graph
runs += 1

# the post-processing launchers' host code (the kernels themselves are no-ops here)
B, P = 3, 1024
f32 = lambda *s: np.zeros(s, np.float32)      # noqa: E731
f64 = lambda *s: np.zeros(s, np.float64)      # noqa: E731
nocs, dep, r, ch = f32(B, P, 3), f32(B, P), f32(B, 3, 3), np.zeros((B, P), np.int32)
K, E, bbox, ts, valid = f64(B, 3, 3), f64(B, 4, 4), f64(B, 8, 3), f64(B, 8), np.zeros(B, np.int32)
_lib.check(lib.rgbm_adapose_postprocess(B, P, 224, vp(nocs), vp(dep), vp(r), vp(ch), vp(K), vp(E), vp(bbox), vp(ts), vp(valid), None),
           "postprocess")
print("ASAN_HOST_OK", runs, "configurations")
