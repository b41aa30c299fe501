"""Two forwards that OVERLAP on the device — the halves of a batch on two HIP streams, each with its own workspace, one handle or two —
against the one-stream forward: a regression check since round 4, when they differed.  What was found (each line one or more gpurun experiments):
  * symptom (library built with hipcc's defaults): 14-19 of 25 overlapped runs differed from the one-stream forward in view*_r / _t / _s (fp16 nets:
    depth outputs too) — a few per cent on pairs of points 4k + 2, 4k + 3 = lane rows 16-31 / 48-63 of a wave, in the blocks resident when a kernel
    starts, never NaN; 0 of 25 with a host synchronisation between the two launch sequences or with GPU_MAX_HW_QUEUES=1; one-stream forwards always
    bit-stable (510 repeated runs);
  * localisation: hashes taken inside `fuse_points_kernel` (debug builds) showed pixel, prob row, depth values and the homography as re-read at the END
    of the kernel to be the reference's, while the warp coordinates computed from the homography words loaded at the START were not; the same coordinates
    evaluated twice in one wave through one `noinline` function (start / end of the kernel) differed from each other in the affected lanes;
  * ruled out: the divisions (approximate ones: same rate), the cross-lane shuffle, occupancy (2 workgroups per CU on one stream: no reproduction),
    leftover register contents (register-poisoning kernel in front), one particular co-tenant kernel (every configuration of the other forward's cost
    volume stage interferes, its PSPNet alone does not), memory contents (a watcher polling the homography through L2 never saw a word change), a
    stand-alone pair of kernels (`tools/micro/valu_under_mfma.hip`; later also `tools/micro/pk_after_load.hip`: a hand-written `v_pk_mul_f32` right behind the
    counted wait of cache-cold loads, 0 of 335 M lanes off next to MFMA / copy co-tenants — the minimal pattern alone does not reproduce it);
  * the instruction: in that kernel hipcc feeds the just-loaded homography registers (`global_load_dwordx4`, the correct `s_waitcnt vmcnt(n)` in front)
    into PACKED fp32 instructions (`v_pk_mul_f32` / `v_pk_fma_f32` / `v_pk_add_f32`).  The same source with the per-lane vector loads kept and packed
    fp32 instructions disabled (`-Xclang -target-feature -Xclang -packed-fp32-ops`): 0 of 60 overlapped runs; with them: 16 of 20.  Loading the record
    through the scalar cache (SGPR operands) also removed it in that kernel, but fp16 nets kept differing until the WHOLE library was built without
    packed fp32 instructions: then bf16 0 of 40, bf16x3 0 of 24, fp16 0 of 40 overlapped runs.
To bring the failure back on demand: tools/patches/README.md (a patch with the vector loads + a build with packed instructions: 13 of 20 runs differ).
So: on this part and toolchain (MI355X, ROCm 7.2 hipcc) a packed-fp32 VALU instruction that consumes registers a vector memory load has just delivered can
compute from other values in half of the lane rows while kernels of another hardware queue share the CU.  The library is built without packed fp32
instructions (build.sh; they bought nothing: same-box A/B within 0.3 %), tests/test_cabi_symbols.py asserts the shipped code objects hold none, and
`fuse_points_kernel` keeps the scalar load (one instruction instead of three per lane).  bench.py's `two_streams` leg compares its outputs with the
one-stream forward (`outputs_bit_identical_to_one_stream`).
usage: python tools/check_two_stream_forwards.py [dtype] [runs]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rgbmanip_amd import synth, _lib
from rgbmanip_amd.adapose import AdaPoseNet

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 25
B, h = 256, 128
dev = torch.device("cuda", 0)
_, d, _ = bench.make_inputs_crop(B, dev, seed=0)
sd = synth.adapose_state_dict(seed=0)
lib = _lib.load()
one = AdaPoseNet(sd, dtype=dt)
args = [d[k] for k in ("img1", "img2", "choose1", "choose2", "P1", "P2", "depths")]
ref = one(d["img1"], d["choose1"], d["img2"], d["choose2"], d["P1"], d["P2"], d["depths"])
torch.cuda.synchronize()
ref = {k: v.cpu().numpy() for k, v in ref.items()}
nets = [AdaPoseNet(sd, dtype=dt), AdaPoseNet(sd, dtype=dt)]
side = [torch.cuda.Stream(), torch.cuda.Stream()]
need = nets[0].workspace_bytes(h) + 256
wss = [torch.empty(need, dtype=torch.uint8, device=dev) for _ in range(2)]


def split(handles, sync_between=False):
    f32 = dict(dtype=torch.float32, device=dev)
    shapes = {"nocs": (B, 1024, 3), "depth": (B, 1024), "r": (B, 3, 3), "t": (B, 3), "s": (B, 3)}
    out = {f"view{v}_{k}": torch.empty(*shp, **f32) for k, shp in shapes.items() for v in (1, 2)}
    cur = torch.cuda.current_stream()
    fork = torch.cuda.Event()
    fork.record(cur)
    for i in range(2):
        si, ws = side[i], wss[i]
        si.wait_event(fork)
        off = (-ws.data_ptr()) % 256
        sl = slice(i * h, (i + 1) * h)
        o = _lib.AdaposeOut(*[out[n][sl].data_ptr() for n, _ in _lib.AdaposeOut._fields_])
        _lib.check(lib.rgbm_adapose_forward_ex(handles[i], h, *[_lib.ptr(t[sl]) for t in args], C.c_void_p(ws.data_ptr() + off), ws.numel() - off,
                                               C.byref(o), 0, _lib.stream_ptr(si)), "rgbm_adapose_forward_ex")
        if sync_between:
            torch.cuda.synchronize()
    for si in side:
        j = torch.cuda.Event()
        j.record(si)
        cur.wait_event(j)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


for name, handles, sb in (("one handle, two streams", [nets[0]._h, nets[0]._h], False), ("two handles, two streams", [nets[0]._h, nets[1]._h], False),
                          ("one handle, host sync between the halves", [nets[0]._h, nets[0]._h], True)):
    bad, which = 0, set()
    for t in range(runs):
        o = split(handles, sb)
        diff = [k for k in ref if not np.array_equal(ref[k], o[k], equal_nan=True)]
        bad += bool(diff)
        which |= set(diff)
    print(f"{dt} {name}: {bad} of {runs} runs differ from the one-stream forward {sorted(which)}")
