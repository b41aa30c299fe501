"""CPU emulation of reduced-precision operand modes through the oracle network (design experiment, not product, not a test).

Every conv of oracle/adapose_ref.py is replaced by an emulation of `fp32 storage, 16-bit split MFMA operands`:
    conv(x, w) ~= conv(xh, wh) + conv(xl, wh) + conv(xh, wl)       (x = xh + xl, w = wh + wl, 16-bit halves, fp32 accumulate)
and the 10 network outputs are compared with the reference golden (tests/golden/adapose_b2.npz).  Answers, before any
kernel is written: which operand format (bf16 / fp16, with or without denormals) keeps the outputs inside 1e-4, and which
layers tolerate single-term operands.    usage: python tools/split_emulation.py [mode ...]
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import adapose_ref  # noqa: E402
from rgbmanip_amd import synth  # noqa: E402

F16_MIN_NORMAL = 6.103515625e-05


def q_f16(x, ftz):
    h = x.clamp(-65504.0, 65504.0).half().float()
    if ftz:
        h = torch.where(h.abs() < F16_MIN_NORMAL, torch.zeros_like(h), h)
    return h


def q_bf16(x, ftz):
    return x.bfloat16().float()


def split(x, q, ftz, terms):
    h = q(x, ftz)
    if terms == 1:
        return h, None
    return h, q(x - h, ftz)


class Emu:
    def __init__(self, fmt="f16", ftz=False, x_terms=2, w_terms=2, only=None, skip=None):
        self.q = q_f16 if fmt == "f16" else q_bf16
        self.ftz, self.xt, self.wt = ftz, x_terms, w_terms
        self.calls = 0
        self.layer_terms = only            # {weight shape: (x_terms, w_terms)} for the layers that deviate from the default

    def op(self, fn, x, w, b, *a, **k):
        self.calls += 1
        xt, wt = self.xt, self.wt
        if self.layer_terms is not None and tuple(w.shape) in self.layer_terms:     # per-layer override, keyed by weight shape
            xt, wt = self.layer_terms[tuple(w.shape)]
        xh, xl = split(x, self.q, self.ftz, xt)
        wh, wl = split(w, self.q, self.ftz, wt)
        y = fn(xh, wh, b, *a, **k)
        if xl is not None:
            y = y + fn(xl, wh, None, *a, **k)
        if wl is not None:
            y = y + fn(xh, wl, None, *a, **k)
        return y


def run(name, emu2d=None, emu3d=None, emu1d=None):
    sd = adapose_ref.to_torch_sd(synth.adapose_state_dict(seed=0))
    gold = np.load(os.path.join(ROOT, "tests", "golden", "adapose_b2.npz"))
    inp = synth.adapose_inputs(2, seed=0)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    o2d, o3d, ot3d, o1d, olin = F.conv2d, F.conv3d, F.conv_transpose3d, F.conv1d, F.linear
    try:
        if emu2d is not None:
            F.conv2d = lambda x, w, b=None, *a, **k: emu2d.op(o2d, x, w, b, *a, **k)
        if emu3d is not None:
            F.conv3d = lambda x, w, b=None, *a, **k: emu3d.op(o3d, x, w, b, *a, **k)
            F.conv_transpose3d = lambda x, w, b=None, *a, **k: emu3d.op(ot3d, x, w, b, *a, **k)
        if emu1d is not None:
            F.conv1d = lambda x, w, b=None, *a, **k: emu1d.op(o1d, x, w, b, *a, **k)
        t0 = time.time()
        out = adapose_ref.adapose_forward(sd, t["img1"], t["choose1"], t["img2"], t["choose2"], t["P1"], t["P2"], t["depths"])
    finally:
        F.conv2d, F.conv3d, F.conv_transpose3d, F.conv1d, F.linear = o2d, o3d, ot3d, o1d, olin
    errs = {}
    for k, v in out.items():
        g = gold[k] if k in gold.files else gold["out_" + k]
        errs[k] = float(np.abs(v.numpy().astype(np.float64) - g).max() / max(np.abs(g).max(), 1e-12))
    grp = {"nocs": max(errs["view1_nocs"], errs["view2_nocs"]), "depth": max(errs["view1_depth"], errs["view2_depth"]),
           "r": max(errs["view1_r"], errs["view2_r"]), "t": max(errs["view1_t"], errs["view2_t"]), "s": max(errs["view1_s"], errs["view2_s"])}
    print(f"{name:44s} " + " ".join(f"{k}={v:.2e}" for k, v in grp.items()) + f"   worst={max(grp.values()):.2e}  ({time.time() - t0:.0f}s)", flush=True)


MODES = {
    "fp32":            lambda: run("fp32 oracle"),
    "f16x1":           lambda: run("f16 single term, all convs", Emu("f16", False, 1, 1), Emu("f16", False, 1, 1)),
    "bf16x3":          lambda: run("bf16 split x3, all convs", Emu("bf16"), Emu("bf16")),
    "f16x3":           lambda: run("f16 split x3 (denormals kept), all convs", Emu("f16"), Emu("f16")),
    "f16x3_ftz":       lambda: run("f16 split x3 (denormals flushed), all convs", Emu("f16", True), Emu("f16", True)),
    "f16x3_3d1":       lambda: run("f16x3 2-D convs, f16 single-term 3-D convs", Emu("f16"), Emu("f16", False, 1, 1)),
    "f16x3_3dx1":      lambda: run("f16x3 2-D, 3-D: x single-term, w split", Emu("f16"), Emu("f16", False, 1, 2)),
    # two products per value pair instead of three: one operand split, the other a single 16-bit term
    "f16_xs_w1":       lambda: run("f16: x split, w single term, all convs", Emu("f16", False, 2, 1), Emu("f16", False, 2, 1)),
    "f16_x1_ws":       lambda: run("f16: x single term, w split, all convs", Emu("f16", False, 1, 2), Emu("f16", False, 1, 2)),
    "bf16_xs_w1":      lambda: run("bf16: x split, w single term, all convs", Emu("bf16", False, 2, 1), Emu("bf16", False, 2, 1)),
    "f16_xs_w1_2d":    lambda: run("f16: 2-D x split / w single, 3-D f16x3", Emu("f16", False, 2, 1), Emu("f16")),
    "f16_2d1_3dx3":    lambda: run("2-D f16 single term (x and w), 3-D f16x3", Emu("f16", False, 1, 1), Emu("f16")),
    "bf16x3_2df16":    lambda: run("2-D f16 single term (x and w), 3-D bf16x3", Emu("f16", False, 1, 1), Emu("bf16")),
    "f16x3_2dx1":      lambda: run("2-D: x single-term w split, 3-D f16x3", Emu("f16", False, 1, 2), Emu("f16")),
}

# Round 3: per-layer term count — bf16 split pairs everywhere, TWO products (one operand single-term) only in
# the layers of one group: layer4 = the five 512 -> 512 3x3 convs, layer3 = the eleven 256 -> 256 3x3 convs, up_1 = 1024 -> 256
L4, L3, UP1 = (512, 512, 3, 3), (256, 256, 3, 3), (256, 1024, 3, 3)
for nm, shp in (("layer4", L4), ("layer3", L3), ("up_1", UP1)):
    MODES[f"bf16x3_{nm}_w1"] = (lambda shp=shp, nm=nm: run(f"bf16x3, {nm}: w single term (2 products)", Emu("bf16", only={shp: (2, 1)}), Emu("bf16")))
    MODES[f"bf16x3_{nm}_x1"] = (lambda shp=shp, nm=nm: run(f"bf16x3, {nm}: x single term (2 products)", Emu("bf16", only={shp: (1, 2)}), Emu("bf16")))
MODES["bf16x3_one_l4_w1"] = lambda: run("bf16x3, ONE layer4 conv (layer4.1.conv1): w single term", _OneCall("bf16", L4, (2, 1)), Emu("bf16"))


class _OneCall(Emu):
    """two products in the FIRST conv whose weight has the given shape only"""
    def __init__(self, fmt, shape, terms):
        super().__init__(fmt)
        self.shape, self.terms, self.done = shape, terms, False

    def op(self, fn, x, w, b, *a, **k):
        if tuple(w.shape) == self.shape and not self.done:
            self.done = True
            self.layer_terms = {self.shape: self.terms}
        else:
            self.layer_terms = None
        return super().op(fn, x, w, b, *a, **k)


if __name__ == "__main__":
    torch.set_num_threads(8)
    for m in (sys.argv[1:] or list(MODES)):
        MODES[m]()
