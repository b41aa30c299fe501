"""Micro-benchmark of the 3-D conv kernels at network shapes (one 32-view chunk), with ablation flags
(1 = skip staging/blend work, 2 = skip the MFMA phase).  Prints conv0 (prof rows 10/11) and conv1..11 (rows 8/9)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rgbmanip_amd import _lib, synth
from rgbmanip_amd.adapose import AdaPoseNet
lib = _lib.load()
B = 16
ci = int(sys.argv[1]) if len(sys.argv) > 1 else 3
net = AdaPoseNet(synth.adapose_state_dict(seed=0), dtype="bf16", cost_impl=ci)
inp = synth.adapose_inputs(4, seed=0)
inp = {k: np.concatenate([v] * (B // 4), 0) for k, v in inp.items()}
def run():
    return net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"])
import ctypes as C
for flags in (0,):
    lib.rgbm_debug_flags(flags)
    run(); torch.cuda.synchronize()
    lib.rgbm_prof_start()
    for _ in range(3): run()
    torch.cuda.synchronize()
    st = (C.c_double * (4 * _lib.PROF_ROWS))(); lib.rgbm_prof_stop(st)
    st = np.array(list(st)).reshape(_lib.PROF_ROWS, 4)
    print(f"cost_impl {ci} flags {flags}: conv0 {(st[11,1]+st[14,1])/3:.3f} ms   conv1..11 {st[17:26,1].sum()/3:.3f} ms  (B={B}: 1 chunk of {2*B} views)")
lib.rgbm_debug_flags(0)
