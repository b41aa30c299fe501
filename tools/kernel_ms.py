"""ms per forward of the in-library profiler's rows (batch 256, bf16 by default, every layer dense), for A/B between builds of the
library (RGBM_HIP_LIB).  usage: kernel_ms.py [dtype] [row-name-substring ...]   e.g.  kernel_ms.py bf16 conv0_sweep"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbmanip_amd import _lib, synth
from rgbmanip_amd.adapose import AdaPoseNet

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
pats = sys.argv[2:] or [""]
lib = _lib.load()
if os.environ.get("RGBM_DEBUG_FLAGS"):
    _lib.check(lib.rgbm_debug_flags(int(os.environ["RGBM_DEBUG_FLAGS"])))      # A/B of a debug-flag switch on one library
B = 256
inp = synth.adapose_inputs(16, seed=0)
inp = {k: torch.from_numpy(np.concatenate([v] * (B // 16), 0)).cuda() for k, v in inp.items()}
net = AdaPoseNet(synth.adapose_state_dict(seed=0), dtype=dtype, options={"sparse_dec": int(os.environ.get("SPARSE_DEC", "0"))})
run = lambda: net(inp["img1"], inp["choose1"], inp["img2"], inp["choose2"], inp["P1"], inp["P2"], inp["depths"])  # noqa: E731
run(); run()
torch.cuda.synchronize()
res = []
for rep in range(3):
    _lib.check(lib.rgbm_prof_start())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    st = (C.c_double * (4 * _lib.PROF_ROWS))()
    _lib.check(lib.rgbm_prof_stop(st))
    st = np.array(list(st)).reshape(_lib.PROF_ROWS, 4)
    res.append({_lib.PROF_KERNELS[v][0]: st[v][1] / 3 for v in range(_lib.PROF_ROWS) if st[v][0] > 0})
for name in res[0]:
    if any(p in name for p in pats):
        print(f"{os.path.basename(os.environ.get('RGBM_HIP_LIB', 'default')):40s} {dtype} {name[:60]:60s} " + " ".join(f"{r[name]:.3f}" for r in res))
