"""How fast can [N,480,640,3] float64 host frames reach HBM?  (a) pinned staging by a thread pool (what estimate() does), (b) the
user's array page-locked in place (hipHostRegister) and copied directly, (c) a plain pageable copy.  usage: check_host_register.py"""
import ctypes
import time

import numpy as np
import torch

n = 128
a = np.random.rand(n, 480, 640, 3)          # 0.94 GB, touched
dev = torch.device("cuda", 0)
out = torch.empty(a.shape, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
rt = torch.cuda.cudart()
t = torch.from_numpy(a)
for rep in range(2):
    t0 = time.perf_counter(); out.copy_(t); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"pageable copy: {(t1 - t0) * 1e3:.1f} ms = {a.nbytes / (t1 - t0) / 1e9:.1f} GB/s")
for rep in range(3):
    t0 = time.perf_counter()
    rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
    t1 = time.perf_counter()
    pinned = t.is_pinned()
    out.copy_(t, non_blocking=True); torch.cuda.synchronize()
    t2 = time.perf_counter()
    rc2 = rt.cudaHostUnregister(a.ctypes.data)
    t3 = time.perf_counter()
    print(f"register rc={int(rc)} {(t1 - t0) * 1e3:.1f} ms, is_pinned={pinned}, copy {(t2 - t1) * 1e3:.1f} ms = {a.nbytes / (t2 - t1) / 1e9:.1f} GB/s, "
          f"unregister rc={int(rc2)} {(t3 - t2) * 1e3:.1f} ms")
pin = torch.empty(a.shape, dtype=torch.float64, pin_memory=True)
for rep in range(2):
    t0 = time.perf_counter(); out.copy_(pin, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"pinned copy: {(t1 - t0) * 1e3:.1f} ms = {a.nbytes / (t1 - t0) / 1e9:.1f} GB/s")
from concurrent.futures import ThreadPoolExecutor
pn = pin.numpy()
for th in (8, 16, 32, 64):
    pool = ThreadPoolExecutor(th)
    parts = [(n * i // th, n * (i + 1) // th) for i in range(th)]
    t0 = time.perf_counter(); list(pool.map(lambda p: np.copyto(pn[p[0]:p[1]], a[p[0]:p[1]]), parts)); t1 = time.perf_counter()
    print(f"staging memcpy, {th} threads: {(t1 - t0) * 1e3:.1f} ms = {a.nbytes / (t1 - t0) / 1e9:.1f} GB/s")
