"""How fast can the host turn float64 frames into float32 in a pinned staging buffer?  (numpy's casting copy runs at ~1 GB/s per thread; torch's
CPU copy_ kernel is vectorised and parallel.)  Measured on the box that runs it.  usage: host_convert_bw.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
n = 32
src = np.random.default_rng(0).random((n, 480, 640, 3))                # float64, 236 MB
ts = torch.from_numpy(src)
pin32 = torch.empty((n, 480, 640, 3), dtype=torch.float32, pin_memory=True)
pin64 = torch.empty((n, 480, 640, 3), dtype=torch.float64, pin_memory=True)
print("cpus", os.cpu_count(), "torch threads", torch.get_num_threads())
def timeit(f, reps=5):
    f(); t = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t) / reps
gb = src.nbytes / 1e9
for th in (1, 4, 8, 16, 32):
    torch.set_num_threads(th)
    t = timeit(lambda: pin32.copy_(ts))
    print(f"torch copy_ f64 -> pinned f32, {th:2d} torch threads: {t * 1e3:7.2f} ms = {gb / t:6.1f} GB/s of float64 read")
torch.set_num_threads(1)
for workers in (4, 8, 16):
    pool = ThreadPoolExecutor(workers)
    parts = [(i * n // workers, (i + 1) * n // workers) for i in range(workers)]
    t = timeit(lambda: list(pool.map(lambda p: pin32[p[0]:p[1]].copy_(ts[p[0]:p[1]]), parts)))
    print(f"thread pool of {workers:2d} x torch copy_ (1 torch thread each) f64 -> pinned f32: {t * 1e3:7.2f} ms = {gb / t:6.1f} GB/s")
    t = timeit(lambda: list(pool.map(lambda p: np.copyto(pin64.numpy()[p[0]:p[1]], src[p[0]:p[1]]), parts)))
    print(f"thread pool of {workers:2d} x np.copyto f64 -> pinned f64 (what ships):              {t * 1e3:7.2f} ms = {gb / t:6.1f} GB/s")
    t = timeit(lambda: list(pool.map(lambda p: np.copyto(pin32.numpy()[p[0]:p[1]], src[p[0]:p[1]], casting="same_kind"), parts)), reps=2)
    print(f"thread pool of {workers:2d} x np.copyto f64 -> pinned f32 (numpy cast):               {t * 1e3:7.2f} ms = {gb / t:6.1f} GB/s")
