"""Where the host-resident path's time goes: multi-threaded pageable -> pinned staging rate by thread count (alone and while a DMA runs),
the DMA rate alone, and the alternative of page-locking the caller's array in place (hipHostRegister) and DMA-ing straight from it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
dev = torch.device("cuda:0")
n = 65536
src = np.random.default_rng(0).random((n, 468, 3), dtype=np.float32)
pin = torch.empty((n, 468, 3), dtype=torch.float32).pin_memory(); pin_np = pin.numpy()
pin2 = torch.empty((n, 468, 3), dtype=torch.float32).pin_memory()
d = torch.empty((n, 468, 3), dtype=torch.float32, device=dev)
GB = src.nbytes / 1e9
def staged(workers, reps=5):
    pool = ThreadPoolExecutor(workers)
    step = (n + workers - 1) // workers
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        futs = [pool.submit(np.copyto, pin_np[a:a + step], src[a:a + step]) for a in range(0, n, step)]
        for f in futs: f.result()
        best = min(best, time.perf_counter() - t0)
    pool.shutdown()
    return GB / best
for w in (1, 2, 4, 8, 16, 32):
    print(f"staging copy, {w:2d} threads: {staged(w):6.1f} GB/s", flush=True)
s = torch.cuda.Stream()
def dma_loop(k):
    with torch.cuda.stream(s):
        for _ in range(k): d.copy_(pin2, non_blocking=True)
torch.cuda.synchronize(); t0 = time.perf_counter(); dma_loop(8); s.synchronize(); print(f"DMA alone: {8 * GB / (time.perf_counter() - t0):.1f} GB/s", flush=True)
for w in (4, 8, 16):
    dma_loop(40); r = staged(w, reps=3); s.synchronize()
    print(f"staging copy, {w:2d} threads, while a DMA runs: {r:6.1f} GB/s", flush=True)
# page-lock the caller's array in place
rt = torch.cuda.cudart()
t0 = time.perf_counter(); rc = rt.cudaHostRegister(src.ctypes.data, src.nbytes, 0); t_reg = time.perf_counter() - t0
print(f"hipHostRegister of {GB:.2f} GB: rc={rc} {t_reg * 1e3:.1f} ms ({GB / t_reg:.1f} GB/s)", flush=True)
ts = torch.from_numpy(src)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): d.copy_(ts, non_blocking=True)
torch.cuda.synchronize(); print(f"DMA from the registered array: {5 * GB / (time.perf_counter() - t0):.1f} GB/s (is_pinned={ts.is_pinned()})", flush=True)
t0 = time.perf_counter(); rt.cudaHostUnregister(src.ctypes.data); print(f"unregister: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
