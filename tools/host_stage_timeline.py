"""The staged host -> device path without the kernel: per batch, how long the threaded pageable -> pinned copy takes and how long the DMA takes
(HIP events on the copy stream), when they run one after the other and when they overlap as in HostPipeline.  usage: host_stage_timeline.py [batch] [threads]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
nb = 16
src = np.random.default_rng(0).random((nb * batch, 468, 3), dtype=np.float32)
src_t = torch.from_numpy(src).clone().numpy()          # a second copy in torch-allocated memory (4-KB pages, typically)
pin = [torch.empty((batch, 468, 3), dtype=torch.float32).pin_memory() for _ in range(2)]
for p in pin: p.zero_()
pin_np = [p.numpy() for p in pin]
d = [torch.empty((batch, 468, 3), dtype=torch.float32, device=dev) for _ in range(2)]
pool = ThreadPoolExecutor(workers)
cs = torch.cuda.Stream()
GB = batch * 5616 / 1e9
def stage(slot, a):
    step = (batch + workers - 1) // workers
    futs = [pool.submit(np.copyto, pin_np[slot][o:o + step], a[o:o + step]) for o in range(0, batch, step)]
    for f in futs: f.result()
for name, arr in (("numpy (mmap, huge pages likely)", src), ("torch CPU allocator", src_t)):
    # serial: stage, then DMA, wait
    ts, td = [], []
    for i in range(nb):
        t0 = time.perf_counter(); stage(i & 1, arr[i * batch:(i + 1) * batch]); ts.append(time.perf_counter() - t0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(cs):
            e0.record(cs); d[i & 1].copy_(pin[i & 1], non_blocking=True); e1.record(cs)
        cs.synchronize(); td.append(e0.elapsed_time(e1) / 1e3)
    print(f"{name}: serial: staging {GB / np.median(ts):6.1f} GB/s, DMA {GB / np.median(td):6.1f} GB/s", flush=True)
    # overlapped as in HostPipeline: stage(i+1) while DMA(i) runs
    evs = []
    t0 = time.perf_counter()
    stage(0, arr[:batch])
    ts = []
    for i in range(nb):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(cs):
            e0.record(cs); d[i & 1].copy_(pin[i & 1], non_blocking=True); e1.record(cs)
        evs.append((e0, e1))
        if i + 1 < nb:
            if i >= 1: evs[i - 1][1].synchronize()     # the slot's previous DMA has read the buffer
            t1 = time.perf_counter(); stage((i + 1) & 1, arr[(i + 1) * batch:(i + 2) * batch]); ts.append(time.perf_counter() - t1)
    cs.synchronize()
    tot = time.perf_counter() - t0
    td = [a.elapsed_time(b) / 1e3 for a, b in evs]
    print(f"{name}: overlapped: {nb * GB / tot:6.1f} GB/s end to end; staging {GB / np.median(ts):6.1f} GB/s, DMA {GB / np.median(td):6.1f} GB/s while the other runs", flush=True)
    # in place
    rt = torch.cuda.cudart()
    t0 = time.perf_counter(); rc = rt.cudaHostRegister(arr.ctypes.data, arr.nbytes, 0); treg = time.perf_counter() - t0
    at = torch.from_numpy(arr)
    t0 = time.perf_counter()
    with torch.cuda.stream(cs):
        for i in range(nb): d[i & 1].copy_(at[i * batch:(i + 1) * batch], non_blocking=True)
    cs.synchronize()
    print(f"{name}: in place: register {treg * 1e3:.1f} ms ({arr.nbytes / treg / 1e9:.0f} GB/s, rc={int(rc)}), DMA {nb * GB / (time.perf_counter() - t0):6.1f} GB/s", flush=True)
    rt.cudaHostUnregister(arr.ctypes.data)
