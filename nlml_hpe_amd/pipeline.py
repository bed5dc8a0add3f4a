"""Host-resident landmarks -> poses with copies overlapped with compute.

The ABI works on device buffers; when a caller's landmarks live in host memory (an .npz of a whole
validation set, frames arriving from decoders) the batches are double-buffered: while the fused kernel
runs batch i on the compute stream, batch i+1 is copied host->device on a copy stream, and the [B,3] poses
are copied back on a third stream.  PCIe (~63 GB/s spec, 57 GB/s measured from pinned memory) caps
this path at about 10 M faces/s (5.6 KB per face), below the kernel's rate, so the overlap matters.

Where the bytes are copied from (round 5, tools/host_stage_timeline.py / host_copy_probe.py on the GPU box): the staged form -- eight threads
copying pageable -> pinned at 85-100 GB/s while the DMA of the previous batch reads the other pinned buffer at 57 GB/s -- runs at 54 GB/s
end to end once staging and DMA really overlap (rounds 2-4 serialised them through an event behind the next batch's H2D: 33-38 GB/s).
Page-locking the CALLER'S array in place (hipHostRegister) and letting the DMA read it directly is as fast (56 GB/s) and needs no CPU
copy, but the registration costs what the memory's pages make it cost: 2.7 ms per 1.5 GB on huge pages (numpy's mmap'd arrays), 55 ms
on 4-KB pages (torch's CPU allocator).  `inplace="auto"` (the default) registers the first batch's range and goes on in place only if
that ran at >= 200 GB/s; a refused registration (read-only mapping, a range registered already) also means the staged form.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


class HostPipeline:
    def __init__(self, model, batch: int = 8192, normalize: bool = True, workers: int | None = None):
        # (8,192 faces = 46 MB per batch: 0.8 ms of DMA against 0.15 ms of kernel, and a 65,536-face call already has eight batches to overlap:
        # 8.7 M faces/s where one 65,536-face batch -- nothing to overlap -- gives 5.6 M; tools/host_pipeline_sweep.py)
        self.model, self.batch, self.normalize = model, int(batch), normalize
        dev = model.device
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.out_stream = torch.cuda.Stream(device=dev)     # poses back to the host: not behind the next batch's landmarks
        self.pin_in = None                      # the staging buffers of the staged form: allocated when that form first runs
        self.dev_in = [torch.empty((self.batch, 468, 3), dtype=torch.float32, device=dev) for _ in range(2)]
        self.pin_out = [torch.empty((self.batch, 3), dtype=torch.float32).pin_memory() for _ in range(2)]
        self.pin_valid = [torch.empty((self.batch,), dtype=torch.bool).pin_memory() for _ in range(2)]
        # the pageable -> pinned staging copy is a plain memcpy; one thread moves ~5 GB/s, so it is split over a few
        # threads (numpy releases the GIL while copying) to keep up with the PCIe link
        self.workers = int(workers) if workers else max(1, min(8, (os.cpu_count() or 2) // 2))
        self.pool = ThreadPoolExecutor(self.workers)
        self.pin_in_np = None
        self.last_mode = None

    def _ensure_staging(self):
        if self.pin_in is None:
            self.pin_in = [torch.empty((self.batch, 468, 3), dtype=torch.float32).pin_memory() for _ in range(2)]
            self.pin_in_np = [t.numpy() for t in self.pin_in]

    # ---- in-place form: the caller's array page-locked where it lies
    @staticmethod
    def _register(arr: np.ndarray) -> bool:
        try:
            rc = torch.cuda.cudart().cudaHostRegister(arr.ctypes.data, arr.nbytes, 0)
            return int(rc) == 0
        except Exception:   # noqa: BLE001 -- (a runtime that refuses raises instead of returning a code)
            return False

    @staticmethod
    def _unregister(arr: np.ndarray) -> None:
        try:
            torch.cuda.cudart().cudaHostUnregister(arr.ctypes.data)
        except Exception:   # noqa: BLE001
            pass

    def run(self, raw: np.ndarray, inplace: str | bool = "auto"):
        """raw f32[N,468,3] (host) -> (pose f32[N,3] radians, valid bool[N]) as numpy arrays.
        inplace: "auto" (page-lock the caller's array in place when that is fast, see the module docstring), True (always try), False (staged)."""
        import time
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        n = raw.shape[0]
        self.last_mode = "staged"
        registered = []
        if inplace and n > 0 and raw.flags.writeable:
            m0 = min(self.batch, n)
            t0 = time.perf_counter()
            first = raw[:m0]
            if self._register(first):
                dt = time.perf_counter() - t0
                fast = inplace is True or first.nbytes / max(dt, 1e-9) >= 200e9
                if fast and m0 == n:
                    registered.append(first)
                    self.last_mode = "in place"
                else:
                    self._unregister(first)          # (two ranges that meet inside a page cannot both be registered: one call for the whole array)
                    if fast and self._register(raw):
                        registered.append(raw)
                        self.last_mode = "in place"
        try:
            return self._run(raw, n, self.last_mode == "in place")
        finally:
            if registered:
                torch.cuda.synchronize(self.model.device)
                for a in registered:
                    self._unregister(a)

    def _run(self, raw: np.ndarray, n: int, in_place: bool):
        if not in_place:
            self._ensure_staging()
        pose = np.empty((n, 3), np.float32)
        valid = np.empty((n,), bool)
        compute = torch.cuda.current_stream(self.model.device)
        starts = list(range(0, n, self.batch))
        h2d_done = [torch.cuda.Event() for _ in starts]
        k_done = [torch.cuda.Event() for _ in starts]
        d2h_done = [torch.cuda.Event() for _ in starts]

        def stage(i):   # host copy into pinned memory, then async H2D on the copy stream
            b0 = starts[i]
            m = min(self.batch, n - b0)
            slot = i & 1
            # (Round 5: until then this waited on the host for d2h_done[i - 2] -- an event that sat BEHIND the next batch's H2D on the one copy
            # stream, so the staging copy of batch i + 1 only began when the H2D of batch i had ended: staging and DMA ran one after the
            # other, 33-38 GB/s.  Now the host waits only for the pinned slot's own previous H2D, the device buffer's release is a
            # stream-side wait, and the outputs travel on a stream of their own: 54 GB/s.)
            if i >= 2 and not in_place:
                h2d_done[i - 2].synchronize()      # the pinned slot's previous contents have been read by the DMA
            if in_place:   # DMA straight out of the caller's (now page-locked) array
                src = torch.from_numpy(raw[b0:b0 + m])
            else:
                step = (m + self.workers - 1) // self.workers
                futs = [self.pool.submit(np.copyto, self.pin_in_np[slot][a:min(a + step, m)], raw[b0 + a:b0 + min(a + step, m)])
                        for a in range(0, m, step)]
                for f in futs:
                    f.result()
                src = self.pin_in[slot][:m]
            with torch.cuda.stream(self.copy_stream):
                if i >= 2:
                    self.copy_stream.wait_event(k_done[i - 2])   # the device slot is no longer being read by batch i - 2's kernel
                self.dev_in[slot][:m].copy_(src, non_blocking=True)
                h2d_done[i].record(self.copy_stream)

        outs = [None] * len(starts)
        if starts:
            stage(0)
        for i, b0 in enumerate(starts):
            m = min(self.batch, n - b0)
            slot = i & 1
            if i + 1 < len(starts):
                stage(i + 1)                        # overlaps with the kernel of batch i
            compute.wait_event(h2d_done[i])
            p, v = self.model.from_landmarks(self.dev_in[slot][:m], self.normalize, return_valid=True)
            k_done[i].record(compute)
            outs[i] = (p, v)
            with torch.cuda.stream(self.out_stream):
                self.out_stream.wait_event(k_done[i])
                self.pin_out[slot][:m].copy_(p, non_blocking=True)
                self.pin_valid[slot][:m].copy_(v, non_blocking=True)
                d2h_done[i].record(self.out_stream)
            if i >= 1:                              # drain batch i-1 while batch i runs
                j = i - 1
                d2h_done[j].synchronize()
                mj = min(self.batch, n - starts[j])
                pose[starts[j]:starts[j] + mj] = self.pin_out[j & 1][:mj].numpy()
                valid[starts[j]:starts[j] + mj] = self.pin_valid[j & 1][:mj].numpy()
        if starts:
            j = len(starts) - 1
            d2h_done[j].synchronize()
            mj = min(self.batch, n - starts[j])
            pose[starts[j]:starts[j] + mj] = self.pin_out[j & 1][:mj].numpy()
            valid[starts[j]:starts[j] + mj] = self.pin_valid[j & 1][:mj].numpy()
        return pose, valid
