"""HostPipeline (host numpy -> pinned staging -> H2D on a copy stream -> fused kernel -> D2H) over batch size and staging threads:
faces/s and GB/s host to device, PCIe-inclusive.  usage: host_pipeline_sweep.py [faces]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import synth, weights
from nlml_hpe_amd.model import HIPPoseModel
from nlml_hpe_amd.pipeline import HostPipeline

N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
mdl = HIPPoseModel(synth.encoder_state_dict(1404, 0), heads, device=dev)
base = synth.raw_landmarks(65536, 1)
raw = np.concatenate([base] * (N // 65536), axis=0) if N > 65536 else base[:N]
print(f"cpu_count {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}, faces {N} ({raw.nbytes / 1e9:.2f} GB)", flush=True)
# plain pinned H2D rate for reference
pin = torch.empty((65536, 468, 3), dtype=torch.float32).pin_memory(); d = torch.empty_like(pin, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): d.copy_(pin, non_blocking=True)
torch.cuda.synchronize(); print(f"pinned -> device alone: {5 * pin.numel() * 4 / (time.perf_counter() - t0) / 1e9:.1f} GB/s", flush=True)
a = np.empty_like(base); t0 = time.perf_counter(); np.copyto(a, base); print(f"one-thread host memcpy: {base.nbytes / (time.perf_counter() - t0) / 1e9:.1f} GB/s", flush=True)
for batch in (8192, 16384, 65536):
    for workers, inplace in ((8, False), (16, False), (8, "auto")):
        pipe = HostPipeline(mdl, batch=batch, workers=workers)
        pipe.run(raw[:2 * batch], inplace=inplace)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); pipe.run(raw, inplace=inplace); best = min(best, time.perf_counter() - t0)
        print(f"batch {batch:6d} workers {workers:2d} inplace={inplace!s:5s} ({pipe.last_mode}): {N / best / 1e6:6.2f} M faces/s  {raw.nbytes / best / 1e9:5.1f} GB/s", flush=True)
        del pipe
