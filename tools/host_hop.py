"""Host-side cost of one K2 call at tick scale (64 faces), by call form: how many microseconds of Python / ctypes /
dispatcher sit in front of the launches.  Each form is called 100 times into an EMPTY stream (synchronised before), host wall clock
around the calls only (not the synchronise after): the enqueue cost per call; `gpu_us` = the same 100 calls including the wait, per call.
    python tools/host_hop.py [faces=64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nlml_hpe_amd import _lib, ops, synth, weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, seed=0)
blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.DEFAULT_MODE)).to(dev)
raw = torch.from_numpy(synth.raw_landmarks(B, seed=1)).to(dev)
out = torch.empty((B, 3), dtype=torch.float32, device=dev)
L = _lib.lib()
ws = ops._small_workspace(B, 1404, dev)
stream = torch.cuda.current_stream(dev).cuda_stream
args_small = (raw.data_ptr(), B, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), stream)
args_fused = (raw.data_ptr(), B, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, stream)
forms = {
    "C ABI nlml_landmarks_to_pose_small via ctypes, preallocated buffers (5 launches)": lambda: L.nlml_landmarks_to_pose_small(*args_small),
    "ops.landmarks_to_pose_small (Python wrapper: checks, torch.empty, device guard, ctypes)": lambda: ops.landmarks_to_pose_small(raw, blob, True),
    "C ABI nlml_landmarks_to_pose via ctypes, preallocated buffers (1 launch)": lambda: L.nlml_landmarks_to_pose(*args_fused),
    "ops.landmarks_to_pose (Python wrapper, 1 launch)": lambda: ops.landmarks_to_pose(raw, blob, True),
    "torch.ops.nlml_hpe.landmarks_to_pose (compiled op, csrc/torch_ops.cpp, 1 launch)": lambda: torch.ops.nlml_hpe.landmarks_to_pose(raw, blob, True),
}
forms["torch.ops.nlml_hpe.landmarks_to_pose_small (compiled op, explicit workspace, 5 launches)"] = lambda: torch.ops.nlml_hpe.landmarks_to_pose_small(raw, blob, True, ws)
forms["torch.ops.nlml_hpe.landmarks_to_pose_valid (compiled op: pose + face mask, 5 launches)"] = lambda: torch.ops.nlml_hpe.landmarks_to_pose_valid(raw, blob, True, ws)
N, REP = 100, 20
print(f"faces {B}, {REP} x {N} calls per form")
for name, fn in forms.items():
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    host, full = [], []
    for _ in range(REP):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) / N * 1e6)
        full.append((t2 - t0) / N * 1e6)
    print(f"{name:100s} host_us {np.median(host):7.2f}   with_wait_us {np.median(full):7.2f}")
# the whole tick (forward + video post) eager vs hipGraph replay
from nlml_hpe_amd.model import HIPPoseModel
from nlml_hpe_amd.video import GraphedTick, VideoPoseTracker
if B == 64:
    mdl = HIPPoseModel(sd, heads, device=dev)
    tr = VideoPoseTracker(mdl, 64, 1920, 1080)
    gt = GraphedTick(VideoPoseTracker(mdl, 64, 1920, 1080))
    for label, fn in (("tick eager (VideoPoseTracker.tick: two compiled ops)", lambda: tr.tick(raw)), ("tick hipGraph replay", lambda: gt.replay())):
        for _ in range(50):
            fn()
        host, full = [], []
        for _ in range(REP):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(N):
                fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append((t1 - t0) / N * 1e6)
            full.append((t2 - t0) / N * 1e6)
        print(f"{label:100s} host_us {np.median(host):7.2f}   with_wait_us {np.median(full):7.2f}")
