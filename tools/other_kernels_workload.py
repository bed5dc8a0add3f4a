"""The kernels that the K2 bench command does not launch, as profiling workloads:
    rocprofv3 ... -- python3 tools/other_kernels_workload.py <which>
which = bf16    encoder_heads_bf16_w8_kernel, fused landmarks->pose, 65,536 faces, 30 + 100 launches (BASELINE config 2's named dtype)
        small   the five layer-per-launch kernels (prepass, layer x3, tail) at 64 and at 2,000 faces, 30 + 200 steps each, default mode
        video   video_post_kernel behind the 64-face layer-per-launch forward: 64 streams, 300 ticks (BASELINE config 5)
        k1      normalize_ipd_kernel, 65,536 faces, 10 + 100 launches"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nlml_hpe_amd import _lib, ops, synth, weights

which = sys.argv[1]
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
sd = synth.encoder_state_dict(1404, seed=0)
raw = torch.from_numpy(synth.raw_landmarks(65536, seed=1)).to(dev)
if which == "bf16":
    blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.MODE_BF16)).to(dev)
    for _ in range(130):
        ops.landmarks_to_pose(raw, blob, True)
elif which == "small":
    blob = torch.from_numpy(weights.pack_blob(sd, heads, _lib.DEFAULT_MODE)).to(dev)
    for n in (64, 2000):
        r = raw[:n].contiguous()
        for _ in range(230):
            ops.landmarks_to_pose_small(r, blob, True)
        torch.cuda.synchronize()
elif which == "video":
    from nlml_hpe_amd.model import HIPPoseModel
    from nlml_hpe_amd.video import VideoPoseTracker
    mdl = HIPPoseModel(sd, heads, device=dev)
    tr = VideoPoseTracker(mdl, 64, 1920, 1080)
    clips = raw[:64 * 8].reshape(8, 64, 468, 3)
    for t in range(300):
        tr.tick(clips[t % 8])
elif which == "k1":
    for _ in range(110):
        ops.normalize_ipd(raw, True)
else:
    raise SystemExit(__doc__)
torch.cuda.synchronize()
print("done", which)
