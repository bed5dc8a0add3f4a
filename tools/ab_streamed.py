"""A/B of the trunk + streamed-tail path across builds on ONE box: per build (NLML_HPE_LIB, "-" = in-tree) a child process times the fused
kernel and the streamed path alternately at 65,536 faces.  usage: python tools/ab_streamed.py <rounds> libA.so libB.so ..."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from nlml_hpe_amd import _lib, ops, synth, weights
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(%r, "models"))
blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(1404, seed=0), heads, _lib.MODE_F16X2S)).to(dev)
raw = torch.from_numpy(synth.raw_landmarks(65536, seed=1)).to(dev)
def t(fn, n=150, w=60):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
f = [t(lambda: ops.landmarks_to_pose(raw, blob, True)) for _ in range(1)]
s = [t(lambda: ops.landmarks_to_pose_streamed(raw, blob, True)) for _ in range(1)]
f.append(t(lambda: ops.landmarks_to_pose(raw, blob, True))); s.append(t(lambda: ops.landmarks_to_pose_streamed(raw, blob, True)))
print("RES %%.4f %%.4f" %% (min(f), min(s)))
''' % (ROOT, ROOT)
rounds, libs = int(sys.argv[1]), sys.argv[2:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        if l != "-":
            env["NLML_HPE_LIB"] = os.path.abspath(l)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        m = re.search(r"RES ([\d.]+) ([\d.]+)", out.stdout)
        if not m:
            print(l, "FAILED", out.stdout[-300:], out.stderr[-300:], flush=True)
            continue
        f, s = float(m.group(1)), float(m.group(2))
        res[l].append((f, s))
        print(f"round {r} {l}: fused {f:.4f} ms, streamed {s:.4f} ms  ({100 * (f / s - 1):+.2f} %)", flush=True)
