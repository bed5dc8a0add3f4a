"""Time K3 (tucker objective) alone; NLML_HPE_LIB selects an experiment build.  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights

dev = torch.device("cuda:0")
art = weights.load_tucker_artefacts("outputs/features")
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
for N in (256, 2048, 4096, 8192, 16384, 65536):
    P = torch.from_numpy(synth.tucker_params(N)).to(dev)
    X = torch.from_numpy(synth.features(N, 1404, 3)).to(dev)
    for _ in range(3): ops.tucker_objective(Wm, X, P, cp, order="fast")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.tucker_objective(Wm, X, P, cp, order="fast")
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{os.environ.get('NLML_HPE_LIB','default')}: K3 N={N}: {ms*1e3:.1f} us  {N/ms*1e3/1e6:.2f} Mevals/s  {N*383.7e3/ms/1e9/78.6*100:.1f}% f64 peak", flush=True)
