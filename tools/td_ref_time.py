"""Time the TD path in the REFERENCE's operation order: K3 (objective) at several N and the device Powell on BASELINE config 3.
usage: python tools/td_ref_time.py [objective|powell|both]   (NLML_HPE_LIB selects an experiment build).  Development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights

OPS_PER_EVAL = 135 * 1404 * 5            # separately rounded f64 operations of one evaluation in the reference's order
PEAK_OPS = 39.3e12                       # f64 vector issue rate, non-fma (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz)
what = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = torch.device("cuda:0")
art = weights.load_tucker_artefacts("outputs/features")
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
tag = os.environ.get("NLML_HPE_LIB", "default")
if what in ("objective", "both"):
    for N in (256, 2048, 4096, 16384):
        P = torch.from_numpy(synth.tucker_params(N)).to(dev)
        X = torch.from_numpy(synth.features(N, 1404, 3)).to(dev)
        for _ in range(2): ops.tucker_objective(Wm, X, P, cp, order="reference")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.tucker_objective(Wm, X, P, cp, order="reference")
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        r = N / ms * 1e3
        print(f"{tag}: K3 reference order N={N}: {ms*1e3:.1f} us  {r/1e6:.2f} M evals/s  {r*OPS_PER_EVAL/PEAK_OPS:.3f} of the f64 vector issue rate", flush=True)
if what in ("powell", "both"):
    idx = synth.tucker_grid_indices(4096, seed=2)
    Xg = torch.from_numpy(synth.tucker_grid_faces(art, idx, 1e-3, seed=2)).to(dev)
    ops.tucker_powell(Wm, Xg[:64], cp, order="reference")
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        res = ops.tucker_powell(Wm, Xg, cp, order="reference")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nf = res["nfev"].double()
        r = float(nf.sum()) / dt
        print(f"{tag}: Powell reference order, 4096 faces: {dt*1e3:.1f} ms  {r/1e6:.1f} M face-evals/s ({r*OPS_PER_EVAL/PEAK_OPS:.3f} of the issue rate)  "
              f"mean nfev {float(nf.mean()):.0f} max {int(nf.max())}", flush=True)
