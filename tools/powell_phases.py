"""Per-phase cycles of a Powell round from the -DPW_STAMPS diagnostic build.  usage: NLML_HPE_LIB=exp_libs/pw_stamps.so python tools/powell_phases.py [fast|reference]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nlml_hpe_amd import ops, synth, weights
dev = torch.device("cuda:0")
art = weights.load_tucker_artefacts("outputs/features")
Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
idx = synth.tucker_grid_indices(64, seed=2)
Xg = torch.from_numpy(synth.tucker_grid_faces(art, idx, 1e-3, seed=2)).to(dev)
for copies in (1, 3, 16):
    X = Xg[:1].repeat(copies, 1).contiguous()
    if copies < 8:
        X = torch.cat([X, Xg[1:9 - copies]])        # pad the workgroup to >= 8 faces so that fval[0..7] exist (the extra faces finish early)
    res = ops.tucker_powell(Wm, X, cp, order=(sys.argv[1] if len(sys.argv) > 1 else "fast"))
    torch.cuda.synchronize()
    ph8 = res["fun"][:8].cpu().numpy()
    ph = ph8[:4]
    nf = int(res["nfev"].max())
    print(f"      inside the last phase (lane 0 = machine 0): objective value {ph8[4]/nf:6.0f}  state machine call {ph8[5]/nf:6.0f}  trial point -> LDS {ph8[6]/nf:6.0f}  barrier {ph8[3]/nf:6.0f}")
    print(f"{copies:2d} long-lived machines, {nf} rounds: cycles per round: live-mask {ph[0]/nf:7.0f}  coefficients {ph[1]/nf:7.0f}  evaluation {ph[2]/nf:7.0f}  state machines + barrier {ph8[3:7].sum()/nf:7.0f}  total {(ph8[:4].sum()+ph8[4:7].sum())/nf:7.0f}")
