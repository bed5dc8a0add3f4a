"""Parity at the reference's operating range on a sample 64 times FX3c's: 1,048,576 Philox faces through FX3b's weights (latent over the
rows of U_yaw / U_pitch / U_roll, poses over the trained +-50 / 40 / 30 degree bins), per face max |pose - f64 truth| for the three K2
parity modes on the device and for the reference's arithmetic on the host (the oracle's torch-CPU f32 restatement, batched as
NLML_HPE_Test.py would be with a batch of 16,384), and kernel against that reference.  The f64 truth is the oracle's numpy forward.
Uses oracle/ as the checker: a measurement tool, not product code.   python tools/parity_soak.py [faces=1048576] > profiles/r04_parity_soak.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nlml_hpe_amd import _lib, ops, synth, weights
from oracle import encoder_heads as EH

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
CH = 65536
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
g3b = np.load(os.path.join(ROOT, "tests", "golden", "fx3b_reference_range.npz"))
sd = synth.encoder_state_dict(1404, seed=0, hidden_weight_gain=2.0)
sd["encoder.10.weight"], sd["encoder.10.bias"] = g3b["enc10_weight"], g3b["enc10_bias"]
P = EH.Params(sd, heads)
modes = {"f16x2s": _lib.MODE_F16X2S, "f32": _lib.MODE_F32, "f16x2": _lib.MODE_F16X2}
blobs = {k: torch.from_numpy(weights.pack_blob(sd, heads, m)).to(dev) for k, m in modes.items()}
torch.set_num_threads(os.cpu_count() or 1)
d_truth = {k: [] for k in list(modes) + ["reference_f32_batched"]}
d_ref = {k: [] for k in modes}
span = [np.inf, -np.inf]
t0 = time.perf_counter()
for i in range(0, N, CH):
    n = min(CH, N - i)
    x = synth.features(n, 1404, seed=1000 + i // CH)
    truth = EH.forward_numpy(x, P, np.float64)
    ref = np.concatenate([EH.forward_torch(x[j:j + 16384], P) for j in range(0, n, 16384)]).astype(np.float64)
    span = [min(span[0], float(np.degrees(truth.min()))), max(span[1], float(np.degrees(truth.max())))]
    d_truth["reference_f32_batched"].append(np.degrees(np.abs(ref - truth)).max(axis=1))
    xt = torch.from_numpy(x).to(dev)
    for k in modes:
        got = ops.encoder_heads_fwd(xt, blobs[k], 1404).cpu().numpy().astype(np.float64)
        d_truth[k].append(np.degrees(np.abs(got - truth)).max(axis=1))
        d_ref[k].append(np.degrees(np.abs(got - ref)).max(axis=1))
    print(f"{i + n} faces, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)


def stats(parts):
    d = np.concatenate(parts)
    return {"p50_deg": float(np.percentile(d, 50)), "p99_deg": float(np.percentile(d, 99)), "p99.99_deg": float(np.percentile(d, 99.99)),
            "max_deg": float(d.max()), "frac_above_1e-4_deg": float((d > 1e-4).mean())}


print(json.dumps({"faces": N, "weights": "FX3b (tests/golden/fx3b_reference_range.npz) over synth.encoder_state_dict(1404, seed=0, hidden_weight_gain=2.0)",
                  "inputs": "synth.features(65536, 1404, seed=1000 + chunk)", "pose_span_deg": span,
                  "vs_f64_truth": {k: stats(v) for k, v in d_truth.items()},
                  "kernel_vs_reference_f32_batched": {k: stats(v) for k, v in d_ref.items()},
                  "note": "reference_f32_batched = the oracle's torch-CPU f32 restatement of the reference's forward in batches of 16,384 on this host "
                          "(the arithmetic NLML_HPE_Test.py runs, at a larger batch); FX3c (16,384 faces, the reference itself) is the pinned form of this table"},
                 indent=1))
