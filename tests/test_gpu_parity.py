"""GPU parity tests: the HIP path, called through the C ABI, against the oracle and the golden
fixtures.  Tolerances: pose 1e-4 degrees absolute (BASELINE.json north_star); normalisation and
validity bit-exact; Tucker objective 1e-12 relative (f64)."""
import json
import os

import numpy as np
import pytest
import torch

from nlml_hpe_amd import ops, synth, weights
from oracle import encoder_heads as EH
from oracle import feature_norm as FN
from oracle import tucker as TK

pytestmark = pytest.mark.gpu

POSE_TOL_DEG = 1e-4


def _report(name, **kv):
    """Append measured margins to gpurun_out/parity_margins.jsonl (read back after the GPU call)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_margins.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **{k: float(v) for k, v in kv.items()}}) + "\n")
    except OSError:
        pass


def _blob(sd, head_sds, device):
    return torch.from_numpy(weights.pack_blob(sd, head_sds)).to(device)


@pytest.mark.parametrize("F", [1404, 136])
def test_fx3_encoder_heads_golden(F, head_sds, golden_dir, device):
    g = np.load(os.path.join(golden_dir, "fx3_encoder_heads.npz"))
    sd = synth.encoder_state_dict(F, seed=0)
    x = synth.features(256, F, seed=1)
    x[7] = 0.0
    assert float(x.astype(np.float64).sum()) == g[f"x_crc_F{F}"][0]
    out, valid = ops.encoder_heads_fwd(torch.from_numpy(x).to(device), _blob(sd, head_sds, device), F, return_valid=True)
    out = out.cpu().numpy()
    err = np.degrees(np.abs(out - g[f"rad_F{F}"]).max())
    _report(f"fx3_golden_F{F}", max_abs_deg=err)
    assert err <= POSE_TOL_DEG, err
    v = valid.cpu().numpy()
    assert not v[7] and v.sum() == 255


@pytest.mark.parametrize("F,B", [(1404, 1), (1404, 31), (1404, 33), (1404, 1000), (136, 77), (64, 50), (10, 40), (1407, 65)])
def test_encoder_heads_vs_oracle(F, B, head_sds, device):
    sd = synth.encoder_state_dict(F, seed=3)
    x = synth.features(B, F, seed=9)
    P = EH.Params(sd, head_sds)
    ref64 = EH.forward_numpy(x, P, np.float64)          # arithmetic truth
    ref32 = EH.forward_numpy(x, P, np.float32)          # what an f32 CPU path (the reference's) gives
    lat64 = EH.encoder_latent_numpy(x, P, np.float64)
    out, lat = ops.encoder_heads_fwd(torch.from_numpy(x).to(device), _blob(sd, head_sds, device), F, return_latent=True)
    out = out.cpu().numpy()
    e64 = np.degrees(np.abs(out - ref64).max())
    e32 = np.degrees(np.abs(out - ref32).max())
    _report(f"vs_oracle_F{F}_B{B}", hip_vs_f64_deg=e64, hip_vs_f32_deg=e32,
            cpu32_vs_f64_deg=np.degrees(np.abs(ref32 - ref64).max()))
    assert e64 <= POSE_TOL_DEG and e32 <= POSE_TOL_DEG
    assert np.abs(lat.cpu().numpy() - lat64).max() <= 2e-6


def test_encoder_heads_strided_input(head_sds, device):
    F = 136
    sd = synth.encoder_state_dict(F, seed=0)
    xfull = synth.features(100, 200, seed=4)
    xt = torch.from_numpy(xfull).to(device)[:, 8:8 + F]          # row stride 200, offset 32 B
    P = EH.Params(sd, head_sds)
    ref = EH.forward_numpy(xfull[:, 8:8 + F], P, np.float64)
    out = ops.encoder_heads_fwd(xt, _blob(sd, head_sds, device), F)
    assert np.degrees(np.abs(out.cpu().numpy() - ref).max()) <= POSE_TOL_DEG


def test_fx1_normalise_golden_bitexact(golden_dir, device):
    g = np.load(os.path.join(golden_dir, "fx1_normalise.npz"))
    raw = torch.from_numpy(g["landmarks"]).to(device)
    out, valid = ops.normalize_ipd(raw, True, return_valid=True)
    assert np.array_equal(out.cpu().numpy(), g["features_norm"])
    assert valid.all()
    out0 = ops.normalize_ipd(raw, False)
    assert np.array_equal(out0.cpu().numpy(), g["features_raw"])


def test_normalise_vs_oracle_bitexact_and_no_face(device):
    raw = synth.raw_landmarks(1003, seed=21)
    raw[5] = 0.0                      # all-zero landmarks: (0-0)/1e-6 = 0 -> "no face" row
    raw[17, 263] = raw[17, 33]        # ipd == 0 on a non-zero face
    ref = FN.normalize_ipd(raw, True)
    out, valid = ops.normalize_ipd(torch.from_numpy(raw).to(device), True, return_valid=True)
    assert np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(valid.cpu().numpy(), ~FN.no_face_mask(ref))


def test_fused_landmarks_to_pose(head_sds, device):
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = _blob(sd, head_sds, device)
    raw = synth.raw_landmarks(333, seed=5)
    raw[11] = 0.0
    rt = torch.from_numpy(raw).to(device)
    feats = ops.normalize_ipd(rt, True)
    two_step = ops.encoder_heads_fwd(feats, blob, 1404)
    fused, valid = ops.landmarks_to_pose(rt, blob, True, return_valid=True)
    assert torch.equal(fused, two_step)                     # same arithmetic, same order => same bits
    assert not bool(valid[11]) and int(valid.sum()) == 332
    P = EH.Params(sd, head_sds)
    ref = EH.forward_numpy(FN.normalize_ipd(raw, True), P, np.float64)
    assert np.degrees(np.abs(fused.cpu().numpy() - ref).max()) <= POSE_TOL_DEG
    raw_mode = ops.landmarks_to_pose(rt, blob, False)
    assert torch.equal(raw_mode, ops.encoder_heads_fwd(rt.reshape(333, 1404), blob, 1404))


def test_determinism(head_sds, device):
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = _blob(sd, head_sds, device)
    x = torch.from_numpy(synth.features(4096, 1404, seed=2)).to(device)
    a = ops.encoder_heads_fwd(x, blob, 1404)
    b = ops.encoder_heads_fwd(x, blob, 1404)
    assert torch.equal(a, b)


def _cos_params(art):
    return np.stack([art["optimized_yaw"][0:3], art["optimized_pitch"][0:3], art["optimized_roll"][0:3]])


def test_fx4_tucker_objective_golden(tucker_art, golden_dir, device):
    g = np.load(os.path.join(golden_dir, "fx4_td_objective.npz"))
    Wm = torch.from_numpy(tucker_art["W"].reshape(135, 1404)).to(device)
    err, xh = ops.tucker_objective(Wm, torch.from_numpy(g["x"]).to(device), torch.from_numpy(g["params"]).to(device),
                                   torch.from_numpy(_cos_params(tucker_art)).to(device), return_xhat=True, order="fast")
    err = err.cpu().numpy()
    assert np.max(np.abs(err - g["err"]) / np.abs(g["err"])) <= 1e-12
    xh = xh.cpu().numpy()[:8]
    assert np.max(np.abs(xh - g["x_hat"])) <= 1e-12 * np.max(np.abs(g["x_hat"]))


def test_tucker_objective_shared_rows_and_ragged(tucker_art, device):
    N = 1001                                  # not a multiple of the 8 evaluations per workgroup
    P = synth.tucker_params(N, 5, seed=6)
    X = synth.features(7, 1404, seed=6)
    idx = (np.arange(N) % 7).astype(np.int32)
    cp = _cos_params(tucker_art)
    ref = TK.objective_batch(P, tucker_art["W"], X[idx], cp[0], cp[1], cp[2])
    Wm = torch.from_numpy(tucker_art["W"].reshape(135, 1404)).to(device)
    err = ops.tucker_objective(Wm, torch.from_numpy(X).to(device), torch.from_numpy(P).to(device),
                               torch.from_numpy(cp).to(device), x_index=torch.from_numpy(idx).to(device), order="fast")
    assert np.max(np.abs(err.cpu().numpy() - ref) / np.abs(ref)) <= 1e-12


def test_errors_are_loud(head_sds, device):
    from nlml_hpe_amd import _lib
    sd = synth.encoder_state_dict(136, seed=0)
    blob = _blob(sd, head_sds, device)
    with pytest.raises(ValueError):
        ops.encoder_heads_fwd(torch.zeros(4, 135, device=device), blob, 136)
    with pytest.raises(_lib.NlmlError):
        ops.encoder_heads_fwd(torch.zeros(4, 136), blob, 136)          # CPU tensor: no fallback
    with pytest.raises(_lib.NlmlError):
        ops.encoder_heads_fwd(torch.zeros(4, 1404, device=device), blob, 1404)   # blob packed for another F
    assert ops.encoder_heads_fwd(torch.zeros(0, 136, device=device), blob, 136).shape == (0, 3)


@pytest.mark.parametrize("F,B,gain", [(1404, 257, 1.0), (1404, 64, 2.4), (136, 100, 2.4), (12, 33, 2.4)])
def test_mfma_chain_bitexact_vs_c_oracle(F, B, gain, head_sds, device):
    """Everything before the Tanh is pure f32 fma: the MFMA chains must equal the C oracle's
    fmaf chains (same k order) bit for bit, whatever the weight gain.  After the Tanh (ocml vs
    glibc tanhf, <= 2 ulp apart) agreement stays far inside the pose tolerance."""
    from oracle import c_oracle as CO
    sd = synth.encoder_state_dict(F, seed=5)
    for k in sd:
        if k.endswith("weight"):
            sd[k] = (sd[k] * np.float32(gain)).astype(np.float32)     # ~variance-preserving at 2.4
    x = synth.features(B, F, seed=13)
    P = EH.Params(sd, head_sds)
    c_out, c_lat, c_pre = CO.encoder_heads(x, P, order=2, want_latent=True, want_pre_tanh=True)
    out, lat, pre = ops.encoder_heads_fwd_debug(torch.from_numpy(x).to(device), _blob(sd, head_sds, device), F)
    assert np.array_equal(pre.cpu().numpy(), c_pre)
    _report(f"c_oracle_F{F}_gain{gain}", latent_abs=np.abs(lat.cpu().numpy() - c_lat).max(),
            out_abs_deg=np.degrees(np.abs(out.cpu().numpy() - c_out).max()))
    assert np.abs(lat.cpu().numpy() - c_lat).max() <= 1e-6
    assert np.degrees(np.abs(out.cpu().numpy() - c_out).max()) <= POSE_TOL_DEG


TD_FAST_TOL_DEG = 2e-2      # fast (matrix-core) TD mode ON FX5's CLEAN GRID FACES ONLY (not a parity mode: on noisy faces its end point
                            # can be degrees away, bench.py extra.td_powell_fast_order); tests/test_powell_sm.py
TD_REF_TOL_DEG = 1e-4       # reference-order TD mode: the bar of north_star (measured: identical bits)


def test_fx5_powell_on_device(tucker_art, golden_dir, device):
    """TD end-to-end (TD_Tester.Test) in the FAST order: final angles within the optimiser tolerance of scipy's.  The minimum is
    flat and Powell's termination is rounding-sensitive: scipy itself ends up to ~1e-2 deg elsewhere when only the f64 summation
    order of its objective changes (tests/test_powell_sm.py::test_powell_final_angles_are_sensitive_...), and the matrix-core
    objective differs from np.einsum in exactly that way."""
    from nlml_hpe_amd import TD_Tester as TD
    g = np.load(os.path.join(golden_dir, "fx5_td_end_to_end.npz"))
    Py, Pp, Pr = tucker_art["optimized_yaw"][:3], tucker_art["optimized_pitch"][:3], tucker_art["optimized_roll"][:3]
    deg, info = TD.Test_batch(tucker_art["W"], g["x"], 5, Py, Pp, Pr, return_info=True, order="fast")
    d = np.abs(deg - g["deg"]).max()
    _report("fx5_powell", max_abs_deg=d, nfev_dev_max=info["nfev"].max(), nfev_ref_max=g["nfev"].max())
    assert (info["status"] == 1).all()
    assert d <= TD_FAST_TOL_DEG, (deg, g["deg"])
    # the objective at the device's minimiser is as low as at scipy's (both ~0 for grid faces)
    f_dev = TD.objective_batch(info["x"], tucker_art["W"], g["x"], Py, Pp, Pr, order="fast")
    assert np.allclose(f_dev, info["fun"], rtol=1e-9, atol=1e-15)
    y, p, r, uid = TD.Test(tucker_art["W"], torch.from_numpy(g["x"][1]), 5, Py, Pp, Pr, None, None, None, None, order="fast")
    assert uid is None and abs(y - g["deg"][1][0]) <= TD_FAST_TOL_DEG


def test_fx4_objective_reference_order_is_bit_exact(tucker_art, golden_dir, device):
    """NLML_TD_ORDER_REFERENCE: np.einsum's operation order and numpy's pairwise sum on the device -- err and x_hat equal the
    reference's own outputs (FX4) BIT FOR BIT, and the C oracle in the same order on ragged / shared-row batches."""
    from oracle import c_oracle as CO
    g = np.load(os.path.join(golden_dir, "fx4_td_objective.npz"))
    cp = np.stack([tucker_art["optimized_yaw"][:3], tucker_art["optimized_pitch"][:3], tucker_art["optimized_roll"][:3]])
    Wm = torch.from_numpy(tucker_art["W"].reshape(135, 1404)).to(device)
    cpt = torch.from_numpy(cp).to(device)
    err, xh = ops.tucker_objective(Wm, torch.from_numpy(g["x"]).to(device), torch.from_numpy(g["params"]).to(device), cpt,
                                   return_xhat=True, order="reference")
    assert np.array_equal(err.cpu().numpy(), g["err"])
    assert np.array_equal(xh.cpu().numpy()[:8], g["x_hat"])
    for N in (1, 2, 3, 5, 6, 7, 9, 11, 13, 16, 45):            # passes of every size 1..8 (9..16 go as two), several workgroups
        P = synth.tucker_params(N, 5, seed=30 + N)
        X = synth.features(5, 1404, seed=31)
        idx = (np.arange(N) * 3 % 5).astype(np.int32)
        ref = CO.tucker_objective(tucker_art["W"], X[idx], P, cp, reference_order=True)
        got = ops.tucker_objective(Wm, torch.from_numpy(X).to(device), torch.from_numpy(P).to(device), cpt,
                                   x_index=torch.from_numpy(idx).to(device), order="reference")
        assert np.array_equal(got.cpu().numpy(), ref), N
    assert ops.tucker_objective(Wm, torch.zeros((0, 1404), device=device), torch.zeros((0, 8), dtype=torch.float64, device=device),
                                cpt, order="reference").shape == (0,)


def test_fx5_powell_reference_order_walks_scipys_trajectory(tucker_art, golden_dir, device):
    """TD end-to-end in the REFERENCE order: every machine is fed the reference's objective values bit for bit, so it makes
    scipy's own evaluations -- the evaluation counts are FX5's and the final angles meet the 1e-4 deg bar (they are identical)."""
    from nlml_hpe_amd import TD_Tester as TD
    g = np.load(os.path.join(golden_dir, "fx5_td_end_to_end.npz"))
    Py, Pp, Pr = tucker_art["optimized_yaw"][:3], tucker_art["optimized_pitch"][:3], tucker_art["optimized_roll"][:3]
    deg, info = TD.Test_batch(tucker_art["W"], g["x"], 5, Py, Pp, Pr, return_info=True, order="reference")
    d = np.abs(deg - g["deg"]).max()
    _report("fx5_powell_reference_order", max_abs_deg=d, nfev_dev_max=info["nfev"].max(), nfev_ref_max=g["nfev"].max())
    assert (info["status"] == 1).all()
    assert np.array_equal(info["nfev"], g["nfev"])
    assert d <= TD_REF_TOL_DEG, (deg, g["deg"])
    # the reference-named single-face entry point defaults to this order
    y, p, r, uid = TD.Test(tucker_art["W"], torch.from_numpy(g["x"][2]), 5, Py, Pp, Pr, None, None, None, None)
    assert uid is None and max(abs(y - g["deg"][2][0]), abs(p - g["deg"][2][1]), abs(r - g["deg"][2][2])) <= TD_REF_TOL_DEG
    # 21 faces (two workgroups, the second ragged): each face ends where it ends alone
    X = np.concatenate([g["x"]] * 5 + [g["x"][:1]])
    deg21 = TD.Test_batch(tucker_art["W"], X, 5, Py, Pp, Pr, order="reference")
    assert np.array_equal(deg21, np.concatenate([deg] * 5 + [deg[:1]]))


def test_powell_batch_ragged_and_independent(tucker_art, device):
    """13 faces (not a multiple of the 8 per workgroup): each result equals the result of minimising it alone."""
    from nlml_hpe_amd import TD_Tester as TD
    from oracle import tucker as TK
    fm = tucker_art
    Py, Pp, Pr = fm["optimized_yaw"][:3], fm["optimized_pitch"][:3], fm["optimized_roll"][:3]
    idx = synth.tucker_grid_indices(13, seed=8)
    X = np.stack([TK.grid_reconstruction(fm["W"], fm["U_id"][i], fm["U_yaw"][j], fm["U_pitch"][k], fm["U_roll"][l])
                  for i, j, k, l in idx])
    deg, info = TD.Test_batch(fm["W"], X, 5, Py, Pp, Pr, return_info=True, order="fast")
    solo, info1 = TD.Test_batch(fm["W"], X[9:10], 5, Py, Pp, Pr, return_info=True, order="fast")
    assert np.array_equal(deg[9], solo[0]) and info["nfev"][9] == info1["nfev"][0]
    # reported only: how far the TD method lands from the grid pose (a property of the reference's model --
    # extreme bins fall into other local minima -- not of this implementation)
    bins = np.stack([-50 + 10 * idx[:, 1], -40 + 10 * idx[:, 2], -30 + 10 * idx[:, 3]], axis=1)
    _report("powell_grid_recovery", median_abs_deg=np.median(np.abs(deg - bins)), mean_nfev=info["nfev"].mean())
    assert (info["status"] == 1).all() and np.isfinite(deg).all()


def test_device_powell_replays_exactly_on_the_cpu(tucker_art, golden_dir, device):
    """The device minimiser IS scipy's Powell applied to the device objective: stepping the same state
    machine on the host with the C oracle's objective summed in the kernel's order reproduces the GPU
    run bit for bit (same number of evaluations, same minimiser)."""
    from nlml_hpe_amd import TD_Tester as TD
    from nlml_hpe_amd.powell_host import minimize_powell
    from oracle import c_oracle as CO
    g = np.load(os.path.join(golden_dir, "fx5_td_end_to_end.npz"))
    W = tucker_art["W"]
    Py, Pp, Pr = tucker_art["optimized_yaw"][:3], tucker_art["optimized_pitch"][:3], tucker_art["optimized_roll"][:3]
    cp = np.stack([Py, Pp, Pr])
    X = g["x"][:2]
    deg, info = TD.Test_batch(W, X, 5, Py, Pp, Pr, return_info=True, order="fast")
    # objective level first: device == C oracle in device order, bitwise
    P = synth.tucker_params(64, 5, seed=12)
    e_dev = TD.objective_batch(P, W, np.repeat(X[:1], 64, axis=0), Py, Pp, Pr, order="fast")
    e_c = CO.tucker_objective(W, np.repeat(X[:1], 64, axis=0), P, cp, device_order=True)
    assert np.array_equal(e_dev, e_c)
    for i in range(2):
        host = minimize_powell(lambda p: float(CO.tucker_objective(W, X[i:i + 1], p[None], cp, device_order=True)[0]), np.zeros(8))
        assert host.nfev == info["nfev"][i]
        assert np.array_equal(host.x, info["x"][i])
        assert host.fun == info["fun"][i]


def test_fx7_video_post_on_device(head_sds, golden_dir, device):
    """K4 against what the reference's own process_video loop produced (FX7): smoothed angles, centre
    with the jump gate, integer axis end points, skipped no-face frames."""
    import json
    from nlml_hpe_amd.video import VideoPoseTracker
    g = json.load(open(os.path.join(golden_dir, "fx7_video_math.json")))
    gin = np.load(os.path.join(golden_dir, "fx7_video_in.npz"))
    lm, pose = gin["landmarks"], gin["pose_rad"]
    T = lm.shape[0]

    class _M:                      # the tracker only needs .device from the model when post() is driven directly
        pass
    m = _M()
    m.device = device
    tr = VideoPoseTracker(m, 1, g["width"], g["height"])
    recs = {f["frame"]: f for f in g["frames"]}
    no_face = set(gin["no_face"].tolist())
    for t in range(T):
        valid = torch.tensor([t not in no_face], device=device)
        sm, c, ep = tr.post(torch.from_numpy(pose[t:t + 1]).to(device), torch.from_numpy(lm[t:t + 1]).to(device), valid)
        if t in no_face:
            continue
        fr = recs[t]
        assert np.allclose(sm.cpu().numpy()[0], fr["smoothed"], rtol=0, atol=1e-9)
        cc = c.cpu().numpy()[0]
        assert np.allclose(cc, fr["centre"], rtol=0, atol=1e-9)
        e = ep.cpu().numpy()[0]
        got = [[[int(cc[0]), int(cc[1])], [int(e[k, 0]), int(e[k, 1])]] for k in range(3)]
        assert got == fr["lines"], (t, got, fr["lines"])


def test_metrics_on_device_match_fx6(golden_dir, device):
    import json
    from nlml_hpe_amd import metrics
    g = json.load(open(os.path.join(golden_dir, "fx6_metrics.json")))
    gt = torch.tensor(g["gt"], dtype=torch.float64, device=device)
    pr = torch.tensor(g["pred"], dtype=torch.float64, device=device)
    assert np.allclose(metrics.compute_maev(gt, pr), g["maev"], rtol=0, atol=1e-9)
    d = metrics.compute_errors(gt, pr, verbose=False)
    printed = dict(line.split(": ") for line in g["printed"])
    assert f'{d["mae_yaw"]:.2f}' == printed["MAE (Yaw)"] and f'{d["std_roll"]:.2f}' == printed["std (Roll)"]
    assert f'{d["maev"]:.2f}' == printed["MAEV"]


def test_entry_points_run(repo_root, device, tmp_path, capsys):
    import subprocess
    import sys
    lm = synth.raw_landmarks(3, seed=31)
    np.save(tmp_path / "faces.npy", lm)
    env = dict(os.environ, PYTHONPATH=repo_root)
    for cmd in (["NLML_HPE_Test.py"], ["TD_Inference.py", "--image_path", str(tmp_path / "faces.npy")],
                ["generatePose_on_video.py", "--source", "synthetic", "--save_output", "True", "--output_path", str(tmp_path / "p.npz")],
                ["NLML_HPE_Model_Builder.py", "--synthetic-encoder-seed", "0", "--out", str(tmp_path / "m.nlml")]):
        res = subprocess.run([sys.executable] + cmd, cwd=repo_root, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, (cmd, res.stdout[-2000:], res.stderr[-2000:])
        if cmd[0] == "NLML_HPE_Test.py":
            # BASELINE config 1 shape (256 rows): the printed metrics block must equal what the oracle computes from
            # the same synthetic landmarks, weights and ground truth (reference: NLML_HPE_Test.py:95-130, :273)
            from oracle import metrics as MT
            raw = synth.raw_landmarks(256, seed=1)
            gt = synth.poses_deg(256, seed=4)
            P = EH.Params(synth.encoder_state_dict(1404, seed=0), weights.load_head_state_dicts(os.path.join(repo_root, "models")))
            pred = np.round(np.degrees(EH.forward_numpy(FN.normalize_ipd(raw, True), P, np.float32).astype(np.float64)), 3)
            lo, hi = np.array([-50, -40, -30.0]), np.array([51, 41, 31.0])
            keep = ((gt >= lo) & (gt <= hi)).all(axis=1)
            exp = MT.compute_errors(gt[keep], pred[keep])
            printed = dict(line.split(": ", 1) for line in res.stdout.splitlines() if ": " in line)
            for key, label in (("mae_yaw", "MAE (Yaw)"), ("mae_pitch", "MAE (Pitch)"), ("mae_roll", "MAE (Roll)"),
                               ("mae_total", "Total MAE"), ("maev", "MAEV"), ("std_yaw", "std (Yaw)")):
                assert printed[label] == f"{exp[key]:.2f}", (label, printed[label], exp[key])
        if cmd[0] == "TD_Inference.py":
            assert res.stdout.count("Estimated yaw in degree") == 3
    out = np.load(tmp_path / "p.npz")
    assert out["smoothed_deg"].shape == (90, 64, 3) and np.isfinite(out["endpoints"]).all()


def test_normalise_full_batch_bitexact(device):
    """BASELINE size: 65,536 faces = 92 M elements.  The kernels divide through a reciprocal with two fma
    corrections (correctly rounded in f64, then one rounding to f32); this checks every element against
    the C oracle's true IEEE division, and the fused path against K1 -> K2."""
    from oracle import c_oracle as CO
    raw = synth.raw_landmarks(65536, seed=3)
    raw[::97] *= np.float32(1e-3)
    raw[1::97] *= np.float32(1920.0)
    ref = CO.normalize_ipd(raw, True)
    out = ops.normalize_ipd(torch.from_numpy(raw).to(device), True).cpu().numpy()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_fused_valid_mask_edge_cases(head_sds, device):
    """A face whose 468 landmarks all coincide normalises to an all-zero row ("no face", FeatureExtractor.py:105-106)
    even though the raw landmarks are non-zero; tiles that are partially filled; B not a multiple of 64."""
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = _blob(sd, head_sds, device)
    raw = synth.raw_landmarks(131, seed=17)
    raw[3] = np.array([0.25, 0.5, 0.75], np.float32)      # all landmarks identical, x != y != z
    raw[64] = 0.0
    raw[130] = np.array([0.1, 0.1, 0.1], np.float32)
    rt = torch.from_numpy(raw).to(device)
    feats, v1 = ops.normalize_ipd(rt, True, return_valid=True)
    pose, v2 = ops.landmarks_to_pose(rt, blob, True, return_valid=True)
    ref_valid = ~FN.no_face_mask(FN.normalize_ipd(raw, True))
    assert not ref_valid[3] and not ref_valid[64] and not ref_valid[130] and ref_valid.sum() == 128
    assert np.array_equal(v1.cpu().numpy(), ref_valid) and np.array_equal(v2.cpu().numpy(), ref_valid)
    assert torch.equal(pose, ops.encoder_heads_fwd(feats, blob, 1404))


def test_full_batch_65536_properties(head_sds, device):
    """BASELINE.json size (65,536 faces, F = 1404): (i) a face's pose does not depend on where it sits in the
    batch or on its tile neighbours (bit-exact against re-running sampled rows alone, in another order);
    (ii) the sampled rows match the f64 oracle to 1e-4 deg; (iii) partial last tile: B-1 rows give the same bits."""
    F, B = 1404, 65536
    sd = synth.encoder_state_dict(F, seed=0)
    blob = _blob(sd, head_sds, device)
    x = synth.features(B, F, seed=42)
    xt = torch.from_numpy(x).to(device)
    full = ops.encoder_heads_fwd(xt, blob, F)
    idx = synth.rng(42, 9).permutation(B)[:777]
    sub = ops.encoder_heads_fwd(xt[torch.from_numpy(idx).to(device)], blob, F)
    assert torch.equal(sub, full[torch.from_numpy(idx).to(device)])
    ref = EH.forward_numpy(x[idx], EH.Params(sd, head_sds), np.float64)
    err = np.degrees(np.abs(sub.cpu().numpy() - ref).max())
    _report("full_batch_sampled_vs_f64", max_abs_deg=err)
    assert err <= POSE_TOL_DEG
    assert torch.equal(ops.encoder_heads_fwd(xt[:B - 1], blob, F), full[:B - 1])


def test_nan_and_inf_stay_in_their_face(head_sds, device):
    """Faces are MFMA columns: a NaN/Inf face must not leak into its tile neighbours (the reference's rows are
    independent, NLML_HPE_Model_Builder.py:55-68), including through the zero-padded K columns."""
    F = 1404
    sd = synth.encoder_state_dict(F, seed=0)
    blob = _blob(sd, head_sds, device)
    x = synth.features(200, F, seed=23)
    clean = ops.encoder_heads_fwd(torch.from_numpy(x).to(device), blob, F)
    bad = x.copy()
    bad[5, 100] = np.nan
    bad[70, 1403] = np.inf            # last real column: its clamped re-read feeds the padded K columns
    bad[133, 0] = -np.inf
    out = ops.encoder_heads_fwd(torch.from_numpy(bad).to(device), blob, F)
    rows = torch.ones(200, dtype=torch.bool, device=device)
    rows[[5, 70, 133]] = False
    assert torch.equal(out[rows], clean[rows])
    assert not torch.isfinite(out[5]).all() and not torch.isfinite(out[70]).all() and not torch.isfinite(out[133]).all()


def test_launches_are_graph_capturable(head_sds, device):
    """The launch functions allocate nothing and never synchronise, so a whole tick (fused forward + video
    post-processing) can be captured into a hipGraph and replayed."""
    from nlml_hpe_amd.model import HIPPoseModel
    sd = synth.encoder_state_dict(1404, seed=0)
    model = HIPPoseModel(sd, head_sds, device=device)
    S = 64
    raw = torch.from_numpy(synth.raw_landmarks(S, seed=3)).to(device)
    eager = model.from_landmarks(raw, True).clone()
    static_in = raw.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model.from_landmarks(static_in, True)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_out = model.from_landmarks(static_in, True)
    static_in.copy_(torch.from_numpy(synth.raw_landmarks(S, seed=4)).to(device))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, model.from_landmarks(static_in, True))
    static_in.copy_(raw)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, eager)


@pytest.mark.parametrize("F,B", [(1404, 300), (136, 129)])
def test_bf16_throughput_mode(F, B, head_sds, device):
    """NLML_MODE_BF16 is a THROUGHPUT mode, not a parity path: its error against the f64 truth is
    measured and reported (~0.1 deg); its indexing/rounding points are pinned against a bf16-emulating model."""
    from nlml_hpe_amd import _lib
    sd = synth.encoder_state_dict(F, seed=0)
    blob = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.MODE_BF16)).to(device)
    assert blob.numel() == _lib.lib().nlml_encoder_heads_packed_bytes(F, _lib.MODE_BF16)
    x = synth.features(B, F, seed=21)
    x[3] = 0.0
    P = EH.Params(sd, head_sds)
    out, lat, valid = ops.encoder_heads_fwd(torch.from_numpy(x).to(device), blob, F, return_latent=True, return_valid=True)
    out, lat = out.cpu().numpy(), lat.cpu().numpy()
    emu, emu_lat = EH.forward_bf16_emulated(x, P)
    truth = EH.forward_numpy(x, P, np.float64)
    e_emu = np.degrees(np.abs(out - emu).max())
    e_truth = np.degrees(np.abs(out - truth).max())
    _report(f"bf16_mode_F{F}", vs_bf16_model_deg=e_emu, vs_f64_truth_deg=e_truth, mean_vs_truth_deg=np.degrees(np.abs(out - truth).mean()),
            latent_vs_model=np.abs(lat - emu_lat).max())
    assert np.abs(lat - emu_lat).max() <= 2e-3        # f32 vs f64 accumulation under bf16 re-rounding of activations
    # a bf16 rounding flip (f32 vs f64 accumulation landing on different sides of a tie) moves an activation by 2^-8
    # relative, so the model is matched closely at the latent and only to bf16 noise at the pose
    assert e_emu <= 0.25 and e_truth <= 0.5
    v = valid.cpu().numpy()
    assert not v[3] and v.sum() == B - 1


def test_bf16_fused_landmarks(head_sds, device):
    from nlml_hpe_amd import _lib
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.MODE_BF16)).to(device)
    raw = synth.raw_landmarks(200, seed=5)
    raw[11] = 0.0
    rt = torch.from_numpy(raw).to(device)
    fused, valid = ops.landmarks_to_pose(rt, blob, True, return_valid=True)
    two_step = ops.encoder_heads_fwd(ops.normalize_ipd(rt, True), blob, 1404)
    assert torch.equal(fused, two_step)
    assert not bool(valid[11]) and int(valid.sum()) == 199
    ref = EH.forward_numpy(FN.normalize_ipd(raw, True), EH.Params(sd, head_sds), np.float64)
    assert np.degrees(np.abs(fused.cpu().numpy() - ref).max()) <= 0.5


@pytest.mark.parametrize("mode_name", ["bf16", "f16x2s", "f16x2", "f32"])
def test_result_does_not_depend_on_where_a_row_starts_in_its_cache_line(mode_name, head_sds, device):
    """The kernels stage x in 16-byte units and (bf16: always; strict-fast: with -DW8_XLINE) let the units of a slab's first 128-byte line
    be loaded one slab ahead, by the row's phase within its line.  Same features behind every phase (base offsets of 0 / 16 / 48 / 112
    bytes in line-aligned rows, and the packed 5,616-byte rows whose phase changes from row to row) must give the same bits."""
    from nlml_hpe_amd import _lib
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.mode_from_name(mode_name))).to(device)
    B = 333
    feats = torch.from_numpy(synth.features(B, 1404, seed=31)).to(device)
    want, want_lat = ops.encoder_heads_fwd(feats, blob, 1404, return_latent=True)
    for off in (0, 4, 12, 28):
        buf = torch.zeros((B, 1408 + 32), dtype=torch.float32, device=device)   # 5,760-byte rows: 45 whole lines
        view = buf[:, off:off + 1404]
        view.copy_(feats)
        got, got_lat = ops.encoder_heads_fwd(view, blob, 1404, return_latent=True)
        assert torch.equal(got, want) and torch.equal(got_lat, want_lat), (mode_name, off)


def test_host_pipeline_overlapped_copies(head_sds, device):
    """Host-resident landmarks through the double-buffered copy/compute pipeline == the direct device call."""
    from nlml_hpe_amd.model import HIPPoseModel
    from nlml_hpe_amd.pipeline import HostPipeline
    model = HIPPoseModel(synth.encoder_state_dict(1404, seed=0), head_sds, device=device)
    raw = synth.raw_landmarks(5000, seed=77)
    raw[123] = 0.0
    pipe = HostPipeline(model, batch=1024)
    pose, valid = pipe.run(raw)
    ref, vref = model.from_landmarks(torch.from_numpy(raw).to(device), True, return_valid=True)
    assert np.array_equal(pose, ref.cpu().numpy()) and np.array_equal(valid, vref.cpu().numpy())
    assert not valid[123]
    pose2, _ = pipe.run(raw[:10])           # reuse with a short tail
    assert np.array_equal(pose2, pose[:10])
    # both forms of the host -> device copy give the same bits: staged through pinned memory, and straight out of the caller's array
    # page-locked in place; a read-only array cannot be registered for the DMA here and falls back to the staged form
    for inplace, want_mode in ((False, "staged"), (True, "in place")):
        p3, v3 = pipe.run(raw, inplace=inplace)
        assert pipe.last_mode == want_mode
        assert np.array_equal(p3, pose) and np.array_equal(v3, valid)
    ro = raw.copy()
    ro.setflags(write=False)
    p4, v4 = pipe.run(ro)
    assert pipe.last_mode == "staged" and np.array_equal(p4, pose) and np.array_equal(v4, valid)
    p5, _ = pipe.run(raw[1:4000], inplace=True)     # a range that starts and ends inside pages, registered again right after
    assert np.array_equal(p5, pose[1:4000])


def test_graphed_video_tick_matches_eager(head_sds, device):
    from nlml_hpe_amd.model import HIPPoseModel
    from nlml_hpe_amd.video import GraphedTick, VideoPoseTracker
    model = HIPPoseModel(synth.encoder_state_dict(1404, seed=0), head_sds, device=device)
    S = 64
    frames = torch.from_numpy(synth.raw_landmarks(S * 5, seed=8).reshape(5, S, 468, 3) * 0.2 + 0.4).to(device)
    frames[2, 7] = 0.0                                   # a stream without a face in one tick
    a, b = VideoPoseTracker(model, S, 1920, 1080), VideoPoseTracker(model, S, 1920, 1080)
    g = GraphedTick(b)
    for t in range(5):
        sm_a, c_a, ep_a, v_a = a.tick(frames[t])
        g.static_raw.copy_(frames[t])
        sm_b, c_b, ep_b, v_b = g.replay()
        torch.cuda.synchronize()
        assert torch.equal(sm_a, sm_b) and torch.equal(c_a, c_b) and torch.equal(ep_a, ep_b) and torch.equal(v_a, v_b)
    assert torch.equal(a.state, b.state)


# ---- split-f16 parity mode (NLML_MODE_F16X2) ---------------------------------------------------------------------
HX_MODES = ["f16x2", "f16x2s"]     # the fast mode and the strict-fast mode (split accumulators): same operands, same tests


def _blob_hx(sd, head_sds, device, hx="f16x2"):
    from nlml_hpe_amd import _lib
    return torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.mode_from_name(hx))).to(device)


@pytest.mark.parametrize("hx", HX_MODES)
@pytest.mark.parametrize("F,B", [(1404, 1), (1404, 31), (1404, 33), (1404, 1000), (136, 77), (64, 50), (10, 40), (1407, 65)])
def test_split_f16_mode_vs_oracle(F, B, hx, head_sds, device):
    """NLML_MODE_F16X2 is a PARITY mode: same 1e-4 degree bar as the f32 kernel (tolerance: POSE_TOL_DEG, absolute,
    against the f64 arithmetic truth and against the f32 CPU restatement of the reference)."""
    sd = synth.encoder_state_dict(F, seed=3)
    x = synth.features(B, F, seed=9)
    P = EH.Params(sd, head_sds)
    ref64 = EH.forward_numpy(x, P, np.float64)
    ref32 = EH.forward_numpy(x, P, np.float32)
    lat64 = EH.encoder_latent_numpy(x, P, np.float64)
    out, lat = ops.encoder_heads_fwd(torch.from_numpy(x).to(device), _blob_hx(sd, head_sds, device, hx), F, return_latent=True)
    out, lat = out.cpu().numpy(), lat.cpu().numpy()
    e64 = np.degrees(np.abs(out - ref64).max())
    e32 = np.degrees(np.abs(out - ref32).max())
    _report(f"split_f16_vs_oracle_{hx}_F{F}_B{B}", hip_vs_f64_deg=e64, hip_vs_f32_deg=e32, latent_abs=np.abs(lat - lat64).max())
    assert e64 <= POSE_TOL_DEG and e32 <= POSE_TOL_DEG, (e64, e32)
    assert np.abs(lat - lat64).max() <= 5e-6


@pytest.mark.parametrize("hx", HX_MODES)
def test_split_f16_mode_golden_and_blob_walk(hx, head_sds, golden_dir, device):
    """FX3 golden vectors (generated by the reference itself) within the bar, and the kernel agrees with the numpy
    walk of its own blob (tests/blob_emulator.py forward_f16x2: same pieces, same three products) far more tightly."""
    import blob_emulator as BE
    from nlml_hpe_amd import _lib
    g = np.load(os.path.join(golden_dir, "fx3_encoder_heads.npz"))
    for F in (1404, 136):
        sd = synth.encoder_state_dict(F, seed=0)
        x = synth.features(256, F, seed=1)
        x[7] = 0.0
        blob_np = weights.pack_blob(sd, head_sds, _lib.mode_from_name(hx))
        out, valid = ops.encoder_heads_fwd(torch.from_numpy(x).to(device), torch.from_numpy(blob_np).to(device), F, return_valid=True)
        out = out.cpu().numpy()
        err = np.degrees(np.abs(out - g[f"rad_F{F}"]).max())
        emu, _ = BE.forward_f16x2(blob_np, x[32:64])
        e_emu = np.degrees(np.abs(out[32:64] - emu).max())
        _report(f"split_f16_fx3_{hx}_F{F}", max_abs_deg=err, vs_blob_walk_deg=e_emu)
        assert err <= POSE_TOL_DEG, err
        assert e_emu <= 2e-5, e_emu            # f32 vs f64 accumulation of identical products
        v = valid.cpu().numpy()
        assert not v[7] and v.sum() == 255


@pytest.mark.parametrize("hx", HX_MODES)
def test_split_f16_fused_landmarks_and_valid_mask(hx, head_sds, device):
    """Fused landmarks->pose in split-f16 mode: IPD normalisation stays exact (f64 division as in K1), so fused ==
    normalise + forward bit for bit; validity mask and partial tiles as in the f32 mode."""
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = _blob_hx(sd, head_sds, device, hx)
    raw = synth.raw_landmarks(131, seed=17)
    raw[3] = np.array([0.25, 0.5, 0.75], np.float32)
    raw[64] = 0.0
    raw[130] = np.array([0.1, 0.1, 0.1], np.float32)
    rt = torch.from_numpy(raw).to(device)
    feats = ops.normalize_ipd(rt, True)
    pose, valid = ops.landmarks_to_pose(rt, blob, True, return_valid=True)
    ref_valid = ~FN.no_face_mask(FN.normalize_ipd(raw, True))
    assert np.array_equal(valid.cpu().numpy(), ref_valid)
    assert torch.equal(pose, ops.encoder_heads_fwd(feats, blob, 1404))
    ok = torch.from_numpy(ref_valid).to(device)
    ref = EH.forward_numpy(FN.normalize_ipd(raw, True), EH.Params(sd, head_sds), np.float64)
    err = np.degrees(np.abs(pose.cpu().numpy() - ref)[ref_valid].max())
    _report("split_f16_fused", max_abs_deg=err)
    assert err <= POSE_TOL_DEG, err
    unnorm = ops.landmarks_to_pose(rt, blob, False)
    ref_u = EH.forward_numpy(FN.normalize_ipd(raw, False), EH.Params(sd, head_sds), np.float64)
    assert np.degrees(np.abs(unnorm.cpu().numpy() - ref_u).max()) <= POSE_TOL_DEG
    assert ok.sum() == 128


@pytest.mark.parametrize("hx", HX_MODES)
def test_split_f16_full_batch_65536_properties(hx, head_sds, device):
    """BASELINE.json size in split-f16 mode: batch-position independence (bit-exact), sampled rows within 1e-4 deg of
    the f64 oracle, and agreement with the f32 parity kernel on every one of the 65,536 faces within the same bar."""
    F, B = 1404, 65536
    sd = synth.encoder_state_dict(F, seed=0)
    blob = _blob_hx(sd, head_sds, device, hx)
    x = synth.features(B, F, seed=42)
    xt = torch.from_numpy(x).to(device)
    full = ops.encoder_heads_fwd(xt, blob, F)
    idx = synth.rng(42, 9).permutation(B)[:777]
    it = torch.from_numpy(idx).to(device)
    assert torch.equal(ops.encoder_heads_fwd(xt[it], blob, F), full[it])
    ref = EH.forward_numpy(x[idx], EH.Params(sd, head_sds), np.float64)
    err = np.degrees(np.abs(full[it].cpu().numpy() - ref).max())
    f32 = ops.encoder_heads_fwd(xt, _blob(sd, head_sds, device), F)
    d = torch.rad2deg((full - f32).abs()).max().item()
    _report(f"split_f16_full_batch_{hx}", sampled_vs_f64_deg=err, all_faces_vs_f32_kernel_deg=d)
    assert err <= POSE_TOL_DEG and d <= POSE_TOL_DEG, (err, d)
    assert torch.equal(ops.encoder_heads_fwd(xt[:B - 1], blob, F), full[:B - 1])


@pytest.mark.parametrize("hx", HX_MODES)
def test_split_f16_nan_inf_stay_loud_and_f16_overflow_is_rescued(hx, head_sds, device):
    """NaN/Inf inputs stay in their face (non-finite pose, as in the reference).  A FINITE input whose activations leave f16's
    range (|v| >= 65520) is re-evaluated by the kernel's f32 slow path (encoder_heads_f16x2_rescue.h): finite, accurate, and
    the other faces of the tile keep their bits."""
    F = 1404
    sd = synth.encoder_state_dict(F, seed=0)
    blob = _blob_hx(sd, head_sds, device, hx)
    x = synth.features(200, F, seed=23)
    clean = ops.encoder_heads_fwd(torch.from_numpy(x).to(device), blob, F)
    bad = x.copy()
    bad[5, 100] = np.nan
    bad[70, 1403] = np.inf
    bad[133, 0] = -np.inf
    bad[150, 7] = 7.0e4               # finite in f32, beyond f16: slow path
    bad[151, 7] = 6.0e4               # still inside f16: fast path, must stay accurate
    bad[152] *= 3.0e4                 # overflow in the hidden layers too
    bad[199] = -2.5e5                 # last face of a partial tile
    out, lat = ops.encoder_heads_fwd(torch.from_numpy(bad).to(device), blob, F, return_latent=True)
    rows = torch.ones(200, dtype=torch.bool, device=device)
    rows[[5, 70, 133, 150, 151, 152, 199]] = False
    assert torch.equal(out[rows], clean[rows])
    for r in (5, 70, 133):
        assert not torch.isfinite(out[r]).all(), r
    P = EH.Params(sd, head_sds)
    for r in (150, 151, 152, 199):
        ref = EH.forward_numpy(bad[r:r + 1], P, np.float64)
        ref32 = EH.forward_torch(bad[r:r + 1], P)
        e = np.degrees(np.abs(out[r].cpu().numpy() - ref[0]).max())
        _report(f"split_f16_rescue_{hx}_row{r}", max_abs_deg=e, torch_f32_vs_f64_deg=np.degrees(np.abs(np.asarray(ref32) - ref).max()))
        assert torch.isfinite(out[r]).all() and e <= POSE_TOL_DEG, (r, e)
        assert np.abs(lat[r].cpu().numpy() - EH.encoder_latent_numpy(bad[r:r + 1], P, np.float64)[0]).max() <= 5e-6
    # the layer-per-launch path takes the same slow path: identical bits
    out_s = ops.encoder_heads_fwd_small(torch.from_numpy(bad).to(device), blob, F)
    fin = torch.isfinite(out).all(dim=1)
    assert torch.equal(out_s[fin], out[fin]) and torch.equal(torch.isfinite(out_s).all(dim=1), fin)


@pytest.mark.parametrize("mode", ["f16x2", "f16x2s", "f32"])
@pytest.mark.parametrize("normalize", [True, False])
def test_fx1_degenerate_faces_through_the_fused_path(mode, normalize, head_sds, golden_dir, device):
    """FX1's faces -- ipd == 0 (the reference's 1e-6 branch, FeatureExtractor.py:47-48: features ~1e6), near-degenerate ipd,
    a tiny face, pixel-scale coordinates -- from raw landmarks to pose in ONE launch, both parity modes: finite, and within
    1e-4 deg of the f64 oracle on the reference's own feature rows (the fixture).  One stated exception: the un-normalised
    pixel-scale face (features to 1.9e3, |pre-tanh| ~ 1e2) where the reference's own f32 forward is 1.2e-4 deg off the truth."""
    from nlml_hpe_amd import _lib
    g = np.load(os.path.join(golden_dir, "fx1_normalise.npz"))
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.mode_from_name(mode))).to(device)
    raw = torch.from_numpy(g["landmarks"]).to(device)
    feats = g["features_norm" if normalize else "features_raw"]
    P = EH.Params(sd, head_sds)
    truth = EH.forward_numpy(feats, P, np.float64)
    ref32 = np.asarray(EH.forward_torch(feats, P, num_threads=1))
    pose, valid = ops.landmarks_to_pose(raw, blob, normalize, return_valid=True)
    pose = pose.cpu().numpy()
    assert np.isfinite(pose).all() and valid.all()
    err = np.degrees(np.abs(pose - truth).max(1))
    tol = np.full(16, POSE_TOL_DEG)
    if not normalize:
        tol[5] = 5e-4
    _report(f"fx1_fused_{mode}_norm{int(normalize)}", max_abs_deg=err.max(), max_abs_deg_wo_face5=np.delete(err, 5).max(),
            torch_f32_vs_f64_deg=np.degrees(np.abs(ref32 - truth).max()))
    assert (err <= tol).all(), err
    if mode in HX_MODES:
        small = ops.landmarks_to_pose_small(raw, blob, normalize)
        assert torch.equal(small, torch.from_numpy(pose).to(device))
        two_step = ops.encoder_heads_fwd(torch.from_numpy(feats).to(device), blob, 1404)
        assert torch.equal(two_step, torch.from_numpy(pose).to(device))     # fused == K1 -> K2, slow-path faces included


@pytest.mark.parametrize("hx", HX_MODES)
def test_split_f16_strided_unaligned_input_and_determinism(hx, head_sds, device):
    """Row stride > F, a start that is not 16-byte aligned (scalar-load path), and bit-identical repeat launches."""
    F = 136
    sd = synth.encoder_state_dict(F, seed=0)
    blob = _blob_hx(sd, head_sds, device, hx)
    xfull = synth.features(100, 200, seed=4)
    P = EH.Params(sd, head_sds)
    for off in (8, 3):                                  # 32-byte offset (vector path) and 12-byte offset (scalar path)
        xt = torch.from_numpy(xfull).to(device)[:, off:off + F]
        ref = EH.forward_numpy(xfull[:, off:off + F], P, np.float64)
        out = ops.encoder_heads_fwd(xt, blob, F)
        assert np.degrees(np.abs(out.cpu().numpy() - ref).max()) <= POSE_TOL_DEG
        assert torch.equal(out, ops.encoder_heads_fwd(xt, blob, F))
    # the vector and the scalar load paths stage the same values: identical bits
    xa = torch.from_numpy(np.ascontiguousarray(xfull[:, 3:3 + F])).to(device)
    assert torch.equal(ops.encoder_heads_fwd(xa, blob, F), ops.encoder_heads_fwd(torch.from_numpy(xfull).to(device)[:, 3:3 + F], blob, F))


@pytest.mark.parametrize("mode", ["f16x2", "f16x2s", "f32"])
def test_repeat_launches_are_bit_identical_under_load(mode, head_sds, device):
    """200 back-to-back launches (all 256 CUs busy, clocks and power moving) of the fused path on the same 16,384 faces
    return the same bits every time: the LDS slab rotation, the two-pass layer 0 and the prefetch rings have no
    timing-dependent hazard."""
    from nlml_hpe_amd import _lib
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.mode_from_name(mode))).to(device)
    raw = torch.from_numpy(synth.raw_landmarks(16384, seed=31)).to(device)
    first = ops.landmarks_to_pose(raw, blob, True).clone()
    bad = torch.zeros((), dtype=torch.int64, device=device)
    for _ in range(200):
        bad += (ops.landmarks_to_pose(raw, blob, True) != first).sum()
    assert int(bad.item()) == 0


# ---- artefact producers ---------------------------------------------------------------------
def test_fx8_cosine_table_on_device(tucker_art, golden_dir, repo_root, device):
    """The heads' training table against FX8 (the reference's own cosine() on its config grids): f64; the device's cos
    and numpy's may differ in the last place, which a*cos()+d turns into an ulp of |a|+|d| (yaw row 3: -9.56*cos+9.25),
    so the tolerance is 4 eps * max_j(|a_j| + |d_j|) absolute."""
    import yaml
    from nlml_hpe_amd import artefacts
    g = np.load(os.path.join(golden_dir, "fx8_cosine_table.npz"))
    cfg = yaml.safe_load(open(os.path.join(repo_root, "configs", "config_MlpHeads.yaml")))
    opt = {n: tucker_art[f"optimized_{n}"] for n in ("yaw", "pitch", "roll")}
    tab = artefacts.heads_training_table(cfg, opt, device=device)
    for name in ("yaw", "pitch", "roll"):
        U, ang = tab[name]
        assert np.array_equal(ang, g[f"angles_{name}"])
        ref = g[f"U_{name}"]
        err = np.abs(U.cpu().numpy() - ref).max()
        _report(f"fx8_cosine_table_{name}", max_abs=err)
        scale = float((np.abs(opt[name][:, 0]) + np.abs(opt[name][:, 3])).max())
        assert err <= 4 * np.finfo(np.float64).eps * scale, (err, scale)


def test_mode5_product_on_device(tucker_art, device):
    """W = core x_5 U_feat: bit-exact against the r-ascending f32 fma chain on a small case; at the shipped shape
    ([135,1404] x [1404,1404], orthonormal synthetic U_feat) within f32 accumulation error of the f64 product, and the
    product undoes itself (W x_5 U_feat^T == core).  U_feat is not among the shipped artefacts (parity unpinned)."""
    from oracle import artefacts as AR
    rng = np.random.default_rng(5)
    core_s = rng.standard_normal((2, 3, 3, 3, 70)).astype(np.float32)
    U_s = rng.standard_normal((45, 70)).astype(np.float32)
    got = ops.mode5_product(torch.from_numpy(core_s).to(device), torch.from_numpy(U_s).to(device)).cpu().numpy()
    assert got.shape == (2, 3, 3, 3, 45)
    assert np.array_equal(got.reshape(-1, 45), AR.mode5_product_chain_f32(core_s.reshape(-1, 70), U_s))
    core = tucker_art["CoreTensor"].astype(np.float32)                     # [5,3,3,3,1404]
    Uf, _ = np.linalg.qr(rng.standard_normal((1404, 1404)))
    Uf = Uf.astype(np.float32)
    W = ops.mode5_product(torch.from_numpy(core).to(device), torch.from_numpy(Uf).to(device))
    W64 = AR.mode5_product(core, Uf)
    rel = np.abs(W.cpu().numpy() - W64).max() / np.abs(W64).max()
    back = ops.mode5_product(W, torch.from_numpy(np.ascontiguousarray(Uf.T)).to(device)).cpu().numpy()
    rel_back = np.abs(back - core).max() / np.abs(core).max()
    _report("mode5_product", rel_vs_f64=rel, roundtrip_rel=rel_back)
    assert rel <= 2e-5 and rel_back <= 1e-4


def test_fx9_td_gradient_on_device(tucker_art, golden_dir, device):
    """compute_gradient (TD_Tester.py:60-102) on the device against FX9 (the reference's own numbers).  The three angle
    derivatives are f64 end to end: tolerance 1e-10 of their scale (einsums regrouped into GEMMs, so rounding, not bits).
    The identity-mode part contains an einsum whose operands are all f32 (:96), which numpy evaluates in f32 in its own
    summation order: that part can only agree to f32 rounding, tolerance 1e-6 of its scale."""
    from nlml_hpe_amd import TD_Tester as HT
    g = np.load(os.path.join(golden_dir, "fx9_td_gradient.npz"))["grad"]
    f = np.load(os.path.join(golden_dir, "fx4_td_objective.npz"))
    Py, Pp, Pr = tucker_art["optimized_yaw"][:3], tucker_art["optimized_pitch"][:3], tucker_art["optimized_roll"][:3]
    G = HT.compute_gradient_batch(f["params"], tucker_art["W"], f["x"], Py, Pp, Pr)
    e_ang = np.abs(G[:, :3] - g[:, :3]).max() / np.abs(g[:, :3]).max()
    e_uid = np.abs(G[:, 3:] - g[:, 3:]).max() / np.abs(g[:, 3:]).max()
    one = HT.compute_gradient(f["params"][3], tucker_art["W"], torch.from_numpy(f["x"][3]), Py, Pp, Pr)
    _report("fx9_td_gradient", angles_rel_of_scale=e_ang, u_id_rel_of_scale=e_uid)
    assert e_ang <= 1e-10 and e_uid <= 1e-6, (e_ang, e_uid)
    assert np.abs(one[:3] - g[3][:3]).max() <= 1e-10 * np.abs(g[:, :3]).max()
    assert np.abs(one[3:] - g[3][3:]).max() <= 1e-6 * np.abs(g[:, 3:]).max()


# ---- split-f16 mode, one launch per layer (small batches) ---------------------------------------------------------
@pytest.mark.parametrize("hx", HX_MODES)
@pytest.mark.parametrize("F,B", [(1404, 1), (1404, 64), (1404, 65), (1404, 200), (1404, 500), (1404, 2000), (136, 77), (13, 5), (1407, 130)])
def test_small_batch_path_is_bit_identical_to_the_fused_kernel(F, B, hx, head_sds, device):
    """nlml_encoder_heads_fwd_small runs every layer as its own launch over (neuron blocks x tiles) with activations in a
    workspace; per output it issues the same MFMAs in the same order as the fused split-f16 kernel, so pose, latent and
    validity must be the same bits (and with them every parity result of the fused kernel carries over)."""
    sd = synth.encoder_state_dict(F, seed=3)
    blob = _blob_hx(sd, head_sds, device, hx)
    x = synth.features(B, F, seed=9)
    if B > 3:
        x[3] = 0.0
    xt = torch.from_numpy(x).to(device)
    a, la, va = ops.encoder_heads_fwd(xt, blob, F, return_latent=True, return_valid=True)
    b, lb, vb = ops.encoder_heads_fwd_small(xt, blob, F, return_latent=True, return_valid=True)
    assert torch.equal(a, b) and torch.equal(la, lb) and torch.equal(va, vb)
    ref = EH.forward_numpy(x, EH.Params(sd, head_sds), np.float64)
    assert np.degrees(np.abs(b.cpu().numpy() - ref).max()) <= POSE_TOL_DEG


@pytest.mark.parametrize("hx", HX_MODES)
def test_small_batch_path_strided_and_unaligned_rows(hx, head_sds, device):
    """The layer-per-launch path's pre-pass has a 16-byte-load form (aligned base, stride and width multiples of four) and a scalar
    form: both stage the same values, whatever the row stride and the start's alignment -- same bits as the fused kernel on the
    same view, and as the contiguous copy."""
    F = 1404
    sd = synth.encoder_state_dict(F, seed=0)
    blob = _blob_hx(sd, head_sds, device, hx)
    xfull = torch.from_numpy(synth.features(150, F + 12, seed=6)).to(device)
    for off in (8, 3):                                  # 32-byte offset, stride F + 12 (vector form); 12-byte offset (scalar form)
        view = xfull[:, off:off + F]
        a = ops.encoder_heads_fwd(view, blob, F)
        b = ops.encoder_heads_fwd_small(view, blob, F)
        c = ops.encoder_heads_fwd_small(view.contiguous(), blob, F)
        assert torch.equal(a, b) and torch.equal(b, c)


@pytest.mark.parametrize("F,B", [(1404, 300), (1404, 64), (136, 77), (13, 5), (1407, 130)])
def test_strict_eight_wave_kernel_equals_the_layer_per_launch_path(F, B, head_sds, device):
    """NLML_MODE_F16X2S runs on the eight-wave kernel (encoder_heads_f16x2_w8.hip: a trunk job shared by a pair of waves, layer 0 in
    two passes with layer 1's accumulators parked in LDS, waves 4-7 ending before the tail).  The layer-per-launch path is an
    independent implementation of the same MFMA order per accumulator (one wave per neuron block, operands streamed from global
    memory, no LDS in the big layers): pose, latent and validity are the same bits -- from features (incl. widths that are not a
    multiple of 4: the scalar staging path) and, at the reference width, from raw landmarks with the fused normalisation.  (Until
    the end of round 4 this test compared the eight-wave kernel with the four-wave strict instantiation, which it replaced.)"""
    sd = synth.encoder_state_dict(F, seed=3)
    blob = _blob_hx(sd, head_sds, device, "f16x2s")
    x = synth.features(B, F, seed=9)
    x[3] = 0.0
    xt = torch.from_numpy(x).to(device)
    a = ops.encoder_heads_fwd(xt, blob, F, return_latent=True, return_valid=True)
    b = ops.encoder_heads_fwd_small(xt, blob, F, return_latent=True, return_valid=True)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    if F == 1404:
        raw = torch.from_numpy(synth.raw_landmarks(B, seed=4)).to(device)
        ar = ops.landmarks_to_pose(raw, blob, True, return_latent=True, return_valid=True)
        br = ops.landmarks_to_pose_small(raw, blob, True, return_latent=True, return_valid=True)
        for u, v in zip(ar, br):
            assert torch.equal(u, v)
    ref = EH.forward_numpy(x, EH.Params(sd, head_sds), np.float64)
    assert np.degrees(np.abs(a[0].cpu().numpy() - ref).max()) <= POSE_TOL_DEG


@pytest.mark.parametrize("hx", HX_MODES)
def test_small_batch_path_landmarks_workspace_and_errors(hx, head_sds, device):
    """Raw-landmark entry of the small-batch path: same bits as the fused launch incl. the validity mask; garbage (NaN)
    in the workspace and in the dead faces of a partial tile does not leak; wrong blob mode / short workspace are refused."""
    from nlml_hpe_amd import _lib
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = _blob_hx(sd, head_sds, device, hx)
    raw = synth.raw_landmarks(131, seed=17)
    raw[3] = np.array([0.25, 0.5, 0.75], np.float32)
    raw[64] = 0.0
    rt = torch.from_numpy(raw).to(device)
    a, va = ops.landmarks_to_pose(rt, blob, True, return_valid=True)
    need = _lib.lib().nlml_encoder_heads_small_workspace_bytes(131, 1404)
    ws = torch.full((need,), 0xFF, dtype=torch.uint8, device=device)          # every f16 in it is a NaN
    b, vb = ops.landmarks_to_pose_small(rt, blob, True, return_valid=True, workspace=ws)
    assert torch.equal(a, b) and torch.equal(va, vb)
    c = ops.landmarks_to_pose_small(rt, blob, False, workspace=ws)
    assert torch.equal(c, ops.landmarks_to_pose(rt, blob, False))
    with pytest.raises(_lib.NlmlError):
        ops.landmarks_to_pose_small(rt, blob, True, workspace=ws[: need // 2])
    with pytest.raises(_lib.NlmlError):
        ops.landmarks_to_pose_small(rt, _blob(sd, head_sds, device), True)    # f32 blob: split-f16 only


@pytest.mark.parametrize("hx", HX_MODES)
def test_model_dispatches_small_batches_and_graph_replays(hx, head_sds, device):
    """HIPPoseModel in split-f16 mode uses the small-batch path up to SMALL_BATCH_MAX faces (same bits either way), and
    the launch sequence (five launches, seven in the strict mode) replays from a hipGraph."""
    from nlml_hpe_amd.model import HIPPoseModel
    sd = synth.encoder_state_dict(1404, seed=0)
    model = HIPPoseModel(sd, head_sds, device=device, mode=hx)
    raw = torch.from_numpy(synth.raw_landmarks(300, seed=8)).to(device)
    assert model._small(300) and not model._small(model.small_batch_max(hx) + 1) and model.small_batch_max("f32") == 0
    assert torch.equal(model.from_landmarks(raw), ops.landmarks_to_pose(raw, model.blob, True))
    feats = ops.normalize_ipd(raw, True)
    assert torch.equal(model.forward_packed(feats), ops.encoder_heads_fwd(feats, model.blob, 1404))
    static_in = raw.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model.from_landmarks(static_in)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_out = model.from_landmarks(static_in)
    static_in.copy_(torch.from_numpy(synth.raw_landmarks(300, seed=9)).to(device))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, ops.landmarks_to_pose(static_in, model.blob, True))


def test_empty_batches_are_no_ops_in_every_k2_path(head_sds, device):
    """B = 0 (an empty shard, a tick with no faces): every K2 entry returns empty results without launching."""
    from nlml_hpe_amd import _lib
    from nlml_hpe_amd.model import HIPPoseModel
    sd = synth.encoder_state_dict(1404, seed=0)
    x0 = torch.empty((0, 1404), dtype=torch.float32, device=device)
    r0 = torch.empty((0, 468, 3), dtype=torch.float32, device=device)
    for mode in ("f32", "f16x2", "f16x2s", "bf16"):
        blob = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.mode_from_name(mode))).to(device)
        assert tuple(ops.encoder_heads_fwd(x0, blob, 1404).shape) == (0, 3)
        out, lat, val = ops.landmarks_to_pose(r0, blob, True, return_latent=True, return_valid=True)
        assert tuple(out.shape) == (0, 3) and tuple(lat.shape) == (0, 9) and tuple(val.shape) == (0,)
    blob = _blob_hx(sd, head_sds, device)
    assert tuple(ops.encoder_heads_fwd_small(x0, blob, 1404).shape) == (0, 3)
    assert tuple(ops.landmarks_to_pose_small(r0, blob, True).shape) == (0, 3)
    model = HIPPoseModel(sd, head_sds, device=device)
    assert tuple(model.from_landmarks(r0).shape) == (0, 3) and tuple(model.forward_packed(x0).shape) == (0, 3)


# ---- the reference's operating range: FX3b, FX2b -------------------------------------------------
# At poses over +-45 deg two correct f32 evaluations of this network differ by ~1e-4 deg (tests/test_oracle_golden.py,
# FX3B_F32_ORDER_BAND_DEG): the reference differs from itself by 9.1e-5 deg (batched vs batch-1 calls) and from the f64 truth
# by 8.2e-5 deg.  So the assertions are: (1) against the f64 TRUTH each kernel is at least as good as the stated bound,
# (2) against the reference's batched output within the bound + the reference's own distance from the truth.
FX3B_REF_VS_TRUTH_DEG = 8.3e-5


@pytest.mark.parametrize("mode", ["f16x2", "f16x2s", "f32"])
def test_fx3b_reference_range_golden(mode, head_sds, golden_dir, device):
    import fixture_models
    from nlml_hpe_amd import _lib
    g, sd, x = fixture_models.fx3b(golden_dir)
    blob = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.mode_from_name(mode))).to(device)
    out, lat = ops.encoder_heads_fwd(torch.from_numpy(x).to(device), blob, 1404, return_latent=True)
    out, lat = out.cpu().numpy(), lat.cpu().numpy()
    P = EH.Params(sd, head_sds)
    truth = EH.forward_numpy(x, P, np.float64)
    e_ref = np.degrees(np.abs(out - g["rad"]).max())
    e_b1 = np.degrees(np.abs(out - g["rad_b1"]).max())
    e_truth = np.degrees(np.abs(out - truth).max())
    e_lat = np.abs(lat - EH.encoder_latent_numpy(x, P, np.float64)).max()
    _report(f"fx3b_{mode}", vs_reference_batched_deg=e_ref, vs_reference_batch1_deg=e_b1, vs_f64_truth_deg=e_truth,
            latent_abs=e_lat, reference_vs_truth_deg=np.degrees(np.abs(g["rad"] - truth).max()),
            reference_batched_vs_batch1_deg=np.degrees(np.abs(g["rad"] - g["rad_b1"]).max()))
    bound = FX3B_KERNEL_VS_TRUTH_DEG[mode]
    assert e_truth <= bound, (mode, e_truth)
    assert e_ref <= bound + FX3B_REF_VS_TRUTH_DEG, (mode, e_ref)


FX3B_KERNEL_VS_TRUTH_DEG = {"f16x2": 1e-4, "f16x2s": 1e-4, "f32": 1e-4}


@pytest.mark.parametrize("mode", ["f16x2", "f16x2s", "f32"])
def test_fx3c_no_worse_than_the_reference_itself(mode, head_sds, golden_dir, device):
    """Parity where the reference operates, stated as a statistic over 16,384 faces (FX3c = FX3b's model, poses over the trained
    bins): the kernel's distance from the f64 truth is NO WORSE THAN THE REFERENCE'S OWN in p50, p99 and max (5 % slack), and the
    fraction of faces that differ from the reference's batched output by more than 1e-4 deg is reported next to the fraction by
    which the reference differs from ITSELF between its batched and its one-face calls (NLML_HPE_Test.py:262-272 makes the latter)."""
    import fixture_models
    from nlml_hpe_amd import _lib
    g, sd, x = fixture_models.fx3c(golden_dir)
    blob = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.mode_from_name(mode))).to(device)
    out = ops.encoder_heads_fwd(torch.from_numpy(x).to(device), blob, 1404).cpu().numpy()
    truth = EH.forward_numpy(x, EH.Params(sd, head_sds), np.float64)
    k = fixture_models.error_stats(out, truth)
    r = fixture_models.error_stats(g["rad"], truth)
    r1 = fixture_models.error_stats(g["rad_b1"], truth)
    vs_ref = fixture_models.error_stats(out, g["rad"])
    vs_b1 = fixture_models.error_stats(out, g["rad_b1"])       # the reference's REAL call shape: one face per call (NLML_HPE_Test.py:262-272)
    ref_self = fixture_models.error_stats(g["rad_b1"], g["rad"])
    _report(f"fx3c_{mode}", kernel_p50=k["p50"], kernel_p99=k["p99"], kernel_max=k["max"], kernel_frac_above=k["frac_above_1e-4"],
            ref_p50=r["p50"], ref_p99=r["p99"], ref_max=r["max"], ref_frac_above=r["frac_above_1e-4"],
            ref_batch1_p50=r1["p50"], ref_batch1_p99=r1["p99"], ref_batch1_max=r1["max"],
            kernel_vs_ref_max=vs_ref["max"], kernel_vs_ref_frac_above=vs_ref["frac_above_1e-4"],
            kernel_vs_ref_p50=vs_ref["p50"], kernel_vs_ref_p99=vs_ref["p99"],
            kernel_vs_batch1_p50=vs_b1["p50"], kernel_vs_batch1_p99=vs_b1["p99"], kernel_vs_batch1_max=vs_b1["max"],
            kernel_vs_batch1_frac_above=vs_b1["frac_above_1e-4"],
            ref_batch1_vs_batched_max=ref_self["max"], ref_batch1_vs_batched_frac_above=ref_self["frac_above_1e-4"])
    # f32 kernel (layers 0 to 3 summed in blocks of 128 k): no worse than the reference in every statistic, nothing beyond 1e-4 deg.
    # split-f16 kernel (the fast mode): 1.10 / 1.08 / 1.24x the reference's distance from the truth in p50 / p99 / max (measured: 1.86e-5 /
    # 5.88e-5 / 1.22e-4 deg; layer 0 and layer 1's first K half run on single accumulators -- no register for a second set in the
    # one-pass form -- the rest on split ones), 0.024 % of the faces beyond 1e-4 deg of the truth.  Bounds = measured + 20 %, not looser.
    # strict-fast kernel (f16x2s: the small products of each K step accumulate apart in layers 0 to 2): measured 1.49e-5 /
    # 4.65e-5 / 9.2e-5 deg = 0.85-0.93x the reference's distance: held to the f32 kernel's bounds.
    ratio = {"f32": (1.05, 1.05, 1.05), "f16x2s": (1.05, 1.05, 1.05), "f16x2": (1.32, 1.30, 1.49)}[mode]
    for s_, q_ in zip(("p50", "p99", "max"), ratio):
        assert k[s_] <= q_ * r[s_], (mode, s_, k, r)
    # The ratio above is against the PINNED reference (the fixture's host); the reference's own distance moves with its BLAS (1.69e-5 deg
    # p50 in the build container, 1.31e-5 on the GPU box's host: profiles/r04_parity_soak_1M.json).  So the strict modes are also held to
    # ABSOLUTE bounds against the f64 truth, which no reference host enters: p50 <= 1.6e-5, p99 <= 5.0e-5, max <= 1e-4 deg at 16,384 faces
    # (measured: f16x2s 1.50e-5 / 4.65e-5 / 9.2e-5, f32 1.25e-5 / 4.0e-5 / 8.7e-5).
    if mode != "f16x2":
        assert k["p50"] <= 1.6e-5 and k["p99"] <= 5.0e-5 and k["max"] <= 1.0e-4, (mode, k)
    # Against the reference's REAL call shape (one face per call, NLML_HPE_Test.py:262-272; its results are 9.2e-6 deg p50 from the
    # truth, closer than its batched call's 1.69e-5): measured p50 / p99 / max / fraction beyond 1e-4 deg -- f16x2s 1.77e-5 / 5.52e-5 /
    # 9.65e-5 / 0, f32 1.54e-5 / 4.70e-5 / 8.54e-5 / 0, f16x2 (opt-in) 2.05e-5 / 6.57e-5 / 1.35e-4 / 0.043 %.  Bounds = measured + 20 %
    # (the fraction: + 2 faces).  For comparison the reference's batched call against its own one-face calls: 1.88e-5 / 6.06e-5 / 1.23e-4 / 0.012 %.
    b1_bound = {"f16x2s": (2.13e-5, 6.63e-5, 1.16e-4, 2.0 / len(x)), "f32": (1.85e-5, 5.64e-5, 1.03e-4, 2.0 / len(x)),
                "f16x2": (2.46e-5, 7.89e-5, 1.63e-4, 5.2e-4)}[mode]
    for s_, q_ in zip(("p50", "p99", "max", "frac_above_1e-4"), b1_bound):
        assert vs_b1[s_] <= q_, (mode, s_, vs_b1)
    if mode != "f16x2":   # the strict modes are closer to the reference's one-face results than its own batched call is
        assert vs_b1["p50"] <= ref_self["p50"] and vs_b1["max"] <= ref_self["max"], (vs_b1, ref_self)
    if mode == "f16x2s":
        # nothing beyond 1e-4 deg of the truth.  Against the reference's batched output: two evaluations with INDEPENDENT errors of
        # 1.5e-5 and 1.7e-5 deg p50 -- 0.073 % of the faces differ by more than 1e-4 deg, max 1.34e-4 deg (measured; bounds +20 %).
        # (The reference against ITSELF, batched vs one-face calls: 0.012 %, 1.23e-4 deg -- its one-face results are 9e-6 deg p50 from the truth.)
        assert k["max"] <= POSE_TOL_DEG and k["frac_above_1e-4"] == 0.0
        assert vs_ref["max"] <= 1.6e-4 and vs_ref["frac_above_1e-4"] <= 9e-4, vs_ref
    elif mode == "f32":
        assert k["max"] <= POSE_TOL_DEG and k["frac_above_1e-4"] == 0.0
        # against the reference's batched output (two f32 evaluations, each ~4e-5 deg from the truth in the tail): 0.012 % of the faces
        # differ by more than 1e-4 deg, max 1.20e-4 deg -- exactly what the reference shows against ITSELF (batched vs one-face calls:
        # 0.012 %, 1.23e-4 deg)
        assert vs_ref["max"] <= 1.45e-4 and vs_ref["frac_above_1e-4"] <= ref_self["frac_above_1e-4"] + 2.0 / len(x), (vs_ref, ref_self)
    else:
        assert k["max"] <= 1.47e-4 and k["frac_above_1e-4"] <= 3e-4, k
        assert vs_ref["max"] <= 2.0e-4 and vs_ref["frac_above_1e-4"] <= 2.1e-3, vs_ref       # measured 1.67e-4 deg, 0.17 %


@pytest.mark.parametrize("mode", ["f16x2", "f16x2s", "f32"])
def test_fx2b_heads_operating_points_golden(mode, head_sds, golden_dir, device):
    """The heads at their own inputs (rows of U_*, trained cosine curves to +-60 deg) through the fused forward, against what
    the reference's CombinedAnglePredictionModel returned on the same x and weights: the 1e-4 deg bar, both parity modes."""
    import fixture_models
    from nlml_hpe_amd import _lib
    g, sd, x = fixture_models.fx2b(golden_dir)
    blob = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.mode_from_name(mode))).to(device)
    out, lat = ops.encoder_heads_fwd(torch.from_numpy(x).to(device), blob, 136, return_latent=True)
    e = np.degrees(np.abs(out.cpu().numpy() - g["rad"]).max())
    e_lat = np.abs(lat.cpu().numpy() - g["latent"]).max()
    _report(f"fx2b_{mode}", max_abs_deg=e, latent_abs=e_lat)
    assert e_lat <= 5e-7, e_lat
    assert e <= POSE_TOL_DEG, e
    if mode in HX_MODES:
        small = ops.encoder_heads_fwd_small(torch.from_numpy(x).to(device), blob, 136)
        assert torch.equal(small, out)


def test_video_post_skips_non_finite_pose(head_sds, device):
    """A stream whose pose is NaN/Inf in one tick keeps its EMA state (like a no-face frame) instead of being poisoned for good."""
    from nlml_hpe_amd.model import HIPPoseModel
    from nlml_hpe_amd.video import VideoPoseTracker
    mdl = HIPPoseModel(synth.encoder_state_dict(1404, seed=0), head_sds, device=device)
    S = 8
    tr = VideoPoseTracker(mdl, S, 1920, 1080)
    raw = torch.from_numpy(synth.raw_landmarks(S, seed=31)).to(device)
    pose = mdl.from_landmarks(raw)
    tr.post(pose, raw)
    before = tr.state.clone()
    bad = pose.clone()
    bad[2, 1] = float("nan")
    bad[5, 0] = float("inf")
    tr.post(bad, raw)
    after = tr.state
    assert torch.equal(after[[2, 5]], before[[2, 5]])                     # untouched, count not advanced
    ok = [0, 1, 3, 4, 6, 7]
    assert torch.isfinite(after).all() and (after[ok, 5] == 2).all()
    assert tr.updated.tolist() == [1, 1, 0, 1, 1, 0, 1, 1]               # ... and the skip is REPORTED
    tr.post(pose, raw)                                                    # and the stream goes on
    assert (tr.state[[2, 5], 5] == 2).all() and torch.isfinite(tr.smoothed).all() and tr.updated.all()
    # tick(): the returned mask is "this tick was applied", not just "a face was found": a NaN landmark gives a NaN pose
    raw_nan = raw.clone()
    raw_nan[4, 10, 0] = float("nan")
    raw_nan[6] = 0.0                                                       # no face
    sm, c, ep, valid = tr.tick(raw_nan)
    assert valid.tolist() == [True, True, True, True, False, True, False, True]


def test_ops_follow_the_tensors_device(head_sds):
    """a model on a GPU that is not the current one must launch THERE (device guard + that device's stream), and
    operands on different GPUs are refused.  Needs two GPUs; the one-GPU box skips it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    from nlml_hpe_amd import _lib
    sd = synth.encoder_state_dict(136, seed=0)
    x = synth.features(100, 136, seed=2)
    d0, d1 = torch.device("cuda:0"), torch.device("cuda:1")
    blob0, blob1 = _blob_hx(sd, head_sds, d0), _blob_hx(sd, head_sds, d1)
    torch.cuda.set_device(0)
    out1 = ops.encoder_heads_fwd(torch.from_numpy(x).to(d1), blob1, 136)       # current device 0, operands on 1
    out0 = ops.encoder_heads_fwd(torch.from_numpy(x).to(d0), blob0, 136)
    assert out1.device == d1 and torch.equal(out1.cpu(), out0.cpu())
    with pytest.raises(_lib.NlmlError):
        ops.encoder_heads_fwd(torch.from_numpy(x).to(d0), blob1, 136)


def test_video_entry_point_shards_streams_over_ranks(repo_root, device, tmp_path):
    """BASELINE config 5 at N > 1: two ranks (one-GPU rehearsal: gloo, shared device) each carry a contiguous block of the 64
    streams; the collated output equals the single-process run bit for bit (streams are independent; no per-tick exchange)."""
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=repo_root)
    one = subprocess.run([sys.executable, "generatePose_on_video.py", "--source", "synthetic", "--save_output", "True",
                          "--output_path", str(tmp_path / "one.npz")], cwd=repo_root, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    env2 = dict(env, NLML_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    port = 29700 + (os.getpid() % 200)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), "generatePose_on_video.py", "--source", "synthetic", "--save_output", "True",
                          "--output_path", str(tmp_path / "two.npz")], cwd=repo_root, env=env2, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, (two.stdout[-2000:], two.stderr[-3000:])
    assert "[rank 1/2, streams 32..63]" in two.stdout
    a, b = np.load(tmp_path / "one.npz"), np.load(tmp_path / "two.npz")
    for k in ("smoothed_deg", "endpoints", "valid"):
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("mode", ["f16x2", "f16x2s", "f32"])
def test_batch_beyond_2_31_elements_is_indexed_in_64_bits(mode, head_sds, device):
    """3,100,000 faces = 4.35e9 input elements (17 GB): every row offset in K1/K2 has to be 64-bit.  A face's result does not
    depend on its position in the batch, so windows of the big batch (first tile, around the 2^31- and 2^32-element marks, a
    ragged end) must equal the same faces run as small batches, bit for bit -- validity mask and no-face rows included."""
    B = 3_100_000
    free, _ = torch.cuda.mem_get_info(device)
    if free < 80 * 2**30:
        pytest.skip("needs ~60 GB of device memory")
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = _blob_hx(sd, head_sds, device, mode) if mode in HX_MODES else _blob(sd, head_sds, device)
    base = synth.raw_landmarks(4096, seed=21)
    base[::97] = 0.0                                                   # "no face" rows (generatePose_on_video.py:118)
    base = torch.from_numpy(base).to(device)
    raw = base.repeat(B // 4096 + 1, 1, 1)[:B].contiguous()           # 4,096 distinct faces tiled to the big batch ...
    bump = torch.arange(B, device=device, dtype=torch.float32) * 1e-7  # ... and made distinct per position
    bump[::97] = 0.0                                                   # (B's tiling period 4,096 is not a multiple of 97: recompute)
    noface = raw.abs().amax(dim=(1, 2)) == 0
    raw[:, 5, 1] += torch.where(noface, torch.zeros_like(bump), bump + 1e-7)
    del bump
    pose, valid = ops.landmarks_to_pose(raw, blob, True, return_valid=True)
    torch.cuda.synchronize()
    assert torch.equal(valid, ~noface)
    mark = 2**31 // 1404                                               # the face at the 2^31-element mark (2^33 bytes)
    for lo, hi in ((0, 64), (mark // 4 - 64, mark // 4 + 64), (mark - 100, mark + 100), (2 * mark - 70, 2 * mark + 61),
                   (B - 1000, B), (B - 37, B)):
        p2, v2 = ops.landmarks_to_pose(raw[lo:hi].contiguous(), blob, True, return_valid=True)
        assert torch.equal(valid[lo:hi], v2), (lo, hi)
        a, b = pose[lo:hi].contiguous(), p2
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (mode, lo, hi, float((a - b).abs().max()))
    assert bool(torch.isfinite(pose[valid]).all())
    if mode in HX_MODES:    # K1 alone and K2 from features at the same size: the two-launch path leaves the fused launch's bits
        feats = ops.normalize_ipd(raw, True)
        for lo, hi in ((mark - 100, mark + 100), (2 * mark - 70, 2 * mark + 61), (B - 37, B)):
            assert torch.equal(feats[lo:hi], ops.normalize_ipd(raw[lo:hi].contiguous(), True)), (lo, hi)
        pose2 = ops.encoder_heads_fwd(feats, blob, 1404)
        assert torch.equal(pose2[valid].view(torch.int32), pose[valid].view(torch.int32))
        del feats, pose2
    del pose, raw
    torch.cuda.empty_cache()


@pytest.mark.parametrize("order", ["fast", "reference"])
def test_tucker_objective_beyond_2_31_elements(order, tucker_art, device):
    """1,600,000 evaluations on 1,600,000 distinct x rows (2.25e9 elements, 9 GB): row offsets are 64-bit.  Evaluations are
    independent, so windows around the 2^31-element mark and at the ragged end equal small launches of the same rows, bit for bit
    -- directly and through x_index."""
    N = 1_600_000
    free, _ = torch.cuda.mem_get_info(device)
    if free < 30 * 2**30:
        pytest.skip("needs ~20 GB of device memory")
    cp = torch.from_numpy(_cos_params(tucker_art)).to(device)
    Wm = torch.from_numpy(tucker_art["W"].reshape(135, 1404)).to(device)
    base = torch.from_numpy(synth.features(4096, 1404, seed=31)).to(device)
    X = base.repeat(N // 4096 + 1, 1)[:N].contiguous()
    X[:, 7] += torch.arange(N, device=device, dtype=torch.float32) * 1e-7
    P = torch.from_numpy(synth.tucker_params(4096, 5, seed=32)).to(device).repeat(N // 4096 + 1, 1)[:N].contiguous()
    err = ops.tucker_objective(Wm, X, P, cp, order=order)
    torch.cuda.synchronize()
    mark = 2**31 // 1404
    for lo, hi in ((0, 16), (mark - 50, mark + 50), (N - 1001, N), (N - 5, N)):
        small = ops.tucker_objective(Wm, X[lo:hi].contiguous(), P[lo:hi].contiguous(), cp, order=order)
        assert torch.equal(err[lo:hi].view(torch.int64), small.view(torch.int64)), (order, lo, hi)
        idx = torch.arange(lo, hi, device=device, dtype=torch.int32)
        via = ops.tucker_objective(Wm, X, P[lo:hi].contiguous(), cp, x_index=idx, order=order)
        assert torch.equal(via.view(torch.int64), small.view(torch.int64)), (order, lo, hi, "x_index")
    assert bool(torch.isfinite(err).all())


# ---- round 3: paths that had never executed (ADVICE) -------------------------------------------------
def test_registered_custom_ops_match_the_python_ops(head_sds, tucker_art, device):
    """torch.ops.nlml_hpe.* (SURVEY.md 8b "Underlying op"), registered from COMPILED code (csrc/torch_ops.cpp, a TORCH_LIBRARY shim over
    the C ABI) -- the schemas reach the same C-ABI launches as nlml_hpe_amd.ops: identical bits, for every registered op."""
    from nlml_hpe_amd import _lib
    raw = torch.from_numpy(synth.raw_landmarks(200, seed=3)).to(device)
    feats = ops.normalize_ipd(raw, True)
    assert torch.equal(torch.ops.nlml_hpe.normalize_ipd(raw, True), feats)
    assert torch.equal(torch.ops.nlml_hpe.normalize_ipd(raw, False), ops.normalize_ipd(raw, False))
    sd = synth.encoder_state_dict(1404, seed=0)
    for mode in ("f16x2s", "f16x2", "f32", "bf16"):
        blob = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.mode_from_name(mode))).to(device)
        assert torch.equal(torch.ops.nlml_hpe.encoder_heads_fwd(feats, blob, 1404), ops.encoder_heads_fwd(feats, blob, 1404)), mode
        assert torch.equal(torch.ops.nlml_hpe.landmarks_to_pose(raw, blob, True), ops.landmarks_to_pose(raw, blob, True)), mode
        if mode in HX_MODES:     # the layer-per-launch forms, workspace passed explicitly
            ws = torch.empty(_lib.lib().nlml_encoder_heads_small_workspace_bytes(200, 1404), dtype=torch.uint8, device=device)
            assert torch.equal(torch.ops.nlml_hpe.encoder_heads_fwd_small(feats, blob, 1404, ws), ops.encoder_heads_fwd(feats, blob, 1404)), mode
            assert torch.equal(torch.ops.nlml_hpe.landmarks_to_pose_small(raw, blob, True, ws), ops.landmarks_to_pose(raw, blob, True)), mode
            with pytest.raises(Exception):
                torch.ops.nlml_hpe.landmarks_to_pose_small(raw, blob, True, ws[:1024])               # too small a workspace: the ABI's error, raised
    Wm = torch.from_numpy(tucker_art["W"].reshape(135, 1404)).to(device)
    cp = torch.from_numpy(_cos_params(tucker_art)).to(device)
    P = torch.from_numpy(synth.tucker_params(40, 5, seed=5)).to(device)
    X = feats[:40].contiguous()
    ref_order = ops.tucker_objective(Wm, X, P, cp, order="reference")
    assert torch.equal(torch.ops.nlml_hpe.tucker_objective(Wm, X, P, cp), ref_order)                 # the default IS the reference order
    assert torch.equal(torch.ops.nlml_hpe.tucker_objective(Wm, X, P, cp, "fast"), ops.tucker_objective(Wm, X, P, cp, order="fast"))
    with pytest.raises(Exception):
        torch.ops.nlml_hpe.encoder_heads_fwd(feats.cpu(), blob, 1404)                                 # no CPU path behind the op either
    with pytest.raises(Exception):
        torch.ops.nlml_hpe.tucker_objective(Wm, X, P, cp, "sideways")
    # TD end to end: FX5-sized run, both the op and the Python wrapper walk the same device Powell
    Xg = feats[:16].contiguous()
    got = torch.ops.nlml_hpe.tucker_powell(Wm, Xg, cp)
    want = ops.tucker_powell(Wm, Xg, cp)
    for g_, k_ in zip(got, ("x", "fun", "nfev", "nit", "status")):
        assert torch.equal(g_, want[k_]), k_
    # one video tick: state updated in place, outputs and the "applied" mask as nlml_video_post_ex gives them to video.py
    from nlml_hpe_amd.model import HIPPoseModel
    from nlml_hpe_amd.video import ALPHA, AXIS_SIZE, MAX_CENTER_JUMP, VideoPoseTracker
    mdl = HIPPoseModel(sd, head_sds, device=device)
    tr = VideoPoseTracker(mdl, 64, 1920, 1080)
    state = torch.zeros((64, 6), dtype=torch.float64, device=device)
    sm2, c2, ep2 = (torch.zeros(sh, dtype=torch.float64, device=device) for sh in ((64, 3), (64, 2), (64, 3, 2)))
    upd2 = torch.zeros((64,), dtype=torch.uint8, device=device)
    r64 = raw[:64].contiguous()
    r64[5] = 0.0                                                                                       # a stream without a face
    for _ in range(3):
        pose, valid = mdl.from_landmarks(r64, normalize=True, return_valid=True)
        sm, c, ep = tr.post(pose, r64, valid)
        torch.ops.nlml_hpe.video_post(pose, r64, valid.to(torch.uint8), 1920.0, 1080.0, ALPHA, MAX_CENTER_JUMP, AXIS_SIZE, state, sm2, c2, ep2, upd2)
        assert torch.equal(sm, sm2) and torch.equal(c, c2) and torch.equal(ep, ep2) and torch.equal(tr.updated, upd2)
        assert torch.equal(state, tr.state) and not bool(upd2[5]) and bool(upd2[4])
    ang = torch.linspace(-1.0, 1.0, 101, device=device)
    assert torch.equal(torch.ops.nlml_hpe.cosine_table(ang, cp[0]), ops.cosine_table(ang, cp[0]))


@pytest.mark.parametrize("hx", HX_MODES)
def test_split_f16_every_face_of_a_tile_overflows(hx, head_sds, device):
    """The slow path's worst case: EVERY face of a tile (and of a 4,096-face batch) leaves f16's range, e.g. un-normalised
    pixel-scale landmarks against trained-scale weights (`normalize=False` is a legal reference input, FeatureExtractor.py:30,52).
    All faces are re-evaluated in f32: finite, accurate, and faces of OTHER tiles keep their bits.
      f16x2s (the default): a tile with such faces goes, whole, through the f32 MATRIX cores in the re-evaluation launch
             behind the kernel (encoder_heads.hip, from the f32 image inside the strict blob): at most 4x the time of the same batch
             inside f16's range, and as close to the f64 truth as an f32 evaluation gets there (1.3x torch's own f32 forward, measured);
      f16x2  (opt-in): four faces at a time on the vector ALUs inside the launch, weights rebuilt from their hi + lo pieces (22 bits):
             ~40x and 1.7x torch's error -- its documented cliff (include/nlml_hpe.h)."""
    F = 1404
    sd = synth.encoder_state_dict(F, seed=0)
    blob = _blob_hx(sd, head_sds, device, hx)
    P = EH.Params(sd, head_sds)
    x = synth.features(192, F, seed=41)
    clean = ops.encoder_heads_fwd(torch.from_numpy(x).to(device), blob, F)
    bad = x.copy()
    bad[64:128] *= 3.4e4                      # the whole middle tile: features to 6.8e4, beyond f16's 65,504 in every face
    out = ops.encoder_heads_fwd(torch.from_numpy(bad).to(device), blob, F)
    assert torch.equal(out[:64], clean[:64]) and torch.equal(out[128:], clean[128:])
    truth = EH.forward_numpy(bad[64:128], P, np.float64)
    ref32 = np.asarray(EH.forward_torch(bad[64:128], P, num_threads=1))
    got = out[64:128].cpu().numpy()
    assert np.isfinite(got).all()
    e = np.degrees(np.abs(got - truth).max())
    e_ref = np.degrees(np.abs(ref32 - truth).max())
    # the batch of them, timed against the same batch inside f16's range
    big = np.tile(bad[64:128], (64, 1))      # 4,096 faces, every one on the slow path
    xb, xc = torch.from_numpy(big).to(device), torch.from_numpy(np.tile(x[64:128], (64, 1))).to(device)

    def ms(t):
        for _ in range(2):
            ops.encoder_heads_fwd(t, blob, F)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        o = ops.encoder_heads_fwd(t, blob, F)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b), o
    t_slow, ob = ms(xb)
    t_fast, _ = ms(xc)
    assert torch.equal(ob, out[64:128].repeat(64, 1))                      # batch position does not matter on the slow path either
    _report(f"split_f16_all_faces_rescued_{hx}", max_abs_deg=e, torch_f32_vs_f64_deg=e_ref, ms_4096_faces_all_slow=t_slow,
            ms_4096_faces_fast=t_fast, slowdown=t_slow / t_fast)
    assert (np.abs(bad[64:128]).max(axis=1) > 65504.0).all()
    # inputs of 7e4 put the network far outside its range (pre-activations of 1e5): torch's own f32 forward is 9e-4 deg from the
    # f64 truth on these faces
    if hx == "f16x2s":
        # measured 1.23e-3 deg against torch's 9.4e-4 (1.32x): two f32 evaluations of pre-activations of 1e5, each with its own
        # rounding, maximum over 64 faces -- the f32 kernel's accuracy class (f16x2's vector-ALU path on 22-bit weights: 1.6e-3 deg, 1.7x)
        assert e <= max(POSE_TOL_DEG, 1.5 * e_ref), (e, e_ref)
        assert t_slow <= 4.0 * t_fast, (t_slow, t_fast)
        # the same faces through the strict parity mode's own kernel: the re-evaluation launch IS that kernel
        blob32 = torch.from_numpy(weights.pack_blob(sd, head_sds)).to(device)
        assert torch.equal(out[64:128], ops.encoder_heads_fwd(torch.from_numpy(bad[64:128]).to(device), blob32, F))
        # ANY number of such faces in a tile goes to the re-evaluation launch (the strict kernels have no slow path of their own): the
        # flagged faces get the strict parity kernel's bits, the others keep theirs
        for nbad in (1, 5):
            mix = x.copy()
            mix[64:64 + nbad] *= 3.4e4
            om = ops.encoder_heads_fwd(torch.from_numpy(mix).to(device), blob, F)
            assert torch.equal(om[64 + nbad:], clean[64 + nbad:]) and torch.equal(om[:64], clean[:64]) and torch.isfinite(om).all()
            em = np.degrees(np.abs(om[64:64 + nbad].cpu().numpy() - EH.forward_numpy(mix[64:64 + nbad], P, np.float64)).max())
            assert em <= max(POSE_TOL_DEG, 1.5 * e_ref), (nbad, em)
            assert torch.equal(om[64:64 + nbad], ops.encoder_heads_fwd(torch.from_numpy(mix[64:64 + nbad]).to(device), blob32, F))
            assert torch.equal(ops.encoder_heads_fwd_small(torch.from_numpy(mix).to(device), blob, F), om)     # layered path: same rule, same bits
    else:   # the slow path on 22-bit weights is within 2x of torch's error; measured 38x in time (6.1 ms for 4,096 all-slow faces)
        assert e <= max(POSE_TOL_DEG, 2.5 * e_ref), (e, e_ref)
        assert t_slow <= 80.0 * t_fast


def test_config3_reference_order_matches_scipy_on_the_c_oracle(tucker_art, device):
    """BASELINE config 3 (4,096 noisy grid faces) in the REFERENCE order, the TD default: eight of the faces -- the one with
    the most evaluations, the one with the fewest and six more -- against scipy's own Powell on the C oracle's objective in the same
    operation order, on the CPU: identical evaluation counts, identical minimisers.  (The fast order would fail this: 10 % of
    these faces end more than 0.02 deg away, bench.py extra.td_powell_fast_order.)"""
    from scipy.optimize import minimize
    from nlml_hpe_amd import TD_Tester as TD
    from oracle import c_oracle as CO
    W = tucker_art["W"]
    Py, Pp, Pr = tucker_art["optimized_yaw"][:3], tucker_art["optimized_pitch"][:3], tucker_art["optimized_roll"][:3]
    cp = np.stack([Py, Pp, Pr])
    idx = synth.tucker_grid_indices(4096, seed=2)
    X = synth.tucker_grid_faces(tucker_art, idx, 1e-3, seed=2)
    deg, info = TD.Test_batch(W, X, 5, Py, Pp, Pr, return_info=True)            # default order = reference
    assert (info["status"] == 1).all()
    picks = sorted({int(info["nfev"].argmax()), int(info["nfev"].argmin()), 0, 1, 17, 1000, 2500, 4095})
    worst = 0.0
    for i in picks:
        res = minimize(lambda p: float(CO.tucker_objective(W, X[i:i + 1], p[None, :], cp, reference_order=True)[0]), np.zeros(8),
                       method="Powell")
        assert res.nfev == info["nfev"][i], (i, res.nfev, info["nfev"][i])
        worst = max(worst, float(np.abs(np.degrees(res.x[:3]) - deg[i]).max()))
    _report("config3_reference_order_vs_scipy", faces_checked=len(picks), max_abs_deg=worst, max_nfev=info["nfev"].max())
    assert worst <= TD_REF_TOL_DEG, worst


def test_device_f_vectors_match_numpy_over_a_sweep(tucker_art, device):
    """The reference-order objective is bit-identical to the reference UP TO the cos() inside the f-vectors
    (f = float32(a*cos(b*w+c)+d), TD_Tester.py:25-28,37).  The device evaluates a CORRECTLY ROUNDED cos (csrc/cr_cos.h); the
    reference's is the host libm's (numpy calls it for scalars; `math.cos` here, so that a numpy build with its own SIMD cos
    cannot move this test) -- accurate to an ulp, not always correctly rounded -- so the two can differ by one unit in the last
    place of cos, and where a*cos+d nearly cancels that could flip the f32 rounding of f.  Sweep 1e6 angles: (1) the bare cos (row
    a=1,b=1,c=0,d=0) and general arguments b*w + c are within 1 ulp of libm's everywhere; (2) with the shipped cosine rows the
    f32 roundings that differ are counted (round 2's form -- the library cos AND an fma-contracted b*w + c -- had 5 per 9e6;
    each such flip can move ONE evaluation of one face off scipy's trajectory)."""
    import math
    n = 1_000_000
    w = np.linspace(-1.6, 1.6, n).astype(np.float32)
    w64 = w.astype(np.float64)
    wt = torch.from_numpy(w).to(device)

    def libm_cos(arg):
        return np.fromiter((math.cos(v) for v in arg.ravel()), dtype=np.float64, count=arg.size).reshape(arg.shape)
    worst_ulp, frac_differ = 0.0, 0.0
    for b_, c_ in ((1.0, 0.0), (1.388305, -1.5476), (2.93, 0.0064), (0.486, 0.0011)):
        row = torch.tensor([[1.0, b_, c_, 0.0]], dtype=torch.float64, device=device)
        got_ = ops.cosine_table(wt, row).cpu().numpy()[:, 0]
        ref_ = libm_cos(np.float64(b_) * w64 + np.float64(c_))      # (a fused multiply-add here would move the argument by an ulp
        worst_ulp = max(worst_ulp, float((np.abs(got_ - ref_) / np.spacing(np.abs(ref_))).max()))   # and the cos by hundreds near its zeros)
        frac_differ = max(frac_differ, float((got_ != ref_).mean()))
    cp = _cos_params(tucker_art).reshape(9, 4)
    got = ops.cosine_table(wt, torch.from_numpy(cp).to(device)).cpu().numpy()
    ref = cp[None, :, 0] * libm_cos(cp[None, :, 1] * w64[:, None] + cp[None, :, 2]) + cp[None, :, 3]
    flips = int((got.astype(np.float32) != ref.astype(np.float32)).sum())
    _report("device_cos_vs_libm", values=9 * n, f32_roundings_that_differ=flips, cos_max_ulp=worst_ulp, cos_frac_not_equal=frac_differ)
    assert worst_ulp <= 1.0 and frac_differ <= 0.01
    assert flips <= 2


def test_reference_order_objective_on_200k_random_evaluations(tucker_art, device):
    """Bit-identity of the reference-order objective on a sample large enough to see a one-in-a-million event: 200,000 random
    evaluations (1.8 M f-vector entries) against the C oracle in the same order.  Before round 3 the f-vector arguments b*w + c
    were contracted into an fma on the device, which moved one f32 rounding in ~2e6 -- invisible to FX4's 32 evaluations."""
    from oracle import c_oracle as CO
    n = 200_000
    cp = _cos_params(tucker_art)
    P = synth.tucker_params(n, 5, seed=77)
    X = synth.features(64, 1404, seed=78)
    idx = (np.arange(n) % 64).astype(np.int32)
    Wm = torch.from_numpy(tucker_art["W"].reshape(135, 1404)).to(device)
    got = ops.tucker_objective(Wm, torch.from_numpy(X).to(device), torch.from_numpy(P).to(device), torch.from_numpy(cp).to(device),
                               x_index=torch.from_numpy(idx).to(device), order="reference").cpu().numpy()
    ref = CO.tucker_objective(tucker_art["W"], X[idx], P, cp, reference_order=True)
    bad = int((got != ref).sum())
    _report("reference_order_200k", evaluations=n, differing=bad)
    assert bad <= 1, bad      # (0 measured; 1 allowed for a cos value where libm is not correctly rounded AND the f32 rounding flips)


@pytest.mark.parametrize("mode", [None, "f32", "f16x2"])
def test_bench_line_carries_the_contract_fields(mode, repo_root, device):
    """`python bench.py` (short run): ONE JSON line with the contract's fields, a roofline block, the CPU baseline with both live parity
    checks -- the seed-0 sample of the timed batch and the operating-range statistics (FX3c) next to the reference's own.  mode None =
    no --mode flag: the default must be the strict-fast mode, and that run also measures roofline.traffic live (two rocprofv3 --pmc
    child passes on this box); the other modes run with --no-traffic and must say `traffic: null`, never a number from elsewhere."""
    import subprocess
    import sys
    flags = ["--no-traffic", "--mode", mode] if mode else []
    mode = mode or "f16x2s"
    res = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--steps", "3", "--warmup", "1", "--settle-ms", "0",
                          "--no-extra", "--cpu-seconds", "1"] + flags, capture_output=True, text=True, timeout=900, cwd=repo_root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["value"] > 1e6 and "workload" in rec["config"]
    rf = rec["roofline"]
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rec["dtype"] == mode
    assert rec["warmup"] == 1 and rec["warmup_effective"] == 1      # --settle-ms 0: --warmup is the only warm-up
    if flags:
        assert rf["traffic"] is None and "not measured" in rf["traffic_source"]
    else:   # measured on this box in this run: at least the algorithmic bytes (65,536 x 5,628 B), at most a few times that
        assert rf["traffic"] is not None, rf["traffic_source"]
        assert 0.9 * 65536 * 5628 <= rf["traffic"] <= 8 * 65536 * 5628 and "measured in this run" in rf["traffic_source"]
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["parity_check"]["max_abs_deg_vs_f64_oracle"] <= POSE_TOL_DEG
    rng = cb["parity_check_operating_range"]
    k_, r_ = rng["kernel_vs_f64_truth"], rng["reference_batched_vs_f64_truth"]
    assert rng["faces"] == 16384 and rng["mode"] == mode and rec["config"]["mode"] == mode
    assert "kernel_vs_reference_batch1" in rng and "kernel_vs_reference_batched" in rng
    if mode in ("f16x2s", "f32"):    # the strict modes: inside the reference's own distance from the truth
        assert k_["p50_deg"] <= 1.05 * r_["p50_deg"] and k_["max_deg"] <= POSE_TOL_DEG and k_["frac_above_1e-4_deg"] == 0.0
    else:
        assert k_["p50_deg"] <= 1.32 * r_["p50_deg"] and k_["max_deg"] <= 1.47e-4


def test_bench_two_rank_path_on_one_gpu(repo_root, device):
    """The N > 1 path of bench.py -- self-launched ranks, the untimed VERIFIED all-gather, the timed steps with and without the
    collective -- rehearsed with two ranks that share this GPU over gloo (NLML_BENCH_REHEARSAL=1: plumbing only, never a
    measurement; on a multi-GPU node the same code runs over RCCL)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["NLML_BENCH_REHEARSAL"] = "1"
    res = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--settle-ms", "0", "--batch", "2000", "--no-extra", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=repo_root, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and "REHEARSAL" in rec["data"]
    assert rec["comm"]["verified_all_gather"] is True and rec["value_no_collective"] > 0
    assert rec["config"]["collective"].startswith("all_gather")


# ---- round 5: the workspace-taking entry points, the 64-face-tile form of the f32 re-evaluation launch with flagged faces, the trunk + streamed-tail
# path, and the RCCL tests that run by themselves on the first multi-GPU box.  (The 128-face-tile path these tests were first written for measured
# slower than the fused kernel and was deleted at the round's end: DESIGN.md section 3, DESIGN_APPENDIX.md A.5.)
def _wide_call(fn_name, first, B, blob, device, *, want_latent=True, want_valid=True, extra=()):
    """One call of a workspace-taking C entry point (nlml_*_streamed / nlml_*_ws) -> (pose, latent, valid)."""
    from nlml_hpe_amd import _lib
    L = _lib.lib()
    F = 1404
    ws = torch.empty((max(16, L.nlml_encoder_heads_workspace_bytes(B, F)),), dtype=torch.uint8, device=device)
    out = torch.full((B, 3), float("nan"), dtype=torch.float32, device=device)
    lat = torch.empty((B, 9), dtype=torch.float32, device=device) if want_latent else None
    val = torch.empty((B,), dtype=torch.uint8, device=device) if want_valid else None
    st = torch.cuda.current_stream(device).cuda_stream
    args = [first.data_ptr(), *extra, blob.data_ptr(), blob.numel(), out.data_ptr(), lat.data_ptr() if want_latent else None,
            val.data_ptr() if want_valid else None, ws.data_ptr(), ws.numel(), st]
    _lib.check(getattr(L, fn_name)(*args), fn_name)
    return out, lat, val


def test_ws_entry_points_dispatch(head_sds, device):
    """nlml_*_ws (what a host may call for every batch size): same bits as the fused entry points at 64, 4,096, 4,097 and 9,000 faces in both
    split-f16 modes and in f32 (which ignores the workspace)."""
    from nlml_hpe_amd import _lib
    sd = synth.encoder_state_dict(1404, seed=0)
    raw = torch.from_numpy(synth.raw_landmarks(9000, seed=8)).to(device)
    for mode in ("f16x2s", "f16x2", "f32"):
        blob = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.mode_from_name(mode))).to(device)
        for B in (64, 4096, 4097, 9000):
            want = ops.landmarks_to_pose(raw[:B], blob, True)
            got, _, _ = _wide_call("nlml_landmarks_to_pose_ws", raw[:B], B, blob, device, extra=(B, 1))
            assert torch.equal(want, got), (mode, B)


def test_strict_reevaluation_launch_in_its_wide_form_with_flagged_faces(head_sds, device):
    """behind a large batch the f32 re-evaluation launch runs 64-face tiles (two column blocks per workgroup), a form the
    flagged-face tests at <= 4,096 faces never reach.  16,384 - 63 faces (a partial last tile): one flagged face in the low half of a
    tile, a few in the high half of another, a whole tile, and the last, partial, tile -- the flagged faces take the strict parity (f32)
    kernel's bits, pose and latent, every other face keeps the strict-fast kernel's (the trunk + streamed-tail path: its own test below)."""
    F, B = 1404, 16384 - 63
    sd = synth.encoder_state_dict(F, seed=0)
    blob = _blob_hx(sd, head_sds, device, "f16x2s")
    blob32 = torch.from_numpy(weights.pack_blob(sd, head_sds)).to(device)
    x = synth.features(B, F, seed=77)
    clean = torch.from_numpy(x).to(device)
    flagged = np.zeros(B, bool)
    flagged[64 * 10 + 5] = True                      # one face, low column block of tile 10
    flagged[64 * 99 + 40:64 * 99 + 44] = True        # four faces, high column block of tile 99
    flagged[64 * 200:64 * 201] = True                # the whole of tile 200
    flagged[B - 3:] = True                           # three faces of the last, partial, tile
    bad = x.copy()
    bad[flagged] *= 3.4e4                            # features to 6.8e4: beyond f16's range in layer 0
    xb = torch.from_numpy(bad).to(device)
    o_clean, l_clean = ops.encoder_heads_fwd(clean, blob, F, return_latent=True)
    o, l = ops.encoder_heads_fwd(xb, blob, F, return_latent=True)
    o32, l32 = ops.encoder_heads_fwd(xb, blob32, F, return_latent=True)
    fl = torch.from_numpy(flagged).to(device)
    assert torch.isfinite(o).all()
    assert torch.equal(o[~fl], o_clean[~fl]) and torch.equal(l[~fl], l_clean[~fl])          # neighbours, same tiles included, untouched
    assert torch.equal(o[fl], o32[fl]) and torch.equal(l[fl], l32[fl])                      # the strict parity kernel's bits
    o_again = ops.encoder_heads_fwd(xb, blob, F)
    assert torch.equal(o, o_again)                                                          # run to run


# ---- round 5, second half: the trunk launch + streamed tail launch (encoder_heads_f16x2_w8.hip TRUNK = true, encoder_heads_f16x2_tailws.hip)
@pytest.mark.parametrize("B", [1, 31, 64, 65, 257, 700, 4096 + 37, 16384, 65536 - 63])
def test_streamed_tail_path_is_bit_identical_to_the_fused_kernel(B, head_sds, device):
    """Layers 0-2 by the eight-wave kernel ending with layer 2's output in a workspace as MFMA operand fragments, then E3..E5 and the heads by
    the streamed-tail kernel (a wave keeps a 32-face block's activations in registers, the weights pass through LDS once per 256 faces):
    pose, latent and the "no face" mask bit for bit against the fused kernel -- raw landmarks (normalised in the launch and not), features,
    partial 32-face blocks, partial workgroups of the tail kernel (B not a multiple of 256), waves beyond the batch."""
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = _blob_hx(sd, head_sds, device, "f16x2s")
    raw_np = synth.raw_landmarks(B, seed=6)
    if B >= 65:
        raw_np[7] = 0.0                       # a "no face" row
        raw_np[B - 1] = raw_np[3]             # (and a duplicate in the last, partial, tile)
    raw = torch.from_numpy(raw_np).to(device)
    for normalize in (True, False):
        o, l, v = ops.landmarks_to_pose(raw, blob, normalize, return_latent=True, return_valid=True)
        o2, l2, v2 = _wide_call("nlml_landmarks_to_pose_streamed", raw, B, blob, device, extra=(B, int(normalize)))
        assert torch.equal(o, o2) and torch.equal(l, l2) and torch.equal(v.to(torch.uint8), v2), (B, normalize)
        o4, _, _ = _wide_call("nlml_landmarks_to_pose_streamed", raw, B, blob, device, extra=(B, int(normalize)), want_latent=False, want_valid=False)
        assert torch.equal(o, o4)             # (no latent, no mask asked for)
    feats = ops.normalize_ipd(raw, True)
    o, l, v = ops.encoder_heads_fwd(feats, blob, 1404, return_latent=True, return_valid=True)
    o3, l3, v3 = _wide_call("nlml_encoder_heads_fwd_streamed", feats, B, blob, device, extra=(feats.stride(0) if B > 1 else 1404, B, 1404))
    assert torch.equal(o, o3) and torch.equal(l, l3) and torch.equal(v.to(torch.uint8), v3)


def test_streamed_tail_path_with_the_shipped_heads_and_a_padded_row_stride(head_sds, device):
    """Features with a row stride larger than F (ldx = 1,408) through the trunk launch, and the layer-per-launch path as a third opinion."""
    sd = synth.encoder_state_dict(1404, seed=3)
    blob = _blob_hx(sd, head_sds, device, "f16x2s")
    B = 1000
    buf = torch.zeros((B, 1408), dtype=torch.float32, device=device)
    buf[:, :1404] = torch.from_numpy(synth.features(B, 1404, seed=12)).to(device)
    x = buf[:, :1404]
    want = ops.encoder_heads_fwd(x, blob, 1404, return_latent=True)
    o, l, _ = _wide_call("nlml_encoder_heads_fwd_streamed", x, B, blob, device, extra=(1408, B, 1404), want_valid=False)
    assert torch.equal(want[0], o) and torch.equal(want[1], l)
    small = ops.encoder_heads_fwd_small(x, blob, 1404, return_latent=True)
    assert torch.equal(small[0], o) and torch.equal(small[1], l)


def test_streamed_tail_path_refuses_what_it_does_not_take(head_sds, device):
    """Strict-fast blob only, 16-byte aligned rows with F % 4 == 0, a workspace of the documented size: anything else is NLML_E_BADARG."""
    from nlml_hpe_amd import _lib
    L = _lib.lib()
    sd = synth.encoder_state_dict(1404, seed=0)
    raw = torch.from_numpy(synth.raw_landmarks(256, seed=8)).to(device)
    st = torch.cuda.current_stream(device).cuda_stream
    blob_fast = torch.from_numpy(weights.pack_blob(sd, head_sds, _lib.MODE_F16X2)).to(device)
    with pytest.raises(_lib.NlmlError, match="F16X2S"):
        _wide_call("nlml_landmarks_to_pose_streamed", raw, 256, blob_fast, device, extra=(256, 1))
    blob = _blob_hx(sd, head_sds, device, "f16x2s")
    out = torch.empty((256, 3), dtype=torch.float32, device=device)
    ws = torch.empty((1024,), dtype=torch.uint8, device=device)
    rc = L.nlml_landmarks_to_pose_streamed(raw.data_ptr(), 256, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, ws.data_ptr(),
                                           ws.numel(), st)
    assert rc == -1 and b"workspace" in L.nlml_last_error()
    ws = torch.empty((L.nlml_encoder_heads_workspace_bytes(256, 1404),), dtype=torch.uint8, device=device)
    assert ws.numel() >= 4 * 65536            # 1 KB per face of whole 64-face tiles
    x = torch.from_numpy(synth.features(257, 1404, seed=2)).to(device)
    rc = L.nlml_encoder_heads_fwd_streamed(x.data_ptr() + 4, 1404, 256, 1400, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None,
                                           ws.data_ptr(), ws.numel(), st)
    assert rc < 0                             # (blob size does not match F = 1,400: refused before any launch)
    rc = L.nlml_encoder_heads_fwd_streamed(x.data_ptr() + 4, 1404, 256, 1404, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None,
                                           ws.data_ptr(), ws.numel(), st)
    assert rc == -1 and b"aligned" in L.nlml_last_error()
    assert L.nlml_landmarks_to_pose_streamed(raw.data_ptr(), 0, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, None, 0, st) == 0


def test_streamed_tail_path_flagged_faces_go_to_the_reevaluation_launch(head_sds, device):
    """A face beyond f16's range leaves the streamed tail with a non-finite pose, exactly as it leaves the fused kernel, and the f32
    re-evaluation launch behind it rewrites its tile: flagged faces = the strict parity kernel's bits, the others untouched, same answer as
    the fused path (16,384 - 63 faces: the re-evaluation launch in its 64-face-tile form)."""
    F, B = 1404, 16384 - 63
    sd = synth.encoder_state_dict(F, seed=0)
    blob = _blob_hx(sd, head_sds, device, "f16x2s")
    blob32 = torch.from_numpy(weights.pack_blob(sd, head_sds)).to(device)
    x = synth.features(B, F, seed=78)
    flagged = np.zeros(B, bool)
    flagged[64 * 3 + 33] = True
    flagged[64 * 120:64 * 121] = True
    flagged[B - 2:] = True
    bad = x.copy()
    bad[flagged] *= 3.4e4
    xb = torch.from_numpy(bad).to(device)
    o, l, _ = _wide_call("nlml_encoder_heads_fwd_streamed", xb, B, blob, device, extra=(F, B, F), want_valid=False)
    o_f, l_f = ops.encoder_heads_fwd(xb, blob, F, return_latent=True)
    o32, l32 = ops.encoder_heads_fwd(xb, blob32, F, return_latent=True)
    fl = torch.from_numpy(flagged).to(device)
    assert torch.isfinite(o).all()
    assert torch.equal(o, o_f) and torch.equal(l, l_f)
    assert torch.equal(o[fl], o32[fl]) and torch.equal(l[fl], l32[fl])


def test_ops_landmarks_to_pose_streamed_wrapper(head_sds, device):
    """The Python wrapper: cached hand-over buffer per (device, stream), a caller's own workspace, the three return forms, and the refusals
    (CPU tensor, wrong shape, a blob of another mode)."""
    from nlml_hpe_amd import _lib
    sd = synth.encoder_state_dict(1404, seed=0)
    blob = _blob_hx(sd, head_sds, device, "f16x2s")
    raw = torch.from_numpy(synth.raw_landmarks(3000, seed=21)).to(device)
    want = ops.landmarks_to_pose(raw, blob, True, return_latent=True, return_valid=True)
    got = ops.landmarks_to_pose_streamed(raw, blob, True, return_latent=True, return_valid=True)
    assert all(torch.equal(a, b) for a, b in zip(want, got))
    assert torch.equal(ops.landmarks_to_pose_streamed(raw[:100], blob), want[0][:100])       # the cached buffer serves a smaller batch
    ws = torch.empty((_lib.lib().nlml_encoder_heads_workspace_bytes(3000, 1404),), dtype=torch.uint8, device=device)
    assert torch.equal(ops.landmarks_to_pose_streamed(raw, blob, workspace=ws), want[0])
    with pytest.raises((_lib.NlmlError, ValueError, TypeError, RuntimeError)):
        ops.landmarks_to_pose_streamed(raw.cpu(), blob)
    with pytest.raises(ValueError):
        ops.landmarks_to_pose_streamed(raw[:, :400], blob)
    blob32 = torch.from_numpy(weights.pack_blob(sd, head_sds)).to(device)
    with pytest.raises(_lib.NlmlError, match="F16X2S"):
        ops.landmarks_to_pose_streamed(raw, blob32)


def test_ws_dispatcher_routes_large_batches_through_the_streamed_path_on_request(repo_root):
    """NLML_K2_STREAMED_MIN=<faces> (read once per process): nlml_landmarks_to_pose_ws takes the trunk + streamed-tail path from that size on --
    seen here as the launches it makes (the tail kernel's name in a rocprofv3-free way: the call must fail with a workspace that is
    large enough for the layer-per-launch path but too small for the hand-over) -- and computes the fused kernel's bits."""
    import subprocess
    import sys
    code = r"""
import os, sys
sys.path.insert(0, %r)
import torch
from nlml_hpe_amd import _lib, ops, synth, weights
dev = torch.device("cuda:0")
heads = weights.load_head_state_dicts(os.path.join(%r, "models"))
blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(1404, seed=0), heads, _lib.MODE_F16X2S)).to(dev)
L = _lib.lib()
B = 9000
raw = torch.from_numpy(synth.raw_landmarks(B, seed=4)).to(dev)
want = ops.landmarks_to_pose(raw, blob, True)
out = torch.empty((B, 3), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty((L.nlml_encoder_heads_workspace_bytes(B, 1404),), dtype=torch.uint8, device=dev)
rc = L.nlml_landmarks_to_pose_ws(raw.data_ptr(), B, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), st)
torch.cuda.synchronize()
small = torch.empty((1 << 20,), dtype=torch.uint8, device=dev)      # 1 MB: less than the hand-over's 1 KB per face
rc2 = L.nlml_landmarks_to_pose_ws(raw.data_ptr(), B, 1, blob.data_ptr(), blob.numel(), out.data_ptr(), None, None, small.data_ptr(), small.numel(), st)
print("RESULT", rc, bool(torch.equal(want, out)) if rc == 0 else None, rc2, L.nlml_last_error().decode()[:60] if rc2 else "")
""" % (repo_root, repo_root)
    def run(env_extra):
        env = {k: v for k, v in os.environ.items() if k != "NLML_K2_STREAMED_MIN"}
        env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0]
    routed = run({"NLML_K2_STREAMED_MIN": "8192"})
    assert routed.startswith("RESULT 0 True -1") and "workspace" in routed, routed      # streamed path: right bits, and it needs its hand-over buffer
    plain = run({})
    assert plain.startswith("RESULT 0 True 0"), plain                                    # fused kernel: ignores the workspace


needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL over xGMI (skips on the one-GPU box)")


@needs_two_gpus
def test_rccl_bench_two_ranks(repo_root):
    """The first multi-GPU box runs this by itself: `python bench.py --gpus 2` self-launches two ranks over the nccl backend (RCCL), verifies
    one untimed all-gather and times the steps with and without the collective (BASELINE config 4's shape, 2,000 faces per GPU)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NLML_BENCH_REHEARSAL")}
    res = subprocess.run([sys.executable, os.path.join(repo_root, "bench.py"), "--gpus", "2", "--batch", "2000", "--steps", "3", "--warmup", "1",
                          "--settle-ms", "0", "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=repo_root, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["data"] == "synthetic"
    assert "RCCL" in rec["comm"]["backend"] and rec["comm"]["verified_all_gather"] is True and rec["comm"]["visible_devices"] >= 2
    assert rec["value"] > 0 and rec["value_no_collective"] > 0


@needs_two_gpus
def test_rccl_video_entry_point_shards_streams(repo_root, tmp_path):
    """BASELINE config 5 on two GPUs over the nccl backend: each rank carries a contiguous block of the 64 streams on its own GPU, the
    collated output equals the one-process run bit for bit."""
    import subprocess
    import sys
    env = {k: v for k, v in dict(os.environ, PYTHONPATH=repo_root, MASTER_ADDR="127.0.0.1").items() if k != "NLML_BENCH_REHEARSAL"}
    one = subprocess.run([sys.executable, "generatePose_on_video.py", "--source", "synthetic", "--save_output", "True",
                          "--output_path", str(tmp_path / "one.npz")], cwd=repo_root, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    port = 29500 + (os.getpid() % 200)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), "generatePose_on_video.py", "--source", "synthetic", "--save_output", "True",
                          "--output_path", str(tmp_path / "two.npz")], cwd=repo_root, env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, (two.stdout[-2000:], two.stderr[-3000:])
    assert "[rank 1/2, streams 32..63]" in two.stdout
    a, b = np.load(tmp_path / "one.npz"), np.load(tmp_path / "two.npz")
    for k in ("smoothed_deg", "endpoints", "valid"):
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k


@needs_two_gpus
def test_rccl_pose_gatherer_collates_the_shards(repo_root, tmp_path):
    """nlml_hpe_amd.distributed over RCCL: two ranks, each its own GPU and its own shard of 2,000 faces through the default mode; the
    gathered [4000, 3] block equals the single-GPU run of the same faces bit for bit on both ranks."""
    import subprocess
    import sys
    script = tmp_path / "gather2.py"
    script.write_text(
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {repo_root!r})\n"
        "from nlml_hpe_amd import ops, synth, weights, _lib\n"
        "from nlml_hpe_amd.distributed import PoseGatherer, shard_bounds\n"
        "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "torch.cuda.set_device(rank); dev = torch.device('cuda', rank)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        f"heads = weights.load_head_state_dicts(os.path.join({repo_root!r}, 'models'))\n"
        "blob = torch.from_numpy(weights.pack_blob(synth.encoder_state_dict(1404, 0), heads, _lib.DEFAULT_MODE)).to(dev)\n"
        "raw = torch.from_numpy(synth.raw_landmarks(4000, seed=3)).to(dev)\n"
        "lo, hi, per = shard_bounds(4000, world, rank)\n"
        "g = PoseGatherer(hi - lo, world, dev)\n"
        "g.submit(ops.landmarks_to_pose_small(raw[lo:hi].contiguous(), blob, True))\n"
        "got = g.drain(); torch.cuda.synchronize()\n"
        "want = ops.landmarks_to_pose_small(raw, blob, True)\n"
        "assert torch.equal(got[:4000], want), 'gathered poses differ from the one-GPU run'\n"
        "dist.barrier(); dist.destroy_process_group(); print('RANK_OK', rank)\n")
    env = {k: v for k, v in dict(os.environ, MASTER_ADDR="127.0.0.1").items() if k != "NLML_BENCH_REHEARSAL"}
    port = 29300 + (os.getpid() % 200)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    assert "RANK_OK 0" in res.stdout and "RANK_OK 1" in res.stdout
