"""CPU: the C-ABI library loads and exports every symbol include/nlml_hpe.h declares; the host-side
packer, loaders and sharding logic (no GPU compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import blob_emulator as BE
from nlml_hpe_amd import _lib, synth, weights
from nlml_hpe_amd.distributed import shard_bounds
from oracle import encoder_heads as EH


def _declared_symbols(repo_root):
    txt = open(os.path.join(repo_root, "include", "nlml_hpe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nlml_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(repo_root):
    names = _declared_symbols(repo_root)
    assert len(names) >= 8
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/nlml_hpe.h but not exported"
    assert set(names) == set(_lib.SYMBOLS), "ctypes table and header disagree"
    assert _lib.lib().nlml_abi_version() == 2


@pytest.mark.parametrize("F", [1404, 136, 13])
def test_packed_blob_walk_matches_oracle(F, head_sds):
    """Decode the blob with the kernel's index arithmetic (tests/blob_emulator.py) and compare with the oracle."""
    sd = synth.encoder_state_dict(F, seed=0)
    blob = weights.pack_blob(sd, head_sds)
    assert blob.nbytes == _lib.lib().nlml_encoder_heads_packed_bytes(F, 0)
    x = synth.features(32, F, seed=5)
    out, lat = BE.forward(blob, x)
    P = EH.Params(sd, head_sds)
    assert np.abs(out - EH.forward_numpy(x, P, np.float64)).max() <= 1e-12
    assert np.abs(lat - EH.encoder_latent_numpy(x, P, np.float64)).max() <= 1e-12


@pytest.mark.parametrize("F", [1404, 136, 13])
def test_packed_blob_walk_split_f16_mode(F, head_sds):
    """NLML_MODE_F16X2: the blob's hi+lo f16 weight pieces, per-stage power-of-two scale and scaled bias, walked with
    the kernel's index arithmetic and its three-product formula, land within the parity bar of the f64 oracle."""
    sd = synth.encoder_state_dict(F, seed=0)
    blob = weights.pack_blob(sd, head_sds, _lib.MODE_F16X2)
    L = _lib.lib()
    sizes = {m: L.nlml_encoder_heads_packed_bytes(F, m) for m in (0, 1, 2)}
    assert blob.nbytes == sizes[2] and len(set(sizes.values())) == 3, "modes are recognised by blob size"
    hdr = BE._header(blob)
    inv = hdr["inv_scale"]
    assert np.all(np.log2(inv) == np.round(np.log2(inv))), "scales are powers of two"
    x = synth.features(32, F, seed=5)
    out, lat = BE.forward_f16x2(blob, x)
    P = EH.Params(sd, head_sds)
    ref = EH.forward_numpy(x, P, np.float64)
    err_deg = np.abs(np.degrees(out - ref)).max()
    assert err_deg <= 2e-5, err_deg                       # parity bar is 1e-4 deg; the split leaves ~1e-5
    assert np.abs(lat - EH.encoder_latent_numpy(x, P, np.float64)).max() <= 2e-6
    # NLML_MODE_F16X2S: the same image (the header's mode and total-size words apart), 256 bytes of zero padding, then a complete
    # NLML_MODE_F32 image: what the strict-fast mode's f32 re-evaluation launch reads (tiles with many faces beyond f16's range)
    blob_s = weights.pack_blob(sd, head_sds, _lib.MODE_F16X2S)
    blob_f32 = weights.pack_blob(sd, head_sds, _lib.MODE_F32)
    assert blob_s.nbytes == blob.nbytes + 256 + blob_f32.nbytes
    a, b = blob.view(np.uint8), blob_s.view(np.uint8)
    diff = np.flatnonzero(a != b[:a.size])
    assert set(diff // 4) == {3, 5} and a[12] == _lib.MODE_F16X2 and b[12] == _lib.MODE_F16X2S and not b[a.size:a.size + 256].any()
    assert np.array_equal(b[a.size + 256:], blob_f32.view(np.uint8))
    assert blob_s.view(np.uint32)[5] == blob.view(np.uint32)[5] + 16 + blob_f32.nbytes // 16


def test_pack_rejects_bad_input(head_sds):
    sd = synth.encoder_state_dict(136, seed=0)
    bad = dict(sd)
    bad["encoder.2.weight"] = bad["encoder.2.weight"][:, :100]
    with pytest.raises(ValueError):
        weights.pack_blob(bad, head_sds)
    L = _lib.lib()
    assert L.nlml_encoder_heads_packed_bytes(0, 0) == 0
    assert L.nlml_encoder_heads_packed_bytes(136, 7) == 0
    buf = np.zeros(16, np.uint8)
    rc = L.nlml_encoder_heads_pack(136, 0, None, None, None, None, buf.ctypes.data_as(ctypes.c_void_p), 16)
    assert rc != 0 and b"null" in L.nlml_last_error()


def test_launch_entry_points_validate_before_touching_the_gpu():
    L = _lib.lib()
    assert L.nlml_normalize_ipd(None, 4, 1, None, None, None) == -1
    assert L.nlml_encoder_heads_fwd(None, 1404, -1, 1404, None, 0, None, None, None, None) == -1
    assert L.nlml_tucker_objective(None, None, 1404, None, None, None, 5, None, None, None) == -1
    assert L.nlml_normalize_ipd(None, 0, 1, None, None, None) == 0          # empty batch is a no-op
    # the workspace-taking entry points (round 5): sizes on the host, argument checks before any launch
    assert L.nlml_encoder_heads_workspace_bytes(0, 1404) == 0 and L.nlml_encoder_heads_workspace_bytes(64, 0) == 0
    assert L.nlml_encoder_heads_workspace_bytes(65536, 1404) >= L.nlml_encoder_heads_small_workspace_bytes(65536, 1404)
    assert L.nlml_encoder_heads_workspace_bytes(65536, 1404) >= 65536 * 1024                                 # the streamed tail's hand-over: 1 KB per face
    assert L.nlml_landmarks_to_pose_ws(None, -1, 1, None, 0, None, None, None, None, 0, None) == -1
    assert L.nlml_landmarks_to_pose_streamed(None, 5, 1, None, 0, None, None, None, None, 0, None) == -1 and b"null" in L.nlml_last_error()
    assert L.nlml_tucker_objective(None, None, 1404, None, None, None, 0, None, None, None) == 0


def test_scripted_state_dict_split_roundtrip(head_sds):
    enc = {k: torch.from_numpy(v) for k, v in synth.encoder_state_dict(136, 0).items()}
    combined = {f"encoder.{k}": v for k, v in enc.items()}
    for n, sd in head_sds.items():
        combined.update({f"{n}_network.{k}": v for k, v in sd.items()})
    e2, h2 = weights.split_scripted_state_dict(combined)
    assert set(e2) == set(enc) and all(torch.equal(e2[k], enc[k]) for k in enc)
    assert all(torch.equal(h2[n][k], head_sds[n][k]) for n in head_sds for k in head_sds[n])
    assert weights.validate_shapes(e2, h2) == 136


def test_artefact_layout(tucker_art, repo_root):
    assert tucker_art["W"].shape == (5, 3, 3, 3, 1404) and tucker_art["W"].dtype == np.float32
    assert tucker_art["optimized_yaw"].shape == (3, 4) and tucker_art["U_id"].shape == (1620, 5)
    with pytest.raises(FileNotFoundError):
        weights.load_encoder_state_dict(os.path.join(repo_root, "models"))     # not shipped by the reference


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 16000, 65536, 65537):
        for world in (1, 2, 3, 8):
            rows = [shard_bounds(n, world, r) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == n
            assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
            assert all(b - a <= per for a, b, per in rows)


def test_no_cpu_fallback():
    from nlml_hpe_amd import ops
    with pytest.raises(_lib.NlmlError):
        ops.normalize_ipd(torch.zeros(2, 468, 3))
    with pytest.raises(_lib.NlmlError):
        ops.encoder_heads_fwd(torch.zeros(2, 136), torch.zeros(16, dtype=torch.uint8), 136)


def test_synth_generator_is_reproducible():
    a, b = synth.features(64, 136, seed=3), synth.features(64, 136, seed=3)
    assert np.array_equal(a, b) and not np.array_equal(a, synth.features(64, 136, seed=4))
    assert (a[:, 3:6] == 0).all()
    sd = synth.encoder_state_dict(136, 0)
    assert sd["encoder.0.weight"].shape == (1024, 136) and abs(sd["encoder.0.weight"]).max() <= 1 / np.sqrt(136)


def test_mode_names_and_default_are_the_parity_modes():
    """Host layer: mode constants mirror include/nlml_hpe.h; names map to them; the model default is a PARITY mode."""
    import inspect
    import re
    from nlml_hpe_amd import model as M
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "nlml_hpe.h")).read()
    consts = {k: int(v) for k, v in re.findall(r"#define\s+NLML_MODE_(\w+)\s+(\d+)", hdr)}
    assert consts == {"F32": _lib.MODE_F32, "BF16": _lib.MODE_BF16, "F16X2": _lib.MODE_F16X2, "F16X2S": _lib.MODE_F16X2S}
    assert _lib.mode_from_name("f16x2s") == _lib.MODE_F16X2S == 3
    assert _lib.mode_from_name("f16x2") == _lib.MODE_F16X2 and _lib.mode_from_name("f32") == 0 and _lib.mode_from_name(1) == 1
    with pytest.raises(ValueError):
        _lib.mode_from_name("fp8")
    with pytest.raises(ValueError):
        _lib.mode_from_name(9)
    # the forward entry points read a blob's mode off its size: the four sizes must differ, at every F the tests use
    L = _lib.lib()
    for F in (10, 13, 64, 136, 1404, 1407):
        sizes = [L.nlml_encoder_heads_packed_bytes(F, m) for m in (0, 1, 2, 3)]
        assert len(set(sizes)) == 4 and all(sizes), (F, sizes)
        assert sizes[3] == sizes[2] + 256 + sizes[0]      # strict-fast: the split-f16 image + 256 bytes + an f32 image
    default = inspect.signature(M.HIPPoseModel.__init__).parameters["mode"].default
    assert default == _lib.DEFAULT_MODE and default in (_lib.MODE_F16X2S, _lib.MODE_F32), \
        "the default must be a STRICT parity mode (inside the reference's own error): never f16x2 (1.10x), never bf16"
    assert inspect.signature(M.load_model).parameters["mode"].default == _lib.DEFAULT_MODE
    assert _lib.mode_from_name(_lib.DEFAULT_MODE_NAME) == _lib.DEFAULT_MODE
    with pytest.raises(_lib.NlmlError):
        M.HIPPoseModel(synth.encoder_state_dict(136, 0), weights.load_head_state_dicts("models"), device="cpu")


def test_torch_ops_come_from_the_compiled_library():
    """torch.ops.nlml_hpe.* are registered by libnlml_torch_ops.so (csrc/torch_ops.cpp), not by Python closures: the library exists
    next to the C ABI one, importing the package loads it, every op resolves with its schema, a CPU tensor finds no kernel (there is
    no CPU path behind the ops) and the Meta kernels give shapes without a GPU."""
    import torch
    from nlml_hpe_amd import ops
    assert os.path.exists(ops.TORCH_OPS_PATH)
    schema = str(torch.ops.nlml_hpe.landmarks_to_pose_small.default._schema)
    assert "Tensor workspace" in schema and "bool normalize" in schema
    assert "Tensor(a!) state" in str(torch.ops.nlml_hpe.video_post.default._schema) and "Tensor(e!) updated" in str(torch.ops.nlml_hpe.video_post.default._schema)
    with pytest.raises(NotImplementedError):
        torch.ops.nlml_hpe.landmarks_to_pose(torch.zeros(2, 468, 3), torch.zeros(16, dtype=torch.uint8), True)
    m = torch.ops.nlml_hpe.landmarks_to_pose(torch.zeros(7, 468, 3, device="meta"), torch.zeros(16, dtype=torch.uint8, device="meta"), True)
    assert tuple(m.shape) == (7, 3)
    # every op has a Meta kernel (the video tick's ops had none, so fake-tensor tracing of a tick raised NotImplemented)
    meta = lambda *shape, dtype=torch.float32: torch.empty(*shape, dtype=dtype, device="meta")
    pose, valid = torch.ops.nlml_hpe.landmarks_to_pose_valid(meta(5, 468, 3), meta(16, dtype=torch.uint8), True, None)
    assert tuple(pose.shape) == (5, 3) and tuple(valid.shape) == (5,) and valid.dtype == torch.uint8
    res = torch.ops.nlml_hpe.tucker_powell(meta(135, 1404), meta(9, 1404), meta(3, 3, 4, dtype=torch.float64), "reference")
    assert [tuple(t.shape) for t in res] == [(9, 8), (9,), (9,), (9,), (9,)] and res[2].dtype == torch.int32
    assert tuple(torch.ops.nlml_hpe.cosine_table(meta(7), meta(3, 4, dtype=torch.float64)).shape) == (7, 3)
    S = 4
    torch.ops.nlml_hpe.video_post(meta(S, 3), meta(S, 468, 3), None, 1920.0, 1080.0, 0.4, 100.0, 80.0, meta(S, 6, dtype=torch.float64),
                                  meta(S, 3, dtype=torch.float64), meta(S, 2, dtype=torch.float64), meta(S, 3, 2, dtype=torch.float64),
                                  meta(S, dtype=torch.uint8))


def test_mode_and_order_names():
    assert _lib.mode_from_name("f16x2") == _lib.MODE_F16X2 and _lib.mode_from_name(0) == _lib.MODE_F32
    assert _lib.td_order_from_name("reference") == _lib.TD_ORDER_REFERENCE and _lib.td_order_from_name(0) == _lib.TD_ORDER_FAST
    for bad in ("fp8", 7):
        with pytest.raises(ValueError):
            _lib.mode_from_name(bad)
        with pytest.raises(ValueError):
            _lib.td_order_from_name(bad)


def test_cr_cos_is_correctly_rounded(tmp_path, repo_root):
    """The cos behind the Tucker f-vectors (csrc/cr_cos.h, the same header on host and device): double-double evaluation rounded
    once.  Against numpy's libm cos it may differ by one unit in the last place -- and wherever it does, 60-digit decimal
    arithmetic says it is the one that is correctly rounded; a random sample is within half an ulp of the exact value."""
    import ctypes as C
    import subprocess
    from decimal import Decimal, getcontext
    so = tmp_path / "crcos.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so),
                    os.path.join(repo_root, "tests", "native", "cr_cos_host.cpp")], check=True, capture_output=True, text=True)
    lib = C.CDLL(str(so))
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-10, 10, 400_000), rng.uniform(-1e5, 1e5, 40_000),
                        np.array([0.0, 1e-300, -1e-9, np.pi / 2, np.pi, 1.5 * np.pi, -np.pi / 2, 7.0, np.inf, np.nan])])
    y = np.empty_like(x)
    lib.cr_cos_array(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_long(len(x)))
    import math
    ref = np.fromiter((math.cos(v) for v in x[:-2]), dtype=np.float64, count=len(x) - 2)   # the host libm itself, whatever numpy's array loop uses
    assert np.isnan(y[-2:]).all()
    y = y[:-2]
    assert (np.abs(y - ref) <= np.spacing(np.abs(ref))).all()                       # never more than one ulp from libm
    differ = np.nonzero(y != ref)[0]
    assert len(differ) <= 5e-3 * len(ref)
    getcontext().prec = 70
    pi = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899862803482534211706798")

    def exact_cos(xf):
        X = Decimal(xf)
        k = (X / (pi / 2)).to_integral_value()
        r = X - k * (pi / 2)

        def series(start):
            t = Decimal(1) if start == 0 else r
            s, n = t, start
            while abs(t) > Decimal(10) ** -65:
                n += 2
                t = -t * r * r / (n * (n - 1))
                s += t
            return s
        return [series(0), -series(1), -series(0), series(1)][int(k) % 4]

    for i in differ[:100]:                                                         # where the two disagree, cr_cos is the closer one
        t = exact_cos(float(x[i]))
        assert abs(Decimal(float(y[i])) - t) <= abs(Decimal(float(ref[i])) - t), x[i]
    for i in rng.integers(0, len(y), 300):                                         # and it is within half an ulp of the exact value
        t = exact_cos(float(x[i]))
        assert abs(Decimal(float(y[i])) - t) <= Decimal(float(np.spacing(abs(y[i])))) / 2, x[i]


def test_f_vector_entry_fast_form_equals_the_correctly_rounded_one(tmp_path, repo_root):
    """cr_f32_a_cos_d (csrc/cr_cos.h) returns float32(a * cos(t) + d) from the library cos wherever the float cannot depend on the
    cos's last bits and from the double-double cos otherwise: the same float as the slow path alone on 3 M values -- random rows,
    the shipped rows' ranges, and rows built to cancel (d = -a * cos(t0) with t next to t0), where the decision is closest."""
    import ctypes as C
    import subprocess
    so = tmp_path / "crcos.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so),
                    os.path.join(repo_root, "tests", "native", "cr_cos_host.cpp")], check=True, capture_output=True, text=True)
    lib = C.CDLL(str(so))
    rng = np.random.default_rng(5)
    n = 1_000_000
    a = np.concatenate([rng.uniform(-12, 12, n), rng.uniform(-12, 12, n), rng.uniform(-12, 12, n)])
    t = np.concatenate([rng.uniform(-10, 10, n), rng.uniform(-3.5, 3.5, n), rng.uniform(-3.5, 3.5, n)])
    d = np.concatenate([rng.uniform(-12, 12, n), rng.uniform(-12, 12, n), np.zeros(n)])
    t0 = t[2 * n:] + rng.uniform(-1e-7, 1e-7, n) * rng.integers(0, 2, n)       # half of them exactly at t: a * cos + d ~ 0
    d[2 * n:] = -a[2 * n:] * np.cos(t0)
    a[:4], t[:4], d[:4] = [1.0, 0.0, np.nan, 3.0], [np.nan, 1.0, 1.0, np.inf], [0.0, 0.0, 0.0, 1.0]
    fast, slow = np.empty(len(a), np.float32), np.empty(len(a), np.float32)
    P = lambda v: v.ctypes.data_as(C.c_void_p)
    lib.cr_fvalue_arrays(P(a), P(t), P(d), P(fast), P(slow), C.c_long(len(a)))
    assert np.array_equal(fast.view(np.uint32)[4:], slow.view(np.uint32)[4:])
    assert np.isnan(fast[0]) and np.isnan(fast[2]) and np.isnan(fast[3]) and np.isnan(slow[0]) and fast[1] == slow[1] == 0.0


def test_td_fast_order_checks_alignment_and_default_is_the_reference_order():
    """the matrix-core order reads Wm and x with 16-byte vector loads, so the ABI refuses a misaligned base or an
    ldx that is not a multiple of 4 in THAT order (the reference order, the default, reads dwords and accepts both).  The checks
    run before any launch, so fake device addresses do on the CPU."""
    L = _lib.lib()
    a16, a4 = ctypes.c_void_p(0x10000), ctypes.c_void_p(0x10004)
    ok = (a16, a16, 1404, None, a16, a16, 1, a16, None)
    assert L.nlml_tucker_objective_ex(a16, a4, 1404, None, a16, a16, 1, a16, None, _lib.TD_ORDER_FAST, None) == -1
    assert b"16-byte" in L.nlml_last_error()
    assert L.nlml_tucker_objective_ex(a4, a16, 1404, None, a16, a16, 1, a16, None, _lib.TD_ORDER_FAST, None) == -1
    assert L.nlml_tucker_objective_ex(a16, a16, 1405, None, a16, a16, 1, a16, None, _lib.TD_ORDER_FAST, None) == -1
    assert L.nlml_tucker_powell_ex(a16, a4, 1404, a16, 1, None, a16, None, None, None, None, _lib.TD_ORDER_FAST, None) == -1
    assert L.nlml_tucker_objective_ex(a16, a16, 1404, None, a16, a16, 1, a16, None, 7, None) == -1          # unknown order
    assert L.nlml_tucker_objective_ex(a16, a16, 1404, None, a16, a16, 0, a16, None, _lib.TD_ORDER_FAST, None) == 0   # N = 0: no launch
    del ok
