"""The fixture recipe is reproducible: tests/golden/make_golden.py, run against /root/reference, regenerates the committed
fixtures bit for bit.  Skipped where the reference is absent (the GPU box): the fixtures themselves travel, the reference does not."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

# FX5 (four scipy Powell runs on the reference objective, ~1 min) is left out to keep the CPU suite short
WHICH = ["1", "2", "2b", "3", "3b", "3c", "4", "6", "7", "8", "9"]
FILES = ["fx1_normalise.npz", "fx2_heads.npz", "fx2b_heads_through_model.npz", "fx3_encoder_heads.npz",
         "fx3b_reference_range.npz", "fx3c_reference_range_16k.npz", "fx4_td_objective.npz", "fx6_metrics.json", "fx7_video_in.npz", "fx7_video_math.json",
         "fx8_cosine_table.npz", "fx9_td_gradient.npz"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout")
def test_fixtures_regenerate_bit_identically(tmp_path, repo_root):
    env = dict(os.environ, NLML_GOLDEN_OUT=str(tmp_path), MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1")
    # run from the repo root ON PURPOSE: it holds files named like the reference's modules, which must not shadow them
    res = subprocess.run([sys.executable, os.path.join(GOLD, "make_golden.py")] + WHICH, cwd=repo_root, env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    for f in FILES:
        new, old = os.path.join(str(tmp_path), f), os.path.join(GOLD, f)
        assert os.path.exists(new), f"{f} was not regenerated"
        if f.endswith(".json"):
            assert json.load(open(new)) == json.load(open(old)), f
            continue
        a, b = np.load(new), np.load(old)
        assert set(a.files) == set(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (f, k)
            assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), (f, k)
