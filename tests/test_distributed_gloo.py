"""CPU, world_size 2, gloo: the sharding + all-gather layer the multi-GPU bench uses (RCCL on the GPU box)."""
import os
import subprocess
import sys

WORKER = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["NLML_REPO"])
from nlml_hpe_amd.distributed import PoseGatherer, gather_poses, shard_bounds
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
for n in (11, 16, 1):
    full = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3)
    s, e, per = shard_bounds(n, w, r)
    got = gather_poses(full[s:e], n)
    assert torch.equal(got, full), (n, got)
g = PoseGatherer(4, w, "cpu")
for i in range(5):                                     # more submissions than ring slots
    g.submit(torch.full((4, 3), float(r * 10 + i)))
last = g.drain()
assert last.shape == (4 * w, 3)
assert all(float(last[4 * k, 0]) == 10 * k + 4 for k in range(w)), last
try:
    g.submit(torch.zeros(3, 3))
    raise SystemExit("shape check missing")
except ValueError:
    pass
dist.barrier()
dist.destroy_process_group()
print("rank", r, "ok")
'''


def test_shard_and_gather_world2(tmp_path, repo_root):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, NLML_REPO=repo_root, MASTER_ADDR="127.0.0.1")
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.count("ok") == 2
