#!/usr/bin/env python3
"""bench.py -- faces/sec of the NLML_HPE batched-inference hot path on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W
    N > 1 works both ways: under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
    (RANK/LOCAL_RANK/WORLD_SIZE from the env), or as a plain `python bench.py --gpus N ...`, which starts the N ranks itself
    as child processes BEFORE anything touches the GPU and relays rank 0's JSON line.
    BASELINE config 4 (2,000 faces per GPU, looped, with and without the collective):  python bench.py --gpus 8 --batch 2000

One STEP = one pass of the hot path over one batch that is already resident in HBM:
raw landmarks f32[B,468,3] -> IPD normalisation -> encoder -> 3 heads -> (yaw,pitch,roll) f32[B,3],
one fused HIP launch per rank (nlml_landmarks_to_pose), B = 65,536 faces per GPU (weak scaling),
F = 1404 (the reference's real feature width).  --mode picks the kernel: f16x2s (DEFAULT: split-f16 operands on the
f16 matrix cores with split accumulators, f32 accumulate -- the strict-fast mode: at the reference's operating range no farther from
the exact result than the reference's own f32 forward), f32 (f32 matrix cores with blocked sums: the strict parity mode) or f16x2
(opt-in: single accumulators where registers are short, 1.10x the reference's error at +-45 deg poses).  With N > 1 every
step also all-gathers the [B,3] poses of all ranks over RCCL (the only collective the path has),
on the communication stream, overlapped with the next step's compute; a second timed region of K steps without
the collective gives `value_no_collective` (--no-collective: time only that one).

Prints ONE JSON line (rank 0): the contract fields plus
  roofline      the fused kernel against the dense MFMA peak of its operand type,
                achieved = algorithmic FLOP per launch / average launch time from HIP events;
  cpu_baseline  the oracle's torch-CPU restatement of the same arithmetic on this host's cores, plus
                parity_check: the measured kernel's poses against the f64 oracle on a sample of the batch;
  extra         the other K2 modes and secondary workloads (features-in K2, F=136, K1, K3, video, TD Powell).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_FACE = {1404: 4_714_240, 136: 2_117_376}     # SURVEY.md section 8 table (2 x MACs)
BYTES_PER_FACE_K2 = {1404: 5_628, 136: 556}            # f32 features in + 3 x f32 out
BYTES_PER_FACE_K1 = 11_232                             # 5616 read + 5616 written
PEAK_F32_MFMA_TFLOPS = 157.3                           # MI355X_MICROARCH.md, dense f32 matrix
PEAK_F16_MFMA_TFLOPS = 2500.0                          # MI355X_MICROARCH.md, dense bf16/f16 MFMA
SPLIT_PRODUCTS = 3                                     # f16x2 mode: hi*hi + hi*lo + lo*hi per algorithmic product
PEAK_HBM_GBS = 8000.0
PEAK_F64_TFLOPS = 78.6
TUCKER_FLOP_PER_EVAL = 383_700                          # algorithmic: 2*135*1404 + residual
TUCKER_REF_OPS_PER_EVAL = 135 * 1404 * 5                # the reference's order: 5 separately rounded f64 operations per (q, m)
PEAK_F64_VALU_TOPS = 39.3                               # f64 vector issue rate, non-fma: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults long enough (~0.25 s of GPU time) to reach the sustained rate: the split-f16 kernel runs at the board's
    # power limit and its first ~50 launches after idle are up to 15 % slower
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=65536, help="faces per GPU per step")
    ap.add_argument("--path", choices=["fused", "features"], default="fused",
                    help="fused: raw landmarks in (K1+K2 in one launch); features: normalised features in (K2)")
    ap.add_argument("--mode", choices=["f16x2s", "f32", "f16x2"], default="f16x2s",
                    help="kernel: f16x2s = split-f16 operands on the f16 matrix cores, split accumulators (strict-fast, the default); "
                         "f32 = f32 matrix cores, blocked sums (strict parity); f16x2 = opt-in, 1.10x the reference's error")
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="that many untimed steps (~1 ms each) BEFORE the W warm-up steps, so that short K/W also measure "
                         "the sustained (power-limited) rate; reported in the JSON as config.settle_ms")
    ap.add_argument("--no-collective", action="store_true",
                    help="N > 1: no all-gather in the timed steps (default: time with it and report value_no_collective beside)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic on this box (traffic: null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the child of measure_traffic(): a few launches, no output
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def time_kernel(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs]))  # ms


def time_stream(fn, iters, warm=3):
    """ms per call of `iters` calls issued back to back (one event pair around all of them)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def spawn_ranks(n: int, script: str = None, argv: list = None) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of THIS process, which has not
    touched the GPU (importing torch does not initialise HIP; nothing above calls torch.cuda), wait for them, relay rank 0's
    stdout (the JSON line) and return non-zero if any rank failed.  Children are started, never exec'ed into."""
    import socket
    import subprocess
    if os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        # a profiler's preloaded library has initialised the GPU in THIS process already: starting the ranks from here is the
        # exec-after-HIP-init hop this pool forbids.  Multi-GPU profiling needs a launcher that starts before any GPU call.
        print("bench.py: refusing to self-launch ranks under a profiler preload (ROCP_TOOL_LIBRARIES / LD_PRELOAD); "
              "use `python -m torch.distributed.run ... bench.py --gpus N` and profile the ranks themselves", file=sys.stderr)
        return 2
    script = os.path.abspath(__file__) if script is None else script
    argv = sys.argv[1:] if argv is None else argv
    deadline = time.time() + float(os.environ.get("NLML_BENCH_DEADLINE_S", "1500"))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # rank 0 inherits stdout (its JSON line is the result); the other ranks print nothing there, keep them on stderr
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in pending:              # exactly the PIDs started above
                    procs[q].terminate()
        if pending and time.time() > deadline:     # a rank stuck in a collective or a GPU wait must not hang the launcher
            print(f"bench.py: ranks {sorted(pending)} still running at the deadline; terminating them", file=sys.stderr, flush=True)
            for q in pending:
                procs[q].terminate()
            t_kill = time.time() + 10
            while any(procs[q].poll() is None for q in pending) and time.time() < t_kill:
                time.sleep(0.1)
            for q in pending:
                if procs[q].poll() is None:
                    procs[q].kill()
            return rc or 124
        time.sleep(0.05)
    return rc


# the kernels of one K2 step (every kernel of the library's nlml:: namespace that a forward launches)
K2_KERNEL_MARKS = ("encoder_heads_", "prepass_kernel", "layer_kernel", "tail_kernel", "tail_encoder_kernel", "head_kernel", "tail_ws_kernel")


def expected_k2_launches(mode: str, layered: bool) -> int:
    """Launches per forward by (mode, path): strict fused = the eight-wave kernel + the f32 re-evaluation launch; strict layered = pre-pass,
    three layers, tail_encoder_kernel, head_kernel, re-evaluation; opt-in fast mode 1 / 5; f32 and bf16 one fused launch."""
    if mode == "f16x2s":
        return 7 if layered else 2
    if mode == "f16x2":
        return 5 if layered else 1
    return 1
PMC_CHILD_WARM, PMC_CHILD_STEPS = 3, 10


def under_profiler() -> bool:
    return bool(os.environ.get("ROCP_TOOL_LIBRARIES")) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def measure_traffic(args):
    """roofline.traffic, measured ON THIS BOX IN THIS RUN: HBM-side bytes per step of the timed K2 kernel(s) from the TCC counters,
    collected and corrected as MI355X_MICROARCH.md's HBM section prescribes -- FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc
    passes (nothing else traced), FETCH_SIZE doubled on gfx950 (it tallies the 128-byte requests of a wide coalesced stream at 64 bytes),
    both in KiB: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Each pass is a CHILD process (`rocprofv3 --pmc C -- python3 bench.py
    --pmc-child ...`: the same batch, blob and launch, 3 + 10 steps, nothing printed) started before this process has touched the
    GPU.  Returns (bytes_per_step or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if under_profiler():
        return None, "not measured: this process runs under a profiler preload (a child start from here would be an exec after GPU init)"
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return None, "not measured: rocprofv3 not found"
    per_step = {}
    t0 = time.time()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix=f"nlml_pmc_{counter}_", dir="/tmp")
        cmd = [tool, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
               "--pmc-child", "--mode", args.mode, "--path", args.path, "--batch", str(args.batch)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, timeout=240)
        except subprocess.TimeoutExpired:
            shutil.rmtree(out, ignore_errors=True)
            return None, f"not measured: the {counter} pass did not finish in 240 s"
        vals = []
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == counter and any(m in row.get("Kernel_Name", "") for m in K2_KERNEL_MARKS):
                    vals.append((int(row.get("Dispatch_Id", 0)), float(row["Counter_Value"])))
        shutil.rmtree(out, ignore_errors=True)
        if r.returncode != 0 or not vals:
            tail = (r.stderr or b"")[-300:].decode(errors="replace").replace("\n", " | ")
            return None, f"not measured: the {counter} pass returned {r.returncode} with {len(vals)} kernel rows ({tail})"
        vals.sort()
        n_total = PMC_CHILD_WARM + PMC_CHILD_STEPS
        if len(vals) % n_total:
            return None, f"not measured: {len(vals)} kernel rows in the {counter} pass do not divide into {n_total} steps"
        per = len(vals) // n_total                          # kernels per step
        want = expected_k2_launches(args.mode, 0 < args.batch <= 4096 and args.mode in ("f16x2", "f16x2s"))
        if per != want:
            return None, f"not measured: {per} kernel rows per step in the {counter} pass, {want} launches expected for mode {args.mode} (a kernel name is missing from K2_KERNEL_MARKS?)"
        timed = vals[PMC_CHILD_WARM * per:]
        per_step[counter] = sum(v for _, v in timed) / PMC_CHILD_STEPS
    b = (2.0 * per_step["FETCH_SIZE"] + per_step["WRITE_SIZE"]) * 1024.0
    return b, (f"measured in this run on this box: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate child passes of the same "
               f"launch ({PMC_CHILD_STEPS} steps after {PMC_CHILD_WARM}), (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md; "
               f"FETCH_SIZE {per_step['FETCH_SIZE']:.0f} KiB, WRITE_SIZE {per_step['WRITE_SIZE']:.0f} KiB per step; {time.time() - t0:.0f} s")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    traffic, traffic_source = None, "not measured (N > 1, --no-traffic or --pmc-child)"
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_traffic and not args.pmc_child:
        traffic, traffic_source = measure_traffic(args)     # child processes, before anything here touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # NLML_BENCH_REHEARSAL=1: exercise the N>1 code path on a box with fewer GPUs than ranks (gloo instead of
    # RCCL, ranks share devices).  For checking the plumbing only -- never a measurement.
    rehearsal = os.environ.get("NLML_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from nlml_hpe_amd import ops, synth, weights
    from nlml_hpe_amd.distributed import PoseGatherer

    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist_mod.init_process_group(backend="gloo")
        else:
            dist_mod.init_process_group(backend="nccl", device_id=dev)
        dist = dist_mod

    F, B = 1404, args.batch
    heads = weights.load_head_state_dicts(os.path.join(ROOT, "models"))
    sd = synth.encoder_state_dict(F, seed=0)
    from nlml_hpe_amd import _lib
    from nlml_hpe_amd.model import HIPPoseModel
    mode = _lib.mode_from_name(args.mode)
    blob = torch.from_numpy(weights.pack_blob(sd, heads, mode)).to(dev)
    raw_np = synth.raw_landmarks(B, seed=1 + rank)           # each rank owns its own shard of faces
    raw = torch.from_numpy(raw_np).to(dev)
    feats = ops.normalize_ipd(raw, True)

    # up to SMALL_BATCH_MAX faces the split-f16 mode runs layer per launch (what HIPPoseModel does; same bits as the fused kernel)
    layered = 0 < B <= HIPPoseModel.small_batch_max(args.mode)
    if args.path == "fused":
        fwd = ops.landmarks_to_pose_small if layered else ops.landmarks_to_pose
        step_fn = lambda: fwd(raw, blob, True)
    else:
        fwd = ops.encoder_heads_fwd_small if layered else ops.encoder_heads_fwd
        step_fn = lambda: fwd(feats, blob, F)

    if args.pmc_child:        # measure_traffic()'s child: the timed launch a few times under rocprofv3 --pmc, nothing else
        for _ in range(PMC_CHILD_WARM + PMC_CHILD_STEPS):
            step_fn()
        torch.cuda.synchronize()
        return

    gatherer = PoseGatherer(B, world, dev) if world > 1 else None
    comm_info = None
    if world > 1:
        # one UNTIMED all-gather whose result every rank verifies (rank r's block carries r): a broken RCCL / xGMI path fails
        # here, loudly and with rc != 0, instead of producing a number
        probe = PoseGatherer(B, world, dev)
        probe.submit(torch.full((B, 3), float(rank), dtype=torch.float32, device=dev))
        got = probe.drain()
        torch.cuda.synchronize()
        want = torch.arange(world, dtype=torch.float32, device=dev).repeat_interleave(B)
        if not (torch.equal(got[:, 0], want) and torch.equal(got[:, 2], want)):
            raise SystemExit(f"bench.py: rank {rank}: the all-gather returned wrong data (collective path broken)")
        ok = torch.ones(1, device=dev)
        dist.all_reduce(ok)
        if int(ok.item()) != world:
            raise SystemExit(f"bench.py: rank {rank}: only {int(ok.item())} of {world} ranks verified the all-gather")
        try:
            nccl_v = ".".join(map(str, torch.cuda.nccl.version())) if not rehearsal else "gloo"
        except Exception:
            nccl_v = "unknown"
        comm_info = {"backend": "gloo (rehearsal)" if rehearsal else "nccl (RCCL)", "rccl_version": nccl_v,
                     "visible_devices": torch.cuda.device_count(), "verified_all_gather": True}
        del probe

    def barrier():
        if dist is not None:
            dist.barrier()

    def run_steps(n, events=None, collective=True):
        g = gatherer if collective else None
        for i in range(n):
            if events is not None:
                events[i][0].record()
            out = step_fn()
            if events is not None:
                events[i][1].record()
            if g is not None:
                g.submit(out)
        if g is not None:
            g.drain()

    def timed_region(collective):
        """W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + synchronize; MAX over ranks."""
        run_steps(args.warmup, collective=collective)
        torch.cuda.synchronize()
        barrier()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps, evs, collective=collective)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].item()), float(t[1].item())

    with_coll = world > 1 and not args.no_collective
    # 1. the cold figure: W warm-ups + K steps straight after start-up, before any settling (what `--settle-ms 0` reads)
    cold_elapsed, cold_kern_ms = timed_region(with_coll) if args.settle_ms > 0 else (None, None)
    # 2. clock/power settling, untimed (see --settle-ms).  A FIXED number of steps, identical on every rank: the steps
    #    contain a collective
    if args.settle_ms > 0:
        run_steps(int(args.settle_ms), collective=with_coll)
        torch.cuda.synchronize()
    # 3. the reported region
    elapsed, kern_ms = timed_region(with_coll)
    # 4. BASELINE config 4 "with and without the collective": the same K steps with no all-gather
    nocoll_elapsed = timed_region(False)[0] if with_coll else None

    if rank == 0:
        value = world * B * args.steps / elapsed
        achieved = B * FLOP_PER_FACE[F] / (kern_ms * 1e-3) / 1e12
        if args.mode in ("f16x2", "f16x2s"):
            peak = PEAK_F16_MFMA_TFLOPS
            kernel = "layer-per-launch path (pre-pass, E0, E1, E2, tail; in the strict mode the tail is two launches and the f32 re-evaluation launch follows)" if layered else "encoder_heads_f16x2_kernel<.., SPLIT=false> (four waves)"
            if args.mode == "f16x2s":
                kernel = ("prepass + 3 x layer_kernel + tail_encoder_kernel + head_kernel + f32 re-evaluation launch (7 launches)" if layered else
                          "encoder_heads_f16x2_w8_kernel (eight waves) + encoder_heads_f32_kernel in re-evaluation mode (a workgroup ends at once "
                          "where its tile has no face beyond f16's range); kernel_ms covers both launches")
            # the stream that bounds the fused split-f16 kernels in fact: every 64-face tile pulls the whole split-f16 weight
            # image (header to tail pad: 9.6 MB at F = 1404) from L2 into its CU once
            wbytes = float(_lib.lib().nlml_encoder_heads_packed_bytes(F, _lib.MODE_F16X2))   # the split-f16 image alone (the strict blob also carries the f32 image the re-evaluation launch reads)
            l2_stream = None if layered else {
                "bytes_per_launch": wbytes * ((B + 63) // 64), "achieved_TBps": wbytes * ((B + 63) // 64) / (kern_ms * 1e-3) / 1e12,
                "peak_TBps": 34.5, "frac_of_l2_peak": wbytes * ((B + 63) // 64) / (kern_ms * 1e-3) / 1e12 / 34.5,
                "per_cu_bytes_per_clk_at_1p76GHz": wbytes * ((B + 63) // 64) / (kern_ms * 1e-3) / 256 / 1.76e9,
                "note": "weights only (x and the hand-overs inside LDS are not in it); peak = MI355X_MICROARCH.md's aggregate L2 rate.  The L1 / L2 counters "
                        "(profiles/r05l1_summary.md) put the rate at which this kernel -- and the bf16 one -- is served at 0.23 128-byte lines per cycle and CU "
                        "(30 B/clk, x included) whatever the queue depth, against 0.37 for a bare all-hit stream: the L2's rate under the kernel's own miss stream "
                        "(x in both layer-0 passes, the weight stream once per XCD and tile round).  A tile's 86 k line requests at that rate are its ~330 k cycles"}
            roof_extra = {"l2_to_cu_weight_stream": l2_stream, "executed_flop_per_launch": SPLIT_PRODUCTS * B * FLOP_PER_FACE[F],
                          "executed_frac": SPLIT_PRODUCTS * achieved / peak,
                          "note": "each algorithmic product runs as 3 f16 MFMA products (hi*hi + hi*lo + lo*hi, f32 accumulate); "
                                  "frac counts the algorithmic FLOP only; at 65,536 faces the kernel runs at the board's power "
                                  "limit (clock ~1.9 GHz instead of 2.4), see DESIGN.md section 3"}
            dtype = args.mode
            what = ("split-f16 strict-fast mode (two f16 pieces per f32 operand on the f16 matrix cores, f32 accumulate, the small products of each K step "
                    "in accumulators of their own; the default.  Parity at the operating range (FX3c, 16,384 faces): distance from the f64 truth 0.88x the PINNED "
                    "reference's (fixture generated in the build container) / 1.14x torch-f32's on the GPU box's host (profiles/r04_parity_soak_1M.json); "
                    "0.03-0.07 % of the faces differ from the reference's batched output by more than 1e-4 deg (f32 mode: 0.002-0.012 %) -- "
                    "cpu_baseline.parity_check_operating_range has this run's figures)")
            if args.mode == "f16x2":
                what = "split-f16 OPT-IN fast mode (single accumulators where registers are short): 1.10x the reference's error at the operating range -- not the parity default"
        else:
            peak, kernel, roof_extra, dtype, what = PEAK_F32_MFMA_TFLOPS, "encoder_heads_f32_kernel", {}, "f32", "f32 strict parity mode (layers 0-3 summed in blocks of 128 k)"
        rec = {
            "metric": "faces_per_sec", "value": value,
            "value_cold": (world * B * args.steps / cold_elapsed) if cold_elapsed is not None else None,   # straight after start-up, W warm-ups only
            "unit": "faces/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: gloo, shared devices -- not a measurement)",
            "config": {"workload": f"landmarks->pose, batch {B}/GPU, F=1404 (468x3 landmarks), encoder+3 heads "
                                   f"{'layer-per-launch' if layered else 'fused'} HIP forward, {what}, path={args.path}",
                       "faces_per_gpu": B, "F": F, "path": args.path, "mode": args.mode, "settle_ms": args.settle_ms,
                       "seeds": {"encoder": 0, "landmarks": "1+rank"},
                       "collective": "all_gather f32[B,3] per step" if with_coll else "none"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel, "kernel_ms": kern_ms,
                         "flop_per_launch": B * FLOP_PER_FACE[F],
                         "hbm_frac": B * (BYTES_PER_FACE_K2[F]) / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, **roof_extra},
        }
        if cold_elapsed is not None:
            rec["warmup_effective"] = args.warmup + args.steps + int(args.settle_ms) + args.warmup
            rec["cold_note"] = (f"`warmup` is the flag as passed; the reported K steps are preceded by warmup_effective = {rec['warmup_effective']} "
                                f"untimed-or-separately-timed steps: W={args.warmup} warm-ups + K={args.steps} steps timed as value_cold (straight after "
                                f"start-up), {int(args.settle_ms)} settling steps (--settle-ms; the kernel runs at the board's power limit and the clock "
                                f"needs ~0.25 s of load to reach its sustained level), then W={args.warmup} warm-ups again.  --settle-ms 0 makes "
                                f"--warmup the only warm-up (value then equals value_cold's definition).")
        else:
            rec["warmup_effective"] = args.warmup
        if comm_info is not None:
            rec["comm"] = comm_info
        if nocoll_elapsed is not None:
            rec["value_no_collective"] = world * B * args.steps / nocoll_elapsed
            rec["ms_per_step_no_collective"] = nocoll_elapsed / args.steps * 1e3
        if world == 1 and not args.no_cpu_baseline:
            sample = np.arange(0, B, max(1, B // 2048))[:2048]
            got = step_fn()[torch.from_numpy(sample).to(dev)].cpu().numpy()
            rec["cpu_baseline"] = cpu_baseline(raw_np, sd, heads, args.cpu_seconds, sample, got)
            others = [] if args.no_extra else [(_lib.mode_from_name(m), m) for m in ("f16x2", "f16x2s", "f32") if m != args.mode]
            rec["cpu_baseline"]["parity_check_operating_range"] = parity_operating_range(ops, weights, dev, heads, mode, args.mode, others)
            if not args.no_extra:
                rec["cpu_baseline"]["td_path"] = td_cpu_baseline(weights, synth)
        if world == 1 and not args.no_extra and B == 65536:    # the secondary workloads are defined on the 65,536-face batch
            rec["extra"] = extra_workloads(ops, synth, weights, dev, heads, sd, raw, feats, B)
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def usable_cores() -> int:
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(raw_np, sd, heads, seconds, sample, got):
    """The oracle ("port": same ATen ops as the reference's CPU path) on this host's cores; `got` = the timed kernel's
    poses for rows `sample`, checked here against the f64 oracle (the only place bench.py touches oracle/)."""
    from oracle import encoder_heads as EH
    from oracle import feature_norm as FN
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    P = EH.Params(sd, heads)
    n = 16384
    x = FN.normalize_ipd(raw_np[:n], True)
    EH.forward_torch(x[:256], P)
    done, t0 = 0, time.perf_counter()
    while True:
        EH.forward_torch(x, P)
        done += n
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for i in range(64):                                  # the way the reference actually runs: batch 1
        EH.forward_torch(x[i:i + 1], P)
    b1 = 64 / (time.perf_counter() - t1)
    truth = EH.forward_numpy(FN.normalize_ipd(raw_np[sample], True), P, np.float64)
    err = np.degrees(np.abs(got.astype(np.float64) - truth))
    return {"value": done / dt, "unit": "faces/s", "cores": ncores, "kind": "port",
            "sample": f"{done} faces in batches of {n} (encoder+heads on pre-normalised rows, torch CPU f32, {dt:.1f} s)",
            "batch1_faces_per_sec": b1,
            "parity_check": {"faces": int(len(sample)), "max_abs_deg_vs_f64_oracle": float(err.max()),
                             "mean_abs_deg": float(err.mean()), "tolerance_deg": 1e-4}}


def parity_operating_range(ops, weights, dev, heads, mode, mode_name, also=()):
    """The measured kernel where the reference operates (FX3c: FX3b's weights -- latent over the rows of U_yaw/U_pitch/U_roll,
    poses over the trained +-50/40/30 deg bins -- on 16,384 Philox faces): p50 / p99 / max of |kernel - f64 truth| per face, next
    to the same statistics of the REFERENCE's own outputs (tests/golden/fx3c_reference_range_16k.npz: its batched and its
    one-face calls).  The seed-0 `parity_check` above is the easy regime (poses under 11 deg)."""
    from nlml_hpe_amd import synth
    from oracle import encoder_heads as EH
    gdir = os.path.join(ROOT, "tests", "golden")
    g3b, g3c = np.load(os.path.join(gdir, "fx3b_reference_range.npz")), np.load(os.path.join(gdir, "fx3c_reference_range_16k.npz"))
    sd = synth.encoder_state_dict(1404, seed=0, hidden_weight_gain=2.0)
    sd["encoder.10.weight"], sd["encoder.10.bias"] = g3b["enc10_weight"], g3b["enc10_bias"]
    x = synth.features(16384, 1404, seed=23)
    blob = torch.from_numpy(weights.pack_blob(sd, heads, mode)).to(dev)
    got = ops.encoder_heads_fwd(torch.from_numpy(x).to(dev), blob, 1404).cpu().numpy()
    truth = EH.forward_numpy(x, EH.Params(sd, heads), np.float64)

    def stats(y, ref):
        d = np.degrees(np.abs(y.astype(np.float64) - ref.astype(np.float64))).max(axis=1)
        return {"p50_deg": float(np.percentile(d, 50)), "p99_deg": float(np.percentile(d, 99)), "max_deg": float(d.max()),
                "frac_above_1e-4_deg": float((d > 1e-4).mean())}
    other = {}
    for m2, name2 in also:     # the other K2 modes on the same faces (not the timed kernel): kernel vs truth and vs the reference
        b2 = torch.from_numpy(weights.pack_blob(sd, heads, m2)).to(dev)
        g2 = ops.encoder_heads_fwd(torch.from_numpy(x).to(dev), b2, 1404).cpu().numpy()
        other[name2] = {"kernel_vs_f64_truth": stats(g2, truth), "kernel_vs_reference_batched": stats(g2, g3c["rad"]),
                        "kernel_vs_reference_batch1": stats(g2, g3c["rad_b1"])}
    vs_batched = stats(got, g3c["rad"])
    f32_frac = vs_batched["frac_above_1e-4_deg"] if mode_name == "f32" else other.get("f32", {}).get("kernel_vs_reference_batched", {}).get("frac_above_1e-4_deg")
    k_truth, r_truth = stats(got, truth), stats(g3c["rad"], truth)
    return {"faces": 16384, "mode": mode_name, "pose_span_deg": [float(np.degrees(truth.min())), float(np.degrees(truth.max()))],
            # the two figures north_star's "within 1e-4 deg of the reference CPU path" turns into at this range, side by side:
            "frac_above_1e-4_deg_vs_reference_batched": {"timed_kernel": vs_batched["frac_above_1e-4_deg"], "f32_kernel": f32_frac,
                                                         "reference_batch1_vs_itself_batched": stats(g3c["rad_b1"], g3c["rad"])["frac_above_1e-4_deg"]},
            "distance_ratio_vs_pinned_reference": {q: k_truth[q] / r_truth[q] for q in ("p50_deg", "p99_deg", "max_deg")},
            "kernel_vs_f64_truth": k_truth, "other_modes": other,
            "reference_batched_vs_f64_truth": r_truth,
            "reference_batch1_vs_f64_truth": stats(g3c["rad_b1"], truth),
            "kernel_vs_reference_batched": vs_batched,
            "kernel_vs_reference_batch1": stats(got, g3c["rad_b1"]),
            "reference_batch1_vs_reference_batched": stats(g3c["rad_b1"], g3c["rad"]),
            "note": "north_star's bar is 1e-4 deg against the reference's CPU output; at this range the reference's own two call "
                    "shapes differ by up to 1.2e-4 deg, so the statement is statistical: kernel vs truth next to reference vs truth; "
                    "kernel_vs_reference_batch1 is against the call shape the reference's entry points really use (one face per call, "
                    "NLML_HPE_Test.py:262-272), kernel_vs_reference_batched against one batched call.  The reference's own distance depends on its "
                    "host's BLAS: these are ratios to the PINNED fixture (generated in the build container); against torch-f32 on the GPU box's host the "
                    "strict-fast mode read 1.14 / 1.12 / 1.10x and the f32 kernel 0.95 / 0.91 / 0.93x on 1,048,576 faces (profiles/r04_parity_soak_1M.json)"}


def td_cpu_baseline(weights, synth):
    """The reference's CPU path for TD, restated by the oracle and timed here on ONE core (numpy's einsum
    and scipy's Powell are single-threaded): np.einsum('ijklm,i,j,k,l->m') + 0.5*sum((x - x_hat)**2) (TD_Tester.py:46,49) on
    BASELINE config 3's 4,096 faces, one evaluation each, and scipy.optimize.minimize(method='Powell') (TD_Tester.py:191-194) on
    the first 8 of them."""
    from scipy.optimize import minimize
    from oracle import tucker as TK
    art = weights.load_tucker_artefacts(os.path.join(ROOT, "outputs", "features"))
    Py, Pp, Pr = art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]
    idx = synth.tucker_grid_indices(4096, seed=2)
    Xg = synth.tucker_grid_faces(art, idx, 1e-3, seed=2)
    P = synth.tucker_params(4096, 5, seed=2)
    W = art["W"]
    TK.objective(P[0], W, Xg[0], Py, Pp, Pr)
    t0 = time.perf_counter()
    for i in range(4096):
        TK.objective(P[i], W, Xg[i], Py, Pp, Pr)
    dt_obj = time.perf_counter() - t0
    nfev, t0 = [], time.perf_counter()
    for i in range(8):
        res = minimize(lambda p: TK.objective(p, W, Xg[i], Py, Pp, Pr), np.zeros(8), method="Powell")
        nfev.append(int(res.nfev))
    dt_pw = time.perf_counter() - t0
    return {"kind": "port", "cores": 1,
            "objective_evals_per_sec": 4096 / dt_obj, "objective_sample": f"4096 evaluations (config 3's faces, one each), numpy f64 einsum, {dt_obj:.1f} s",
            "powell_seconds_per_face": dt_pw / 8, "powell_faces_per_sec": 8 / dt_pw, "powell_mean_nfev": float(np.mean(nfev)),
            "powell_sample": f"scipy Powell on 8 of config 3's faces, {dt_pw:.1f} s"}


def extra_workloads(ops, synth, weights, dev, heads, sd1404, raw, feats, B):
    from nlml_hpe_amd import _lib
    ex = {}
    blob = torch.from_numpy(weights.pack_blob(sd1404, heads, _lib.MODE_F32)).to(dev)
    blob_hx = torch.from_numpy(weights.pack_blob(sd1404, heads, _lib.MODE_F16X2)).to(dev)
    sd136 = synth.encoder_state_dict(136, seed=0)
    x136 = torch.from_numpy(synth.features(B, 136, seed=1)).to(dev)

    EV = "mean of one HIP event pair per launch, {n} launches after {w} warm-ups (torch's current stream)"

    def k2(fn, F, peak, products=1):
        ms = time_kernel(fn, 100, warm=30)
        tf = B * FLOP_PER_FACE[F] / ms / 1e9
        d = {"faces_per_sec": B / ms * 1e3, "tflops": tf, "mfma_frac": tf / peak, "timing": EV.format(n=100, w=30)}
        if products > 1:
            d["executed_frac"] = products * tf / peak
        return d

    # f32 parity kernel (f32 matrix cores; the MFMA-bound formulation)
    ex["k2_f32_fused_F1404"] = k2(lambda: ops.landmarks_to_pose(raw, blob, True), 1404, PEAK_F32_MFMA_TFLOPS)
    ex["k2_f32_features_F1404"] = k2(lambda: ops.encoder_heads_fwd(feats, blob, 1404), 1404, PEAK_F32_MFMA_TFLOPS)
    blob136 = torch.from_numpy(weights.pack_blob(sd136, heads, _lib.MODE_F32)).to(dev)
    ex["k2_f32_features_F136"] = k2(lambda: ops.encoder_heads_fwd(x136, blob136, 136), 136, PEAK_F32_MFMA_TFLOPS)
    # split-f16 parity kernel (f16 matrix cores, three products per algorithmic product)
    ex["k2_f16x2_fused_F1404"] = k2(lambda: ops.landmarks_to_pose(raw, blob_hx, True), 1404, PEAK_F16_MFMA_TFLOPS, SPLIT_PRODUCTS)
    ex["k2_f16x2_features_F1404"] = k2(lambda: ops.encoder_heads_fwd(feats, blob_hx, 1404), 1404, PEAK_F16_MFMA_TFLOPS, SPLIT_PRODUCTS)
    blob136_hx = torch.from_numpy(weights.pack_blob(sd136, heads, _lib.MODE_F16X2)).to(dev)
    ex["k2_f16x2_features_F136"] = k2(lambda: ops.encoder_heads_fwd(x136, blob136_hx, 136), 136, PEAK_F16_MFMA_TFLOPS, SPLIT_PRODUCTS)
    for k in ("k2_f16x2_fused_F1404", "k2_f16x2_features_F1404", "k2_f16x2_features_F136"):
        ex[k]["parity_class"] = "OPT-IN fast mode: 1.10x the reference's error at the operating range (FX3c) -- not a strict parity figure"
    # strict-fast mode (the default): the same operands and MFMAs, split accumulators
    blob_hxs = torch.from_numpy(weights.pack_blob(sd1404, heads, _lib.MODE_F16X2S)).to(dev)
    ex["k2_f16x2s_fused_F1404"] = k2(lambda: ops.landmarks_to_pose(raw, blob_hxs, True), 1404, PEAK_F16_MFMA_TFLOPS, SPLIT_PRODUCTS)
    ex["k2_f16x2s_features_F1404"] = k2(lambda: ops.encoder_heads_fwd(feats, blob_hxs, 1404), 1404, PEAK_F16_MFMA_TFLOPS, SPLIT_PRODUCTS)
    blob136_hxs = torch.from_numpy(weights.pack_blob(sd136, heads, _lib.MODE_F16X2S)).to(dev)
    ex["k2_f16x2s_features_F136"] = k2(lambda: ops.encoder_heads_fwd(x136, blob136_hxs, 136), 136, PEAK_F16_MFMA_TFLOPS, SPLIT_PRODUCTS)
    # The same forward as trunk launch + streamed tail launch (nlml_landmarks_to_pose_streamed: bit-identical, opt-in), alternating with the
    # fused kernel in this process: what taking the tail out of the tile buys.  The trunk launch alone costs ~0.90 of the fused kernel --
    # its share of the L2 -> CU bytes -- which is the measurement behind "the fused kernel's time is its L2 traffic" (DESIGN.md section 3).
    ab = {"fused_ms": [], "streamed_ms": []}
    for _ in range(2):
        ab["fused_ms"].append(time_kernel(lambda: ops.landmarks_to_pose(raw, blob_hxs, True), 100, warm=30))
        ab["streamed_ms"].append(time_kernel(lambda: ops.landmarks_to_pose_streamed(raw, blob_hxs, True), 100, warm=30))
    same = bool(torch.equal(ops.landmarks_to_pose(raw, blob_hxs, True), ops.landmarks_to_pose_streamed(raw, blob_hxs, True)))
    ms_f, ms_s = float(np.mean(ab["fused_ms"])), float(np.mean(ab["streamed_ms"]))
    ex["k2_f16x2s_streamed_tail_F1404"] = {"faces_per_sec": B / ms_s * 1e3, "fused_faces_per_sec_same_run": B / ms_f * 1e3, "speedup_vs_fused": ms_f / ms_s,
                                           "bit_identical_to_fused": same, **ab, "timing": EV.format(n=100, w=30) + ", fused / streamed alternating twice",
                                           "note": "opt-in (explicit entry points, NLML_K2_STREAMED_MIN); not the timed step, not `value`"}
    # Batch-size sweep in the default mode, through the dispatch the host layer uses (HIPPoseModel: layer-per-launch path up to 4,096 faces,
    # fused kernel above): one forward per launch sequence, HIP events.  A forward is a chain of 11 dependent layers: the layer-per-launch
    # path costs ~45 us for ANY batch it takes (seven launches), the fused kernel one tile time (~0.15 ms) for any batch up to one round
    # of tiles (16,384 faces) -- so faces/s climbs with the batch until the chip is full; `frac_of_full` is against the 65,536-face rate.
    from nlml_hpe_amd.model import HIPPoseModel
    sweep, full_rate = [], None
    for Bs in (65536, 32768, 16384, 8192, 5120, 4096, 2000, 512, 64):
        r = raw[:Bs].contiguous()
        small = 0 < Bs <= HIPPoseModel.small_batch_max("f16x2s")
        f = (lambda r=r: ops.landmarks_to_pose_small(r, blob_hxs, True)) if small else (lambda r=r: ops.landmarks_to_pose(r, blob_hxs, True))
        ms = time_kernel(f, 60 if Bs >= 16384 else 200, warm=30)
        rate = Bs / ms * 1e3
        full_rate = rate if full_rate is None else full_rate
        sweep.append({"faces": Bs, "ms": ms, "faces_per_sec": rate, "frac_of_full": rate / full_rate, "path": "layer-per-launch" if small else "fused"})
    ex["k2_batch_sweep"] = {"mode": "f16x2s", "entry": "landmarks_to_pose (raw landmarks, normalisation fused)", "points": sweep,
                            "timing": EV.format(n="60-200", w=30),
                            "note": "below ~6,800 faces a launch cannot reach 60 % of the full-batch rate: 2,000 faces in 45 us (the seven-launch "
                                    "floor of the layer-per-launch path) would be 44 M faces/s; see DESIGN.md section 3, small and mid-size batches"}
    # (the blocks the two outputs will most likely be carved from are poisoned first: a launch that wrote nothing would otherwise
    # leave the previous call's pose in its torch.empty output and could show up as a difference of exactly 0)
    poison = [torch.full((B, 3), float("nan"), device=dev) for _ in range(2)]
    del poison
    pose_hx, pose_f32 = ops.landmarks_to_pose(raw, blob_hxs, True), ops.landmarks_to_pose(raw, blob, True)
    torch.cuda.synchronize()
    d = torch.rad2deg((pose_hx - pose_f32).abs())
    fin = bool(torch.isfinite(d).all())
    ex["k2_f16x2s_vs_f32_kernel_all_faces"] = {"max_abs_deg": float(d.max()) if fin else None,
                                              "mean_abs_deg": float(d.mean()) if fin else None, "faces": B, "all_finite": fin,
                                              "distinct_buffers": pose_hx.data_ptr() != pose_f32.data_ptr()}
    # throughput mode (bf16 operands, f32 accumulate): NOT a parity result -- its measured error is reported with it
    blob_bf = torch.from_numpy(weights.pack_blob(sd1404, heads, _lib.MODE_BF16)).to(dev)
    ms = time_kernel(lambda: ops.landmarks_to_pose(raw, blob_bf, True), 100, warm=30)
    sub = slice(0, 4096)
    d = (ops.landmarks_to_pose(raw[sub], blob_bf, True) - ops.landmarks_to_pose(raw[sub], blob, True)).abs()
    ex["k2_bf16_throughput_mode"] = {"faces_per_sec": B / ms * 1e3, "tflops": B * FLOP_PER_FACE[1404] / ms / 1e9,
                                     "bf16_mfma_frac": B * FLOP_PER_FACE[1404] / ms / 1e9 / PEAK_F16_MFMA_TFLOPS,
                                     "max_abs_deg_vs_f32_mode": float(torch.rad2deg(d.max())),
                                     "mean_abs_deg_vs_f32_mode": float(torch.rad2deg(d.mean())),
                                     "timing": EV.format(n=100, w=30),
                                     "note": "throughput mode; fails the 1e-4 deg parity bar by design (SURVEY D3)"}
    # BASELINE config 4 per GPU: an AFLW2000-shaped stream, 2,000 faces per step (32 tiles: 1/8 of the CUs busy, so the
    # rate is set by one tile's latency; the all-gather of config 4 exists only at N > 1: --batch 2000 --gpus N)
    r2k = raw[:2000].contiguous()
    for name, b in (("f16x2s", blob_hxs), ("f32", blob)):
        ms = time_kernel(lambda: ops.landmarks_to_pose(r2k, b, True), 200, warm=50)
        ex[f"config4_2000_faces_step_{name}"] = {"ms_per_step": ms, "faces_per_sec": 2000 / ms * 1e3, "timing": EV.format(n=200, w=50)}
    # the same step on the layer-per-launch path (same bits as the fused f16x2 kernel; what HIPPoseModel uses up to 4,096 faces)
    ms = time_kernel(lambda: ops.landmarks_to_pose_small(r2k, blob_hxs, True), 200, warm=50)
    ex["config4_2000_faces_step_f16x2s_layered"] = {"ms_per_step": ms, "faces_per_sec": 2000 / ms * 1e3, "timing": EV.format(n=200, w=50)}
    r64 = raw[:64].contiguous()
    ms_f = time_kernel(lambda: ops.landmarks_to_pose(r64, blob_hxs, True), 200, warm=50)
    ms_l = time_kernel(lambda: ops.landmarks_to_pose_small(r64, blob_hxs, True), 200, warm=50)
    ex["tick_64_faces_f16x2s"] = {"fused_ms": ms_f, "layered_ms": ms_l, "timing": EV.format(n=200, w=50)}
    ms = time_kernel(lambda: ops.normalize_ipd(raw, True), 20)
    ex["k1_normalize"] = {"faces_per_sec": B / ms * 1e3, "gbs": B * BYTES_PER_FACE_K1 / ms / 1e6,
                          "hbm_frac": B * BYTES_PER_FACE_K1 / ms / 1e6 / PEAK_HBM_GBS, "timing": EV.format(n=20, w=3)}
    art = weights.load_tucker_artefacts(os.path.join(ROOT, "outputs", "features"))
    cp = torch.from_numpy(np.stack([art["optimized_yaw"][:3], art["optimized_pitch"][:3], art["optimized_roll"][:3]])).to(dev)
    Wm = torch.from_numpy(art["W"].reshape(135, 1404)).to(dev)
    N = 4096
    P = torch.from_numpy(synth.tucker_params(N, 5, seed=2)).to(dev)
    ms = time_kernel(lambda: ops.tucker_objective(Wm, feats[:N], P, cp, order="fast"), 20)
    ms_s = time_stream(lambda: ops.tucker_objective(Wm, feats[:N], P, cp, order="fast"), 200, warm=400)   # sustained, see the reference order below
    tf = N * TUCKER_FLOP_PER_EVAL / ms_s / 1e9
    ex["k3_tucker_objective_fast_order"] = {"evals_per_sec": N / ms_s * 1e3, "tflops_f64": tf,
                                 "f64_frac": tf / PEAK_F64_TFLOPS, "n": N,
                                 "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s",
                                              "frac": tf / PEAK_F64_TFLOPS, "traffic": None, "kernel": "tucker_objective_kernel",
                                              "flop_per_launch": N * TUCKER_FLOP_PER_EVAL, "kernel_ms": ms_s},
                                 "timing": "200 launches back to back between one event pair after 400 warm-up launches (sustained clock; an event pair "
                                           "around every launch adds ~3 us of launch gap to each, and a burst after idle runs at a lower clock)",
                                 "f64_frac_event_pair_per_launch": N * TUCKER_FLOP_PER_EVAL / ms / 1e9 / PEAK_F64_TFLOPS}
    NL = 65536
    PL = torch.from_numpy(synth.tucker_params(NL, 5, seed=3)).to(dev)
    ms_l = time_stream(lambda: ops.tucker_objective(Wm, feats[:NL], PL, cp, order="fast"), 10)
    ex["k3_tucker_objective_fast_order_65536"] = {"evals_per_sec": NL / ms_l * 1e3, "timing": "10 launches back to back between one HIP event pair, 3 warm-ups",
                                       "f64_frac": NL * TUCKER_FLOP_PER_EVAL / ms_l / 1e9 / PEAK_F64_TFLOPS, "n": NL}
    # host-resident batch: pinned staging + copy stream overlapped with compute (PCIe-inclusive; never `value`)
    from nlml_hpe_amd.model import HIPPoseModel
    from nlml_hpe_amd.pipeline import HostPipeline
    mdl = HIPPoseModel(sd1404, heads, device=dev)
    pipe = HostPipeline(mdl, batch=8192)
    raw_host = raw.cpu().numpy()
    pipe.run(raw_host[:32768])
    t0 = time.perf_counter()
    pipe.run(raw_host)
    dt = time.perf_counter() - t0
    ex["host_resident_pcie_inclusive"] = {"faces_per_sec": B / dt, "gb_per_s_h2d": B * 5616 / dt / 1e9, "timing": "host wall clock around one run over the 65,536 faces, after a 32,768-face warm-up run",
                                          "h2d_form": pipe.last_mode,
                                          "note": "host numpy -> H2D on a copy stream (the caller's array page-locked in place and read by the DMA directly when that is fast, "
                                                  "else staged through pinned buffers: h2d_form says which) -> fused kernel -> D2H, double-buffered"}
    # BASELINE config 5 (per GPU): 64 concurrent streams, one tick = 64 raw-landmark sets -> smoothed pose + axes
    from nlml_hpe_amd.video import GraphedTick, VideoPoseTracker
    S, T = 64, 300
    clips = raw[:S * 8].reshape(8, S, 468, 3)
    for label in ("eager", "hipgraph"):
        tr = VideoPoseTracker(mdl, S, 1920, 1080)
        gt = GraphedTick(tr) if label == "hipgraph" else None
        lat = []
        for t in range(T + 20):
            frame = clips[t % 8]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if gt is None:
                tr.tick(frame)
            else:
                gt.static_raw.copy_(frame)
                gt.replay()
            torch.cuda.synchronize()
            if t >= 20:
                lat.append(time.perf_counter() - t0)
        lat = np.array(lat)
        ex[f"video_64_streams_{label}"] = {"tick_ms_p50": float(np.percentile(lat, 50) * 1e3), "tick_ms_p99": float(np.percentile(lat, 99) * 1e3),
                                           "faces_per_sec_sustained": S / float(lat.mean()), "offered_load_faces_per_sec": 64 * 30,
                                           "timing": "host wall clock per tick, synchronize on both sides, 300 ticks after 20"}
    # TD end-to-end (TD_Tester.Test): device-side lock-step Powell, one minimisation per face, BASELINE config 3 (4,096 faces).
    # The headline is the REFERENCE order (the default and the parity mode): the reference's objective bits, scipy's own
    # trajectory and end point (FX4 / FX5 bit-exact); the fast (matrix-core) order is reported beside it with how far its end
    # points land from the reference order's.
    idx = synth.tucker_grid_indices(4096, seed=2)
    Xg = torch.from_numpy(synth.tucker_grid_faces(art, idx, 1e-3, seed=2)).to(dev)

    POWELL_RUNS = 3
    POWELL_TIMING = ("mean of {n} launches, host wall clock around launch + synchronize (one launch is the whole workload), "
                     "after a 64-face warm-up launch; min and max beside it").format(n=POWELL_RUNS)

    def powell(order):
        ops.tucker_powell(Wm, Xg[:64], cp, order=order)
        torch.cuda.synchronize()
        dts = []
        for _ in range(POWELL_RUNS):
            t0 = time.perf_counter()
            res = ops.tucker_powell(Wm, Xg, cp, order=order)
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        return res, float(np.mean(dts)), {"seconds_min": min(dts), "seconds_max": max(dts), "runs": POWELL_RUNS, "timing": POWELL_TIMING}

    res_r, dt_r, spread_r = powell("reference")
    nfr = res_r["nfev"].double()
    ev_r = float(nfr.sum()) / dt_r
    ex["td_powell_end_to_end"] = {
        "order": "reference (parity mode, the default)", "faces": int(Xg.shape[0]), "seconds": dt_r, **spread_r, "faces_per_sec": Xg.shape[0] / dt_r,
        "mean_nfev": float(nfr.mean()), "max_nfev": float(nfr.max()), "face_evals_per_sec": ev_r,
        "converged_frac": float((res_r["status"] == 1).double().mean()),
        "roofline": {"bound": "valu_f64", "achieved": ev_r * TUCKER_REF_OPS_PER_EVAL / 1e12, "peak": PEAK_F64_VALU_TOPS, "unit": "T op/s",
                     "frac": ev_r * TUCKER_REF_OPS_PER_EVAL / 1e12 / PEAK_F64_VALU_TOPS,
                     "frac_in_algorithmic_flop_of_f64_peak": ev_r * TUCKER_FLOP_PER_EVAL / 1e12 / PEAK_F64_TFLOPS, "traffic": None,
                     "kernel": "tucker_powell_kernel<NLML_TD_ORDER_REFERENCE>", "kernel_ms": dt_r * 1e3,
                     "note": "947,700 separately rounded f64 vector operations per evaluation against the f64 vector issue rate "
                             "(non-fma); the launch lasts as long as its slowest face"}}
    res, dt, spread_f = powell("fast")
    nf = res["nfev"].double()
    dd = torch.rad2deg((res_r["x"][:, :3] - res["x"][:, :3]).abs()).max(dim=1).values
    ex["td_powell_fast_order"] = {
        "order": "fast (f64 matrix cores; opt-in, NOT a parity mode)", "faces": int(Xg.shape[0]), "seconds": dt, **spread_f, "faces_per_sec": Xg.shape[0] / dt,
        "mean_nfev": float(nf.mean()), "max_nfev": float(nf.max()), "face_evals_per_sec": float(nf.sum()) / dt,
        "converged_frac": float((res["status"] == 1).double().mean()),
        "end_point_vs_reference_order": {"median_deg": float(dd.median()), "max_deg": float(dd.max()),
                                         "frac_above_0.02deg": float((dd > 0.02).double().mean()),
                                         "frac_above_1deg": float((dd > 1.0).double().mean())},
        "roofline": {"bound": "mfma", "achieved": float(nf.sum()) / dt * TUCKER_FLOP_PER_EVAL / 1e12, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s",
                     "frac": float(nf.sum()) / dt * TUCKER_FLOP_PER_EVAL / 1e12 / PEAK_F64_TFLOPS, "traffic": None,
                     "kernel": "tucker_powell_kernel<NLML_TD_ORDER_FAST>", "kernel_ms": dt * 1e3},
        "note": "noisy grid faces (sigma 1e-3): where the objective has several shallow minima Powell's end point is chaotic in the "
                "last bits of the objective, so a re-ordered objective lands elsewhere on some faces"}
    # sustained rate: the clock needs tens of milliseconds of load to reach its sustained level (a burst of 20 launches after idle
    # runs at ~2.0 GHz and reads 0.58 of the issue rate), so 400 untimed launches, then 200 back to back -- as `value` is measured
    ms_burst = time_stream(lambda: ops.tucker_objective(Wm, feats[:N], P, cp, order="reference"), 20, warm=5)
    ms = time_stream(lambda: ops.tucker_objective(Wm, feats[:N], P, cp, order="reference"), 200, warm=400)
    evs = N / ms * 1e3
    ex["k3_tucker_objective"] = {
        "order": "reference (parity mode, the default)", "evals_per_sec": evs, "n": N,
        "evals_per_sec_burst_of_20_after_idle": N / ms_burst * 1e3,
        "roofline": {"bound": "valu_f64", "achieved": evs * TUCKER_REF_OPS_PER_EVAL / 1e12, "peak": PEAK_F64_VALU_TOPS, "unit": "T op/s",
                     "frac": evs * TUCKER_REF_OPS_PER_EVAL / 1e12 / PEAK_F64_VALU_TOPS,
                     "frac_in_algorithmic_flop_of_f64_peak": evs * TUCKER_FLOP_PER_EVAL / 1e12 / PEAK_F64_TFLOPS, "traffic": None,
                     "kernel": "tucker_objective_ref_kernel", "kernel_ms": ms,
                     "note": "5 separately rounded f64 operations per (q, m) on the vector ALUs; balanced passes: every SIMD issues the same number of (column, evaluation) chains; "
                             "sustained rate (400 warm-up launches, 200 timed back to back); a burst after idle runs at ~2.0 GHz (tools/td_ref_stamps.py)"}}
    return ex


if __name__ == "__main__":
    main()
