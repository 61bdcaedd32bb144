"""The N > 1 path of bench.py (batch-sharded replicas, barrier + max-over-ranks timing, one JSON line from rank 0, the
DDP-wrapped real model as `model_step`) with world_size 2 on CPU: gloo backend, kernels under tests/hipemu."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, port):
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--emu", "--seq-len", "3000", "--d-model", "4",
           "--batch", "1", "--dtype", "bf16", "--model-layers", "2"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout          # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_world_size_2_gloo():
    r = _run(2, 29611)
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["sweep"] is None                          # the sweep belongs to the N = 1 run only
    assert r["unit"] == "nt/s" and r["higher_is_better"] is True and r["vs_baseline"] is None
    assert r["config"]["seq_len"] == 3000 and "dp2" in r["config"]["parallelism"]
    # whole-job aggregate: 2 ranks x 1 sequence x 3000 nt per step
    assert abs(r["value"] - 2 * 3000 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6
    assert r["roofline"]["bound"] == "hbm" and r["roofline"]["algorithmic_bytes_per_step"] == 5 * 4 * 3000 * 2 + 12 * 4 * 3000 + 32
    # the N > 1 leg's real collective: HyenaDNALM under DDP (train.py:611-620), whole-job nt/s over both ranks
    m = r["model_step"]
    assert "error" not in m, m
    assert m["n_gpus"] == 2 and "DDP" in m["parallelism"] and "gradient_as_bucket_view=True" in m["parallelism"]
    assert abs(m["value"] - 2 * 3000 / (m["ms_per_step"] * 1e-3)) / m["value"] < 1e-6 and m["loss"] == m["loss"]
    # `dist`: a reader of the line can tell that N ranks met -- backend, world size, every rank's own step time, one timed
    # gradient-sized all_reduce whose SUM proves all ranks took part (here: gloo on the CPU; on the GPU node: RCCL + its transport lines)
    d = r["dist"]
    assert d is not None and "error" not in d, d
    assert d["backend"] == "gloo" and d["is_rccl"] is False and d["world_size"] == 2
    assert len(d["per_rank_ms"]) == 2 and all(t > 0 for t in d["per_rank_ms"])
    assert abs(max(d["per_rank_ms"]) - r["ms_per_step"]) <= 1e-6 * r["ms_per_step"] + 1e-9       # `value` is the slowest rank's time
    assert d["allreduce_probe_ms"] > 0 and d["allreduce_probe_sum_ok"] is True and d["allreduce_probe_bytes"] > 0
    for key in ("devices_visible", "nccl_version", "max_rank_ms", "allreduce_probe_busbw_GBs"):
        assert key in d, key


def test_rccl_log_summary_counts_transports(tmp_path):
    """what rank 0 makes of RCCL's INFO log on a GPU node (bench._rccl_log_summary), on lines of the published format"""
    sys.path.insert(0, ROOT)
    import bench
    log = tmp_path / "rccl.log"
    log.write_text("host:1:1 [0] NCCL INFO NCCL version 2.22.3+hip6.3 RCCL\n"
                   "host:1:1 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC\n"
                   "host:1:1 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC\n"
                   "host:1:1 [0] NCCL INFO Channel 00/0 : 1[1] -> 0[0] via SHM/direct/direct\n"
                   "host:1:1 [0] NCCL INFO comm 0x1 rank 0 nranks 2 cudaDev 0 busId 1000 - Init COMPLETE\n")
    s = bench._rccl_log_summary(str(log))
    assert "version" in s["version_line"] and s["transports"] == {"P2P/IPC": 2, "SHM/direct/direct": 1} and len(s["sample"]) >= 3
    assert "error" in bench._rccl_log_summary(str(tmp_path / "missing.log"))


def test_bench_single_process_line_shape():
    env = dict(os.environ, OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emu", "--steps", "1", "--warmup", "0",
                        "--seq-len", "1024", "--d-model", "2"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r
    assert r["n_gpus"] == 1 and r["data"] == "synthetic" and "workload" in r["config"]
    # roofline: the floor this plan can reach and where the traffic figure comes from travel with the line
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "floor_frac", "floor_derivation"):
        assert key in r["roofline"], key
    # `sweep`: the other BASELINE.json configurations through the same timed loop (N = 1 runs; the emulated run times one small stand-in)
    assert isinstance(r["sweep"], list) and len(r["sweep"]) >= 1
    for c in r["sweep"]:
        assert "error" not in c, c
        for key in ("seq_len", "batch_per_gpu", "channels", "steps", "ms_per_step", "value", "unit", "frac", "valu_frac", "floor_frac",
                    "traffic", "traffic_source", "hipgraph_replay"):
            assert key in c, key
        assert c["unit"] == "nt/s" and c["ms_per_step"] > 0
        assert abs(c["value"] - c["batch_per_gpu"] * c["seq_len"] / (c["ms_per_step"] * 1e-3)) / c["value"] < 1e-6


def test_bench_sweep_names_the_four_other_contract_configurations():
    """BASELINE.json configs 1-4 (the headline is config 5), as (L, B per GPU, d)"""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.SWEEP == [(1024, 8, 128), (32768, 8, 256), (160000, 2, 256), (450560, 1, 256)]
    # a PMC record taken on another generation of a plan's kernels is reported as stale, not as a measurement
    t, why = bench.measured_traffic(1048576, 256, 1, "bf16", True, "twolevel")
    assert (t is None) == why.startswith(("stale", "no PMC"))
    saved = dict(bench.KERNEL_SET)
    try:
        bench.KERNEL_SET["twolevel"] = "never"
        t, why = bench.measured_traffic(1048576, 256, 1, "bf16", True, "twolevel")
        assert t is None and why.startswith("stale")
    finally:
        bench.KERNEL_SET.update(saved)


def test_ddp_operator_world_size_2():
    """batch-sharded DDP over the operator (fused mixer shell, fused filter, long conv -- all through autograd Functions):
    all-reduced gradients == single-process gradients over the whole batch"""
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", os.path.join(ROOT, "tests", "_ddp_worker.py")]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    assert "DDP_OK world=2" in p.stdout, p.stdout[-1500:]


def test_ddp_lm_world_size_2():
    """bench.py's N > 1 model: HyenaDNALM under DDP(find_unused_parameters=False, gradient_as_bucket_view=True) -- all-reduced
    gradients of every parameter == single-process gradients over the whole batch"""
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29615", os.path.join(ROOT, "tests", "_ddp_lm_worker.py")]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    assert "DDP_LM_OK world=2" in p.stdout, p.stdout[-1500:]
