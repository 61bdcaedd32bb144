#!/usr/bin/env python
"""bench.py -- the hot path on N GPUs of one node:  Hyena long convolution (fftconv) forward + backward.

A "step" is one forward + one backward pass of the fused long convolution over one batch of synthetic
activations already resident in HBM (one Hyena layer call: out = causal_conv(u, k) + bias * u, then
du, dk, dbias for a random upstream gradient).  Default workload = the configuration BASELINE.json quotes the
metric on: L = 1,048,576, d = 256, B = 1 per GPU, bf16 activations, fp32 filter (hyenadna-large-1m's layer).

    python bench.py                      # 1 GPU, default K / W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU: the path shards on the batch axis (independent sequences): every rank runs the same
per-GPU workload on its own seeded batch (weak scaling).  The convolution itself has no exchange step, so the timed
region (`value`) carries no collective.  The only collective of the training step this path belongs to is DDP's gradient
all-reduce, and that is measured where it lives: at every N the line also carries `model_step` -- the REAL
hyenadna model (hyena_dna_amd.lm.HyenaDNALM, north_star configuration 5: d_model 256, n_layer 8, d_inner 1024) on this
rank's synthetic token batch, forward + backward + AdamW, for N > 1 wrapped in torch DistributedDataParallel exactly as
the reference's trainer does (train.py:611-620: find_unused_parameters=False, gradient_as_bucket_view=True; RCCL carries
the model's ~6.6 M fp32 gradients in DDP's buckets, overlapped with the backward), barrier-bracketed, max over ranks.

One JSON line on stdout (rank 0).  `roofline` is computed from algorithmic bytes (SURVEY.md 8d:
5*B*D*L*s + 12*D*L per step) over the HIP-event time of the timed region; `roofline_valu` is the second roofline of
SURVEY.md 8d (algorithmic flops vs the fp32 vector peak) together with the LDS bytes the transforms' exchanges move;
`cpu_baseline` is the oracle (`oracle/hyena_oracle.py`, a restatement of the reference's torch.fft path) timed on this
box's host cores on a bounded sample.
"""
import argparse
import json
import os
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime starts: see hyena_dna_amd/__init__.py
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured copy)
VALU_PEAK_TFLOPS = 157.3         # fp32 vector peak (MI355X_MICROARCH.md; v_fma_f32 measured at 117 TF by scripts/valu_rate.hip)
LDS_PEAK_TBS = 45.0              # aggregate ds_write rate, the tighter half of an exchange (MI355X_MICROARCH.md: 38-51 TB/s)
METRIC = "nucleotides/sec fwd+bwd at L=1M d=256; fftconv achieved HBM GB/s vs peak"


def algorithmic_bytes(B, D, L, s):
    """SURVEY.md 8d, operator boundary of the reference's fftconv: fwd reads u, k and writes out; bwd reads dout, u,
    k and writes du, dk (+ dbias)."""
    return 5 * B * D * L * s + 12 * D * L + 8 * D


def algorithmic_flops(B, D, L, M):
    """a real FFT of 2M points done as a complex FFT of M points costs ~5 M log2 M flops (nominal radix-2
    count); transforms per (b, d) row, fwd+bwd: U, y; G, du, U again = 5, plus K and dk once per channel = 2 / B."""
    import math
    per = 5.0 * M * math.log2(max(M, 2))
    return (5 * B + 2) * D * per


def lds_exchange_bytes(B, D, M):
    """Bytes the transforms move through LDS per step: every transform is three (M <= 32768: 32 x 32 x R on chip) or four
    (two-level: 32 x M1/32 columns, 32 x 32 rows) register passes, i.e. two exchanges, each writing and reading 8 M bytes."""
    return (5 * B + 2) * D * 2 * 2 * 8 * M


# Which generation of the long-convolution kernels the tree holds: a hash of the plan's source files (round 5; until then a label bumped by
# hand, and missed twice).  A PMC record (profiles/pmc_traffic.json, `kernel_set`) taken on other sources is reported as stale (traffic null)
# instead of being passed off as a measurement of the current kernels.
PLAN_SOURCES = {"onchip": ("onchip_kernels.h", "onchip.hip", "onchip_dk.hip", "onchip_host.h", "launch.h"),
                "twolevel": ("fftconv_kernels.h", "fftconv.hip", "launch.h")}


def _plan_hash(plan):
    import hashlib
    h = hashlib.sha1()
    for name in PLAN_SOURCES[plan]:
        with open(os.path.join(ROOT, "hyena_dna_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:12]


KERNEL_SET = {plan: _plan_hash(plan) for plan in PLAN_SOURCES}


def measured_traffic(L, D, B, io_dtype, save, plan):
    """HBM bytes per step from the committed PMC run (profiles/pmc_traffic.json, made by scripts/gpu_pmc_cfg.sh + pmc_traffic_json.py) if
    one was taken on this exact configuration AND on the current generation of this plan's kernels; (None, why) otherwise (rocprofv3
    cannot run inside this process).  Returns (bytes or None, provenance string)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            doc = json.load(f)
        for t in doc["configs"]:
            c = t["config"]
            if (c["seq_len"], c["channels"], c["batch_per_gpu"], c["io_dtype"], c["save_spectra"]) == (L, D, B, io_dtype, bool(save)):
                ks = t.get("kernel_set", "?")
                if ks != f"{plan}-{KERNEL_SET[plan]}":
                    return None, f"stale: PMC run {t.get('source', '?')} was taken on kernel set {ks}, the tree holds {plan}-{KERNEL_SET[plan]}"
                return t["traffic_bytes_per_step"], f"{t.get('source', '?')}; kernel set {ks}"
    except (OSError, KeyError, ValueError):
        pass
    return None, "no PMC run committed for this configuration"


# The other four BASELINE.json configurations (L, B per GPU, d): timed by the same loop as the headline in the default N = 1 run
SWEEP = [(1024, 8, 128), (32768, 8, 256), (160000, 2, 256), (450560, 1, 256)]
# ... and one length between the plans' home grounds (the former cliff above 32768: VERDICT r3 item 8), not a contract configuration
SWEEP_EXTRA = [(65536, 4, 256)]
# The lengths the reference's trainer really hands the operator: hg38_dataset.py:220-223 (`data = seq[:-1]`) makes L = max_length - 1, odd, so
# every channel-major row starts 2 bytes off a 4- / 16-byte boundary.  (L, B per GPU, d, the aligned neighbour it is compared with)
SWEEP_REAL = [(32767, 8, 256, 32768), (159999, 2, 256, 160000), (449999, 1, 256, 450000), (999999, 1, 256, 1000000),
              (1048575, 1, 256, 1048576)]
# The SHIPPED experiment's shapes at the operator / model level: configs/experiment/hg38/hg38_hyena.yaml:47-48 is batch_size 256,
# max_length 1024 -> the operator sees (B, L, D) = (256, 1023, 128) in a 2-layer model (hg38_dataset.py:222, `data = seq[:-1]`; README default
# recipe); the 32k and 160k models likewise at (8, 32767, 256) and (2, 159999, 256).  B > 1 at odd L is the only thing the trainer ever runs.
# (L, B per GPU, d_model, n_layer, aligned neighbour)
REAL_SHAPES = [(1023, 256, 128, 2, 1024), (32767, 8, 256, 8, 32768), (159999, 2, 256, 8, 160000)]
# What the part sustains for the mixed read + write streams of the two-level plan (profiles/cpol_bw_r2.txt: 5.0-5.3 TB/s typical, 5.8 TB/s
# the single best case): the floor of ANY exact-fp32 two-pass transform is its real traffic / this rate
MIXED_STREAM_TBS = 5.8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seq-len", type=int, default=1048576)
    ap.add_argument("--d-model", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1, help="sequences per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--chunk", type=int, default=0, help="channels per kernel-chain pass (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-model", action="store_true", help="skip the secondary full-model step (for N > 1: the DDP-wrapped model)")
    ap.add_argument("--model-layers", type=int, default=8, help="n_layer of the secondary full-model step (hyenadna-large-1m: 8)")
    ap.add_argument("--no-operator", action="store_true", help="skip the secondary whole-layer measurement")
    ap.add_argument("--no-sweep", action="store_true", help="skip the `sweep` field (the other four BASELINE.json configurations, N = 1)")
    ap.add_argument("--no-save-spectra", action="store_true",
                    help="backward recomputes the column spectra of u and k instead of reusing the forward's")
    ap.add_argument("--fwd-only", action="store_true", help="diagnostic; the reported metric needs fwd+bwd")
    ap.add_argument("--no-graph", action="store_true", help="launch-bound sizes: time eager calls instead of hipGraph replays of the step")
    ap.add_argument("--share-gpu0", action="store_true",
                    help="TEST ONLY: every rank on cuda:0 with the gloo backend (exercises the N > 1 GPU leg on a 1-GPU box)")
    ap.add_argument("--emu", action="store_true",
                    help="TEST ONLY: run the host logic on the CPU emulation of the kernels with the gloo backend")
    ap.add_argument("--cpu-worker", default=None, help="INTERNAL: one pinned channel group of cpu_baseline (a JSON spec); no GPU is touched")
    return ap.parse_args()


def _cpu_once(u, k, bias, dout):
    """one forward + backward of the oracle's fftconv_ref on host tensors; returns seconds"""
    from oracle import hyena_oracle as O
    u_ = u.clone().requires_grad_(True)
    k_ = k.clone().requires_grad_(True)
    b_ = bias.clone().requires_grad_(True)
    t0 = time.perf_counter()
    out = O.fftconv_ref(u_, k_, b_, None, gelu=False)
    out.backward(dout)
    return time.perf_counter() - t0


def _cpu_sample(L, Ds, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(1, Ds, L, generator=g).to(dtype)
    k = torch.randn(Ds, L, generator=g) * torch.exp(-5.0 * torch.linspace(0, 1, L))[None] * 0.1
    bias = torch.randn(Ds, generator=g)
    dout = torch.randn(1, Ds, L, generator=g).to(dtype)
    return u, k, bias, dout


def cpu_worker(spec):
    """`python bench.py --cpu-worker '<json>'`: ONE channel group of the host-cores baseline -- pinned to its own cores, the oracle's
    fftconv_ref fwd+bwd on `channels` channels, `reps` timed passes after a warm-up, all groups released together through a `go` file.
    Prints one JSON line {elapsed_s, reps, channels}.  (The oracle is used here as cpu_baseline's measured CPU path, nowhere else.)"""
    spec = json.loads(spec)
    cores = spec["cores"]
    try:
        os.sched_setaffinity(0, set(cores))
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(len(cores))
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[spec["dtype"]]
    ops = _cpu_sample(spec["L"], spec["channels"], dtype, seed=spec["seed"])
    _cpu_once(*ops)                                            # warm-up: FFT plans, thread pool, page faults
    open(spec["ready"], "w").close()
    deadline = time.time() + 120.0
    while not os.path.exists(spec["go"]) and time.time() < deadline:
        time.sleep(0.005)
    t0 = time.perf_counter()
    for _ in range(spec["reps"]):
        _cpu_once(*ops)
    print(json.dumps({"elapsed_s": time.perf_counter() - t0, "reps": spec["reps"], "channels": spec["channels"]}), flush=True)


def _cpu_groups(L, D, dtype_name, t_best, Ds, host, reps, timeout_s):
    """`groups` = host // t_best concurrent worker processes (cpu_worker), each pinned to t_best cores of its own and convolving its own Ds
    channels: the aggregate nt/s of the host with `groups * t_best` cores busy.  None if it cannot be run (memory, spawn failure, timeout)."""
    import subprocess
    import tempfile
    try:
        avail = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = list(range(host))
    groups = len(avail) // t_best
    # memory: a group holds ~Ds rows of u, k, dout, their 2L-point spectra and autograd's copies: ~30 fp32 rows of 2L points per channel
    need = 30 * Ds * 2 * L * 4
    try:
        with open("/proc/meminfo") as f:
            mem_avail = next(int(l.split()[1]) * 1024 for l in f if l.startswith("MemAvailable"))
        for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                         ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
            try:                                                 # a container's own limit can sit far below the host's free memory
                with open(lim) as f1, open(cur) as f2:
                    v = f1.read().strip()
                    if v != "max":
                        mem_avail = min(mem_avail, int(v) - int(f2.read().strip()))
            except (OSError, ValueError):
                pass
        groups = min(groups, int(mem_avail * 0.5 // need))
    except (OSError, StopIteration, ValueError):
        groups = min(groups, 4)
    if groups < 2:
        return None
    tmp = tempfile.mkdtemp(prefix="bench_cpu_")
    go = os.path.join(tmp, "go")
    procs = []
    try:
        for gidx in range(groups):
            spec = {"L": L, "channels": Ds, "dtype": dtype_name, "seed": gidx, "reps": reps, "go": go,
                    "ready": os.path.join(tmp, f"ready{gidx}"), "cores": avail[gidx * t_best:(gidx + 1) * t_best]}
            env = dict(os.environ, OMP_NUM_THREADS=str(t_best), MKL_NUM_THREADS=str(t_best), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", json.dumps(spec)],
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True))
        t_end = time.time() + timeout_s
        while time.time() < t_end and not all(os.path.exists(os.path.join(tmp, f"ready{g}")) for g in range(groups)):
            if any(p.poll() is not None for p in procs):
                return None
            time.sleep(0.02)
        open(go, "w").close()
        outs = []
        for p in procs:
            out, _ = p.communicate(timeout=max(1.0, t_end - time.time()) + 30.0)
            outs.append(json.loads(out.strip().splitlines()[-1]))
        slowest = max(o["elapsed_s"] for o in outs)
        total_channel_passes = sum(o["channels"] * o["reps"] for o in outs)
        return {"groups": groups, "threads_per_group": t_best, "cores": groups * t_best, "channels_per_group": Ds, "reps": reps,
                "slowest_group_s": slowest, "value": L * (total_channel_passes / D) / slowest}
    except Exception:
        return None
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(L, D, dtype, budget_s=20.0):
    """The oracle (reference torch.fft path) fwd+bwd on the host cores, on a bounded sample of the same workload: the same L, B = 1
    (SURVEY.md 8d).  Two stages.  (1) ONE channel group (Ds <= 64 channels; channels are independent) at several thread counts -- {8, 32,
    all cores}, best of 3 each after a warm-up: a 64-row FFT job does not fill a many-core host and oversubscribing it is slower than using
    fewer threads, so this finds the thread count t* one group runs best at.  (2) host_cores // t* such groups AT THE SAME
    TIME, one process each, pinned to disjoint cores, each convolving its own channels, released together: `value` is the
    aggregate -- all groups' channel passes over the slowest group's time -- and `cores` the cores actually busy.  If stage 2 cannot run (one
    group only, too little host memory, a worker failed) the stage-1 figure is reported and `cores` says so."""
    host = os.cpu_count() or 1
    Ds = max(1, min(D, 64, int(4.0e8 / max(L, 1024) / 12 * 8)))
    ops = _cpu_sample(L, Ds, dtype, seed=0)
    t_start = time.perf_counter()
    tried = {}
    saved_threads = torch.get_num_threads()
    for t in sorted({min(8, host), min(32, host), host}):
        if tried and time.perf_counter() - t_start > budget_s:
            break
        torch.set_num_threads(t)
        _cpu_once(*ops)                                                    # warm-up at this thread count
        best = None
        for _ in range(3):
            dt = _cpu_once(*ops)
            best = dt if best is None else min(best, dt)
            if time.perf_counter() - t_start > 1.5 * budget_s:
                break
        tried[t] = best
    torch.set_num_threads(saved_threads)
    t_best = min(tried, key=tried.get)
    best = tried[t_best]
    one_group = L * (Ds / D) / best           # a nucleotide = one position through all D channels
    dtype_name = {torch.bfloat16: "bf16", torch.float16: "fp16", torch.float32: "fp32"}[dtype]
    reps = max(1, min(3, int(6.0 / max(best, 1e-3))))
    multi = _cpu_groups(L, D, dtype_name, t_best, Ds, host, reps, timeout_s=40.0) if host // t_best >= 2 else None
    res = {"value": one_group, "unit": "nt/s", "cores": t_best, "host_cores": host, "kind": "port", "repetitions": 3,
           "by_threads": {str(t): L * (Ds / D) / v for t, v in tried.items()},
           "one_group": {"value": one_group, "threads": t_best, "channels": Ds, "best_s": best},
           "sample": f"oracle fftconv_ref fwd+bwd (torch.fft, fp32 math), B=1, L={L}, {Ds} of {D} channels, timed at "
                     f"{sorted(tried)} threads (best of 3 after a warm-up each); best {best * 1e3:.0f} ms at {t_best} threads on a "
                     f"{host}-core host; scaled by {Ds}/{D} channels"}
    if multi is not None:
        res.update(value=multi["value"], cores=multi["cores"], concurrent=multi,
                   sample=f"oracle fftconv_ref fwd+bwd (torch.fft, fp32 math), B=1, L={L}: {multi['groups']} channel groups of {Ds} channels "
                          f"at the same time, one process each pinned to {t_best} cores of its own ({multi['cores']} of {host} host cores busy), "
                          f"{multi['reps']} passes per group after a warm-up, released together; all groups' channel passes / the slowest "
                          f"group's {multi['slowest_group_s']:.2f} s, scaled to {D} channels.  One group alone (the thread count was chosen on it, "
                          f"best of {sorted(tried)} threads): {one_group:.0f} nt/s")
    return res


def operator_algorithmic_bytes(B, D, L, s):
    """What ONE HyenaOperator layer has to move, forward + backward, if every tensor crossed HBM once (the roofline a whole layer is
    judged against): SURVEY.md 8d's secondary boundary for the mixer core between the projections, 11 B L D s + 12 D L, plus the
    projections' own I/O at the layer boundary -- forward: read u, write out (2 B L D s); backward: read dout, read u again for
    in_proj's weight gradient, write du (3 B L D s) -- plus the layer's weights and their gradients (negligible: 4 D^2 * 2 s)."""
    return 11 * B * L * D * s + 12 * D * L + 5 * B * L * D * s + 16 * D * D * s


def _timed_steps(step, steps, dev):
    """per-step HIP-event times (ms) of `steps` back-to-back calls: events between the steps, one synchronisation at the end"""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize(dev)
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]


def _without_stalls(times):
    """(mean of the steps that are not stalls, [(step, ms) of the stalls]): a step more than three times the median long is a stall -- the secondary legs
    build and drop models of up to 200 GiB of reserved memory in one process, and the driver's unmapping of a released pool has been seen to hold one
    later step for 3 - 6 s (profiles/r6af_filter_side_stream.txt: never in a steady training loop, 0 of 290 + 1141 steps).  The stalls are listed, not hidden."""
    med = _median(times)
    stalls = [(i, round(t, 1)) for i, t in enumerate(times) if t > 3.0 * med]
    kept = [t for t in times if t <= 3.0 * med]
    return sum(kept) / len(kept), stalls


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def _release_and_settle(dev, budget_s=12.0):
    """Between secondary legs: hand the previous leg's cached blocks back to the device and WAIT until the device is quiet again.  Releasing ~200 GiB
    (the 2^20 model step reserves that much since the filter's kernels run on a second stream with an allocator pool of its own) keeps the driver
    busy unmapping for several seconds AFTER empty_cache() has returned, and kernels issued meanwhile run late: one 5.7 s step landed in the timed
    region of whichever leg came next (profiles/r6af_filter_side_stream.txt).  A small timed copy is repeated until three consecutive runs take
    less than twice the fastest one seen."""
    import gc
    gc.collect()
    torch.cuda.synchronize(dev)
    before = torch.cuda.memory_reserved(dev)
    torch.cuda.empty_cache()
    torch.cuda.synchronize(dev)
    released = before - torch.cuda.memory_reserved(dev)
    probe = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    # the hold comes a few hundred ms AFTER a big release (seen at step 11 - 17 of the next leg, always ~5.7 s): after one, keep the probe running for 8 s
    # so that it meets the probe and not a timed step
    t_min = time.perf_counter() + (8.0 if released > (32 << 30) else 0.0)
    best, calm, t_end = None, 0, time.perf_counter() + budget_s
    while time.perf_counter() < t_end and (calm < 3 or time.perf_counter() < t_min):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        probe.fill_(1)
        b.record()
        torch.cuda.synchronize(dev)
        t = a.elapsed_time(b)
        best = t if best is None else min(best, t)
        calm = calm + 1 if t < 2.0 * best + 0.05 else 0
    del probe


def operator_layer(L, D, B, dtype, dev, steps=20, warmup=4, graph=False, order=2):
    """Secondary figure (not `value`): one whole HyenaOperator layer -- in_proj, short conv, gates, implicit filter, long
    conv, out_proj -- forward + backward under autocast, HyenaDNA configuration (hg38_hyena.yaml:20-30), random init.
    `ms_per_step` is the mean of `steps` steps, with the minimum and the median beside it (box noise is +- 4 %: VERDICT r4 item 4),
    and a roofline against operator_algorithmic_bytes.  graph: the step is captured into one hipGraph and replayed (shapes whose eager
    step is bound by Python issuing its ~60 launches, not by the GPU); falls back to eager calls if the capture fails."""
    from hyena_dna_amd.hyena import HyenaOperator
    _release_and_settle(dev)
    torch.manual_seed(0)
    op = HyenaOperator(d_model=D, l_max=L + 2, order=order, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10,
                       lr=6e-4, wd=0.0, lr_pos_emb=0.0).to(dev)
    u = torch.randn(B, L, D, device=dev, dtype=dtype, requires_grad=True)
    dy = torch.randn(B, L, D, device=dev, dtype=dtype)

    def step():
        op.zero_grad(set_to_none=True)
        u.grad = None
        with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            y = op(u)
        y.backward(dy)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    run, graphed = step, False
    if graph:
        try:
            import gc
            gc.collect()
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):                   # this stream's workspaces / keep-the-spectra decisions exist before the capture
                    step()
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    step()
            torch.cuda.current_stream(dev).wait_stream(side)
            for _ in range(warmup):
                g.replay()
            torch.cuda.synchronize(dev)
            run, graphed = g.replay, True
        except Exception:                            # secondary figure: the eager loop is always available
            torch.cuda.synchronize(dev)
            run, graphed = step, False
    times = _timed_steps(run, steps, dev)
    ms, stalls = _without_stalls(times)
    s = 4 if dtype == torch.float32 else 2
    best = min(times)
    if order != 2:             # (no roofline model for the deeper recurrence: the leg reports times and the route taken)
        return {"ms_per_step": ms, "min_ms": best, "median_ms": _median(times), "stalled_steps": stalls, "value": B * L / ms * 1e3, "unit": "nt/s", "steps": steps, "order": order,
                "route": op._route(L), "workload": f"one HyenaOperator layer of order {order} (configs/model/layer/hyena_dna.yaml) fwd+bwd, L={L}, d={D}, B={B}, "
                                                   f"{str(dtype).split('.')[-1]} autocast; secondary figure"}
    abytes = operator_algorithmic_bytes(B, D, L, s)
    return {"ms_per_step": ms, "min_ms": best, "median_ms": _median(times), "stalled_steps": stalls, "value": B * L / ms * 1e3, "unit": "nt/s", "steps": steps,
            "hipgraph_replay": graphed,
            "roofline": {"bound": "hbm", "achieved": abytes / (best * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": abytes / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": abytes,
                         "of": "min_ms", "boundary": "SURVEY 8d secondary (mixer core, 11 B L D s + 12 D L) + the projections' layer I/O "
                                                     "(5 B L D s) + weights: every tensor of the layer boundary once"},
            "workload": f"one HyenaOperator layer fwd+bwd (projections + short conv + gates + implicit filter + long conv), "
                        f"L={L}, d={D}, B={B}, {str(dtype).split('.')[-1]} autocast; secondary figure, not `value`"}


def model_step(L, D, B, dtype, dev, rank=0, world=1, n_layer=8, steps=8, warmup=3, emu=False, graphed_ok=True):
    """Secondary figure (not `value`): the full hyenadna pre-training step of north_star configuration 5 on synthetic tokens --
    embedding -> n_layer x [add+LayerNorm -> HyenaOperator -> add+LayerNorm -> MLP (d -> 4d -> d, tanh-GELU)] -> LayerNorm ->
    tied LM head -> cross entropy, backward, AdamW step -- random init, autocast (hg38_hyena.yaml: d_model 256, n_layer 8,
    d_inner 1024, vocab 12 padded to 16)."""
    from hyena_dna_amd.lm import HyenaDNALM, token_cross_entropy
    if not emu:
        _release_and_settle(dev)
    torch.manual_seed(0)
    layer = dict(l_max=L + 2, order=2, filter_order=64, emb_dim=5, short_filter_order=3, modulate=True, w=10, lr=6e-4, wd=0.0,
                 lr_pos_emb=0.0)
    model = HyenaDNALM(d_model=D, n_layer=n_layer, d_inner=4 * D, vocab_size=12, layer=layer, resid_dropout=0.0, embed_dropout=0.1,
                       pad_vocab_size_multiple=8, fused_dropout_add_ln=True, residual_in_fp32=True).to(dev)
    net = model
    if world > 1:
        # train.py:611-620 (DDPStrategy(find_unused_parameters=False, gradient_as_bucket_view=True)): identical replicas
        # (same seed above), gradients averaged over the ranks bucket by bucket while the backward still runs
        from torch.nn.parallel import DistributedDataParallel
        net = DistributedDataParallel(model, device_ids=None if emu else [dev.index], find_unused_parameters=False,
                                      gradient_as_bucket_view=True)
    opt = torch.optim.AdamW(model.parameters(), lr=6e-4, weight_decay=0.1, betas=(0.9, 0.999))
    g = torch.Generator(device=dev).manual_seed(2222 + rank)
    ids = torch.randint(7, 11, (B, L), generator=g, device=dev)           # A, C, G, T (hg38_char_tokenizer.py:59-66)
    tgt = torch.roll(ids, -1, 1)
    dev_type = "cpu" if emu else "cuda"

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast(dev_type, dtype=dtype, enabled=dtype != torch.float32 and not emu):   # (the CPU test double runs fp32)
            logits = net(ids)[0].logits              # through DDP's forward, so that its reducer is armed for the backward
            loss = token_cross_entropy(logits, tgt)           # = F.cross_entropy over the flattened logits (lm.py)
        loss.backward()
        opt.step()
        return loss

    def sync():
        if not emu:
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        if not emu:
            torch.cuda.synchronize(dev)

    for _ in range(warmup):
        step()
    sync()
    ev = None if emu else [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    if ev:
        ev[0].record()
    for i in range(steps):
        loss = step()
        if ev:
            ev[i + 1].record()
    sync()
    wall = time.perf_counter() - t0
    per_step = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)] if ev else [wall * 1e3 / steps] * steps
    if world > 1:                                    # the step ends when the slowest rank's does
        tm = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        wall = tm.item()
    ms = wall * 1e3 / steps
    stalls = []
    if ev and world == 1:
        # one GPU: the mean of the per-step event times without stalled steps (listed in `stalled_steps`; `wall_ms_per_step` keeps the raw figure).  N > 1
        # keeps the barrier-bracketed wall clock, max over ranks, as the contract asks.
        ms, stalls = _without_stalls(per_step)
    loss = float(loss.detach())                              # (also drops the last autograd graph before the capture below)
    graphed = None
    if L <= 65536 and world == 1 and graphed_ok and not emu:
        try:
            # launch-bound regime: the same step captured into one hipGraph (lm.GraphedTrainStep) and replayed
            from hyena_dna_amd.lm import GraphedTrainStep
            del opt
            opt_g = torch.optim.AdamW(model.parameters(), lr=6e-4, weight_decay=0.1, betas=(0.9, 0.999), capturable=True)
            gstep = GraphedTrainStep(model, opt_g, ids, tgt, autocast_dtype=dtype, warmup=2)
            for _ in range(2):
                gstep()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = max(steps, 10)
            e0.record()
            for _ in range(n):
                gl = gstep()
            e1.record()
            torch.cuda.synchronize(dev)
            gms = e0.elapsed_time(e1) / n
            graphed = {"ms_per_step": gms, "value": B * L / gms * 1e3, "unit": "nt/s", "steps": n, "loss": float(gl),
                       "how": "forward + loss + backward + AdamW captured into one hipGraph (hyena_dna_amd.lm.GraphedTrainStep)"}
        except Exception as e:                                     # never lose the eager figure over the capture
            graphed = {"error": repr(e)[:200]}
    return {"ms_per_step": ms, "median_ms": _median(per_step), "min_ms": min(per_step), "wall_ms_per_step": wall * 1e3 / steps, "stalled_steps": stalls,
            "value": B * L * world / ms * 1e3, "unit": "nt/s", "n_gpus": world, "steps": steps, "loss": float(loss),
            "graphed": graphed, "params": sum(p.numel() for p in model.parameters()),
            "peak_mem_GB": None if emu else torch.cuda.max_memory_allocated(dev) / 2 ** 30,
            "parallelism": "single GPU" if world == 1 else
                           f"dp{world}: torch DDP (find_unused_parameters=False, gradient_as_bucket_view=True), one sequence batch per "
                           f"rank, gradients all-reduced in DDP's buckets over {'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend() + ' (test)'}",
            "workload": f"full model step (fwd + bwd + AdamW): hyenadna d_model={D}, n_layer={n_layer}, d_inner={4 * D}, L={L}, B={B}/GPU, "
                        f"{str(dtype).split('.')[-1]} autocast, synthetic tokens; whole-job nt/s, barrier-bracketed, max over ranks; "
                        f"secondary figure, not `value`"}


def make_conv_step(L, B, D, dtype, dev, seed, chunk=None, save=True, fwd_only=False, pitched=True):
    """Synthetic operands resident in HBM + the step closure: what hyena_dna_amd.fftconv.FFTConvFunc does per layer call -- forward
    (keeping its spectra for the backward unless save is off), then the backward for a given upstream gradient.
    pitched: the operands are laid out as the operator's fused path lays them out (hyena_dna_amd._lib.empty_rows: rows 64 elements apart
    -- identical to the packed layout whenever L is a multiple of 64, i.e. at every aligned length); False: packed rows, what a caller of
    the bare op seam (fftconv_func on its own contiguous tensors) gets."""
    from hyena_dna_amd import _lib
    g = torch.Generator(device=dev).manual_seed(seed)

    def rows(t):
        if not pitched:
            return t
        r = _lib.empty_rows(t.shape[:-1], t.shape[-1], t.dtype, t.device)
        r.copy_(t)
        return r

    u = rows(torch.randn(B, D, L, generator=g, device=dev).to(dtype))
    k = rows(torch.randn(D, L, generator=g, device=dev) * torch.exp(-5.0 * torch.linspace(0, 1, L, device=dev))[None] * 0.1)
    bias = torch.randn(D, generator=g, device=dev)
    dout = rows(torch.randn(B, D, L, generator=g, device=dev).to(dtype))

    def step():
        if save:
            out, saved = _lib.fftconv_fwd(u, k, bias, chunk=chunk, save=True)
        else:
            out, saved = _lib.fftconv_fwd(u, k, bias, chunk=chunk), None
        if fwd_only:
            return out
        return _lib.fftconv_bwd(dout, u, k, bias, chunk=chunk, saved=saved)

    return step


def maybe_graph(step, L, B, D, dtype, dev, warmup, enabled):
    """Launch-bound sizes (a hyenadna-tiny-1k layer call is ~15 us of GPU work behind ~35 us of Python + ctypes per call): the step is
    captured into ONE hipGraph and replayed -- the same launches on the same buffers, issued by the runtime instead of the interpreter
    (what lm.GraphedTrainStep does for the whole training step)."""
    if not enabled or algorithmic_bytes(B, D, L, 2 if dtype != torch.float32 else 4) > 64 * 2 ** 20:
        return step, False
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):                       # this stream's workspace and the allocator's blocks exist before the capture
            step()
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            step()
    torch.cuda.current_stream(dev).wait_stream(side)
    for _ in range(warmup):
        graph.replay()
    return graph.replay, True


def conv_rooflines(L, B, D, dtype_name, save, ev_ms_step):
    """Both rooflines of SURVEY.md 8d for one fftconv fwd+bwd step that took ev_ms_step (HIP events), plus the floor this plan can reach
    on this part with its one-line derivation."""
    from hyena_dna_amd import _lib
    s = 4 if dtype_name == "fp32" else 2
    abytes = algorithmic_bytes(B, D, L, s)
    achieved = abytes / (ev_ms_step * 1e-3) / 1e9
    M = int(_lib.lib().hyena_fftconv_fft_size(L))
    onchip = int(_lib.lib().hyena_fftconv_plan(L)) == _lib.PLAN_ONCHIP
    plan = "onchip" if onchip else "twolevel"
    aflops = algorithmic_flops(B, D, L, M)
    tflops = aflops / (ev_ms_step * 1e-3) / 1e12
    lds_b = lds_exchange_bytes(B, D, M)
    traffic, src = measured_traffic(L, D, B, dtype_name, save, plan)
    if onchip:
        # the row never leaves the CU: traffic is the algorithmic bytes, the kernels are bound by VALU issue.  Floor = the transforms'
        # nominal instruction count at the fp32 issue rate the part sustains (v_fma_f32: 117 of the 157 TF, profiles/valu_rate_r2.txt)
        floor_ms = aflops / (117.0e12) * 1e3
        floor_how = "VALU issue: algorithmic flops / 117 TF (measured v_fma_f32 rate, profiles/valu_rate_r2.txt); the row stays on chip"
    elif traffic is not None:
        floor_ms = traffic / (MIXED_STREAM_TBS * 1e12) * 1e3
        floor_how = (f"two-pass exact-fp32 transform: measured HBM traffic {traffic / 1e9:.2f} GB (six transform units x 2 crossings of 8 M B "
                     f"+ saved spectra + I/O = 128 M B per row) / {MIXED_STREAM_TBS} TB/s, the best mixed read+write rate measured on this part "
                     f"(profiles/cpol_bw_r2.txt); DESIGN.md section 5")
    else:
        floor_ms, floor_how = None, "no PMC traffic record for this configuration"
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src,
            "floor_frac": None if floor_ms is None else abytes / (floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "floor_ms": floor_ms, "floor_derivation": floor_how,
            "kernel": "all launches of one fftconv fwd+bwd step (" +
                      ("spec / conv / dk kernels of the workspace-free plan)" if onchip
                       else "col_fwd / row_* / col_inv chain of the two-level plan)"),
            "algorithmic_bytes_per_step": abytes, "event_ms_per_step": ev_ms_step}
    # the second roofline of SURVEY.md 8d: the fused op sits at / above the fp32 ridge, so VALU (and LDS) co-bind
    valu = {"bound": "valu_fp32", "achieved": tflops, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / VALU_PEAK_TFLOPS,
            "algorithmic_flops_per_step": aflops, "fft_points": M, "transforms_per_step": (5 * B + 2) * D,
            "lds_bytes_per_step": lds_b, "lds_achieved_TBs": lds_b / (ev_ms_step * 1e-3) / 1e12, "lds_peak_TBs": LDS_PEAK_TBS}
    return roof, valu


def sweep_configs(dtype, dtype_name, dev, steps, warmup, graph_ok, emu=False, configs=None, pitched=True):
    """The other BASELINE.json configurations through the same timed loop as the headline (HIP events around `steps` steps after
    `warmup`, operands resident): {ms_per_step, nt/s, both roofline fractions, floor, PMC traffic} each.  N = 1 only."""
    res = []
    for (L, B, D) in (configs or SWEEP):
        try:
            step = make_conv_step(L, B, D, dtype, dev, seed=2222, pitched=pitched)
            for _ in range(warmup):
                step()
            run, graphed = (step, False) if emu else maybe_graph(step, L, B, D, dtype, dev, warmup, graph_ok)
            if emu:
                t0 = time.perf_counter()
                for _ in range(steps):
                    run()
                ms = (time.perf_counter() - t0) * 1e3 / steps
            else:
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(steps):
                    run()
                e1.record()
                torch.cuda.synchronize(dev)
                ms = e0.elapsed_time(e1) / steps
            roof, valu = conv_rooflines(L, B, D, dtype_name, True, ms)
            res.append({"seq_len": L, "batch_per_gpu": B, "channels": D, "io_dtype": dtype_name, "steps": steps, "warmup": warmup,
                        "hipgraph_replay": bool(graphed), "ms_per_step": ms, "value": B * L / ms * 1e3, "unit": "nt/s",
                        "frac": roof["frac"], "valu_frac": valu["frac"], "floor_frac": roof["floor_frac"],
                        "traffic": roof["traffic"], "traffic_source": roof["traffic_source"],
                        "algorithmic_bytes_per_step": roof["algorithmic_bytes_per_step"]})
            del step, run
            if not emu:
                torch.cuda.empty_cache()
        except Exception as e:                                       # a secondary figure never costs the contract line
            res.append({"seq_len": L, "batch_per_gpu": B, "channels": D, "error": repr(e)[:200]})
    return res


def sweep_real_shapes(dtype, dtype_name, dev, steps, warmup, graph_ok):
    """The sequence lengths the reference's trainer really produces (L = max_length - 1, hg38_dataset.py:220-223), each next to its aligned
    neighbour through the same loop: {ms_per_step, value, frac} + `aligned` {seq_len, ms_per_step} + `vs_aligned` = real / aligned time.
    `ms_per_step` is measured on PITCHED rows -- the layout the operator's fused path hands the convolution (rows 64 elements apart,
    hyena_dna_amd._lib.row_pitch) --, `packed_ms` on packed (B, D, L) tensors, what a caller of the bare op seam passes.  Every leg is measured
    twice, interleaved, and the better time is kept (the comparison is about a 2 % difference)."""
    res = []
    for (L, B, D, La) in SWEEP_REAL:
        try:
            best = {}
            for rep in range(2):
                for key, length, pitched in (("real", L, True), ("aligned", La, True), ("packed", L, False)):
                    r = sweep_configs(dtype, dtype_name, dev, steps, warmup, graph_ok, configs=[(length, B, D)], pitched=pitched)[0]
                    if "error" in r:
                        raise RuntimeError(r["error"])
                    if key not in best or r["ms_per_step"] < best[key]["ms_per_step"]:
                        best[key] = r
            r = {k: best["real"][k] for k in ("seq_len", "batch_per_gpu", "channels", "io_dtype", "steps", "warmup", "hipgraph_replay",
                                              "ms_per_step", "value", "unit", "frac", "valu_frac", "algorithmic_bytes_per_step")}
            from hyena_dna_amd import _lib
            r["row_pitch"] = _lib.row_pitch(L)
            r["aligned"] = {"seq_len": La, "ms_per_step": best["aligned"]["ms_per_step"]}
            r["vs_aligned"] = best["real"]["ms_per_step"] / best["aligned"]["ms_per_step"]
            r["packed_ms"] = best["packed"]["ms_per_step"]
            r["packed_vs_aligned"] = best["packed"]["ms_per_step"] / best["aligned"]["ms_per_step"]
            res.append(r)
        except Exception as e:
            res.append({"seq_len": L, "batch_per_gpu": B, "channels": D, "error": repr(e)[:200]})
    return res


def real_shape_legs(dtype, dev, no_operator=False, no_model=False):
    """The shipped experiments' own shapes at the operator and the model level (REAL_SHAPES), each next to its aligned neighbour through the
    same loops: `operator_layer` {real, aligned, vs_aligned} and `model_step` {real, aligned, vs_aligned} per shape.  Shapes whose eager step is
    bound by the host issuing launches (L < 65536) are ALSO timed as hipGraph replays, and `vs_aligned` is then taken on the replayed (GPU) time --
    the misalignment these legs exist to expose is GPU time; the eager ratio is reported beside it."""
    legs = []
    for (L, B, D, n_layer, La) in REAL_SHAPES:
        leg = {"seq_len": L, "batch_per_gpu": B, "d_model": D, "n_layer": n_layer, "aligned_seq_len": La,
               "source": "configs/experiment/hg38/hg38_hyena.yaml:47-48 + hg38_dataset.py:222 (L = max_length - 1)"}
        launch_bound = L < 65536
        if not no_operator:
            try:
                r = {}
                for key, length in (("real", L), ("aligned", La)):
                    best = None
                    for _ in range(2):               # interleaved twice, the better kept: the comparison is about a 2 % difference
                        o = operator_layer(length, D, B, dtype, dev, steps=20, warmup=3, graph=launch_bound)
                        best = o if best is None or o["min_ms"] < best["min_ms"] else best
                    r[key] = {k: best[k] for k in ("ms_per_step", "min_ms", "median_ms", "hipgraph_replay")}
                    torch.cuda.empty_cache()
                r["vs_aligned"] = {"min": r["real"]["min_ms"] / r["aligned"]["min_ms"],
                                   "median": r["real"]["median_ms"] / r["aligned"]["median_ms"]}
                leg["operator_layer"] = r
            except Exception as e:
                leg["operator_layer"] = {"error": repr(e)[:200]}
        if not no_model:
            try:
                r = {}
                for key, length in (("real", L), ("aligned", La)):
                    torch.cuda.empty_cache()
                    m = model_step(length, D, B, dtype, dev, n_layer=n_layer, steps=8, warmup=2, graphed_ok=launch_bound)
                    r[key] = {k: m.get(k) for k in ("ms_per_step", "min_ms", "median_ms", "value", "loss", "peak_mem_GB")}
                    gr = m.get("graphed") or {}
                    r[key]["graphed_ms"] = gr.get("ms_per_step")
                    if "error" in gr:
                        r[key]["graphed_error"] = gr["error"]
                r["vs_aligned"] = {"min": r["real"]["min_ms"] / r["aligned"]["min_ms"],
                                   "median": r["real"]["median_ms"] / r["aligned"]["median_ms"]}
                if r["real"]["graphed_ms"] and r["aligned"]["graphed_ms"]:
                    r["vs_aligned"]["graphed"] = r["real"]["graphed_ms"] / r["aligned"]["graphed_ms"]
                leg["model_step"] = r
            except Exception as e:
                leg["model_step"] = {"error": repr(e)[:300]}
        legs.append(leg)
        torch.cuda.empty_cache()
    return legs


def _rccl_log_summary(path):
    """RCCL's INFO log of this rank, reduced to what identifies the job: the version line, and how many channels went over which transport."""
    import re
    out = {"file": path, "version_line": None, "transports": {}, "sample": []}
    try:
        with open(path, errors="replace") as f:
            for ln in f:
                ln = ln.rstrip()
                if out["version_line"] is None and re.search(r"(RCCL|NCCL) version", ln):
                    out["version_line"] = ln[-160:]
                m = re.search(r"\bvia (\S+)", ln)
                if m and "Channel" in ln:
                    key = m.group(1)
                    out["transports"][key] = out["transports"].get(key, 0) + 1
                    if len(out["sample"]) < 3:
                        out["sample"].append(ln[-200:])
                elif len(out["sample"]) < 6 and re.search(r"XGMI|xgmi|PCIe|comm 0x.* rank .* nranks", ln):
                    out["sample"].append(ln[-200:])
    except OSError as e:
        out["error"] = repr(e)[:120]
    return out


def _gpu_link_types():
    """rocm-smi's link type between every pair of GPUs of the node ({"XGMI": n, "PCIE": m}), or the reason it is not available"""
    import re
    import subprocess
    try:
        txt = subprocess.run(["rocm-smi", "--showtopotype"], capture_output=True, text=True, timeout=20).stdout
        kinds = {}
        for k in re.findall(r"\b(XGMI|PCIE)\b", txt):
            kinds[k] = kinds.get(k, 0) + 1
        return kinds or {"unparsed": txt[-200:]}
    except Exception as e:
        return {"error": repr(e)[:120]}


def dist_diagnostics(dev, rank, world, ms_max, ms_local, args, nccl_log):
    """Collective on every rank (world > 1): each rank's own ms/step of the timed region gathered to everybody, and one gradient-sized
    all_reduce -- 6.6 M fp32 = 26 MB, hyenadna-large-1m's gradients (train.py:611-620: what DDP's buckets carry) -- timed OUTSIDE the timed
    region: 2 warm-up calls, then 5 calls between barriers, max over ranks.  Rank 0 adds the backend, the library version, the visible
    devices, the node's GPU link types and RCCL's own transport lines."""
    cpu = args.emu
    t = torch.tensor([ms_local], dtype=torch.float64, device="cpu" if cpu else dev)
    per_rank = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(per_rank, t)
    n = 6_600_000 if not cpu else 66_000
    buf = torch.ones(n, dtype=torch.float32, device="cpu" if cpu else dev)
    for _ in range(2):
        dist.all_reduce(buf)
        buf.fill_(1.0)
    if not cpu:
        torch.cuda.synchronize(dev)
    dist.barrier()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(buf)
    if not cpu:
        torch.cuda.synchronize(dev)
    probe = torch.tensor([(time.perf_counter() - t0) * 1e3 / reps], dtype=torch.float64, device="cpu" if cpu else dev)
    dist.all_reduce(probe, op=dist.ReduceOp.MAX)
    ok = bool(abs(float(buf[0]) - float(world) ** reps) < 1e-3 * float(world) ** reps)        # the sum really covered `world` ranks
    if rank != 0:
        return None
    backend = dist.get_backend()
    info = {"backend": backend, "is_rccl": backend == "nccl" and getattr(torch.version, "hip", None) is not None,
            "world_size": dist.get_world_size(), "devices_visible": 0 if cpu else torch.cuda.device_count(),
            "device_name": None if cpu else torch.cuda.get_device_name(dev),
            "nccl_version": None, "per_rank_ms": [float(x) for x in per_rank], "max_rank_ms": ms_max,
            "allreduce_probe_ms": float(probe), "allreduce_probe_bytes": n * 4, "allreduce_probe_sum_ok": ok,
            "allreduce_probe_busbw_GBs": 2 * (world - 1) / world * n * 4 / (float(probe) * 1e-3) / 1e9}
    if backend == "nccl":
        try:
            info["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:
            info["nccl_version"] = repr(e)[:80]
        info["gpu_link_types"] = _gpu_link_types()
        if nccl_log:
            info["rccl_log"] = _rccl_log_summary(nccl_log)
    return info


def main():
    args = parse()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    from hyena_dna_amd import _lib
    if args.emu:
        from tests.hipemu.emu_backend import EmuBackend       # test double, CPU only
        _lib._backend = EmuBackend()
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a ROCm device (there is no CPU fallback)"
        if args.share_gpu0:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    nccl_log = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        cpu_backend = args.emu or args.share_gpu0
        if not cpu_backend and "NCCL_DEBUG" not in os.environ:
            # RCCL's own account of the job -- version, ranks, the transport of every channel (P2P over xGMI, SHM or NET) -- goes to a per-rank
            # file (stdout carries exactly ONE line, the JSON); rank 0 summarises its file into `dist.rccl_log` below
            import tempfile
            nccl_log = os.path.join(tempfile.gettempdir(), f"bench_rccl_{os.getpid()}_r{rank}.log")
            os.environ["NCCL_DEBUG"] = "INFO"
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,ENV")
            os.environ["NCCL_DEBUG_FILE"] = nccl_log
        dist.init_process_group("gloo" if cpu_backend else "nccl", rank=rank, world_size=world,
                                **({} if cpu_backend else {"device_id": dev}))

    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    B, D, L = args.batch, args.d_model, args.seq_len
    chunk = args.chunk if args.chunk > 0 else None
    save = not args.no_save_spectra and not args.fwd_only
    step = make_conv_step(L, B, D, dtype, dev, seed=2222 + rank, chunk=chunk, save=save, fwd_only=args.fwd_only)   # 2222 = the reference's train seed

    def sync():
        if not args.emu:
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        if not args.emu:
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    # --no-graph times the eager calls at the launch-bound sizes too
    run_step, graphed = (step, False) if args.emu else maybe_graph(step, L, B, D, dtype, dev, args.warmup, not args.no_graph)
    sync()
    if not args.emu:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    if not args.emu:
        e1.record()
    sync()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1) if not args.emu else wall * 1e3
    local_wall = wall
    tmax = torch.tensor([wall], dtype=torch.float64, device=dev if not args.emu else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = tmax.item()

    dist_info = None
    if world > 1:
        try:
            dist_info = dist_diagnostics(dev, rank, world, wall * 1e3 / args.steps, local_wall * 1e3 / args.steps, args, nccl_log)
        except Exception as e:                                       # diagnostics never cost the contract line
            dist_info = {"error": repr(e)[:300]}

    sweep = sweep_extra = None
    if world == 1 and not args.no_sweep and not args.fwd_only:
        # the four other contract configurations, same loop (rank 0 of an N = 1 run only: a sweep is not part of the scaling legs)
        sweep = sweep_configs(dtype, args.dtype, dev, max(args.steps, 20) if not args.emu else 1, args.warmup if not args.emu else 0,
                              not args.no_graph, emu=args.emu, configs=[(256, 2, 64)] if args.emu else None)
        if not args.emu:
            sweep_extra = sweep_configs(dtype, args.dtype, dev, max(args.steps, 20), args.warmup, not args.no_graph, configs=SWEEP_EXTRA)

    sweep_real = None
    if world == 1 and not args.no_sweep and not args.fwd_only and not args.emu:
        sweep_real = sweep_real_shapes(dtype, args.dtype, dev, max(args.steps, 20), args.warmup, not args.no_graph)

    model_res = None
    if not args.no_model and not args.fwd_only and (world > 1 or not args.emu):
        # every rank takes part (DDP's collectives); a failure must not lose the contract line, nor leave the other ranks
        # inside a collective: the exception is reported and every rank then moves on to the final barrier
        try:
            model_res = model_step(L, D, B, dtype, dev, rank=rank, world=world, n_layer=args.model_layers, emu=args.emu)
        except Exception as e:
            model_res = {"error": repr(e)[:300]}

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        nt_per_s = B * L * world / (wall / args.steps)
        ev_ms_step = ev_ms / args.steps
        roof, valu = conv_rooflines(L, B, D, args.dtype, save, ev_ms_step)
        line = {
            "metric": METRIC, "value": nt_per_s, "unit": "nt/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Hyena long-conv layer call (fftconv fwd+bwd), L={L}, d={D}, B={B}/GPU, "
                                   f"{args.dtype} activations, fp32 filter and FFT math" +
                                   (" [FWD ONLY -- diagnostic]" if args.fwd_only else ""),
                       "seq_len": L, "channels": D, "batch_per_gpu": B, "io_dtype": args.dtype,
                       "save_spectra": bool(save), "hipgraph_replay": bool(graphed),
                       "chunk": int(_lib.lib().hyena_fftconv_default_chunk(B, D, L, 1)) if chunk is None else chunk,
                       "parallelism": f"dp{world} (batch-sharded replicas: the convolution has no exchange step, so no collective in the "
                                      f"timed region; the DDP gradient all-reduce is measured in `model_step`)"
                                      if world > 1 else "single GPU",
                       "unit_of_work": "one nucleotide through one Hyena long-conv layer call (fwd+bwd), all d channels"},
            # N > 1: what lets a reader of SCALE_r*.json see that N ranks really met over RCCL (backend, library version, devices, every rank's own
            # step time, one timed 26 MB gradient-sized all_reduce outside the timed region, RCCL's transport lines); null at N = 1
            "dist": dist_info,
            "roofline": roof,
            "roofline_valu": valu,
            # the other four BASELINE.json configurations through the same loop (N = 1 runs; null otherwise)
            "sweep": sweep,
            "sweep_extra": sweep_extra,
            # the lengths the reference's trainer really produces (L = max_length - 1), each next to its aligned neighbour
            "sweep_real": sweep_real,
        }
        if world == 1 and not args.emu and not args.no_operator and not args.fwd_only:
            try:
                line["operator_layer"] = operator_layer(L, D, B, dtype, dev)
            except Exception as e:                                   # secondary: never lose the contract line over it
                line["operator_layer"] = {"error": repr(e)[:200]}
            # order 3 (configs/model/layer/hyena_dna.yaml:3) on the same kernels (mixer.HyenaMixerCMOrderNFunc), next to the op-by-op route it replaces
            try:
                import hyena_dna_amd.hyena as _H
                r3 = operator_layer(L, D, B, dtype, dev, steps=6, warmup=2, order=3)
                saved = _H.ORDER_N_FUSED
                _H.ORDER_N_FUSED = False
                try:
                    g3 = operator_layer(L, D, B, dtype, dev, steps=3, warmup=1, order=3)
                finally:
                    _H.ORDER_N_FUSED = saved
                r3["generic_route"] = {k: g3[k] for k in ("ms_per_step", "min_ms", "median_ms", "route")}
                r3["vs_generic_route"] = g3["min_ms"] / r3["min_ms"]
                line["operator_layer_order3"] = r3
                torch.cuda.empty_cache()
            except Exception as e:
                line["operator_layer_order3"] = {"error": repr(e)[:200]}
        line["model_step"] = model_res
        if world == 1 and not args.emu and not args.no_sweep and not args.fwd_only and L % 2 == 0:
            # the layer and the model at the length the reference's trainer feeds them for this max_length: L - 1
            if not args.no_operator:
                try:
                    r = operator_layer(L - 1, D, B, dtype, dev)
                    if "ms_per_step" in line.get("operator_layer", {}):
                        r["vs_aligned"] = {"min": r["min_ms"] / line["operator_layer"]["min_ms"],
                                           "median": r["median_ms"] / line["operator_layer"]["median_ms"]}
                    line["operator_layer_real"] = r
                except Exception as e:
                    line["operator_layer_real"] = {"error": repr(e)[:200]}
            if not args.no_model:
                try:
                    torch.cuda.empty_cache()
                    r = model_step(L - 1, D, B, dtype, dev, n_layer=args.model_layers, graphed_ok=False)
                    if model_res and "median_ms" in model_res:
                        r["vs_aligned"] = {"min": r["min_ms"] / model_res["min_ms"], "median": r["median_ms"] / model_res["median_ms"]}
                    line["model_step_real"] = r
                except Exception as e:
                    line["model_step_real"] = {"error": repr(e)[:300]}
        if world == 1 and not args.emu and not args.no_sweep and not args.fwd_only and not (args.no_operator and args.no_model):
            # the shipped experiments' own (B, L, d_model, n_layer) at the operator and the model level, next to their aligned neighbours
            try:
                line["real_shapes"] = real_shape_legs(dtype, dev, no_operator=args.no_operator, no_model=args.no_model)
            except Exception as e:
                line["real_shapes"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline and not args.emu:
            line["cpu_baseline"] = cpu_baseline(L, D, dtype)          # rank 0 at N = 1 only (other ranks would idle at the barrier)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
