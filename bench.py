#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native AVSeparationTransformer forward path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): SyntheticAVDataset 2-speaker clips, 1 s @ 8 kHz (F=257, T=63,
N=50 lip frames of 32x32), d_model=256, nhead=4, 2 encoder + 2 fusion layers, batch 32 PER GPU, forward
only, fp32.  A "step" is one forward over one resident batch: inputs are in HBM before the timed region,
outputs stay in HBM.  Clips are independent in eval mode, so N GPUs = N independent
shards of the clip stream, no data-path collective; scaling is weak.

`--mode train` (not the headline) times one training step per "step" instead: train-mode forward
(dropout 0.1, BatchNorm batch statistics) + SeparationLoss + backward + gradient all-reduce (N > 1) + clip + Adam on
BASELINE configs[3]'s model (d=512, 6+4 layers, 3 speakers, 16 clips per GPU).

One JSON line on rank 0:
  value          whole-job clips/s (all ranks' clips / max-over-ranks wall time of the K timed steps)
  roofline       dominant kernel of the path, priced live with HIP events on its own stream
                 (avsep_profile_begin/end, include/avsep.h) against the fp32 matrix peak
  cpu_baseline   the reference's CPU path (oracle/torch_cpu.py port, bit-identical to the reference here)
                 timed on this box's host cores, rank 0, N=1 only, bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "av-separation-transformer_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak (spec)
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA dense peak (spec; 16x the fp32 MFMA's)
HBM_PEAK_GBS = 8000.0


def kernel_pipe(name):
    """(pipe, matrix products executed per algorithmic fp32 product, peak TFLOP/s) of a kernel instance, by its name as
    rocprofv3 prints it.  The two-term fp16 GEMM and attention (csrc/gemm_h2.hip, attention_h2_kernel) execute THREE fp16 MFMA products per fp32 product, the
    three-term bf16 kernels (gemm_split.hip, gemm_planes.hip, attention_split.hip, wgrad_split.hip) SIX bf16 ones -- both on
    the 16-bit matrix pipe (2.5 PFLOP/s dense) --, every other matrix kernel one fp32 MFMA; None: no matrix work (HBM-bound)."""
    if name.startswith(("gemm_h2", "attention_h2")):
        return ("fp16 MFMA, 3 products per fp32 product", 3, BF16_MATRIX_PEAK_TFLOPS)
    if name.startswith(("gemm_split", "gemm_planes", "attention_split", "wgrad_split")):
        return ("bf16 MFMA, 6 products per fp32 product", 6, BF16_MATRIX_PEAK_TFLOPS)
    if name.startswith("conv_stack_h2"):
        return ("fp16 MFMA, 3 products per fp32 product (conv2, conv3)", 3, BF16_MATRIX_PEAK_TFLOPS)
    if name.startswith(("gemm_", "attention", "attn_", "conv_stack", "wgrad")):
        return ("fp32 MFMA", 1, FP32_MATRIX_PEAK_TFLOPS)
    return None


def path_floor(kernels, iters):
    """Seconds per forward that the step's kernels would take at the peak of the pipe each one actually runs on: executed matrix
    flops / that pipe's dense peak for the matrix kernels, algorithmic bytes / HBM peak for the rest.  floor / measured <= 1."""
    t = 0.0
    for k in kernels:
        pipe = kernel_pipe(k["name"])
        t += (k["flops"] * pipe[1] / (pipe[2] * 1e12) if pipe else k["bytes"] / (HBM_PEAK_GBS * 1e9)) / iters
    return t

WORKLOADS = {
    # name: model kwargs, dataset kwargs, per-GPU batch          
    "cfg2": dict(model=dict(freq_bins=257, d_model=256, nhead=4, num_encoder_layers=2, num_fusion_layers=2,
                            num_speakers=2),
                 data=dict(sample_rate=8000, duration=1.0, num_frames=25, frame_h=32, frame_w=32,
                           speaker_freqs=(220.0, 440.0)), batch=32),
    "cfg3": dict(model=dict(freq_bins=257, d_model=512, nhead=8, num_encoder_layers=6, num_fusion_layers=4,
                            num_speakers=2),
                 data=dict(sample_rate=16000, duration=2.0, num_frames=25, frame_h=32, frame_w=32,
                           speaker_freqs=(220.0, 440.0)), batch=64),
    "cfg5": dict(model=dict(freq_bins=257, d_model=512, nhead=8, num_encoder_layers=6, num_fusion_layers=8,
                            num_speakers=2),
                 data=dict(sample_rate=16000, duration=4.0, num_frames=25, frame_h=48, frame_w=48,
                           speaker_freqs=(220.0, 440.0)), batch=32),
    # training workloads (--mode train; SURVEY.md §8(f) N1 / BASELINE configs[3]: 3 speakers, d=512, 16 clips per GPU)
    "cfg4": dict(model=dict(freq_bins=257, d_model=512, nhead=8, num_encoder_layers=6, num_fusion_layers=4,
                            num_speakers=3),
                 data=dict(sample_rate=16000, duration=2.0, num_frames=25, frame_h=32, frame_w=32,
                           speaker_freqs=(220.0, 440.0, 660.0)), batch=16),
}


def flops_per_clip(F, T, N, H, W, d, Le, Lf, S):
    """Algorithmic forward FLOPs per clip (1 MAC = 2 FLOP, elementwise work excluded)."""
    def layer(L):
        return 24 * L * d * d + 4 * L * L * d
    co = lambda x: (x - 1) // 2 + 1  # noqa: E731
    h1, w1 = co(H), co(W)
    h2, w2 = co(h1), co(w1)
    h3, w3 = co(h2), co(w2)
    conv = 2 * N * (h1 * w1 * 32 * 9 + h2 * w2 * 64 * 32 * 9 + h3 * w3 * 128 * 64 * 9) + 2 * N * 128 * d
    return (2 * T * F * d * 3 + 2 * T * d * d * 3 + Le * layer(T) + conv + Le * layer(N) + Lf * layer(T)
            + 2 * T * d * 2 * d + 2 * T * 2 * d * F * S)


def metric_string(mk, dk, steps_in_flight):
    """BASELINE.json's metric, spelled for the workload that ran (the cfg3 / cfg5 lines printed cfg2's)."""
    dur, sr = dk["duration"], dk["sample_rate"] // 1000
    return (f"separated clips/sec ({mk['num_speakers']}-spk, {dur:g}s@{sr}kHz, d={mk['d_model']}) forward, fp32" +
            (f", throughput with {steps_in_flight} steps in flight" if steps_in_flight > 1 else ""))


def shard_range(rank, world, batch):
    """Clip indices of this rank: shard r owns clips [r*batch, (r+1)*batch) of the synthetic stream."""
    return range(rank * batch, (rank + 1) * batch)


def max_over_ranks(dist, seconds, dev):
    """The slowest rank defines the step time (one scalar all-reduce outside the timed region)."""
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_ranks(dist, obj):
    """One python object per rank, on every rank (outside the timed region)."""
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def device_identity(dev):
    """What tells two GPUs of a node apart: name, PCI domain:bus:device (the reliable one on this stack), uuid when the
    torch build exposes a non-degenerate one."""
    p = torch.cuda.get_device_properties(dev)
    pci = None
    if hasattr(p, "pci_bus_id"):
        pci = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}"
    uuid = str(getattr(p, "uuid", "") or "")
    if not uuid.strip("0-") or uuid in ("None", ""):
        uuid = None
    return {"index": dev.index, "name": p.name, "pci": pci, "uuid": uuid, "pid": os.getpid(),
            "visible_devices": torch.cuda.device_count()}


def duplicate_devices(idents):
    """Pairs of ranks that report the same physical device: same PCI address (a rank that landed on a neighbour's GPU
    would otherwise be invisible in the one aggregated number).  Only a real address counts -- a missing or all-zero one
    proves nothing, and the uuid field is informational (runtimes have reported the same uuid for every device)."""
    seen, dup = {}, []
    for r, d in enumerate(idents):
        key = d.get("pci")
        if not key or not key.replace(":", "").strip("0"):
            continue
        if key in seen:
            dup.append((seen[key], r, key))
        else:
            seen[key] = r
    return dup


def timed_rounds(dist, dev, rounds, run_round):
    """R rounds of EXACTLY K steps, each bracketed by barrier + device synchronisation on both sides and reduced with MAX
    over ranks; returns the per-round seconds (max over ranks) and this rank's own per-round seconds.  The median round
    is the reported one: a single 8 ms region right after start-up is one sample of a device still settling its clocks
    (the driver's --steps 20 --warmup 5 run of round 2 read 12 % below a 200-step run of the same build)."""
    mine, worst = [], []
    for _ in range(rounds):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_round()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        mine.append(dt)
        worst.append(max_over_ranks(dist, dt, dev))
    return worst, mine


def median(xs):
    s_ = sorted(xs)
    n = len(s_)
    return s_[n // 2] if n % 2 else 0.5 * (s_[n // 2 - 1] + s_[n // 2])


def spawn_ranks(n, argv):
    """Launcher half of `python bench.py --gpus N` (N > 1, no torchrun around it): run
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    <same arguments>` as a CHILD process -- never an exec, and before this process has touched the GPU -- with stdout /
    stderr inherited, so rank 0's JSON line is the launcher's JSON line.  Returns the child's exit status (non-zero when
    any rank failed: torchrun tears the others down)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: required for RCCL between processes on this stack
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def selftest_launch(rank, world, a):
    """--selftest-launch: everything of the N>1 path except the GPU work, on gloo."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("AVSEP_SELFTEST_FAIL_RANK") == str(rank):      # a rank dying mid-run: the launcher must report it
        os._exit(3)
    dist.barrier()
    elapsed = max_over_ranks(dist, 0.010 * (rank + 1), torch.device("cpu"))     # pretend rank r took 10*(r+1) ms
    clips = list(shard_range(rank, world, 4))
    per_rank = gather_ranks(dist, 10.0 * (rank + 1))
    # pretend devices: distinct unless the test asks for two ranks on one GPU
    same = os.environ.get("AVSEP_SELFTEST_SAME_DEVICE") == "1"
    idents = gather_ranks(dist, {"index": 0 if same else rank, "name": "selftest", "pci": f"0000:{1 if same else rank + 1:02x}:00",
                                 "uuid": None, "pid": os.getpid()})
    dup = duplicate_devices(idents)
    dist.barrier()
    dist.destroy_process_group()
    if dup:
        raise SystemExit(f"ranks share a device: {dup}")
    if rank == 0:
        print(json.dumps({"selftest": "launch", "n_gpus": world, "elapsed_max": elapsed, "rank0_clips": clips,
                          "per_rank_ms": per_rank, "devices": idents, "world_size": world}))


def host_cpu_info():
    """Physical cores / logical CPUs / model name of this box's host CPU (BASELINE.md §3 wants them next to the CPU
    baseline)."""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
            elif k == "model name" and model == "unknown":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
                phys.add((pid, cid))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    return {"cpu_model": model, "host_cores": len(phys) or (os.cpu_count() or 1), "host_logical_cpus": logical or usable,
            "usable_cpus": usable}


def stream_leg(model, ds, B, dev, steps, inflight=2):
    """--stream: the same forward fed like a service would feed it -- DISTINCT batches living in pinned host memory,
    H2D of batch i+1 and D2H of batch i-1 on their own streams under the compute of batch i (double-buffered device
    buffers, one hipGraph per buffer set).  Reported next to the resident-batch `value`, never instead of it."""
    nb = 4                                                              # distinct host batches, cycled
    host_in = []
    for j in range(nb):
        items = [ds[(j * B + i) % len(ds)] for i in range(B)]
        host_in.append((torch.stack([it["mixed_spec"] for it in items]).contiguous().pin_memory(),
                        torch.stack([it["lip_frames"] for it in items]).contiguous().pin_memory()))
    _, F, T = host_in[0][0].shape
    S = model.num_speakers
    sets = []
    for _ in range(inflight):
        sets.append(dict(mixed=torch.empty(host_in[0][0].shape, device=dev), lips=torch.empty(host_in[0][1].shape, device=dev),
                         masks=torch.empty(B, T, S, F, device=dev), sep=torch.empty(B, T, S, F, device=dev),
                         h_masks=torch.empty(B, T, S, F).pin_memory(), h_sep=torch.empty(B, T, S, F).pin_memory(),
                         ev_in=torch.cuda.Event(), ev_done=torch.cuda.Event(), ev_out=torch.cuda.Event()))
    s_in, s_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    s_cmps = [torch.cuda.Stream(device=dev) for _ in range(inflight)]

    def run(n):
        for i in range(n):
            st = sets[i % inflight]
            s_cmp = s_cmps[i % inflight]
            hm, hl = host_in[i % nb]
            with torch.cuda.stream(s_in):
                s_in.wait_event(st["ev_out"])                       # the previous occupant's outputs have left
                st["mixed"].copy_(hm, non_blocking=True)
                st["lips"].copy_(hl, non_blocking=True)
                st["ev_in"].record(s_in)
            with torch.cuda.stream(s_cmp), torch.no_grad():
                s_cmp.wait_event(st["ev_in"])
                model.run_static(st["mixed"], st["lips"], st["masks"], st["sep"], graph=True, slot=i % inflight)
                st["ev_done"].record(s_cmp)
            with torch.cuda.stream(s_out):
                s_out.wait_event(st["ev_done"])
                st["h_masks"].copy_(st["masks"], non_blocking=True)
                st["h_sep"].copy_(st["sep"], non_blocking=True)
                st["ev_out"].record(s_out)
        torch.cuda.synchronize()

    for st in sets:
        st["ev_out"].record(s_out)
    run(2 * inflight + 2)
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    in_mb = sum(t.numel() * 4 for t in host_in[0]) / 1e6
    out_mb = 2 * sets[0]["masks"].numel() * 4 / 1e6
    # the last batch's outputs against a resident re-run of the same batch: the streamed path computes the same bits
    st = sets[(steps - 1) % inflight]
    chk_m, chk_s = torch.empty_like(st["masks"]), torch.empty_like(st["sep"])
    with torch.no_grad():
        model.run_static(st["mixed"], st["lips"], chk_m, chk_s, graph=False, slot=(steps - 1) % inflight)
    torch.cuda.synchronize()
    same = bool(torch.equal(chk_m.cpu(), st["h_masks"]) and torch.equal(chk_s.cpu(), st["h_sep"]))
    return {"value": round(B * steps / dt, 2), "unit": "clips/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
            "distinct_batches": nb, "buffers_in_flight": inflight, "h2d_mb_per_step": round(in_mb, 2),
            "d2h_mb_per_step": round(out_mb, 2), "pcie_gbs": round((in_mb + out_mb) * steps / dt / 1e3, 2),
            "outputs_bit_equal_to_resident_run": same,
            "note": "pinned host buffers; H2D stream, one compute stream per buffer set, D2H stream; PCIe-inclusive; "
                    "not the headline value"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event leg (roofline = path only)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--mode", default="forward", choices=("forward", "train"),
                    help="forward = the headline metric; train = one DP training step per 'step' (row N1, not the headline)")
    ap.add_argument("--dropout", type=float, default=0.1, help="train mode only (reference default 0.1)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="forward mode: independent steps kept in flight at once, each on its own stream with its own "
                         "batch, output and workspace buffers (1 = strictly one step after the other)")
    ap.add_argument("--rounds", type=int, default=9,
                    help="timed rounds of exactly --steps steps each (barrier + synchronize on both sides of every round); "
                         "the MEDIAN round is reported as value / ms_per_step, min and max beside it")
    ap.add_argument("--stream", action="store_true",
                    help="also time the PCIe-inclusive streamed mode (distinct pinned host batches, double-buffered "
                         "H2D/D2H on side streams); reported as `stream` next to the resident-batch value")
    ap.add_argument("--stream-steps", type=int, default=200)
    ap.add_argument("--schedule", type=int, default=0,
                    help="launch schedule of the forward (include/avsep.h avsep_set_schedule): 0 = one launch per op, 1 = the "
                         "encoder layers of each branch as one dependency-driven persistent launch")
    ap.add_argument("--chain-group", type=int, default=8, help="schedule 1: clips per group of the work-list order")
    ap.add_argument("--chain-skew", type=float, default=0.0, help="schedule 1: ops between consecutive clip groups (0 = op-major)")
    ap.add_argument("--train-fp32-forward", action="store_true",
                    help="train mode, A/B: the fp32 MFMA GEMM for the Linear forward GEMMs (av_separation._train.SPLIT_GEMM = False; "
                         "default since round 5: split-precision for N, K >= 512)")
    ap.add_argument("--train-fp32-dgrad", action="store_true",
                    help="train mode, A/B: the fp32 MFMA GEMM also for the activation-gradient GEMMs (default: split-precision for N, K >= 512)")
    ap.add_argument("--train-fp32-wgrad", action="store_true",
                    help="train mode, A/B: the fp32 MFMA weight-gradient kernel everywhere (default: split-precision for N, K >= 512)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the `also` block (short cfg3 / cfg5 forward and cfg4 training lines behind the default cfg2 run; "
                         "--no-cpu and --no-profile, the developer tools' flags, skip it too)")
    ap.add_argument("--no-quality", action="store_true",
                    help="skip the trained-weights quality pair of the cpu_baseline block (the reference's demo recipe on the "
                         "HIP path and on the CPU port)")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="CPU-only check of the N>1 launch path (spawn, rendezvous, barrier, max-over-ranks, one JSON "
                         "line): no GPU work, gloo backend; used by tests/test_shards_gloo.py")
    a = ap.parse_args()
    if a.mode == "train" and a.workload == "cfg2" and "--workload" not in " ".join(sys.argv):
        a.workload = "cfg4"

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has made no GPU call yet (importing
        # torch does not initialise HIP), starts one fresh worker per GPU and relays their output and exit status.
        raise SystemExit(spawn_ranks(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus disagree")
    if a.selftest_launch:
        return selftest_launch(rank, world, a)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the HIP path has no CPU fallback)")
    # rehearsal hook for the one-GPU test box (tests/test_bench_gpu.py): all ranks on device 0, gloo instead of RCCL
    rehearsal = os.environ.get("AVSEP_BENCH_REHEARSAL") == "1"
    dev = torch.device("cuda", 0 if rehearsal else local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import av_separation as av
    if a.mode == "train":
        return train_main(a, av, dev, dist, rank, world)
    wl = WORKLOADS[a.workload]
    B = a.batch or wl["batch"]
    mk, dk = wl["model"], wl["data"]
    torch.manual_seed(0)                       # same random-init weights on every rank (replicated model)
    model = av.AVSeparationTransformer(dropout=0.0, **mk).to(dev).eval()
    if a.schedule:
        model.set_schedule(a.schedule, a.chain_group, a.chain_skew)
    # R = --inflight independent steps are kept in flight: step i runs on stream i % R with batch / output / workspace
    # set i % R (R distinct resident batches per rank; a batch of 32 clips alone leaves the 256 CUs latency-bound, a second
    # one fills the bubbles -- what a serving loop does).  Every step is a complete forward of its own 32-clip batch.
    R = max(1, a.inflight)
    ds = av.SyntheticAVDataset(num_samples=world * B * R, **dk)
    sets = []
    for r in range(R):
        its = [ds[r * world * B + i] for i in shard_range(rank, world, B)]
        mx = torch.stack([it["mixed_spec"] for it in its]).to(dev).contiguous()
        lp = torch.stack([it["lip_frames"] for it in its]).to(dev).contiguous()
        sets.append(dict(items=its, mixed=mx, lips=lp, stream=torch.cuda.Stream(device=dev)))
    items, mixed, lips = sets[0]["items"], sets[0]["mixed"], sets[0]["lips"]
    _, F, T = mixed.shape
    _, N, H, W = lips.shape
    S = mk["num_speakers"]
    for st in sets:
        st["masks"] = torch.empty(B, T, S, F, device=dev)
        st["sep"] = torch.empty(B, T, S, F, device=dev)
    masks, sep = sets[0]["masks"], sets[0]["sep"]
    graph = not a.no_graph
    stream = sets[0]["stream"]

    def run_steps(n, nsets):
        # ONE host thread feeds all buffer sets (one thread per set was measured: 0.478 vs 0.378 ms/step -- the HIP
        # runtime serialises concurrent hipGraphLaunch calls)
        for i in range(n):
            st = sets[i % nsets]
            with torch.cuda.stream(st["stream"]):
                model.run_static(st["mixed"], st["lips"], st["masks"], st["sep"], graph=graph, slot=i % nsets)

    # every in-flight slot is a native context of its own (own packed weights, workspace, graph): before anything is
    # timed, each one must reproduce slot 0's outputs on slot 0's batch bit for bit
    slots_equal = True
    with torch.no_grad():
        run_steps(max(a.warmup, R), R)             # W warm-up steps (at least one per buffer set: graph capture)
        torch.cuda.synchronize()
        for r in range(1, R):
            chk_m, chk_s = torch.empty_like(masks), torch.empty_like(sep)
            with torch.cuda.stream(sets[r]["stream"]):
                model.run_static(mixed, lips, chk_m, chk_s, graph=False, slot=r)
            torch.cuda.synchronize()
            slots_equal = slots_equal and bool(torch.equal(chk_m, masks) and torch.equal(chk_s, sep))
        if not slots_equal:
            raise SystemExit("an in-flight slot does not reproduce slot 0's outputs: refusing to time it")
        nrounds = max(1, a.rounds)
        worst, mine = timed_rounds(dist, dev, nrounds, lambda: run_steps(a.steps, R))     # each round: EXACTLY K steps
        if a.schedule:      # a chained launch that gave up waiting for a producer tile returns stale activations with AVSEP_OK: ask
            model.chain_status()
        elapsed = median(worst)
        # the same K steps strictly one after the other on one stream (the latency view; the figure to compare rounds by)
        single = single_rounds = None
        if R > 1:
            run_steps(max(2, a.warmup), 1)
            single_rounds, _ = timed_rounds(dist, dev, max(1, min(nrounds, 5)), lambda: run_steps(a.steps, 1))
            single = median(single_rounds)

    with torch.cuda.stream(stream), torch.no_grad():
        # ---- per-kernel roofline, live: eager forwards with every launch bracketed by HIP events
        prof_iters = 3
        kernels = []
        if not a.no_profile:
            model.run_static(mixed, lips, masks, sep, graph=False)
            stream.synchronize()
            model.profile_begin()
            for _ in range(prof_iters):
                model.run_static(mixed, lips, masks, sep, graph=False)
            stream.synchronize()
            kernels = model.profile_end()
            # a profiled forward's outputs are meaningless (each launch is repeated): leave a clean result behind
            model.run_static(mixed, lips, masks, sep, graph=graph)
            stream.synchronize()

    gflop_clip = flops_per_clip(F, T, N, H, W, mk["d_model"], mk["num_encoder_layers"], mk["num_fusion_layers"], S) / 1e9
    value = world * B * a.steps / elapsed
    ms_step = elapsed / a.steps * 1e3
    for k in kernels:
        k["avg_us"] = k["ms"] / k["calls"] * 1e3
        k["tflops"] = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
        k["gbs"] = k["bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
    if not kernels:   # --no-profile: price the whole path only
        kernels = [{"name": "whole forward (hipGraph)", "calls": prof_iters, "ms": ms_step * prof_iters,
                    "flops": gflop_clip * 1e9 * B * prof_iters, "bytes": 0.0, "avg_us": ms_step * 1e3,
                    "tflops": gflop_clip * B / ms_step, "gbs": 0.0}]
    dom = max(kernels, key=lambda k: k["ms"])
    launches_per_fwd = dom["calls"] / prof_iters
    # The split-precision GEMM runs on the bf16 matrix pipe: it is priced by the bf16 MFMA flops it EXECUTES (6 per algorithmic
    # fp32 product) against the bf16 dense peak; its algorithmic rate and that rate over the fp32 matrix peak are kept beside it.
    dom_pipe = kernel_pipe(dom["name"]) or ("fp32 MFMA", 1, FP32_MATRIX_PEAK_TFLOPS)
    split_dom = dom_pipe[1] > 1
    SPLIT_PRODUCTS = dom_pipe[1]
    dom_peak = dom_pipe[2]
    dom_exec = dom["tflops"] * SPLIT_PRODUCTS
    roofline = {
        "bound": "mfma", "kernel": dom["name"],
        "achieved": round(dom_exec, 3), "peak": dom_peak, "unit": "TFLOP/s",
        "frac": round(dom_exec / dom_peak, 4), "traffic": None,
        "avg_launch_us": round(dom["avg_us"], 3), "launches_per_step": launches_per_fwd,
        "algorithmic_gflop_per_launch": round(dom["flops"] / dom["calls"] / 1e9, 4),
        "share_of_kernel_time": round(dom["ms"] / sum(k["ms"] for k in kernels), 4),
        # whole path: clips/s x GFLOP/clip vs the fp32 matrix peak of ONE GPU
        "path_tflops_per_gpu": round(value / world * gflop_clip / 1e3, 3),
        "path_frac": round(value / world * gflop_clip / 1e3 / FP32_MATRIX_PEAK_TFLOPS, 4),
    }
    floor_s = path_floor(kernels, prof_iters) if not a.no_profile else None
    if floor_s:
        # every kernel of the step at the peak of the pipe it runs on (16-bit matrix pipe: executed products; fp32 matrix pipe;
        # HBM for the row kernels): a fraction <= 1 whatever the arithmetic
        roofline["path_floor_ms"] = round(floor_s * 1e3, 4)
        roofline["path_floor_frac"] = round(floor_s / (ms_step * 1e-3), 4)
    if split_dom:
        roofline["matrix_pipe"] = (dom_pipe[0] + ", fp32 accumulation; `achieved` counts the EXECUTED 16-bit flops")
        roofline["algorithmic_tflops"] = round(dom["tflops"], 3)
        roofline["algorithmic_over_fp32_matrix_peak"] = round(dom["tflops"] / FP32_MATRIX_PEAK_TFLOPS, 4)
        roofline["path_frac_note"] = ("path_frac = algorithmic fp32 flops of the whole step over the FP32 matrix peak (SURVEY.md "
                                      "section 8(d)); the Linear layers of this workload run on the bf16 pipe, so it is not bounded by 1")

    # HBM-side bytes per launch of the dominant kernel, from the committed PMC passes of this same command
    # (tools/pmc_bench.sh -> profiles/pmc_hbm_traffic.json; counters cannot be read from inside the process).  The file
    # carries the build id of the library it was measured on: a different loaded library gets no traffic figure.
    from av_separation import _native
    build_id = _native.load().avsep_build_id().decode()
    roofline["library_build_id"] = build_id
    try:
        pmc_file = json.load(open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")))
        hit = (pmc_file.get("workloads", {}).get(a.workload) or (pmc_file["kernels"] if a.workload == "cfg2" else {})).get(dom["name"])
        if pmc_file.get("build_id") != build_id:
            roofline["traffic_note"] = (f"profiles/pmc_hbm_traffic.json was measured on build {pmc_file.get('build_id')}, "
                                        f"the loaded library is {build_id}: traffic withheld (re-run tools/pmc_bench.sh)")
        elif hit:
            # ONE launch population for both figures: the PMC pass traces the eager TWO-stream
            # forward, whose tail runs as two half-batch launches per site (cfg2: 18 launches of the dominant instance per
            # forward, where the one-stream profile leg above has 13 full-batch ones).  The instance's algorithmic bytes
            # per forward are the same either way (a split site moves the same A and C rows; its W is counted once), so
            # both per-launch figures are per-forward totals over the PMC pass's launch count.
            n_pmc = hit["calls_per_forward"]
            alg_fwd = dom["bytes"] / prof_iters
            traffic_launch = hit["fetch_bytes_per_launch"] + hit["write_bytes_per_launch"]
            roofline["traffic"] = round(traffic_launch)
            roofline["traffic_unit"] = ("HBM-side bytes per launch, averaged over the launches of the eager two-stream "
                                        "forward (PMC, profiles/pmc_hbm_traffic.json, same build id)")
            roofline["algorithmic_bytes_per_launch"] = round(alg_fwd / n_pmc)
            roofline["traffic_population"] = {"launches_per_forward": round(n_pmc, 2),
                                              "traffic_bytes_per_forward": round(traffic_launch * n_pmc),
                                              "fetch_bytes_per_forward": round(hit["fetch_bytes_per_launch"] * n_pmc),
                                              "write_bytes_per_forward": round(hit["write_bytes_per_launch"] * n_pmc),
                                              "algorithmic_bytes_per_forward": round(alg_fwd),
                                              "traffic_over_algorithmic": round(traffic_launch * n_pmc / alg_fwd, 3)}
    except (OSError, KeyError, ValueError):
        pass

    # the same instances INSIDE the replayed graph (other kernels beside them on the chip): from the committed rocprofv3
    # kernel statistics of the timed region (tools/graph_stats.py -> profiles/graph_kernel_stats.json), same build only
    in_graph = {}
    try:
        gs = json.load(open(os.path.join(ROOT, "profiles", "graph_kernel_stats.json")))
        if gs.get("build_id") == build_id:
            in_graph = gs.get("workloads", {}).get(a.workload, {})
    except (OSError, ValueError):
        pass
    if in_graph.get(dom["name"]):
        # inside the replayed step the tail runs as two half-batch launches per site: the instance's flops per STEP are the
        # profile leg's, its time per step is launches x average duration there -> the fraction it reaches beside the
        # other kernels of the step
        ig = in_graph[dom["name"]]
        roofline["in_graph_avg_launch_us"] = round(ig["avg_us"], 3)
        roofline["in_graph_launches_per_step"] = round(ig["calls_per_step"], 2)
        ig_tf = dom["flops"] / prof_iters / (ig["calls_per_step"] * ig["avg_us"] * 1e-6) / 1e12 * (SPLIT_PRODUCTS if split_dom else 1)
        roofline["in_graph_achieved"] = round(ig_tf, 3)
        roofline["in_graph_frac"] = round(ig_tf / dom_peak, 4)

    out = {
        "metric": metric_string(mk, dk, R),
        "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if mk["d_model"] < 512 else ("f32 (nn.Linear products as 3 fp16 MFMA products per fp32 product where the input has a static "
                                                     "bound, else 6 bf16 products like the attention's; fp32 accumulation)"),
        "data": "synthetic",
        "config": {"workload": f"{a.workload}: SyntheticAVDataset {S}-speaker, F={F}, T={T}, N={N}, {H}x{W} lips, "
                               f"d_model={mk['d_model']}, nhead={mk['nhead']}, {mk['num_encoder_layers']}+"
                               f"{mk['num_fusion_layers']} layers, forward-only",
                   "batch_per_gpu": B, "global_batch": world * B, "gflop_per_clip": round(gflop_clip, 4),
                   "launch": ("hipGraph replay" if graph else "eager") +
                             (f", {R} independent steps in flight (one stream + batch / output / workspace set each)" if R > 1
                              else ", one step after the other"),
                   "schedule": ("one launch per op" if not a.schedule else
                                f"chained encoder layers (group {a.chain_group}, skew {a.chain_skew:g})"),
                   "steps_in_flight": R, "slots_bit_equal_on_one_batch": slots_equal,
                   "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                   "parallelism": f"replica x{world} (clip shards)"},
        "timing": {"rounds": nrounds, "steps_per_round": a.steps, "reported": "median round (max over ranks per round)",
                   "ms_per_step_min": round(min(worst) / a.steps * 1e3, 4),
                   "ms_per_step_max": round(max(worst) / a.steps * 1e3, 4),
                   "ms_per_step_rounds": [round(w / a.steps * 1e3, 4) for w in worst]},
        "roofline": roofline,
        # avg_us / tflops / gbs: each instance ALONE on the chip (HIP events, one stream); in_graph_*: the same instance inside
        # the replayed step (rocprofv3 of the timed region), where the two branches' kernels share the CUs
        "kernels": [{"name": k["name"], "calls_per_step": k["calls"] / prof_iters, "avg_us": round(k["avg_us"], 2),
                     "tflops": round(k["tflops"], 2), "gbs": round(k["gbs"], 1),
                     "in_graph_avg_us": round(in_graph[k["name"]]["avg_us"], 2) if k["name"] in in_graph else None,
                     "in_graph_calls_per_step": round(in_graph[k["name"]]["calls_per_step"], 2) if k["name"] in in_graph else None}
                    for k in kernels],
    }

    if dist is not None:
        # what each rank ran on and how long IT took: a slow rank, or two ranks on one device, must not hide in the aggregate
        idents = gather_ranks(dist, device_identity(dev))
        out["ranks"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                        "per_rank_ms": [round(x * 1e3, 4) for x in gather_ranks(dist, median(mine))],
                        "devices": idents}
        dup = duplicate_devices(idents)
        if dup and not rehearsal:
            raise SystemExit(f"ranks share a physical device (rank, rank, id): {dup}")
        out["ranks"]["duplicate_devices"] = dup
    if single is not None:
        out["one_step_at_a_time"] = {"value": round(world * B * a.steps / single, 2), "unit": "clips/s",
                                     "ms_per_step": round(single / a.steps * 1e3, 4),
                                     "ms_per_step_rounds": [round(w / a.steps * 1e3, 4) for w in single_rounds],
                                     "note": "same K steps on ONE stream, each waiting for the previous one (step latency)"}
    if rank == 0 and world == 1 and a.stream:
        out["stream"] = stream_leg(model, ds, B, dev, a.stream_steps, inflight=max(2, R))
        with torch.cuda.stream(stream), torch.no_grad():           # leave the resident batch's outputs behind again
            model.run_static(mixed, lips, masks, sep, graph=graph)
        torch.cuda.synchronize()
    if rank == 0 and world == 1 and not a.no_cpu:
        clean = torch.stack([it["clean_specs"] for it in items])
        out["cpu_baseline"] = cpu_baseline(model, mixed, lips, masks, sep, clean, mk, B, a.cpu_seconds)
        if a.workload == "cfg2" and not a.no_quality:
            out["cpu_baseline"]["quality"] = trained_quality(av, dev)
    if rank == 0 and world == 1 and a.workload == "cfg2" and not (a.no_also or a.no_cpu or a.no_profile):
        # the other BASELINE configs under the same clock: short lines, no CPU legs
        del model, sets
        torch.cuda.empty_cache()
        t_also = time.perf_counter()
        out["also"] = {w: also_forward(av, dev, w) for w in ("cfg3", "cfg5")}
        out["also"]["cfg4_train"] = also_train(av, dev)
        out["also"]["cfg4_train_fp32_forward_gemms"] = also_train(av, dev, split_gemm=False)
        out["also"]["seconds"] = round(time.perf_counter() - t_also, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def also_forward(av, dev, name, steps=20, rounds=5, warmup=4):
    """A short line for another BASELINE forward config inside the default run: the workload's own model and per-GPU batch,
    SyntheticAVDataset clips, hipGraph replay, `rounds` rounds of EXACTLY `steps` steps with two steps in flight (median
    round) and the same steps one at a time.  Both in-flight slots hold the same clips (dataset generation is host time)."""
    wl = WORKLOADS[name]
    B, mk, dk = wl["batch"], wl["model"], wl["data"]
    torch.manual_seed(0)
    model = av.AVSeparationTransformer(dropout=0.0, **mk).to(dev).eval()
    ds = av.SyntheticAVDataset(num_samples=B, **dk)
    its = [ds[i] for i in range(B)]
    S = mk["num_speakers"]
    sets = []
    for _ in range(2):
        mx = torch.stack([it["mixed_spec"] for it in its]).to(dev).contiguous()
        lp = torch.stack([it["lip_frames"] for it in its]).to(dev).contiguous()
        _, F, T = mx.shape
        sets.append(dict(mixed=mx, lips=lp, masks=torch.empty(B, T, S, F, device=dev), sep=torch.empty(B, T, S, F, device=dev),
                         stream=torch.cuda.Stream(device=dev)))
    _, N, H, W = sets[0]["lips"].shape

    def run_steps(n, nsets):
        for i in range(n):
            st = sets[i % nsets]
            with torch.cuda.stream(st["stream"]):
                model.run_static(st["mixed"], st["lips"], st["masks"], st["sep"], graph=True, slot=i % nsets)

    with torch.no_grad():
        run_steps(max(2, warmup), 2)
        torch.cuda.synchronize()
        same = bool(torch.equal(sets[0]["masks"], sets[1]["masks"]))          # two contexts, same clips: same bits
        worst, _ = timed_rounds(None, dev, rounds, lambda: run_steps(steps, 2))
        run_steps(2, 1)
        single, _ = timed_rounds(None, dev, rounds, lambda: run_steps(steps, 1))
        # per-kernel leg (in-library HIP-event profiler, one stream): the dominant kernel against the peak of the pipe it runs
        # on, and the whole step's floor on the pipes actually used
        st0 = sets[0]
        with torch.cuda.stream(st0["stream"]):
            model.run_static(st0["mixed"], st0["lips"], st0["masks"], st0["sep"], graph=False)
            st0["stream"].synchronize()
            model.profile_begin()
            model.run_static(st0["mixed"], st0["lips"], st0["masks"], st0["sep"], graph=False)
            st0["stream"].synchronize()
            kernels = model.profile_end()
            model.run_static(st0["mixed"], st0["lips"], st0["masks"], st0["sep"], graph=True)
            st0["stream"].synchronize()
        # one clip at a time (the reference's own evaluation calls the model at B = 1, demo.py:31-64): the latency a drop-in sees
        one = dict(mixed=st0["mixed"][:1].contiguous(), lips=st0["lips"][:1].contiguous(), masks=torch.empty(1, T, S, F, device=dev),
                   sep=torch.empty(1, T, S, F, device=dev))
        def run_one(n):
            with torch.cuda.stream(st0["stream"]):
                for _ in range(n):
                    model.run_static(one["mixed"], one["lips"], one["masks"], one["sep"], graph=True, slot=0)
        run_one(3)
        torch.cuda.synchronize()
        b1, _ = timed_rounds(None, dev, rounds, lambda: run_one(steps))
        b1_equal = bool(torch.equal(one["masks"][0], sets[0]["masks"][0]))         # the clip alone has the bits it has inside the batch
    gflop = flops_per_clip(F, T, N, H, W, mk["d_model"], mk["num_encoder_layers"], mk["num_fusion_layers"], S) / 1e9
    el2, el1 = median(worst), median(single)
    inflight = 2 if el2 <= el1 else 1                                                 # the faster of the two is the line's value
    el = min(el2, el1)
    value = B * steps / el
    for k in kernels:
        k["tflops"] = k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0
    dom = max(kernels, key=lambda k: k["ms"])
    pipe = kernel_pipe(dom["name"]) or ("fp32 MFMA", 1, FP32_MATRIX_PEAK_TFLOPS)
    floor_s = path_floor(kernels, 1)
    out = {"metric": metric_string(mk, dk, inflight), "value": round(value, 2), "unit": "clips/s", "ms_per_step": round(el / steps * 1e3, 4),
           "steps": steps, "rounds": rounds, "batch_per_gpu": B, "gflop_per_clip": round(gflop, 4), "steps_in_flight": inflight,
           "two_steps_in_flight": {"value": round(B * steps / el2, 2), "ms_per_step": round(el2 / steps * 1e3, 4)},
           "roofline": {"bound": "mfma", "kernel": dom["name"], "matrix_pipe": pipe[0],
                        "achieved": round(dom["tflops"] * pipe[1], 2), "peak": pipe[2], "unit": "TFLOP/s (executed)",
                        "frac": round(dom["tflops"] * pipe[1] / pipe[2], 4), "algorithmic_tflops": round(dom["tflops"], 2),
                        "avg_launch_us": round(dom["ms"] / dom["calls"] * 1e3, 2), "launches_per_step": dom["calls"],
                        "share_of_kernel_time": round(dom["ms"] / sum(k["ms"] for k in kernels), 4),
                        "path_floor_ms": round(floor_s * 1e3, 4), "path_floor_frac": round(floor_s / (el / steps), 4),
                        "path_floor_of": "every kernel at the peak of the pipe it runs on: executed 16-bit matrix flops / 2.5 PFLOP/s, "
                                         "fp32 matrix flops / 157.3 TFLOP/s, row kernels' bytes / 8 TB/s"},
           "path_frac": round(value * gflop / 1e3 / FP32_MATRIX_PEAK_TFLOPS, 4),
           "path_frac_of": "fp32 matrix peak (algorithmic fp32 flops; d_model >= 512: the Linear layers run as split-precision 16-bit "
                           "MFMA products, so the fraction is not bounded by 1 -- roofline.path_floor_frac is)",
           "one_step_at_a_time": {"value": round(B * steps / el1, 2), "ms_per_step": round(el1 / steps * 1e3, 4)},
           "one_clip_at_a_time": {"ms_per_forward": round(median(b1) / steps * 1e3, 4), "clips_per_s": round(steps / median(b1), 1),
                                  "bit_equal_to_the_clip_inside_the_batch": b1_equal},
           "slots_bit_equal": same, "masks_in_unit_interval": bool(float(sets[0]["masks"].min()) >= 0.0 and
                                                                   float(sets[0]["masks"].max()) <= 1.0),
           "note": "short line inside the default run: the workload's own model and per-GPU batch, hipGraph replay, both in-flight "
                   "slots hold the same SyntheticAVDataset clips (own buffers, own native context each); the full-length figures "
                   "are `bench.py --workload " + name + "`"}
    del model, sets
    torch.cuda.empty_cache()
    return out


def also_train(av, dev, steps=5, rounds=3, warmup=6, split_gemm=True):
    """A short line for BASELINE configs[3]'s training step inside the default run: the loop body of `--mode train`
    (zero_grad, train-mode forward with dropout 0.1, SeparationLoss, backward, clip, fused Adam) on one resident 16-clip
    batch, single rank.  split_gemm: av_separation._train.SPLIT_GEMM ("the training step on the split-precision GEMM";
    on by default since round 5) for the duration of the line."""
    from av_separation import _train
    from av_separation.losses import SeparationLoss
    was = _train.SPLIT_GEMM
    _train.SPLIT_GEMM = bool(split_gemm)
    try:
        return _also_train(av, dev, steps, rounds, warmup, split_gemm)
    finally:
        _train.SPLIT_GEMM = was


def _also_train(av, dev, steps, rounds, warmup, split_gemm):
    from av_separation.losses import SeparationLoss
    wl = WORKLOADS["cfg4"]
    B, mk, dk = wl["batch"], wl["model"], wl["data"]
    torch.manual_seed(0)
    model = av.AVSeparationTransformer(dropout=0.1, **mk).to(dev).train()
    ds = av.SyntheticAVDataset(num_samples=B, **dk)
    its = [ds[i] for i in range(B)]
    mixed = torch.stack([it["mixed_spec"] for it in its]).to(dev).contiguous()
    lips = torch.stack([it["lip_frames"] for it in its]).to(dev).contiguous()
    targets = torch.stack([it["clean_specs"] for it in its]).to(dev).contiguous()
    _, F, T = mixed.shape
    _, N, H, W = lips.shape
    crit = SeparationLoss(0.5)
    opt = torch.optim.Adam(model.parameters(), lr=3e-4, fused=True)
    losses = []

    def step():
        opt.zero_grad()
        sep, _ = model(mixed, lips)
        loss = crit(sep, targets)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0, foreach=True)
        opt.step()
        losses.append(loss.detach())

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    worst, _ = timed_rounds(None, dev, rounds, lambda: [step() for _ in range(steps)])
    el = median(worst)
    gflop = 3 * flops_per_clip(F, T, N, H, W, mk["d_model"], mk["num_encoder_layers"], mk["num_fusion_layers"],
                               mk["num_speakers"]) / 1e9
    value = B * steps / el
    out = {"metric": "training clips/sec (forward+backward+Adam step), fp32", "value": round(value, 2), "unit": "clips/s",
           "ms_per_step": round(el / steps * 1e3, 3), "steps": steps, "rounds": rounds, "batch_per_gpu": B,
           "gflop_per_clip": round(gflop, 3), "frac": round(value * gflop / 1e3 / FP32_MATRIX_PEAK_TFLOPS, 4),
           "loss_first_last": [round(float(losses[0]), 4), round(float(losses[-1]), 4)],
           "linear_gemm": ("split-precision forward, activation-gradient and weight-gradient GEMMs (default; gradients gated against the float64 "
                           "oracle run with the step's own ReLU decisions, tests/test_train_gpu.py)"
                           if split_gemm else "fp32 MFMA forward (av_separation._train.SPLIT_GEMM = False: the reference's own ReLU decisions), "
                           "split-precision activation-gradient and weight-gradient GEMMs")}
    del model, opt
    torch.cuda.empty_cache()
    return out


def trained_quality(av, dev):
    """The quality half of BASELINE.json's metric on TRAINED weights: the reference's demo recipe
    (tools/quality_recipe.py; /root/reference/demo.py:116-198) from the same initial weights and batch order on the HIP path
    and on the CPU port of the reference."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import quality_recipe as q
    order = q.batch_order(q.DATA["num_samples"], seed=7)
    items = q.make_items(av, order)
    g, state0 = q.run_gpu(av, dev, items, order)
    c = q.run_cpu_port(state0, items, order)
    r3 = lambda x: round(x, 3)  # noqa: E731
    return {"recipe": "reference demo.py:116-198: d_model 128, 2+2 layers, dropout 0.1, 63 Adam steps (batch 8, lr 3e-4, "
                      "SeparationLoss(0.5), clip 1.0), SNR of the first 20 items, seeded batch order, same initial weights; "
                      "README.md:61-65: input 0.01 / untrained 3.20 / trained 37.24 / improvement +37.23 dB",
            "input_snr_db": r3(g["in_snr"]),
            "gpu": {"untrained_out_snr_db": r3(g["out_snr_untrained"]), "trained_out_snr_db": r3(g["out_snr"]),
                    "snr_improvement_db": r3(g["improvement"]), "train_seconds": round(g["train_seconds"], 2)},
            "cpu_port": {"untrained_out_snr_db": r3(c["out_snr_untrained"]), "trained_out_snr_db": r3(c["out_snr"]),
                         "snr_improvement_db": r3(c["improvement"]), "train_seconds": round(c["train_seconds"], 2)},
            "trained_out_snr_gpu_minus_cpu_db": r3(g["out_snr"] - c["out_snr"])}


def train_main(a, av, dev, dist, rank, world):
    """--mode train: a step = zero_grad, train-mode forward (dropout, BatchNorm batch statistics), PIT loss, backward,
    gradient all-reduce (N>1), clip_grad_norm_(1.0), Adam -- the reference's quick_train loop body (demo.py:96-106)
    on one resident batch per rank.  Not the headline metric; reported for the N1 row."""
    from av_separation import parallel, _train
    from av_separation.losses import SeparationLoss
    if a.train_fp32_forward:
        _train.SPLIT_GEMM = False
    if a.train_fp32_dgrad:
        _train.SPLIT_GEMM_DGRAD = False
    if a.train_fp32_wgrad:
        _train.SPLIT_GEMM_WGRAD = False
    wl = WORKLOADS[a.workload]
    B = a.batch or wl["batch"]
    mk, dk = wl["model"], wl["data"]
    torch.manual_seed(0)
    model = av.AVSeparationTransformer(dropout=a.dropout, **mk).to(dev).train()
    ds = av.SyntheticAVDataset(num_samples=world * B, **dk)
    items = [ds[i] for i in shard_range(rank, world, B)]
    mixed = torch.stack([it["mixed_spec"] for it in items]).to(dev).contiguous()
    lips = torch.stack([it["lip_frames"] for it in items]).to(dev).contiguous()
    targets = torch.stack([it["clean_specs"] for it in items]).to(dev).contiguous()
    _, F, T = mixed.shape
    _, N, H, W = lips.shape
    S = mk["num_speakers"]
    crit = SeparationLoss(0.5)
    dp = parallel.DataParallel(model, timing=True) if dist is not None else None
    opt = torch.optim.Adam(model.parameters(), lr=3e-4, fused=True)
    losses, bwd_events, exposed = [], [], []

    def step():
        if dp is not None:
            dp.zero_grad()
        else:
            opt.zero_grad()      # set_to_none=True, the reference's optimizer.zero_grad() (demo.py:99): the backward then
                                 # ASSIGNS each .grad instead of zero-filling and accumulating (~150 fills + adds per step)
        sep, _ = model(mixed, lips)
        loss = crit(sep, targets, group=dp.group if dp is not None else None)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        loss.backward()
        e1.record()
        if dp is not None:
            dp.reduce_gradients()
            exposed.append(dp.buckets.exposed_wait_s)
        e2.record()
        bwd_events.append((e0, e1, e2))
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0, foreach=True)
        opt.step()
        losses.append(loss.detach())

    for _ in range(max(1, a.warmup)):
        step()
    torch.cuda.synchronize()

    def one_round():
        bwd_events.clear()
        exposed.clear()
        for _ in range(a.steps):
            step()

    nrounds = max(1, min(a.rounds, 5))
    worst, mine = timed_rounds(dist, dev, nrounds, one_round)       # each round: EXACTLY K steps, sync on both sides
    elapsed = median(worst)
    gflop_clip = 3 * flops_per_clip(F, T, N, H, W, mk["d_model"], mk["num_encoder_layers"], mk["num_fusion_layers"], S) / 1e9
    value = world * B * a.steps / elapsed
    tf = value / world * gflop_clip / 1e3
    nparam = sum(p.numel() for p in model.parameters())
    out = {
        "metric": "training clips/sec (forward+backward+Adam step), fp32", "value": round(value, 2), "unit": "clips/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{a.workload} training: SyntheticAVDataset {S}-speaker, F={F}, T={T}, N={N}, {H}x{W} lips, "
                               f"d_model={mk['d_model']}, nhead={mk['nhead']}, {mk['num_encoder_layers']}+"
                               f"{mk['num_fusion_layers']} layers, dropout {a.dropout}, SeparationLoss(0.5), "
                               f"clip 1.0, Adam lr 3e-4",
                   "batch_per_gpu": B, "global_batch": world * B, "gflop_per_clip": round(gflop_clip, 3),
                   "parameters": nparam, "launch": "eager, one launch per op",
                   "gemm": ("Linear layers with N, K >= 512: forward GEMMs " +
                            ("split-precision" if _train.SPLIT_GEMM else "fp32 MFMA (A/B switch)") + ", activation-gradient GEMMs " +
                            ("split-precision" if (_train.SPLIT_GEMM or _train.SPLIT_GEMM_DGRAD) else "fp32 MFMA") + ", weight gradients " +
                            ("split-precision" if _train.SPLIT_GEMM_WGRAD else "fp32 MFMA") +
                            " (split-precision = six bf16 MFMA products per fp32 product, fp32-equivalent); everything else fp32 MFMA"),
                   "parallelism": f"dp{world} (bucketed gradient all-reduce + cross-rank BatchNorm statistics)"
                   if world > 1 else "single rank"},
        "roofline": {"bound": "mfma", "kernel": "whole training step (3 x forward FLOPs)", "achieved": round(tf, 3),
                     "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP32_MATRIX_PEAK_TFLOPS, 4),
                     "traffic": None},
        "loss_first_last": [round(float(losses[0]), 4), round(float(losses[-1]), 4)],
        # the exchange step next to the backward it overlaps: device time of backward() (reduce-scatters are launched
        # from its hooks), device time from its end to the end of reduce_gradients() (= the part of the exchange the
        # backward did not hide, plus the all-gathers), host time spent waiting inside reduce_gradients()
        "exchange": {"backward_ms": round(sum(e0.elapsed_time(e1) for e0, e1, _ in bwd_events) / len(bwd_events), 3),
                     "after_backward_ms": round(sum(e1.elapsed_time(e2) for _, e1, e2 in bwd_events) / len(bwd_events), 3),
                     "host_wait_ms": round(1e3 * sum(exposed) / max(1, len(exposed)), 3),
                     "gradient_mb": round(4 * nparam / 1e6, 1),
                     "buckets": len(dp.buckets.buckets) if dp is not None else 0,
                     "form": "reduce-scatter (from backward hooks) + 1/G on the shard + all-gather" if dp is not None
                     else "single rank: no exchange",
                     # last step of the last round: launch-to-completion span of every bucket's reduce-scatter (it starts
                     # inside the backward, so the span includes the backward it overlaps) and of its all-gather
                     "per_bucket": dp.buckets.bucket_times_ms() if dp is not None else []},
        "timing": {"rounds": nrounds, "steps_per_round": a.steps, "reported": "median round (max over ranks per round)",
                   "ms_per_step_rounds": [round(w / a.steps * 1e3, 3) for w in worst]},
    }
    if dist is not None:
        idents = gather_ranks(dist, device_identity(dev))
        out["ranks"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                        "per_rank_ms": [round(x * 1e3, 3) for x in gather_ranks(dist, median(mine))], "devices": idents}
        dup = duplicate_devices(idents)
        if dup and os.environ.get("AVSEP_BENCH_REHEARSAL") != "1":
            raise SystemExit(f"ranks share a physical device (rank, rank, id): {dup}")
        out["ranks"]["duplicate_devices"] = dup
    if rank == 0 and world == 1 and not a.no_cpu:
        out["cpu_baseline"] = cpu_train_baseline(model, mixed, lips, targets, mk, B, a.dropout, a.cpu_seconds)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def cpu_train_baseline(model, mixed, lips, targets, mk, B, dropout, budget_s):
    """The reference's training step on this box's host cores through oracle/torch_cpu.forward_train (pinned against
    the reference's gradients in tests/test_oracle.py): forward + SeparationLoss + backward + clip + Adam, same batch."""
    from oracle import torch_cpu
    from av_separation.losses import SeparationLoss
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    params = []
    for k, t in state.items():
        if t.is_floating_point() and "running_" not in k and not k.endswith(".pe"):
            t.requires_grad_()
            params.append(t)
    opt = torch.optim.Adam(params, lr=3e-4)
    crit = SeparationLoss(0.5)
    mx, lp, tg = mixed.cpu(), lips.cpu(), targets.cpu()
    avail = torch.get_num_threads()
    threads = min(avail, 32)                       # the forward baseline's sweep lands on 16-32 threads on this host
    torch.set_num_threads(threads)

    def step():
        opt.zero_grad()
        sep, _ = torch_cpu.forward_train(state, mx, lp, mk["nhead"], mk["num_speakers"], dropout)
        loss = crit(sep, tg)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()

    times = []
    t_end = time.perf_counter() + max(budget_s, 1.0) * 2.5
    while len(times) < 3 and (not times or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    torch.set_num_threads(avail)
    best = min(times)
    return {"value": round(B / best, 3), "unit": "clips/s", "cores": threads, "threads": threads, **host_cpu_info(),
            "kind": "port",
            "sample": f"{len(times)} training steps of the same {B}-clip batch, best {best:.2f} s "
                      f"(torch {torch.__version__} CPU autograd, fp32, dropout {dropout})"}


def si_snr_improvement(separated, mixed, clean):
    """SI-SNRi in dB (BASELINE.json's quality metric): best-permutation SI-SNR of the separated spectrograms against the
    clean ones minus the SI-SNR of the unprocessed mixture, batch mean (losses.si_snr semantics)."""
    from itertools import permutations
    from av_separation.losses import si_snr
    S = clean.shape[1]
    best = max(float(si_snr(separated[:, list(p)], clean)) for p in permutations(range(S)))
    return best - float(si_snr(mixed.unsqueeze(1).expand_as(clean), clean))


def cpu_baseline(model, mixed, lips, masks_gpu, sep_gpu, clean, mk, B, budget_s):
    """Reference CPU path (port) on this box's host cores + the parity of this run's GPU outputs against it."""
    import numpy as np
    from oracle import torch_cpu
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    mx, lp = mixed.cpu(), lips.cpu()
    run = lambda: torch_cpu.forward(state, mx, lp, mk["nhead"], mk["num_speakers"])  # noqa: E731
    # the reference sets no thread count; torch's default (= all cores) is far from the best on a big host
    # for these small ops, so pick the fastest of a few counts first and report the one used
    avail = torch.get_num_threads()
    best_t, threads = None, avail
    swept = sorted({c for c in (8, 16, 32, 64, avail) if c <= avail})
    for n in swept:
        torch.set_num_threads(n)
        t0 = time.perf_counter()
        run()
        reps = 3 if time.perf_counter() - t0 < 0.5 else 1      # big configs: one timed run per count bounds the sweep
        dt = float("inf")
        for _ in range(reps):                    # best of 3: single timings on a shared host are noisy
            t0 = time.perf_counter()
            run()
            dt = min(dt, time.perf_counter() - t0)
        if best_t is None or dt < best_t:
            best_t, threads = dt, n
    torch.set_num_threads(threads)
    run()
    times = []
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end and len(times) < 200:
        t0 = time.perf_counter()
        ref_sep, ref_masks = run()
        times.append(time.perf_counter() - t0)
    torch.set_num_threads(avail)
    med = float(np.median(times))
    got = masks_gpu.permute(0, 2, 3, 1).cpu()
    got_sep = sep_gpu.permute(0, 2, 3, 1).cpu()
    quality = {"si_snr_i_db_gpu": round(si_snr_improvement(got_sep, mx, clean), 4),
               "si_snr_i_db_cpu": round(si_snr_improvement(ref_sep, mx, clean), 4),
               "weights": "random init (torch.manual_seed(0)): the pair shows parity of the metric on the bench batch; separation "
                          "quality on TRAINED weights is `quality` below"}
    return {"value": round(B / med, 2), "unit": "clips/s", "cores": threads, "threads": threads, **host_cpu_info(),
            "kind": "port",
            "sample": f"{len(times)} forwards of the same {B}-clip batch, median {med * 1e3:.1f} ms "
                      f"(torch {torch.__version__} CPU, eval/no_grad/fp32, fused encoder fast path; `cores` = the "
                      f"thread count used = fastest of {swept} threads; host_cores = physical cores of the box)",
            "gpu_masks_max_abs_err_vs_cpu": float((got - ref_masks).abs().max()),
            "gpu_separated_max_abs_err_vs_cpu": float((got_sep - ref_sep).abs().max()), **quality}


if __name__ == "__main__":
    main()
