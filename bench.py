#!/usr/bin/env python3
"""ESRGAN hot-path benchmark (see BASELINE.json: LR Mpixels/s of the 4x train step).

    python bench.py --gpus N --steps K --warmup W

N == 1 workload: BASELINE config 2 — 23-RRDB generator warm-up (L1 pixel loss),
forward + backward + Adam, batch 16 of 64x64 LR patches, bf16 MFMA, synthetic
data, default-initialised weights.  For N > 1 the driver launches one rank per GPU
(torch.distributed.run); each rank takes its own batch of 16 (weak scaling) and the
gradients are averaged over RCCL before the optimizer step.

Prints ONE JSON line on rank 0 with the throughput, a roofline object for the
dominant kernel (measured live with events around every launch of one extra,
untimed step) and, at N == 1, the CPU baseline (the oracle's plain-PyTorch
restatement of the same step on the host cores, bounded sample).
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_LR_PX_FWD_BWD = 107_561_088      # 3 x 17,926,848 MAC x 2 (conv MACs only)
# adversarial step: 3 G + D (3 fwd + 2x2 bwd + 1 dgrad) + VGG (2 fwd + 1 dgrad), conv MACs only
FLOP_PER_LR_PX_ADV = 168_921_216
PEAK_BF16_TFLOPS = 2500.0                  # MI355X dense bf16 MFMA peak


def profile_ops(plan, ctypes_mod):
    """Events around the launches of one forward+backward, in the form the timed steps run them:
    the backward's weight-gradient launches on the plan's side stream beside the data-gradient
    chain, each event recorded on the stream its kernels are launched on.
    Returns {kernel: [time_ms, flops, n]}."""
    stats = {}
    main = torch.cuda.current_stream()
    lib = plan.lib
    evs = []

    def timed(ops, meta, stream):
        i = 0
        while i < len(ops):
            # one event pair around each run of consecutive launches of the same kernel (the
            # four 32-channel convolutions of a dense block, ...): an event between two short
            # launches costs about as much as the launch boundary it replaces
            j = i
            while j + 1 < len(ops) and meta[j + 1][0] == meta[i][0] and j + 1 - i < 8:
                j += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = lib.srx_run_ops(ctypes_mod.addressof(ops[i]), j + 1 - i, stream.cuda_stream)
            e1.record(stream)
            assert rc == 0
            evs.append((meta[i][0], sum(m[1] for m in meta[i:j + 1]), j + 1 - i, e0, e1))
            i = j + 1

    timed(plan.fwd_ops, plan.fwd_meta, main)
    if plan.bwd_segments is not None and plan.bwd_seg_meta is not None:
        E = importlib.import_module("super-resolution_amd.engine")
        side = E.side_stream(plan.device)
        for (chain, aside), (cmeta, smeta) in zip(plan.bwd_segments, plan.bwd_seg_meta):
            if chain is not None:
                timed(chain, cmeta, main)
            if aside is not None:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                timed(aside, smeta, side)
        done = torch.cuda.Event()
        done.record(side)
        main.wait_event(done)
    else:
        timed(plan.bwd_ops, plan.bwd_meta, main)
    torch.cuda.synchronize()
    for name, flops, n, e0, e1 in evs:
        s = stats.setdefault(name, [0.0, 0.0, 0])
        s[0] += e0.elapsed_time(e1)
        s[1] += flops
        s[2] += n
    return stats


def host_cpu_info():
    """(model string, physical cores, logical CPUs) of the host, from /proc/cpuinfo."""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    model = v
                elif k == "processor":
                    logical += 1
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                elif not k and phys is not None and core is not None:
                    cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    return model, (len(cores) or None), (logical or os.cpu_count())


def cpu_baseline():
    """Oracle (plain PyTorch fp32 restatement of the reference) on the host cores, BASELINE
    config 1's shape (B=4 of 32x32 LR, 23 blocks): the forward exactly as config 1 states it, and
    the same train step the GPU line measures (fwd + L1 + bwd).  Bounded: about 6 s each.
    Threads: the job's CPU share (OMP_NUM_THREADS, 16 per GPU on the measurement boxes), not
    every core of the host — stated next to the host's core count."""
    from oracle import esrgan_oracle as O
    model, physical, logical = host_cpu_info()
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = min(logical or 1, share) if share > 0 else (physical or logical or 1)
    torch.set_num_threads(threads)
    sd = O.make_state_dict(O.generator_state_dict_spec(23), salt=3)
    lr = O.hash_input("bench.lr", (4, 3, 32, 32))
    hr = O.hash_input("bench.hr", (4, 3, 128, 128))

    def fwd():
        with torch.no_grad():
            return O.generator_forward(lr, sd)

    def fwd_bwd():
        for v in sd.values():
            v.requires_grad_(True)
        out = O.generator_forward(lr, sd)
        return torch.autograd.grad(O.l1_loss(out, hr), list(sd.values()))

    def timed(fn, budget_s):
        t0 = time.perf_counter()
        fn()
        warm = time.perf_counter() - t0
        iters = max(3, min(40, int(budget_s / max(warm, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        return (time.perf_counter() - t0) / iters, iters

    dt_f, it_f = timed(fwd, 6.0)
    dt_s, it_s = timed(fwd_bwd, 6.0)
    px = 4 * 32 * 32
    return {"value": px / dt_s / 1e6, "unit": "LR Mpixels/s", "cores": threads, "kind": "port",
            "threads_used": threads, "host_physical_cores": physical, "host_logical_cpus": logical,
            "cpu_model": model,
            "forward_only_value": px / dt_f / 1e6,
            "sample": "oracle generator, fp32, B=4 of 32x32 LR (config 1's shape), 23 blocks: value = fwd+L1+bwd "
                      f"({it_s} timed iters after 1 warm-up, {dt_s:.3f} s/iter); forward_only_value = config 1 as "
                      f"stated ({it_f} iters, {dt_f:.3f} s/iter); threads = OMP_NUM_THREADS (the job's CPU share)"}


def child_commands(args_list, n, port):
    """(argv, environment additions) of the n ranks this script starts itself for --gpus n when it
    was not started by a launcher: one process per GPU, the same arguments, the rendezvous on
    127.0.0.1 (the container hostname may not resolve)."""
    cmds = []
    for r in range(n):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
        cmds.append(([sys.executable, os.path.abspath(__file__)] + list(args_list), env))
    return cmds


def _tail(path, nbytes=2000):
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            f.seek(max(0, f.tell() - nbytes))
            return f.read().decode("utf-8", "replace")
    except OSError:
        return ""


def supervise_ranks(cmds, rank_timeout, poll_s=0.1, grace_s=5.0, log_dir=None):
    """Start one fresh process per (argv, environment additions) entry and watch all of them.

    Every rank's stdout and stderr go to files of their own (rank<r>.out / rank<r>.err under
    log_dir, a temporary directory by default).  The first rank that exits non-zero, or the
    group running past rank_timeout seconds, ends the job: the remaining ranks (which would
    otherwise sit in the rendezvous or in a collective until some outer limit) are terminated,
    then killed after grace_s, and each rank's stderr tail is written to this process's stderr.
    Returns (exit code, rank 0's stdout): 0 only if every rank exited 0."""
    import subprocess
    import tempfile
    log_dir = log_dir or tempfile.mkdtemp(prefix="bench_ranks_")
    procs, files = [], []
    for r, (argv, env) in enumerate(cmds):
        fo = open(os.path.join(log_dir, f"rank{r}.out"), "wb")
        fe = open(os.path.join(log_dir, f"rank{r}.err"), "wb")
        files += [fo, fe]
        procs.append(subprocess.Popen(argv, env=dict(os.environ, **env), stdout=fo, stderr=fe))
    t0 = time.monotonic()
    why = None
    while why is None:
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            why = f"rank {bad[0]} exited with code {rcs[bad[0]]}"
        elif all(rc == 0 for rc in rcs):
            break
        elif time.monotonic() - t0 > rank_timeout:
            why = f"ranks still running after --rank-timeout {rank_timeout:g} s"
        else:
            time.sleep(poll_s)
    if why is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < grace_s:
            time.sleep(poll_s)
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    for f in files:
        f.close()
    out0 = _tail(os.path.join(log_dir, "rank0.out"), 1 << 20)
    if why is None:
        return 0, out0
    sys.stderr.write(f"bench.py: {why}; the other ranks were stopped (logs: {log_dir})\n")
    for r, p in enumerate(procs):
        sys.stderr.write(f"--- rank {r} (exit code {p.returncode}) stderr tail ---\n"
                         + _tail(os.path.join(log_dir, f"rank{r}.err")) + "\n")
    sys.stderr.flush()
    return 1, out0


def self_launch(args, argv):
    """--gpus N without a launcher: start N fresh rank processes (before this process has made
    any GPU call), forward rank 0's JSON line; a rank that fails, or the group running past
    --rank-timeout, ends the job with a non-zero exit code and every rank's stderr tail."""
    import socket
    port = args.master_port
    if not port:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
    cmds = child_commands([a for a in argv if a != "--dry-launch"], args.gpus, port)
    if args.dry_launch:
        print(json.dumps({"dry_launch": [{"argv": c, "env": e} for c, e in cmds]}), flush=True)
        return 0
    ngpu = torch.cuda.device_count()            # counts devices without initialising the GPU
    if ngpu < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {ngpu} GPU(s); one rank per GPU is "
                         f"required (RCCL does not run two ranks on one device)")
    rc, out0 = supervise_ranks(cmds, args.rank_timeout)
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def bench_inference(args, dev, rank, world):
    """Config 4: generator forward only, bf16, replicas (no collective): B images of patch x patch
    LR per GPU (default 4 of 512x512 = the 8-GPU share of 32 images), whole-image or tiled."""
    I = importlib.import_module("super-resolution_amd.inference")
    G = importlib.import_module("super-resolution_amd.esrgan.generator")
    torch.manual_seed(0)
    gen = G.GeneratorESRGAN(num_blocks=args.blocks).to(dev)
    B = args.batch if args.batch != 16 else 4
    h = args.patch or 512
    eng = I.SuperResolutionInference("esrgan", model=gen, half_precision=True, device=str(dev), window="hann_pos")
    x = torch.rand((B, 3, h, h), generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    for _ in range(max(1, args.warmup // 2)):
        eng.upscale(x, tile_size=args.tile)
    torch.cuda.synchronize()
    steps = max(1, args.steps // 4)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.upscale(x, tile_size=args.tile)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    value = world * B * h * h / dt / 1e6
    out = {"metric": "LR Mpixels/sec ESRGAN 4x inference forward", "value": round(value, 3), "unit": "LR Mpixels/s",
           "n_gpus": world, "steps": steps, "warmup": max(1, args.warmup // 2), "ms_per_step": round(dt * 1e3, 2),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"config 4: inference, {B} images/GPU of {h}x{h} LR -> {4 * h}x{4 * h}, "
                                  + (f"tiled {args.tile}/overlap 32, tiles batched" if args.tile else "whole image"),
                      "parallelism": f"replicas x{world}"},
           "algorithmic_tflops": round(value * 1e6 * 35_853_696 / 1e12, 1) if args.blocks == 23 else None}
    if rank == 0:
        print(json.dumps(out), flush=True)


def run_extras(dev, only=None):
    """The other BASELINE configurations on this GPU, a bounded pass after the headline measurement:
    the config-5 shape (64 x 128x128 warm-up step), config 3 (adversarial step) at 32x32 and 64x64 LR,
    config 4 (inference, the one-GPU share of 32 images: 4 of 512x512) whole and tiled 256 / 32.
    Returns {name: {ms_per_step, lr_mpx_s, pct_bf16_mfma_peak, steps}}; a case that fails is
    reported as {"error": ...} and does not touch the headline keys."""
    import gc
    import warnings
    G = importlib.import_module("super-resolution_amd.esrgan.generator")
    T = importlib.import_module("super-resolution_amd.train")
    out = {}

    def timed(step, warm, steps):
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def record(name, dt, px, flop_per_px, steps):
        tf = px / dt * flop_per_px / 1e12
        out[name] = {"ms_per_step": round(dt * 1e3, 3), "lr_mpx_s": round(px / dt / 1e6, 3),
                     "pct_bf16_mfma_peak": round(100 * tf / PEAK_BF16_TFLOPS, 2) if flop_per_px else None,
                     "steps": steps}

    def batch(B, h, seed):
        g = torch.Generator().manual_seed(seed)
        return ((torch.rand((B, 3, h, h), generator=g) * 2 - 1).to(dev),
                (torch.rand((B, 3, 4 * h, 4 * h), generator=g) * 2 - 1).to(dev))

    def warmup_case(name, B, h, warm, steps):
        torch.manual_seed(0)
        gen = G.GeneratorESRGAN(num_blocks=23).to(dev)
        tr = T.GeneratorWarmupTrainer(gen, (B, 3, h, h), torch.bfloat16)
        lr, hr = batch(B, h, 4321)
        record(name, timed(lambda: tr.step(lr, hr), warm, steps), B * h * h, FLOP_PER_LR_PX_FWD_BWD, steps)

    def adversarial_case(name, B, h, warm, steps):
        Dm = importlib.import_module("super-resolution_amd.esrgan.discriminator")
        Lm = importlib.import_module("super-resolution_amd.utils.loss")
        torch.manual_seed(0)
        gen, disc = G.GeneratorESRGAN(num_blocks=23).to(dev), Dm.Discriminator().to(dev)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            vgg = Lm.VGGLoss(device=dev, allow_synthetic=True)
        tr = T.AdversarialTrainer(gen, disc, vgg, (B, 3, h, h), torch.bfloat16)
        lr, hr = batch(B, h, 4322)
        record(name, timed(lambda: tr.step(lr, hr), warm, steps), B * h * h, FLOP_PER_LR_PX_ADV, steps)

    def inference_case(name, B, h, tile, warm, steps):
        I = importlib.import_module("super-resolution_amd.inference")
        torch.manual_seed(0)
        gen = G.GeneratorESRGAN(num_blocks=23).to(dev)
        eng = I.SuperResolutionInference("esrgan", model=gen, half_precision=True, device=str(dev), window="hann_pos")
        x = torch.rand((B, 3, h, h), generator=torch.Generator().manual_seed(4323)).to(dev)
        # algorithmic FLOPs of the image's own pixels (tile overlap is recomputation, not counted)
        record(name, timed(lambda: eng.upscale(x, tile_size=tile), warm, steps), B * h * h, 35_853_696, steps)

    cases = [("c5_shape_b64_128", lambda n: warmup_case(n, 64, 128, 3, 8)),
             ("c3_b16_32", lambda n: adversarial_case(n, 16, 32, 5, 15)),
             ("c3_b16_64", lambda n: adversarial_case(n, 16, 64, 4, 10)),
             ("c4_b4_512_whole", lambda n: inference_case(n, 4, 512, 0, 2, 4)),
             ("c4_b4_512_tiled256", lambda n: inference_case(n, 4, 512, 256, 2, 4))]
    for name, fn in cases:
        if only and name not in only:
            continue
        try:
            fn(name)
        except Exception as e:                      # an extra must never cost the headline line
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        gc.collect()
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["c2", "c3", "c4"], default="c2",
                    help="c2: generator warm-up step (the headline); c3: full adversarial D+G step; "
                         "c4: inference forward (whole image, or tiled with --tile)")
    ap.add_argument("--tile", type=int, default=0, help="c4: tile size (0 = whole image), overlap 32")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--patch", type=int, default=None, help="LR patch size (default 64 for c2, 32 for c3)")
    ap.add_argument("--blocks", type=int, default=23)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--extras", type=int, default=None,
                    help="after the headline measurement also time the other BASELINE configurations "
                         "(default: on for the default one-GPU config-2 run, off otherwise)")
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--wgrad-overlap", type=int, default=1, help="weight gradients on a side stream (0/1)")
    ap.add_argument("--skip-wgrad", action="store_true", help="diagnostic: drop weight-gradient launches (INVALID as a result)")
    ap.add_argument("--wgrad-slices", type=int, default=None, help="pixel slices per weight-gradient task")
    ap.add_argument("--wgrad-rdbs", type=int, default=None, help="dense blocks per weight-gradient launch")
    ap.add_argument("--wgrad-target-wgs", type=int, default=None, help="workgroups a trunk weight-gradient launch aims for")
    ap.add_argument("--wgrad-tail", type=str, default=None, help="dense blocks of the last weight-gradient launches, e.g. 3,2,1")
    ap.add_argument("--pair-wgrad-tasks", type=int, default=0,
                    help="pair the dense blocks' single-sub-block weight-gradient tasks (0/1)")
    ap.add_argument("--wgrad-walk-last", type=int, default=None,
                    help="column-walk kernel for the last N trunk weight-gradient launches")
    ap.add_argument("--deterministic-wgrad", type=int, default=0,
                    help="weight gradients reduced in a fixed order (per-slice partials) instead of atomics (0/1)")
    ap.add_argument("--adam-in-backward", type=int, default=0,
                    help="single process: Adam applied range by range beside backward (0/1)")
    ap.add_argument("--from-host", action="store_true",
                    help="diagnostic: every step copies its batch from pinned host memory (PCIe-inclusive rate; "
                         "not the headline value)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="diagnostic: run the gradient exchange over RCCL even with one rank")
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 and no launcher: print the rank commands instead of starting them")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-started ranks (0: a free one)")
    ap.add_argument("--rank-timeout", type=float, default=540.0,
                    help="with --gpus N > 1 and no launcher: seconds after which ranks that are still running are "
                         "stopped and the job fails (a rank that exits non-zero stops the others at once)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: start the ranks ourselves, before any GPU call here
        raise SystemExit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         f"(torch.distributed.run --nproc-per-node {args.gpus}) or let --gpus start them")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    exchange = world > 1 or args.force_exchange
    if exchange:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
        # Every stream of a step has the SAME (normal) priority: on this runtime a stream that waits
        # for an event of a lower-priority stream can keep that stream from ever being scheduled
        # (tools/stream_priority_probe.py).  The runtime gives each of the first four streams of a
        # priority a hardware queue of its own, so the weight-gradient stream is created before the
        # process group's: launch chain, weight gradients and collectives then never share a queue.
        importlib.import_module("super-resolution_amd.engine").side_stream(dev)
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=os.environ.get("SRX_PG_HIGH_PRIORITY") == "1")
        import datetime
        import faulthandler
        # a rank that stalls (in the rendezvous, a collective or a device wait) says where and ends,
        # so that its launcher can stop the other ranks, instead of sitting there until an outer limit
        faulthandler.dump_traceback_later(max(30.0, args.rank_timeout - 15.0), exit=True)
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, pg_options=opts,
                                timeout=datetime.timedelta(seconds=max(30.0, args.rank_timeout - 30.0)))
        assert dist.get_world_size() == world

    import ctypes
    G = importlib.import_module("super-resolution_amd.esrgan.generator")
    T = importlib.import_module("super-resolution_amd.train")
    E = importlib.import_module("super-resolution_amd.engine")
    D = importlib.import_module("super-resolution_amd.parallel")

    torch.manual_seed(0)
    gen = G.GeneratorESRGAN(num_blocks=args.blocks).to(dev)
    if args.workload == "c4":
        return bench_inference(args, dev, rank, world)
    adv = args.workload == "c3"
    B, h = args.batch, args.patch or (32 if adv else 64)
    sync = D.OverlappedGradAllReduce(world, bucket_mb=8.0, single_rank_too=True) if exchange else None
    if args.skip_wgrad:
        E.Plan.wgrad = lambda self, ops, *a, **k: None
    E.Plan.pair_wgrad_tasks = bool(args.pair_wgrad_tasks)
    E.Plan.deterministic_wgrad = bool(args.deterministic_wgrad)
    T.GeneratorWarmupTrainer.step_in_backward = bool(args.adam_in_backward)
    T.AdversarialTrainer.step_in_backward = bool(args.adam_in_backward)
    E.GeneratorPlan.overlap_wgrad = bool(args.wgrad_overlap)
    if args.wgrad_slices is not None:
        E.GeneratorPlan.trunk_wgrad_slices = args.wgrad_slices
    if args.wgrad_walk_last is not None:
        E.GeneratorPlan.trunk_wgrad_walk_last = args.wgrad_walk_last
    if args.wgrad_tail is not None:
        E.GeneratorPlan.trunk_wgrad_tail = tuple(int(t) for t in args.wgrad_tail.split(","))
    if args.wgrad_target_wgs is not None:
        E.GeneratorPlan.trunk_wgrad_target_wgs = args.wgrad_target_wgs
    if args.wgrad_rdbs is not None:
        E.GeneratorPlan.trunk_wgrad_rdbs = args.wgrad_rdbs
    if adv:
        import warnings
        Dm = importlib.import_module("super-resolution_amd.esrgan.discriminator")
        Lm = importlib.import_module("super-resolution_amd.utils.loss")
        disc = Dm.Discriminator().to(dev)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")            # synthetic VGG weights: there is no network
            vgg = Lm.VGGLoss(device=dev, allow_synthetic=True)
        sync_d = D.FlatGradAllReduce(world, single_rank_too=True) if exchange else None
        trainer = T.AdversarialTrainer(gen, disc, vgg, (B, 3, h, h), torch.bfloat16, grad_sync_g=sync,
                                       grad_sync_d=sync_d)
    else:
        trainer = T.GeneratorWarmupTrainer(gen, (B, 3, h, h), torch.bfloat16, grad_sync=sync,
                                           n_streams=args.streams)
    gcpu = torch.Generator().manual_seed(1234 + rank)
    lr = (torch.rand((B, 3, h, h), generator=gcpu) * 2 - 1).to(dev)
    hr = (torch.rand((B, 3, 4 * h, 4 * h), generator=gcpu) * 2 - 1).to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.from_host:
        lr_h, hr_h = lr.cpu().pin_memory(), hr.cpu().pin_memory()

        def one_step():
            trainer.step(lr_h.to(dev, non_blocking=True), hr_h.to(dev, non_blocking=True))
    else:
        def one_step():
            trainer.step(lr, hr)

    for _ in range(args.warmup):
        one_step()
    barrier()
    # device events between the steps (on the launch stream, where every step begins and ends):
    # the per-step median next to the wall-clock mean the headline value is computed from
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        one_step()
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = trainer.losses()["loss_gen"] if adv else trainer.mean_loss()

    px_per_step = world * B * h * h
    value = px_per_step * args.steps / dt / 1e6
    # valid for the 23-block generator
    tflops = value * 1e6 * (FLOP_PER_LR_PX_ADV if adv else FLOP_PER_LR_PX_FWD_BWD) / 1e12
    out = {
        "metric": "LR Mpixels/sec ESRGAN 4x train step (fwd+bwd)", "value": round(value, 4),
        "unit": "LR Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "ms_per_step_median_events": round(median_ms, 3),
        "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": ("INVALID: weight gradients skipped" if args.skip_wgrad else
                                                              "synthetic, copied from pinned host memory every step" if args.from_host
                                                              else "synthetic"),
        "config": {"workload": (f"config 3: adversarial step (D step + G step, 3 discriminator forwards, spectral "
                                f"norm, VGG loss, Adam on both), batch {B}/GPU of {h}x{h} LR patches, bf16") if adv
                               else ({(16, 64): "config 2", (64, 128): "config 5 shape"}.get((B, h), "custom shape") +
                                     f": 23-RRDB generator warm-up (L1), fwd+bwd+Adam, batch {B}/GPU "
                                     f"of {h}x{h} LR patches, bf16") if args.blocks == 23 else
                               f"{args.blocks}-RRDB generator warm-up, batch {B}, {h}x{h}",
                   "global_batch": world * B, "lr_patch": h, "parallelism": f"dp{world}",
                   ("final_loss_gen" if adv else "final_l1_loss"): round(loss, 5)},
        "algorithmic_tflops": round(tflops, 2) if args.blocks == 23 else None,
        "pct_bf16_mfma_peak": round(100 * tflops / (PEAK_BF16_TFLOPS * world), 2) if args.blocks == 23 else None,
    }

    if rank == 0 and not args.no_profile:
        stats = profile_ops(trainer.gplan if adv else trainer.plan, ctypes)
        tot = sum(v[0] for v in stats.values())
        name, (ms, flops, n) = max(stats.items(), key=lambda kv: kv[1][0])
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM-side bytes per launch of that kernel come from separate rocprofv3 --pmc passes over this
        # same command (counters cannot be read from inside the process); the summary is kept
        # under profiles/ with the command that produced it
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r3_pmc_traffic.json")) as f:
                k = json.load(f)["kernels"].get(name)
            if k and not adv and (B, h, args.blocks) == (16, 64, 23):
                traffic = k["fetch_bytes"] + k["write_bytes"]
        except (OSError, ValueError, KeyError):
            pass
        out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                           "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r3_pmc_traffic.json)",
                           "launches": n, "avg_us": round(ms * 1e3 / n, 2),
                           "share_of_step_kernel_time": round(ms / tot, 3)}
        out["kernel_breakdown_ms"] = {k: [round(v[0], 3), v[2], round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] > 0 else 0]
                                      for k, v in sorted(stats.items(), key=lambda kv: -kv[1][0])[:8]}
    if dist is not None:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    default_run = (world == 1 and not adv and (B, h, args.blocks) == (16, 64, 23) and not exchange
                   and not args.skip_wgrad)
    if rank == 0 and (args.extras if args.extras is not None else default_run):
        import gc
        del one_step, trainer, gen, lr, hr     # the headline step's buffers go back to the allocator first
        gc.collect()
        torch.cuda.empty_cache()
        out["extras"] = run_extras(dev)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        faulthandler.cancel_dump_traceback_later()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
