#!/usr/bin/env python3
"""Headline benchmark: WaveGlow forward audio-samples/s at batch 8 x 16000 per GPU
(BASELINE.json configs[2]; `config.json` defaults: 12 flows, n_group 8, WN 8 layers x 512 channels).

One "step" = one full WaveGlow.forward over one synthetic batch already resident in HBM,
including the per-forward weight-norm recompute + weight packing the reference also pays
(torch.nn.utils.weight_norm hooks, reference glow.py:123-151).

    python bench.py --gpus N --steps K --warmup W

For N > 1 either the driver launches one rank per GPU with torch.distributed.run, or - when the
command above is run directly (no RANK in the environment) - this script starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process before it
touches the GPU and exits with the child's code (the reference ships the same kind of launcher,
waveglow/distributed.py:145-170).  The forward pass of independent batches needs no collective, so
ranks only meet at the timing barriers ("weak" scaling).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fused in+cond+gate GEMM),
timed live with HIP events on the launch stream; `cpu_baseline` is the CPU oracle (a port of the
reference's torch op sequence) timed on this host at N=1 with BASELINE.md section 4's protocol.
Extra blocks on the same line: `tacotron` (mel-frames/s, the second half of BASELINE.json's metric),
`waveglow_train` / `tacotron_train` (BASELINE configs[3] / configs[1] train steps at N=1) and, at
N > 1, `waveglow_train_dp` (the data-parallel train step over RCCL, BASELINE configs[3]).
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s HBM3E
WG_FWD_FLOP_PER_SAMPLE = 65.36e6  # 8.3666 TFLOP per 8 x 16000 forward
TACO_FWD_FLOP_PER_FRAME = 52.1e6  # 1.334 TFLOP per 32 x 800 teacher-forced forward
GEMM_PMC = "r04_pmc_traffic.json"  # profiles/: HBM-side bytes per launch of the gate GEMM (separate --pmc passes)


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--segment", type=int, default=16000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tacotron", action="store_true", help="skip the Tacotron mel-frames/s block (N=1 only)")
    ap.add_argument("--no-train", action="store_true", help="skip the train-step blocks")
    ap.add_argument("--event-stride", type=int, default=7,
                    help="time every n-th gate-GEMM launch with a HIP event pair inside the timed region (roofline.avg_launch_ms); "
                         "1 = every launch, 0 = none (no roofline block).  7 is coprime with the 8 layers of a flow, so every layer / dilation "
                         "is sampled; an event pair around EVERY launch costs 0.66 ms per forward (3.6 percent), every 7th 0.1 ms")
    ap.add_argument("--mode", choices=["forward", "train"], default="forward",
                    help="forward: the headline metric (default); train: value = zero_grad+forward+loss+backward+Adam, "
                         "data-parallel over RCCL when launched with N > 1 ranks (BASELINE configs[3])")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks as a child process tree.  Runs before this
    process has made any GPU call; it never replaces itself (exec of a GPU-initialised process is not allowed on this pool)."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("no RANK in the environment: launching %d ranks: %s" % (args.gpus, " ".join(cmd[1:])))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def usable_cores():
    """(threads this process may really use, os.cpu_count(), why): min(os.cpu_count(), affinity mask, cgroup CPU quota).
    No other cap: oversubscribing a cgroup quota makes the torch CPU path slower, not faster."""
    total = os.cpu_count() or 1
    n, why = total, "os.cpu_count()"
    try:
        aff = len(os.sched_getaffinity(0))
        if aff < n:
            n, why = aff, "sched_getaffinity"
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    q = max(1, int(int(txt[0]) / int(txt[1])))
                    if q < n:
                        n, why = q, "cgroup cpu.max"
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    if max(1, q // per) < n:
                        n, why = max(1, q // per), "cgroup cfs quota"
        except (OSError, ValueError, IndexError):
            pass
    return n, total, why


def cpu_baseline(cfg, sd, batch, n_samples, reps=5):
    """the oracle (oracle/waveglow_oracle.py, kind "port": the reference's torch op sequence) on the
    host cores, fp32, no_grad, the SAME 8 x 16000 batch shape as the GPU step, one same-shape warm-up call, median of 5."""
    import torch
    from oracle import waveglow_oracle as O
    from text2speech_amd import synth
    cores, total, why = usable_cores()
    torch.set_num_threads(cores)
    log("cpu baseline on %d threads (%s; host reports %d)" % (cores, why, total))
    mel, audio = synth.waveglow_inputs(batch, n_samples, seed=31)
    times = []
    with torch.no_grad():
        t0 = time.perf_counter()
        O.waveglow_forward(sd, cfg, mel, audio)                              # warm-up, same shape
        log("cpu baseline warm-up %.2f s" % (time.perf_counter() - t0))
        for _ in range(reps):
            t0 = time.perf_counter()
            O.waveglow_forward(sd, cfg, mel, audio)
            times.append(time.perf_counter() - t0)
            log("cpu baseline rep %.2f s" % times[-1])
    t = sorted(times)[len(times) // 2]
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": batch * n_samples / t, "unit": "audio samples/s", "cores": cores, "kind": "port",
            "sample": "oracle forward on the full batch %d x %d, one same-shape warm-up then the median of %d calls "
                      "(%.2f s each), fp32 torch CPU ops, %d threads = min(os.cpu_count()=%d, affinity, cgroup quota) [%s]"
                      % (batch, n_samples, reps, t, cores, total, why),
            "cpu_model": model, "host_cpu_count": total}


def tacotron_metrics(dev):
    """Second half of BASELINE.json's metric: Tacotron-2 mel-frames/s (autoregressive B=1; teacher-forced eval forward
    at the configs[1] shape B=32, T_in=256, T_out=800) and the decoder step's weight stream against the HBM roofline."""
    import torch
    from text2speech_amd import synth
    from text2speech_amd.tacotron import Tacotron
    hp = dict(synth.TACOTRON_HPARAMS)
    m = Tacotron(hp, 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state())
    m = m.to(dev).eval()
    out = {}
    ids = (torch.arange(64) % 78 + 2)[None].to(dev)
    # BASELINE configs[4] front half: B=1, 64 symbols, 1000 forced frames.  A second length separates the decode step
    # (slope) from the per-utterance work (intercept: encoder convolutions + BiLSTM, postnet, stop-flag syncs).
    m.decoder.gate_threshold = 2.0
    times = {}
    for n in (200, 1000):
        m.decoder.max_decoder_steps = n
        m.inference(ids, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m.inference(ids, None)
        torch.cuda.synchronize()
        times[n] = (time.perf_counter() - t0) / 3
    n = 1000
    dt = times[n]
    slope = (times[1000] - times[200]) / 800.0
    dec = m.decoder
    A, D, E, P = dec.attention_rnn_dim, dec.decoder_rnn_dim, dec.encoder_embedding_dim, dec.prenet_dim
    lstm_bytes = 4.0 * (4 * A * (P + E + A) + 4 * D * (A + E + D))          # f32 LSTMCell weights streamed per step
    out["inference_B1"] = {"mel_frames_per_s": n / dt, "us_per_step": dt / n * 1e6, "frames": n, "symbols": 64,
                           "decoder": getattr(m._eng(), "decoder_kind", "launch chain"),
                           "ms_200_frames": times[200] * 1e3, "ms_1000_frames": times[1000] * 1e3,
                           "decode_us_per_step": slope * 1e6,
                           "per_utterance_ms": (times[200] - 200 * slope) * 1e3}
    out["roofline"] = {"bound": "hbm", "kernel": "decoder step (2 LSTM cells + attention + projection)",
                       "achieved": lstm_bytes / (dt / n) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": lstm_bytes / (dt / n) / 1e9 / HBM_PEAK_GBS,
                       "algorithmic_bytes_per_step": lstm_bytes,
                       "decode_step_only": {"achieved": lstm_bytes / slope / 1e9, "frac": lstm_bytes / slope / 1e9 / HBM_PEAK_GBS,
                                            "note": "the same bytes over the decode step alone (slope between 200 and 1000 frames)"},
                       "note": "71.3 MB of f32 LSTMCell weights per decoder step (what the reference streams) over the whole "
                               "step time, encoder + postnet included.  Since round 4 the step is four launches: attention cell "
                               "(+ the sparse prenet layer), attention + gate-stream role (50 MB of the weights on the CUs the "
                               "attention leaves idle), decoder cell, projection"}
    try:        # HBM-side bytes of the four launches of a decode step, from the committed PMC passes (not measured in this run)
        pm = json.load(open(os.path.join(ROOT, "profiles", "r04_decode_pmc_traffic.json")))["kernels"]
        names = ("att_fused_mfma_kernelILb1ELb1", "lstm_cell_p2_kernel", "lstm_cell_kernelILi1ELi4ELb0", "gemv_rows_loc_kernelILi7")
        per = [next(v["traffic_bytes_per_launch"] for k, v in pm.items() if n in k) for n in names]
        out["roofline"]["traffic"] = sum(per)
        out["roofline"]["traffic_note"] = ("FROM THE COMMITTED PROFILE, NOT MEASURED IN THIS RUN: profiles/r04_decode_pmc_traffic.json "
                                           "(separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 note): attention + "
                                           "gate stream %.1f MB, attention cell %.1f, decoder cell %.1f, projection %.1f per launch"
                                           % tuple(x / 1e6 for x in per))
    except (OSError, KeyError, ValueError, StopIteration):
        out["roofline"]["traffic"] = None
    B, T_in, T_out = 32, 256, 800
    gen = torch.Generator().manual_seed(21)
    text = torch.randint(2, 80, (B, T_in), generator=gen).to(dev)
    mel = torch.randn(B, 80, T_out, generator=gen).to(dev)
    il = torch.full((B,), T_in, dtype=torch.long, device=dev)
    ol = torch.full((B,), T_out, dtype=torch.long, device=dev)
    inp = (text, il, mel, T_in, torch.zeros(B, device=dev), ol)
    m(inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        m(inp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    out["forward_B32_Tin256_Tout800"] = {"mel_frames_per_s": B * T_out / dt, "ms": dt * 1e3,
                                         "achieved_TFLOPs": B * T_out * TACO_FWD_FLOP_PER_FRAME / dt / 1e12}
    return out


def tacotron_train_metrics(dev, steps=4, warmup=2):
    """BASELINE configs[1]: Tacotron-2 train step, B=32, T_in=256, T_out=800, teacher-forced: zero_grad -> forward (training
    mode, device-drawn dropout) -> Tacotron2Loss -> hand-written backward -> FusedAdam (reference train.py:216-225)."""
    import torch
    from text2speech_amd import synth
    from text2speech_amd.optim import FusedAdam
    from text2speech_amd.tacotron import Tacotron, Tacotron2Loss
    B, T_in, T_out = 32, 256, 800
    m = Tacotron(dict(synth.TACOTRON_HPARAMS), 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state())
    m = m.to(dev).train()
    opt = FusedAdam(list(m.parameters()), lr=1e-4, weight_decay=1e-6)           # train.py:187-189
    gen = torch.Generator().manual_seed(21)
    text = torch.randint(2, 80, (B, T_in), generator=gen).to(dev)
    mel = torch.randn(B, 80, T_out, generator=gen).to(dev)
    gate = torch.zeros(B, T_out, device=dev)
    gate[:, -1] = 1
    il = torch.full((B,), T_in, dtype=torch.long, device=dev)
    ol = torch.full((B,), T_out, dtype=torch.long, device=dev)
    inp = (text, il, mel, T_in, torch.zeros(B, device=dev), ol)
    crit = Tacotron2Loss()

    def step():
        m.zero_grad(set_to_none=True)
        loss = crit(m(inp), (mel, gate))
        loss.backward()
        opt.step()
        return loss

    first = None
    for _ in range(warmup):
        l = step()
        first = float(l) if first is None else first
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        l = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flop = 3.0 * B * T_out * TACO_FWD_FLOP_PER_FRAME
    return {"workload": "Tacotron-2 train step B=32, T_in=256, T_out=800 (BASELINE configs[1]), fixed shapes",
            "ms_per_step": dt * 1e3, "mel_frames_per_s": B * T_out / dt, "steps": steps, "warmup": warmup,
            "algorithmic_TFLOP_per_step": flop / 1e12, "achieved_TFLOPs": flop / dt / 1e12,
            "dtype": "f32 (LSTM cells, attention: exact f32 MFMA / VALU) + bf16x3 (convolutions)",
            "loss_first": first, "loss_last": float(l), "max_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}


WATCHDOG_EXIT_CODE = 3


def watchdog_fired(out, key, rank, seconds, emit=None, leave=None):
    """The train block's timer went off (a collective that never completes, a hung kernel).  The headline line measured before
    it is still printed - with the error recorded under `key` - but the process LEAVES NON-ZERO: a run that gave up must not
    read as a pass (VERDICT r3 "other").  `emit` / `leave` are injectable for the CPU test."""
    emit = emit or (lambda line: print(line, flush=True))
    leave = leave or os._exit
    log("rank %d: train block exceeded %d s: giving up on it (exit code %d)" % (rank, seconds, WATCHDOG_EXIT_CODE))
    if out is not None:
        out[key] = {"error": "timeout after %d s" % seconds}
        out["error"] = "watchdog: %s did not finish within %d s; process exited with code %d" % (key, seconds, WATCHDOG_EXIT_CODE)
        emit(json.dumps(out))
    leave(WATCHDOG_EXIT_CODE)


def run_steps(step, steps, warmup, dist, grad):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides."""
    import torch
    with torch.enable_grad() if grad else torch.no_grad():
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    from text2speech_amd import _lib, synth
    from text2speech_amd.glow import WaveGlow, WaveGlowLoss

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d; launch with torch.distributed.run --nproc-per-node %d "
                         "(or run `python bench.py --gpus %d` without a launcher)" % (world, args.gpus, args.gpus, args.gpus))
    _lib.load()
    # T2S_BENCH_REHEARSE=gloo: functional rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (ranks share
    # the visible devices, collectives over gloo).  The JSON line is tagged; it is not a measurement.
    rehearse = os.environ.get("T2S_BENCH_REHEARSE", "")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1 or os.environ.get("T2S_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints a version banner on stdout when the communicator is created; stdout must carry exactly one
        # JSON line, so fd 1 points at stderr until the communicator exists.
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearse:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    cfg = synth.WAVEGLOW_DEFAULT
    sd = synth.waveglow_state(cfg)
    model = WaveGlow(**cfg)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    mel, audio = synth.waveglow_inputs(args.batch, args.segment, seed=1234 + rank)
    mel, audio = mel.to(dev), audio.to(dev)
    eng = model._eng()
    log("rank %d: model built, starting warm-up" % rank)

    def make_train_step():
        from text2speech_amd.optim import FusedAdam
        from text2speech_amd import distributed as D
        model.train()
        if dist is not None:
            D.apply_gradient_allreduce(model)
        opt = FusedAdam(model.parameters(), lr=1e-4)           # waveglow/train.py:79
        crit = WaveGlowLoss(1.0)

        def step():
            model.zero_grad(set_to_none=True)
            loss = crit(model((mel, audio)))
            loss.backward()
            opt.step()
            return loss
        step.optimizer = opt
        return step

    def fwd_step():
        model((mel, audio))

    def max_over_ranks(dt):
        if dist is None:
            return dt
        tt = torch.tensor([dt], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    step = make_train_step() if args.mode == "train" else fwd_step
    # the gate GEMM's HIP events are recorded on the launch stream during the timed steps only
    eng.gemm_events = None
    with torch.enable_grad() if args.mode == "train" else torch.no_grad():
        for _ in range(args.warmup):
            step()
    torch.cuda.synchronize()
    eng.gemm_events = [] if (rank == 0 and args.mode == "forward" and args.event_stride > 0) else None
    eng.gemm_event_stride = max(1, args.event_stride)
    dt = max_over_ranks(run_steps(step, args.steps, 0, dist, args.mode == "train"))
    evs = eng.gemm_events or []
    eng.gemm_events = None
    log("rank %d: %d steps in %.3f s" % (rank, args.steps, dt))

    out = None
    if rank == 0:
        wn = cfg["WN_config"]
        C, ks = wn["n_channels"], wn["kernel_size"]
        n_cond = cfg["n_mel_channels"] * cfg["n_group"]
        L = args.segment // cfg["n_group"]
        flops_per_launch = 2.0 * (2 * C) * (ks * C + n_cond) * args.batch * L
        roof = None
        if evs:
            ms = [a.elapsed_time(b) for a, b in evs]
            avg_ms = sum(ms) / len(ms)
            achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
            # HBM-side bytes per launch of this kernel come from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
            # --pmc WRITE_SIZE in separate runs, FETCH doubled per the gfx950 note in MI355X_MICROARCH.md)
            traffic, tsrc = None, None
            for name in (GEMM_PMC, "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic_v5.json"):
                try:
                    pm = json.load(open(os.path.join(ROOT, "profiles", name)))
                    # the gate GEMM: the ping-pong kernel (round 2), or the two lockstep instantiations of round 1
                    gk = [v for k, v in pm["kernels"].items() if k.startswith("_Z19gate_gemm_pp_kernel")] or \
                         [v for k, v in pm["kernels"].items() if k.startswith("_Z16conv_gemm_kernelILi0E")]
                    traffic = sum(v["traffic_bytes_per_launch"] * v["launches"] for v in gk) / sum(v["launches"] for v in gk)
                    tsrc = name
                    break
                except (OSError, KeyError, ValueError, ZeroDivisionError):
                    continue
            roof = {"bound": "mfma", "kernel": "gate_gemm_pp_kernel (in_layers + cond_layers + tanh*sigmoid gate, ping-pong schedule)",
                    "achieved": achieved, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / BF16_DENSE_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_note": "FROM THE COMMITTED PROFILE, NOT MEASURED IN THIS RUN: bytes/launch at batch 8x16000 from profiles/%s (separate --pmc FETCH_SIZE / WRITE_SIZE passes, "
                                    "FETCH doubled per the gfx950 note, tools/pmc_traffic.py); algorithmic bytes are 83 MB read + "
                                    "33 MB written; the fetch side is fabric requests: 74 MB of activations once plus the 8.9 MB "
                                    "of packed weights once per XCD (8 private L2s)" % tsrc,
                    "avg_launch_ms": avg_ms, "launches": len(ms),
                    "launches_note": "HIP-event pairs around every %d-th of the %d gate-GEMM launches of the timed region"
                                     % (max(1, args.event_stride), wn["n_layers"] * cfg["n_flows"] * args.steps),
                    "algorithmic_flops_per_launch": flops_per_launch,
                    "note": "split-bf16: 3 bf16 MFMA products per algorithmic MAC, so frac <= 1/3 by construction; in-kernel the "
                            "chip holds 1.85-1.95 GHz under this load (profiles/r02_summary.md), i.e. a 660 TFLOP/s ceiling for "
                            "this scheme, and the K loop runs at 91 % of MFMA-bound"}
        total_samples = args.gpus * args.batch * args.segment * args.steps
        out = {
            "metric": "WaveGlow forward audio samples/sec (batch 8x16000 per GPU)" if args.mode == "forward"
            else "WaveGlow train-step audio samples/sec (batch 8x16000 per GPU, fwd+loss+bwd+Adam, DP all-reduce)",
            "value": total_samples / dt, "unit": "audio samples/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16x3 (split-bf16 products, f32 accumulate)",
            "data": "synthetic (seeded N(0,1) mel, U(-0.5,0.5) audio; seeded random weights, WN.end ~ N(0,0.02^2))",
            "config": {"workload": "WaveGlow forward, batch %d x %d samples per GPU, 12 flows, n_group 8, "
                                   "WN 8 layers x 512 channels (reference waveglow/config.json)" % (args.batch, args.segment),
                       "per_gpu_batch": args.batch, "segment_length": args.segment, "mode": args.mode,
                       "parallelism": "replicas, no collective" if args.mode == "forward"
                       else "dp%d, 13 bucketed RCCL all-reduces overlapped with backward" % args.gpus},
            "roofline": roof,
            "achieved_TFLOPs_whole_step": total_samples * WG_FWD_FLOP_PER_SAMPLE * (3.0 if args.mode == "train" else 1.0) / dt / 1e12,
        }
        if args.mode == "train" and eng.grad_sync is not None:       # per-bucket all-reduce timing of the last step (rank 0's view)
            out["collective"] = eng.grad_sync.stats()
    # ---- data-parallel WaveGlow train step (BASELINE configs[3]) next to the forward figure, every rank takes part ----
    train_block, train_key = None, None
    if args.mode == "forward" and not args.no_train:
        train_key = "waveglow_train" if world == 1 else "waveglow_train_dp"

        def give_up():      # a collective that never completes must not cost the headline line: print it, leave non-zero
            watchdog_fired(out, train_key, rank, 240)
        guard = threading.Timer(240.0, give_up)
        guard.daemon = True
        guard.start()
        try:
            tstep = make_train_step()
            k_tr, w_tr = 6, 2
            single_ms = None
            if world > 1 and eng.grad_sync is not None:
                # the same step WITHOUT the gradient exchange, in this process on this GPU: DP step time minus this is the
                # communication the backward could not hide.  The ranks' weights and Adam moments drift apart during these
                # steps (different data, no averaging), so rank 0's are broadcast again afterwards: the timed DP steps run on
                # synchronised replicas
                gs_keep, eng.grad_sync = eng.grad_sync, None
                single_ms = max_over_ranks(run_steps(tstep, 3, 2, dist, True)) / 3 * 1e3
                eng.grad_sync = gs_keep
                from text2speech_amd import distributed as D
                with torch.no_grad():
                    tensors = [p for p in model.state_dict().values() if torch.is_tensor(p)]
                    for st_ in tstep.optimizer.state.values():
                        tensors += [v for k_, v in st_.items() if torch.is_tensor(v) and v.is_cuda]
                    D._broadcast_flat(tensors)
                    torch.autograd.graph.increment_version(list(model.parameters()))
            dtt = max_over_ranks(run_steps(tstep, k_tr, w_tr, dist, True))
            n_samp = world * args.batch * args.segment
            flop = 3.0 * args.batch * args.segment * WG_FWD_FLOP_PER_SAMPLE          # train step ~ 3 x forward
            train_block = {"workload": "WaveGlow train step (zero_grad, forward with saves, WaveGlowLoss, hand-written backward, "
                                       "FusedAdam), batch %d x %d per GPU, config.json defaults" % (args.batch, args.segment),
                           "n_gpus": world, "ms_per_step": dtt / k_tr * 1e3, "audio_samples_per_s": n_samp * k_tr / dtt,
                           "steps": k_tr, "warmup": w_tr, "algorithmic_TFLOP_per_step_per_gpu": flop / 1e12,
                           "achieved_TFLOPs_per_gpu": flop / (dtt / k_tr) / 1e12,
                           "frac_of_bf16_peak": flop / (dtt / k_tr) / 1e12 / BF16_DENSE_PEAK_TFLOPS,
                           "parallelism": "single GPU" if world == 1 else
                           "dp%d: 13 flat gradient buckets all-reduced (RCCL AVG) from inside the backward" % world}
            gs = eng.grad_sync
            if gs is not None:
                torch.cuda.synchronize()
                st_ = gs.stats()
                train_block["allreduce_bytes_per_step"] = sum(b["bytes"] for b in st_["buckets"])
                train_block["collective"] = {"backend": st_["backend"], "ranks_in_group": st_["world"],
                                             "buckets_last_step": st_["buckets"],
                                             "note": "per bucket: bytes, ms from 'last gradient of the bucket written' to 'all-reduce "
                                                     "complete' (events on the communication stream), ms since the previous bucket "
                                                     "completed; buckets ship in backward order: flows 11 .. 0, then the upsampler"}
                if single_ms is not None:
                    train_block["single_gpu_ms_per_step_same_process"] = single_ms
                    train_block["exposed_comm_ms_per_step"] = dtt / k_tr * 1e3 - single_ms
        except Exception as e:      # noqa: BLE001  (the headline line must still be printed)
            train_block = {"error": "%s: %s" % (type(e).__name__, e)}
        guard.cancel()
        tstep = None
        model.eval()
        model.zero_grad(set_to_none=True)

    if rank == 0:
        if rehearse:
            out["rehearsal"] = "NOT A MEASUREMENT: %d ranks share %d GPU(s), collectives over %s" % (world, torch.cuda.device_count(), rehearse)
        if train_block is not None:
            out[train_key] = train_block
        if args.gpus == 1 and args.mode == "forward":
            if not args.no_tacotron:
                # the vocoder direction (BASELINE configs[4] back half): B = 1, reverse flow, weights packed once
                try:
                    inf = {}
                    with torch.no_grad():
                        for frames in (200, 1000):
                            melv = torch.randn(1, cfg["n_mel_channels"], frames, generator=torch.Generator().manual_seed(frames)).to(dev)
                            for _ in range(2):
                                model.infer(melv, sigma=0.666)
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            for _ in range(5):
                                model.infer(melv, sigma=0.666)
                            torch.cuda.synchronize()
                            dti = (time.perf_counter() - t0) / 5
                            inf["frames_%d" % frames] = {"ms": dti * 1e3, "audio_samples_per_s": frames * 256 / dti}
                    out["waveglow_infer_B1"] = inf
                except Exception as e:      # noqa: BLE001
                    out["waveglow_infer_B1"] = {"error": "%s: %s" % (type(e).__name__, e)}
            del model, eng
            torch.cuda.empty_cache()
            if not args.no_tacotron:
                log("tacotron metrics")
                out["tacotron"] = tacotron_metrics(dev)
                if not args.no_train:
                    log("tacotron train step")
                    torch.cuda.empty_cache()
                    try:
                        out["tacotron_train"] = tacotron_train_metrics(dev)
                    except Exception as e:      # noqa: BLE001
                        out["tacotron_train"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(cfg, sd, args.batch, args.segment)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
