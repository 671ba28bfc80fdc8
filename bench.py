#!/usr/bin/env python3
"""Headline benchmark: WaveGlow forward audio-samples/s at batch 8 x 16000 per GPU
(BASELINE.json configs[2]; `config.json` defaults: 12 flows, n_group 8, WN 8 layers x 512 channels).

One "step" = one full WaveGlow.forward over one synthetic batch already resident in HBM,
including the per-forward weight-norm recompute + weight packing the reference also pays
(torch.nn.utils.weight_norm hooks, reference glow.py:123-151).

    python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU with torch.distributed.run; the forward pass of
independent batches needs no collective, so ranks only meet at the timing barriers ("weak" scaling).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fused
in+cond+gate GEMM), timed live with HIP events on the launch stream; `cpu_baseline` is the CPU
oracle (a port of the reference's torch op sequence) timed on this host at N=1.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from text2speech_amd import _lib, synth  # noqa: E402
from text2speech_amd.glow import WaveGlow  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA


def usable_cores():
    """CPU cores this process may really use: min(affinity, cgroup quota), not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return min(n, 16) if os.environ.get("T2S_BENCH_ALL_CORES") is None else n


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_baseline(cfg, sd, sample_batch=2, n_samples=16000, reps=3):
    """The oracle (oracle/waveglow_oracle.py, kind "port") on the host cores, bounded sample."""
    from oracle import waveglow_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    log("cpu baseline on %d threads" % cores)
    mel, audio = synth.waveglow_inputs(sample_batch, n_samples, seed=31)
    times = []
    with torch.no_grad():
        O.waveglow_forward(sd, cfg, mel[:1, :, :9], audio[:1, :2048])     # warm-up (thread pool, allocator)
        for _ in range(reps):
            t0 = time.perf_counter()
            O.waveglow_forward(sd, cfg, mel, audio)
            times.append(time.perf_counter() - t0)
            log("cpu baseline rep %.2f s" % times[-1])
    t = sorted(times)[len(times) // 2]
    return {"value": sample_batch * n_samples / t, "unit": "audio samples/s", "cores": cores, "kind": "port",
            "sample": "oracle forward on batch %d x %d (1/%d of the GPU batch), median of %d, fp32 torch CPU ops"
                      % (sample_batch, n_samples, 8 // sample_batch, reps)}


def tacotron_metrics(dev):
    """Second half of BASELINE.json's metric: Tacotron-2 mel-frames/s (autoregressive B=1; teacher-forced eval forward
    at the configs[1] shape B=32, T_in=256, T_out=800) and the decoder step's weight stream against the HBM roofline."""
    from text2speech_amd.tacotron import Tacotron
    hp = dict(synth.TACOTRON_HPARAMS)
    m = Tacotron(hp, 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state())
    m = m.to(dev).eval()
    out = {}
    ids = (torch.arange(64) % 78 + 2)[None].to(dev)
    n = 400
    m.decoder.gate_threshold, m.decoder.max_decoder_steps = 2.0, n
    m.inference(ids, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.inference(ids, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    dec = m.decoder
    A, D, E, P = dec.attention_rnn_dim, dec.decoder_rnn_dim, dec.encoder_embedding_dim, dec.prenet_dim
    lstm_bytes = 4.0 * (4 * A * (P + E + A) + 4 * D * (A + E + D))          # f32 LSTMCell weights streamed per step
    out["inference_B1"] = {"mel_frames_per_s": n / dt, "us_per_step": dt / n * 1e6, "frames": n, "symbols": 64}
    out["roofline"] = {"bound": "hbm", "kernel": "decoder step (2 x lstm_cell_kernel + attention + projection)",
                       "achieved": lstm_bytes / (dt / n) / 1e9, "peak": 8000.0, "unit": "GB/s",
                       "frac": lstm_bytes / (dt / n) / 1e9 / 8000.0,
                       "algorithmic_bytes_per_step": lstm_bytes,
                       "note": "71.3 MB of LSTMCell weights per step over the whole step time (5 dependent launches, "
                               "latency-bound at B=1); the two cell kernels alone stream them at ~3.9 TB/s"}
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
    out["forward_B32_Tin256_Tout800"] = {"mel_frames_per_s": B * T_out / dt, "ms": dt * 1e3}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--segment", type=int, default=16000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tacotron", action="store_true", help="skip the Tacotron mel-frames/s block (N=1 only)")
    ap.add_argument("--mode", choices=["forward", "train"], default="forward",
                    help="forward: the headline metric (default); train: zero_grad+forward+loss+backward+Adam, "
                         "data-parallel over RCCL when launched with N > 1 ranks (BASELINE configs[3])")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
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
    if args.mode == "train":
        from text2speech_amd.glow import WaveGlowLoss
        from text2speech_amd.optim import FusedAdam
        from text2speech_amd import distributed as D
        model.train()
        if dist is not None:
            D.apply_gradient_allreduce(model)
        opt = FusedAdam(model.parameters(), lr=1e-4)
        crit = WaveGlowLoss(1.0)

        def step():
            model.zero_grad(set_to_none=True)
            loss = crit(model((mel, audio)))
            loss.backward()
            opt.step()
            return loss
    else:
        def step():
            with torch.no_grad():
                model((mel, audio))
    run_train_or_fwd = step
    with torch.enable_grad() if args.mode == "train" else torch.no_grad():
        for _ in range(args.warmup):
            run_train_or_fwd()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        eng.gemm_events = [] if rank == 0 else None
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run_train_or_fwd()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    log("rank %d: %d steps in %.3f s" % (rank, args.steps, dt))
    if dist is not None:
        tt = torch.tensor([dt], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        evs = eng.gemm_events or []
        eng.gemm_events = None
        wn = cfg["WN_config"]
        C, ks = wn["n_channels"], wn["kernel_size"]
        n_cond = cfg["n_mel_channels"] * cfg["n_group"]
        L = args.segment // cfg["n_group"]
        flops_per_launch = 2.0 * (2 * C) * (ks * C + n_cond) * args.batch * L
        roof = None
        if evs and args.mode == "forward":
            ms = [a.elapsed_time(b) for a, b in evs]
            avg_ms = sum(ms) / len(ms)
            achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
            # HBM-side bytes per launch of this kernel come from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
            # --pmc WRITE_SIZE in separate runs, FETCH doubled per the gfx950 note): profiles/r01_pmc_traffic.json
            traffic = None
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_v5.json")))
                # the gate GEMM runs as two instantiations (shared-B tile for dilation <= 32, plain otherwise): launch-weighted mean
                gk = [v for k, v in pm["kernels"].items() if k.startswith("_Z16conv_gemm_kernelILi0E")]
                traffic = sum(v["traffic_bytes_per_launch"] * v["launches"] for v in gk) / sum(v["launches"] for v in gk)
            except (OSError, KeyError, ValueError, ZeroDivisionError):
                pass
            roof = {"bound": "mfma", "kernel": "conv_gemm_kernel<EPI_GATE> (in_layers+cond_layers+gate)",
                    "achieved": achieved, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / BF16_DENSE_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_note": "bytes/launch at batch 8x16000 from profiles/r01_pmc_traffic_v5.json (separate --pmc passes, tools/pmc_traffic.py; launch-weighted over the two gate-GEMM instantiations); "
                                    "compulsory bytes are 83 MB read + 33 MB written",
                    "avg_launch_ms": avg_ms, "launches": len(ms), "algorithmic_flops_per_launch": flops_per_launch,
                    "note": "split-bf16: 3 bf16 MFMA products per algorithmic MAC, so frac <= 1/3 by construction"}
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
        }
        if rehearse:
            out["rehearsal"] = "NOT A MEASUREMENT: %d ranks share %d GPU(s), collectives over %s" % (world, torch.cuda.device_count(), rehearse)
        if args.gpus == 1 and args.mode == "forward" and not args.no_tacotron:
            log("tacotron metrics")
            del model, eng
            torch.cuda.empty_cache()
            out["tacotron"] = tacotron_metrics(dev)
        if args.gpus == 1 and not args.no_cpu_baseline and args.mode == "forward":
            out["cpu_baseline"] = cpu_baseline(cfg, sd)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
