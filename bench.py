#!/usr/bin/env python3
"""bench.py -- throughput of the demix hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" = one pass of the hot path (asx_demix_dev: chunking -> STFT -> ConvTDFNet
-> iSTFT -> Hann fold -> result/divider) over ONE 4-minute 44.1 kHz stereo song
that is already resident in HBM, on the UVR-MDX-NET-Inst_HQ_3 geometry
(n_fft 6144, hop 1024, dim_f 3072, segment 256, overlap 0.25; ConvTDFNet g=48,
l=3, 11 blocks, bn=8) with seeded synthetic weights and input (no network for
checkpoints or datasets).  All arithmetic is fp32 (fp32-input MFMA).

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): every rank
demixes its own song (weak scaling: config 5 shards files across GPUs) and the
separated stems are gathered to rank 0 over xGMI inside the timed region.

Prints ONE JSON line on rank 0 (see the field notes in DESIGN.md).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 44100
SONG_SECONDS = 240
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
METRIC = "audio-sec separated / wall-sec (RTF), UVR-MDX-NET 44.1kHz stereo, 1/2/4/8 GPU"


def cpu_baseline(seconds: float, seed: int):
    """The CPU oracle (restatement of the reference path) timed on this host."""
    import torch
    from oracle import mdx_oracle as O
    # torch-CPU collapses when oversubscribed on the 256-core GPU host (0.07x RT at 256 threads);
    # 32 threads is where the conv-heavy net stops scaling.  `cores` reports what was used.
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    d = O.NetDims()
    sd = O.make_convtdf_state(d, seed=0)
    mix = O.synth_mix(int(SR * seconds), seed=seed)
    run = O.make_model_run(sd, d)
    t0 = time.perf_counter()
    out = O.demix(mix, O.MDXParams(), run)
    dt = time.perf_counter() - t0
    n_chunks = len(O.chunk_plan(mix.shape[1], O.MDXParams())[5])
    return out, mix, {"value": seconds / dt, "unit": "audio-s/wall-s", "cores": int(torch.get_num_threads()),
                      "kind": "port",
                      "sample": f"{seconds:g} s of the same synthetic song ({n_chunks} chunks), torch-CPU fp32 oracle, "
                                f"{dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=SONG_SECONDS)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="length of the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from oracle import mdx_oracle as O           # synthetic weights/input generator + cpu_baseline leg only
    import audio_separator_amd as A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_FORCE_DIST=1 exercises the RCCL code path (init, gather, all_reduce, barrier) with a single rank
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)

    # ---- engine + synthetic model --------------------------------------------------------
    d = O.NetDims()
    sd = O.make_convtdf_state(d, seed=0)
    eng = A.Engine(A.MDXConfig(max_batch=args.max_batch), device=local_rank)
    eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
    N = int(SR * args.seconds)
    mix = torch.from_numpy(O.synth_mix(N, seed=rank)).to(dev)       # resident in HBM before timing
    out = torch.empty_like(mix)
    gathered = [torch.empty_like(out) for _ in range(world)] if (use_dist and rank == 0) else None
    stream = torch.cuda.current_stream().cuda_stream
    plan = eng.plan(N)

    def step():
        eng.demix_dev(mix.data_ptr(), N, out.data_ptr(), stream=stream)
        if use_dist:
            dist.gather(out, gathered, dst=0)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * args.seconds * args.steps / dt

    # ---- roofline of the dominant kernel (3x3 TFC conv, MFMA bound), HIP events on the launch stream ----
    roofline = None
    prof = None
    if rank == 0:
        eng.profile_enable(True)
        eng.demix_dev(mix.data_ptr(), N, out.data_ptr(), stream=stream)
        prof = eng.profile_read()
        eng.profile_enable(False)
        c = prof["conv3x3"]
        ach = c["flops"] / (c["ms"] * 1e-3) / 1e12
        # HBM bytes per launch from the PMC passes kept under profiles/ (FETCH_SIZE x2 per the gfx950
        # correction + WRITE_SIZE), scaled by this run's algorithmic bytes per launch
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_conv3x3.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                traffic = round(json.load(fh)["traffic_over_algorithmic"] * c["bytes"] / max(1, c["launches"]), 1)
        roofline = {"kernel": "conv_dma_kernel<3,3,1,1,3,8,2,0> (TFC 3x3 convs)", "bound": "mfma", "achieved": round(ach, 2),
                    "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                    "traffic": traffic, "algorithmic_bytes_per_launch": c["bytes"] / max(1, c["launches"]),
                    "launches": c["launches"],
                    "avg_launch_ms": round(c["ms"] / max(1, c["launches"]), 4),
                    "flops_per_launch": c["flops"] / max(1, c["launches"]),
                    "share_of_step_ms": round(c["ms"], 2)}

    # ---- CPU baseline + on-the-fly parity of the same workload (rank 0, N = 1 only) ----
    cpu = None
    parity = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        ref, cmix, cpu = cpu_baseline(args.cpu_seconds, seed=0)
        g = eng.demix(cmix)
        parity = float(np.sqrt(np.mean((g.astype(np.float64) - ref) ** 2)) / np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
        cpu["value"] = round(cpu["value"], 3)

    if rank == 0:
        res = {
            "metric": METRIC, "value": round(value, 2), "unit": "audio-s/wall-s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "UVR-MDX-NET-Inst_HQ_3 geometry (n_fft 6144, hop 1024, dim_f 3072, segment 256, "
                                   "overlap 0.25; ConvTDFNet g48 l3 11 blocks bn8, synthetic weights), "
                                   f"{args.seconds:g}-s 44.1 kHz stereo song per GPU, input resident in HBM",
                       "samples_per_song": N, "chunks_per_song": plan["n_chunks"],
                       "songs_per_step": world, "parallelism": f"files sharded over {world} GPU(s)"
                       + (", stems gathered to rank 0 (RCCL)" if world > 1 else ""),
                       "samples_per_s": round(value * SR * 2, 1),
                       "net_tflops_per_s": round(eng.net_flops(plan["n_chunks"]) * world * args.steps / dt / 1e12, 2)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if parity is not None:
            res["parity_rel_rms_vs_cpu"] = float(f"{parity:.3e}")
        if prof is not None:
            res["kernel_ms"] = {k: round(v["ms"], 3) for k, v in prof.items() if v["launches"]}
            # achieved roofline fraction of every stage: MFMA-shaped classes against the fp32 MFMA peak, the
            # FFT / fold stages against HBM (algorithmic bytes / kernel time; spec 8 TB/s)
            stages = {}
            for k, v in prof.items():
                if not v["launches"] or v["ms"] <= 0:
                    continue
                tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
                gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                if tf / PEAK_FP32_MFMA_TFLOPS >= gb / 8000.0:   # the roof the class sits closer to is the one that binds it
                    stages[k] = {"bound": "mfma", "achieved": round(tf, 2), "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4)}
                else:
                    stages[k] = {"bound": "hbm", "achieved": round(gb, 1), "unit": "GB/s", "frac": round(gb / 8000.0, 4)}
            res["stage_roofline"] = stages
        print(json.dumps(res))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
