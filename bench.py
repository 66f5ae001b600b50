#!/usr/bin/env python3
"""bench.py -- throughput of the demix hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode files|chunks] [--songs-per-rank S]

A "step" = one pass of the hot path (asx_demix_dev: chunking -> STFT -> ConvTDFNet -> iSTFT -> Hann fold ->
result/divider) over 4-minute 44.1 kHz stereo songs that are already resident in HBM, on the UVR-MDX-NET-Inst_HQ_3
geometry (n_fft 6144, hop 1024, dim_f 3072, segment 256, overlap 0.25; ConvTDFNet g=48, l=3, 11 blocks, bn=8) with
seeded synthetic weights and input (no network for checkpoints or datasets).  All arithmetic is fp32 (fp32-input MFMA).

Ranks.  ``--gpus N`` with N > 1 launches N ranks itself (``python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ...``) unless it already runs under such a launcher (WORLD_SIZE set), in
which case WORLD_SIZE must equal N -- anything else is an error, never a silent 1-rank run.  One rank per GPU, RCCL.

  --mode files  (default; BASELINE config 5; weak scaling) every rank demixes its own ``--songs-per-rank`` songs
                (default 1; config 5 is 8 per rank on 8 GPUs = 64 songs) and the rank's stems go to rank 0 in ONE
                gather per step over xGMI, inside the timed region; the gather of step k overlaps the compute of step
                k + 1 (asynchronous collective on RCCL's stream, double-buffered stems).
  --mode chunks (strong scaling) ONE song, its chunk list split across the ranks (sharding.sharded_demix): contiguous
                chunk ranges, every rank folds its own sample range (one 2-MB seam chunk from the left neighbour), one gather
                of the folded [2, N / G] slabs to rank 0.

``--dry-gloo`` replaces the GPU work by a copy and RCCL by gloo so that the launcher / rendezvous / collective plumbing
can be exercised on a CPU-only box (tests/test_bench_launcher.py); its line says ``"dry": true`` and carries no rate.

Prints ONE JSON line on rank 0 (field notes in DESIGN.md 4).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 44100
SONG_SECONDS = 240
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the row GEMMs run six bf16 products per fp32 multiply-add)
METRIC = "audio-sec separated / wall-sec (RTF), UVR-MDX-NET 44.1kHz stereo, 1/2/4/8 GPU"
PMC_FILES = ("r05_pmc_conv3x3.json", "r03_pmc_conv3x3.json", "r02_pmc_conv3x3.json", "r01_pmc_conv3x3.json")
PMC_FILES_WINO = ("r05_pmc_wino3.json", "r04_pmc_wino3.json", "r03_pmc_wino3.json")
PMC_FILES_3H = ("r06_pmc_conv3h.json",)


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
    # the whole 4-minute song through the same oracle has been timed once per round (tools/fullsong_*.py); quote the stored runs
    whole = []
    for name, key in (("r02_fullsong_parity.json", None), ("r03_fullsong_parity.json", "cases"), ("r05_fullsong_parity.json", "cases")):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                rec = json.load(fh)
            c = rec["cases"]["mdx_hq3"]["cpu_oracle"] if key else rec["mdx_hq3_plain"]
            whole.append(f"{c['cpu_wall_s']:g} s at {c['cpu_threads']} threads (profiles/{name})")
        except Exception:
            pass
    note = ("; whole 4-min song on the same oracle: " + ", ".join(whole)) if whole else ""
    return out, mix, {"value": seconds / dt, "unit": "audio-s/wall-s", "cores": int(torch.get_num_threads()),
                      "kind": "port",
                      "kind_note": "oracle loop: oracle/mdx_oracle.py (the pinned restatement of MDXSeparator.demix + the reference's torch ConvTDFNet) -- "
                                   "the reference tree itself is absent on the GPU box, and onnxruntime is absent everywhere",
                      "sample": f"{seconds:g} s of the same synthetic song ({n_chunks} chunks), torch-CPU fp32 oracle, "
                                f"{dt:.1f} s wall" + note}


_JSON_FD = None


def emit(obj):
    """The result line, on the process's original stdout."""
    line = (json.dumps(obj) + "\n").encode()
    os.write(_JSON_FD if _JSON_FD is not None else 1, line)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=("files", "chunks"), default="files")
    ap.add_argument("--songs-per-rank", type=int, default=1)
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=SONG_SECONDS)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="length of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--siblings", type=int, default=1, help="1: append short htdemucs / BS-Roformer / VR / hdemucs lines (N = 1 only)")
    ap.add_argument("--file-level", type=int, default=1, help="1: time Separator-level separate(wav) -> stem files after the timed region (N = 1 only)")
    ap.add_argument("--no-graph", action="store_true", help="--mode chunks: launch the rank's chunk range eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-arith-ab", action="store_true", help="skip the fp32-exact arithmetic A/B after the timed region (N = 1 only)")
    ap.add_argument("--no-overlap", action="store_true", help="blocking gather (A/B of the gather / compute overlap)")
    ap.add_argument("--config5", action="store_true", help="BASELINE config 5 preset: --mode files --songs-per-rank 8 (64 songs on 8 GPUs)")
    ap.add_argument("--traffic", choices=("live", "stored"), default="live",
                    help="roofline.traffic: 'live' = two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a one-song child of this "
                         "script, inside this run (N = 1 only; falls back to 'stored' when rocprofv3 is missing or fails); "
                         "'stored' = the ratio kept under profiles/")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-gloo", action="store_true")
    ap.add_argument("--master-port", type=int, default=0)
    args = ap.parse_args(argv)
    if args.config5:
        args.mode, args.songs_per_rank = "files", 8
    return args


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """Re-exec under torch.distributed.run with one rank per GPU."""
    port = args.master_port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["ASX_BENCH_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def per_level_table(launch_recs, g, wino, num_blocks=11, conv3h=96):
    """The 3x3 TFC convs and the TDF row GEMMs of one profiled pass, grouped by U-Net level (VERDICT r4 next #2c).  A conv's level
    follows from its own algorithmic figures -- 3x3 conv c -> c over a plane P: flops = 18 c^2 P, bytes = 8 c P, so
    c = 4 flops / (9 bytes), level = c / g - 1; the 2 x num_blocks TDF launches come in the net's block order (encoder levels
    0 .. n-1, bottleneck n, decoder n-1 .. 0, two linears each).  Every launch record carries the number of 16-bit MFMA products per
    multiply-add its kernel executed (Engine.profile_launches_ex: 6 = bf16 x 6, 3 = fp16 x 3, 0 = fp32 MFMA), which names the kernel:
    a 3x3 launch on the 16-bit pipe is conv3h_kernel on the levels of 48 n <= `conv3h` channels (direct implicit GEMM: ALL of the convolution's
    FLOPs, each 432-deep slice reduction padded to 15 stages of 32) and conv_wino6_kernel elsewhere (Winograd F(2x2,3x3): 4/9 of them, input
    channels padded to whole 32-channel stages); on the fp32 pipe it is conv_wino3_kernel (4/9) or the direct conv_dma_kernel.
    Per level: launches, summed and average milliseconds, the algorithmic rate, the rate and fraction of the matrix peak the launches
    EXECUTE, and the time the launch's algorithmic bytes take at the 6.29 TB/s a device copy reaches."""
    def add(tab, key, ms, flops, nbytes, nprod):
        r = tab.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "nprod": set()})
        r["launches"] += 1
        r["ms"] += ms
        r["flops"] += flops
        r["bytes"] += nbytes
        r["nprod"].add(nprod)
    conv, tdf = {}, {}
    n = num_blocks // 2
    order = list(range(n)) + [n] + list(range(n - 1, -1, -1))
    tdf_recs = [r for r in launch_recs if r[0] == "tdf"]
    if len(tdf_recs) == 2 * num_blocks:
        for i, (_, ms, flops, nbytes, nprod) in enumerate(tdf_recs):
            add(tdf, (order[i // 2], i % 2), ms, flops, nbytes, nprod)
    for cls, ms, flops, nbytes, nprod in launch_recs:
        if cls == "conv3x3" and nbytes > 0:
            add(conv, int(round(4.0 * flops / (9.0 * nbytes) / g)) - 1, ms, flops, nbytes, nprod)
    out = {"conv3x3": {}, "tdf": {}}
    arith_name = {3: "fp16 x 3", 6: "bf16 x 6"}
    for lvl in sorted(conv):
        r = conv[lvl]
        c = g * (lvl + 1)
        tf = r["flops"] / (r["ms"] * 1e-3) / 1e12
        ent = {"channels": c, "launches": r["launches"], "ms": round(r["ms"], 3), "avg_launch_ms": round(r["ms"] / r["launches"], 4),
               "algorithmic_tflops": round(tf, 1), "algorithmic_gb_per_launch": round(r["bytes"] / r["launches"] / 1e9, 3),
               "hbm_floor_ms_per_launch": round(r["bytes"] / r["launches"] / 6.29e12 * 1e3, 3),
               "flops": r["flops"], "bytes": r["bytes"]}
        npr = max(r["nprod"]) if len(r["nprod"]) == 1 else -1
        if npr > 0 and wino and conv3h and c % 48 == 0 and c <= conv3h:
            pad = 480.0 / 432.0                       # 9 x 48 = 432 reduction elements in 15 stages of 32 (each kernel row ends on a half-empty stage)
            ent.update({"kernel": f"conv3h_kernel (direct implicit GEMM, weights resident in LDS, {arith_name[npr]}" + (f"; {(c // 48) ** 2} launches over 48-channel slices per layer)" if c > 48 else ")"), "nprod": npr,
                        "executed_tflops_16bit": round(tf * npr * pad, 1), "frac": round(tf * npr * pad / PEAK_BF16_MFMA_TFLOPS, 4), "peak": PEAK_BF16_MFMA_TFLOPS})
        elif npr > 0:
            # conv_wino6_kernel: 4/9 of the direct FLOPs as `npr` 16-bit products each, input channels padded to whole 32-channel stages
            pad = (-(-c // 32) * 32) / c
            ent.update({"kernel": f"conv_wino6_kernel (Winograd F(2x2,3x3), {arith_name[npr]})", "nprod": npr,
                        "executed_tflops_16bit": round(tf * (4.0 / 9.0) * npr * pad, 1),
                        "frac": round(tf * (4.0 / 9.0) * npr * pad / PEAK_BF16_MFMA_TFLOPS, 4), "peak": PEAK_BF16_MFMA_TFLOPS})
        elif npr == 0:
            exf = (4.0 / 9.0) if wino else 1.0
            ent.update({"kernel": "conv_wino3_kernel (Winograd F(2x2,3x3), fp32 MFMA)" if wino else "conv_dma_kernel<3,3,...> (direct, fp32 MFMA)", "nprod": 0,
                        "executed_tflops": round(tf * exf, 1), "frac": round(tf * exf / PEAK_FP32_MFMA_TFLOPS, 4), "peak": PEAK_FP32_MFMA_TFLOPS})
        else:
            ent.update({"kernel": "mixed (launches of one level on different kernels)", "nprod": sorted(r["nprod"])})
        out["conv3x3"][f"L{lvl}"] = ent
    for (lvl, which) in sorted(tdf):
        r = tdf[(lvl, which)]
        tf = r["flops"] / (r["ms"] * 1e-3) / 1e12
        ent = {"launches": r["launches"], "ms": round(r["ms"], 3), "avg_launch_ms": round(r["ms"] / r["launches"], 4), "fp32_equivalent_tflops": round(tf, 1),
               "algorithmic_gb_per_launch": round(r["bytes"] / r["launches"] / 1e9, 3),
               "hbm_floor_ms_per_launch": round(r["bytes"] / r["launches"] / 6.29e12 * 1e3, 3)}
        npr = max(r["nprod"]) if len(r["nprod"]) == 1 else -1
        if npr > 0:
            ent.update({"executed_tflops_16bit": round(npr * tf, 1), "frac": round(npr * tf / PEAK_BF16_MFMA_TFLOPS, 4), "arithmetic": arith_name[npr]})
        else:
            ent["frac"] = round(tf / PEAK_FP32_MFMA_TFLOPS, 4)
        out["tdf"][f"L{lvl}.{'F_to_F8' if which == 0 else 'F8_to_F'}"] = ent
    return out


def pmc_child(args):
    """What the live-traffic passes profile: one warm-up and one measured demix of the bench song, nothing else."""
    import torch
    from workload import synth as O              # seeded synthetic weights / song (neutral: not the checker package)
    import audio_separator_amd as A
    d = O.NetDims()
    sd = O.make_convtdf_state(d, seed=0)
    eng = A.Engine(A.MDXConfig(max_batch=args.max_batch), device=0)
    eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
    N = int(SR * args.seconds)
    mix = torch.from_numpy(O.synth_mix(N, seed=0)).to("cuda:0")
    out = torch.empty_like(mix)
    for _ in range(2):
        eng.demix_dev(mix.data_ptr(), N, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    eng.close()


def live_traffic(args, kernel_substr, per_layers=0, dispatches_per_pass=0):
    """HBM bytes per launch of the dominant kernel, measured in THIS run: rocprofv3 --kernel-trace --pmc <counter> around a
    one-song child of this script, one counter per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass),
    FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads), both in KiB.  -> (bytes per launch, launches seen, note)
    or (None, 0, why)."""
    import csv
    import glob
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, 0, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="asx_pmc_", dir="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [rp, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--seconds", str(args.seconds), "--max-batch", str(args.max_batch)]
            env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240)
            tot, cnt = 0.0, 0
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if kernel_substr in row.get("Kernel_Name", "") and row.get("Counter_Name") == ctr:
                            tot += float(row["Counter_Value"])
                            cnt += 1
            if cnt == 0:
                return None, 0, f"rocprofv3 pass {ctr}: rc {r.returncode}, no dispatch of {kernel_substr} in its output ({r.stderr.decode(errors='replace')[-200:]!r})"
            # per_layers: bytes per conv LAYER (several dispatches each); the child runs whole passes (a warm-up and a measured one): cnt / dispatches_per_pass of them
            vals[ctr] = (tot / (per_layers * (cnt / dispatches_per_pass) if per_layers > 0 and dispatches_per_pass > 0 else cnt), cnt)
    except Exception as e:                       # never take the headline line down
        return None, 0, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = vals["FETCH_SIZE"][0] * 1024 * 2, vals["WRITE_SIZE"][0] * 1024
    return fetch + write, vals["FETCH_SIZE"][1], (f"live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (one pass each) around a one-song child of "
                                                   f"bench.py inside this run, {vals['FETCH_SIZE'][1]} dispatches " + (f"({dispatches_per_pass} per pass) summed and divided by the passes and by {per_layers} layers" if per_layers > 0 else "averaged") + "; FETCH_SIZE x 2 per the gfx950 note "
                                                   f"of MI355X_MICROARCH.md (fetch {fetch / 1e9:.3f} GB + write {write / 1e9:.3f} GB per launch)")


def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(self_launch(args))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            sys.exit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}; refusing to report a mislabelled run")

    # exactly ONE line on stdout: RCCL prints a version banner to fd 1 at init, so everything else goes to stderr
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    if args.dry_gloo:
        return dry_run(args, world, rank)

    from workload import synth as O              # seeded synthetic weights / song; oracle/ is imported by cpu_baseline() only
    import audio_separator_amd as A
    from audio_separator_amd.sharding import HipEngineAdapter, sharded_demix

    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} visible")
    # BENCH_FORCE_DIST=1 exercises the RCCL code path (init, gather, all_reduce, barrier) with a single rank
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus
    dev = torch.device("cuda", local_rank)

    # ---- engine + synthetic model --------------------------------------------------------
    d = O.NetDims()
    sd = O.make_convtdf_state(d, seed=0)
    eng = A.Engine(A.MDXConfig(max_batch=args.max_batch), device=local_rank)
    eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
    N = int(SR * args.seconds)
    S = args.songs_per_rank if args.mode == "files" else 1
    stream = torch.cuda.current_stream().cuda_stream
    plan = eng.plan(N)
    if args.mode == "files":
        # resident in HBM before timing: S songs per rank, seeds rank * S + s (64 distinct songs at 8 x 8)
        from audio_separator_amd.sharding import FilesPipeline
        mixes = [torch.from_numpy(O.synth_mix(N, seed=rank * S + s)).to(dev) for s in range(S)]
        pipe = FilesPipeline(lambda m, o: eng.demix_dev(m.data_ptr(), N, o.data_ptr(), stream=stream), mixes, world, rank, use_dist,
                             overlap=not args.no_overlap)
        step, drain = pipe.step, pipe.drain
        songs_per_step = world * S
        scaling = "weak"
    else:
        from audio_separator_amd.sharding import ShardWorkspace
        mix = torch.from_numpy(O.synth_mix(N, seed=0)).to(dev)           # the same song on every rank
        adapter = HipEngineAdapter(eng)
        shard_ws = ShardWorkspace(graph=not args.no_graph)   # local / slab / out allocated once: the loop times compute + gather + fold; the rank's chunk-range compute replays a captured hipGraph from its third call on

        def step(k):
            sharded_demix(adapter, mix, workspace=shard_ws)

        def drain():
            pass
        songs_per_step = 1
        scaling = "strong"

    def fence():
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = songs_per_step * args.seconds * args.steps / dt

    # ---- the same K steps on the fp32-EXACT arithmetic (gemm_f16x3 = 0: every split-operand kernel multiplies exact three-way bf16 splits, six
    # products per multiply-add; conv3h_kernel's level falls back to conv_wino3_kernel), outside the timed region: both headlines are driver-timed
    arithmetic_ab = None
    if world == 1 and eng.option("gemm_bf16x6") > 0 and eng.option("gemm_f16x3") > 0 and not args.no_arith_ab:
        eng.set_option("gemm_f16x3", 0)
        for k in range(2):
            step(k)
        fence()
        t1 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        fence()
        dt6 = time.perf_counter() - t1
        eng.set_option("gemm_f16x3", 1)
        step(0)                                         # (the profiled pass below runs on the default again, its weight images are back)
        fence()
        arithmetic_ab = {"what": "the same timed loop with asx_set_option(gemm_f16x3, 0): six exact bf16 products per multiply-add (fp32-exact split operands) in the "
                                 "row GEMMs, attention and conv_wino6_kernel; the levels up to 144 channels on conv_wino3_kernel (fp32 MFMA) / conv_wino6_kernel instead of conv3h_kernel; "
                                 "the level-change convs (conv_down6_kernel / conv_up6_kernel) run six exact bf16 products in both legs",
                         "value": round(songs_per_step * args.seconds * args.steps / dt6, 2), "ms_per_step": round(dt6 / args.steps * 1e3, 3),
                         "steps": args.steps, "default_over_exact": round(dt6 / dt, 4)}

    # ---- what the exchange costs, outside the timed region (so that a sub-linear N > 1 result can be attributed) ----
    comm = {"gather_bytes_per_step": 0, "gather_ms": None, "fold_ms": None}
    if args.mode == "files":
        comm["gather_bytes_per_step"] = pipe.gather_bytes_per_step
        if use_dist:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            dist.barrier()
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(3):
                dist.gather(pipe.outs[0], pipe.gathered[0], dst=0)          # blocking: ordered with the current stream
            ev[1].record()
            torch.cuda.synchronize()
            comm["gather_ms"] = round(ev[0].elapsed_time(ev[1]) / 3, 3)
            comm["gather_ms_note"] = "one blocking dist.gather of a step's stems, hipEvents on the current stream, mean of 3; in the timed loop it overlaps the next step"
    else:
        shard_ws.timed = True
        sharded_demix(adapter, mix, workspace=shard_ws)
        shard_ws.timed = False
        if shard_ws.timings:
            comm["gather_ms"] = round(shard_ws.timings["gather_ms"], 3)
            comm["fold_ms"] = round(shard_ws.timings["fold_ms"], 3)
            comm["compute_ms"] = round(shard_ws.timings["compute_ms"], 3)
        comm["graph"] = {"enabled": shard_ws.graph, "replays": shard_ws.graph_replays, "error": shard_ws.graph_error}
        cs = plan["chunk_size"]
        # local fold: (world - 1) seam chunks point-to-point + the folded slabs of the other ranks to rank 0
        comm["gather_bytes_per_step"] = (world - 1) * 2 * cs * 4 + int(2 * N * 4 * (world - 1) / world)
        if use_dist:
            dist.barrier()

    # ---- roofline of the dominant kernel (3x3 TFC conv, MFMA bound), HIP events on the launch stream ----
    roofline = None
    prof = None
    if rank == 0:
        m0 = mixes[0] if args.mode == "files" else mix
        o0 = torch.empty_like(m0)
        eng.profile_enable(True)
        eng.demix_dev(m0.data_ptr(), N, o0.data_ptr(), stream=stream)
        prof = eng.profile_read()
        launch_recs = eng.profile_launches_ex()
        eng.profile_enable(False)
        wino = eng.option("winograd") > 0
        x6 = eng.option("gemm_bf16x6") > 0
        w6c = eng.option("winograd_bf16x6") if (eng.option("winograd") == 3 and x6) else 0
        c3h = eng.option("conv_direct_f16x3") if (eng.option("winograd") == 3 and x6 and eng.option("gemm_f16x3") > 0) else 0
        per_level = per_level_table(launch_recs, d.g, wino, d.num_blocks, c3h)
        # the dominant kernel: conv_wino3_kernel (fp32 MFMA) on the levels neither conv3h_kernel (48 channels: level 0) nor conv_wino6_kernel
        # (from the winograd_bf16x6 channel count up) takes -- level 1 of the HQ_3 net, ~48 of the ~120 ms the 3x3 class takes, the largest
        # single kernel of the step; every level is listed in per_level with the kernel that ran it
        lv = per_level["conv3x3"]
        dom3h = [v for v in lv.values() if v["kernel"].startswith("conv3h")]
        dom3 = [v for v in lv.values() if v.get("nprod") == 0]
        is3h = bool(dom3h) and sum(v["ms"] for v in dom3h) >= sum(v["ms"] for v in dom3)
        dom = dom3h if is3h else (dom3 or list(lv.values()))
        c = {"flops": sum(v["flops"] for v in dom), "bytes": sum(v["bytes"] for v in dom), "ms": sum(v["ms"] for v in dom),
             "launches": sum(v["launches"] for v in dom)}
        for v in lv.values():
            v.pop("flops")
            v.pop("bytes")
        call = prof["conv3x3"]
        ach = c["flops"] / (c["ms"] * 1e-3) / 1e12
        # HBM bytes per launch (= per conv LAYER: a 96-channel layer is four dispatches of conv3h_kernel): measured live below (rocprofv3 PMC passes,
        # FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, summed over the kernel's dispatches and divided by the layers) or, failing that,
        # the ratio traffic / algorithmic bytes of the PMC record kept under profiles/ applied to this run's algorithmic bytes; traffic_source says which.
        traffic, source = None, None
        for name in (PMC_FILES_3H if is3h else PMC_FILES_WINO if wino else PMC_FILES):
            pmc_path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc_path):
                with open(pmc_path) as fh:
                    traffic = round(json.load(fh)["traffic_over_algorithmic"] * c["bytes"] / max(1, c["launches"]), 1)
                source = f"stored: profiles/{name} (rocprofv3 --pmc passes), ratio applied to this run's algorithmic bytes"
                break
        # `achieved` / `frac` describe what the matrix pipe EXECUTES (<= 1 by construction).  conv3h_kernel: three fp16 products per multiply-add of
        # the direct convolution, each 432-deep slice reduction in 15 stages of 32 (x 480 / 432), against the dense 16-bit peak.  The Winograd
        # kernels run 16 multiply-adds per 2 x 2 output tile and channel pair instead of the direct convolution's 36, i.e. 4/9 of the algorithmic
        # FLOPs.  The direct-convolution (algorithmic) rate -- SURVEY 8d's per-unit figure over the same launch time -- is kept under
        # `algorithmic`; it can exceed a peak and is NOT a roofline fraction.
        exf = 3.0 * 480.0 / 432.0 if is3h else (4.0 / 9.0) if wino else 1.0
        peak = PEAK_BF16_MFMA_TFLOPS if is3h else PEAK_FP32_MFMA_TFLOPS
        if world == 1 and args.traffic == "live":
            live, nlaunch, why = live_traffic(args, "conv3h_kernel" if is3h else "conv_wino3_kernel" if wino else "conv_dma_kernel<asx::ConvDmaCfg<3, 3, 1, 1",
                                              per_layers=c["launches"] if is3h else 0,
                                              dispatches_per_pass=sum(v["launches"] * (v["channels"] // 48) ** 2 for v in dom3h) if is3h else 0)
            if live is not None:
                traffic, source = round(live, 1), why
            else:
                source = (source or "") + f" [live measurement unavailable: {why}]"
        roofline = {"kernel": (f"conv3h_kernel (TFC 3x3 convs of the levels up to {c3h} channels: direct implicit GEMM on the fp16 pipe, three products on two-part "
                               "operands, 48 x 48 weight slices resident in LDS)" if is3h else
                               "conv_wino3_kernel (TFC 3x3 convs of the levels below "
                               f"{w6c} channels, Winograd F(2x2,3x3) on fp32 MFMA)" if (wino and w6c) else
                               "conv_wino3_kernel (TFC 3x3 convs, Winograd F(2x2,3x3) on fp32 MFMA)" if wino
                               else "conv_dma_kernel<3,3,1,1,3,4,2,0> (TFC 3x3 convs)"),
                    "bound": "mfma", "achieved": round(ach * exf, 2),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(ach * exf / peak, 4),
                    "traffic": traffic, "traffic_source": source,
                    "algorithmic_bytes_per_launch": c["bytes"] / max(1, c["launches"]),
                    "launches": c["launches"],
                    "avg_launch_ms": round(c["ms"] / max(1, c["launches"]), 4),
                    "flops_per_launch": c["flops"] / max(1, c["launches"]) * exf,
                    "share_of_step_ms": round(c["ms"], 2),
                    "conv3x3_class_ms": round(call["ms"], 2), "conv3x3_class_launches": call["launches"]}
        if is3h:
            roofline["launch_is"] = "one conv layer (a 96-channel layer = four dispatches of the kernel over 48-channel slices); hbm floor of the launches: per_level"
            roofline["peak_note"] = ("peak = the guide's dense 16-bit MFMA figure.  Measured on this chip (profiles/r06_micro_mfma.txt, not re-measured by this run): a pure "
                                     "stream of v_mfma_f32_16x16x32_f16 reaches 2412 TFLOP/s on zero operands at 2.37 GHz and 1750-1920 TFLOP/s on operands that toggle the "
                                     "datapath (clock 1.91-1.97 GHz) -- the sustained ceiling of the pipe with nothing else running")
        roofline["per_level"] = per_level
        if wino:
            roofline["note"] = (("achieved / frac = EXECUTED 16-bit MFMA FLOPs (3 products x 480 / 432 stage padding per multiply-add of the direct convolution) / launch time "
                                 "/ dense 16-bit peak; " if is3h else
                                 "achieved / frac = EXECUTED MFMA FLOPs (Winograd F(2x2,3x3): 4/9 of the direct convolution's) / launch time / fp32-MFMA peak; ") +
                                "`algorithmic` = direct-convolution FLOPs over the same time (fp32-equivalent; against the fp32 peak it may exceed 1 and is not a "
                                "roofline fraction); `direct_kernel` = the fp32-MFMA non-Winograd kernel measured in this same run")
            roofline["algorithmic"] = {"flops_per_launch": c["flops"] / max(1, c["launches"]), "achieved": round(ach, 2),
                                       "unit": "TFLOP/s", "over_peak": round(ach / PEAK_FP32_MFMA_TFLOPS, 4)}
            mode = eng.option("winograd")
            eng.set_option("winograd", 0)
            eng.demix_dev(m0.data_ptr(), N, o0.data_ptr(), stream=stream)
            eng.profile_enable(True)
            eng.demix_dev(m0.data_ptr(), N, o0.data_ptr(), stream=stream)
            dprof = eng.profile_read()["conv3x3"]
            eng.profile_enable(False)
            eng.set_option("winograd", mode)
            dach = dprof["flops"] / (dprof["ms"] * 1e-3) / 1e12
            roofline["direct_kernel"] = {"kernel": "conv_dma_kernel<3,3,1,1,3,4,2,0>", "achieved": round(dach, 2), "unit": "TFLOP/s",
                                         "frac": round(dach / PEAK_FP32_MFMA_TFLOPS, 4), "share_of_step_ms": round(dprof["ms"], 2)}

    # ---- CPU baseline + on-the-fly parity of the same workload (rank 0, N = 1 only) ----
    cpu = None
    parity = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        ref, cmix, cpu = cpu_baseline(args.cpu_seconds, seed=0)
        g = eng.demix(cmix)
        parity = float(np.sqrt(np.mean((g.astype(np.float64) - ref) ** 2)) / np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
        cpu["value"] = round(cpu["value"], 3)

    if rank == 0:
        par = (f"files sharded over {world} GPU(s), {S} song(s) per rank" + (", stems gathered to rank 0 (one RCCL gather per step"
               + (", blocking" if args.no_overlap else ", overlapped with the next step's compute") + ")" if world > 1 else "")) \
            if args.mode == "files" else f"chunks of one song sharded over {world} GPU(s), seam chunk to the right neighbour, local fold, one RCCL gather of [2, N / G] slabs"
        res = {
            "metric": METRIC, "value": round(value, 2), "unit": "audio-s/wall-s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": ("f32 (I/O and accumulators; matrix products on the 16-bit pipe from block-scaled two-part fp16 operands, 22-bit products -- see arithmetic; "
                      "fp32-exact setting timed in arithmetic_ab)" if (eng.option("gemm_bf16x6") > 0 and eng.option("gemm_f16x3") > 0) else
                      "f32 (I/O and accumulators; matrix products from exactly split bf16 operands or fp32 MFMA -- see arithmetic)"),
            "data": "synthetic",
            "config": {"workload": "UVR-MDX-NET-Inst_HQ_3 geometry (n_fft 6144, hop 1024, dim_f 3072, segment 256, "
                                   "overlap 0.25; ConvTDFNet g48 l3 11 blocks bn8, synthetic weights), "
                                   f"{args.seconds:g}-s 44.1 kHz stereo song(s), input resident in HBM"
                                   + (f"; BASELINE config 5 preset: batch of {songs_per_step} songs sharded across {world} GPU(s) with one RCCL gather per step"
                                      if args.config5 else ""),
                       "mode": args.mode, "samples_per_song": N, "chunks_per_song": plan["n_chunks"],
                       "songs_per_step": songs_per_step, "parallelism": par,
                       "samples_per_s": round(value * SR * 2, 1),
                       "net_tflops_per_s": round(eng.net_flops(plan["n_chunks"]) * songs_per_step * args.steps / dt / 1e12, 2)},
            "rccl": dict({"world_size": dist.get_world_size() if use_dist else 1, "backend": dist.get_backend() if use_dist else None,
                          "launcher": "self" if os.environ.get("ASX_BENCH_LAUNCHED") == "1" else ("external" if "WORLD_SIZE" in os.environ else None)},
                         **comm),
            "roofline": roofline, "cpu_baseline": cpu, "arithmetic_ab": arithmetic_ab,
            # what "f32" means inside: nothing runs in a reduced-precision mode
            "arithmetic": {"io": "float32",
                           "conv3x3": "48-channel level: direct implicit GEMM on the fp16 pipe, two-part operands, one running power-of-two exponent per walk down T "
                                      "(csrc/kernels_conv3h.h; same arithmetic as row_gemm); other levels below the winograd_bf16x6 channel count (default 144): "
                                      "Winograd F(2x2,3x3) on fp32 MFMA (exact fma chains); from there up: Winograd on split operands on the 16-bit pipe (csrc/kernels_wino6.h)",
                           "attention": ("as row_gemm (one exponent per query, per 64-key tile of K, a running one per tile of V, none for the probabilities)"
                                         if (eng.option("gemm_bf16x6") > 0 and eng.option("gemm_f16x3") > 0) else
                                         "six bf16 products on exactly split operands" if eng.option("gemm_bf16x6") > 0 else "fp32 MFMA"),
                           "row_gemm": ("three fp16 MFMA products on fp32 operands scaled per block by a power of two and split into two fp16 parts each "
                                        "(11 + 11 significand bits, dropped term < 2^-22 of a product; as close to a float64 GEMM as the fp32-MFMA "
                                        "kernel, tests/test_gpu_parity.py::test_rowgemm_bf16x6_vs_float64[f16x3], ::test_rowgemm_f16x3_block_exponent)")
                           if (eng.option("gemm_bf16x6") > 0 and eng.option("gemm_f16x3") > 0) else
                           ("six bf16 MFMA products on fp32 operands split EXACTLY into three bf16 parts each (dropped cross "
                            "terms < 2^-24 of a product; closer to a float64 GEMM than the fp32-MFMA kernel, "
                            "tests/test_gpu_parity.py::test_rowgemm_bf16x6_vs_float64)") if eng.option("gemm_bf16x6") > 0
                           else "fp32 MFMA (gemm_bf16x6 = 0)",
                           "switch": "ASX_GEMM_BF16X6 / asx_set_option(gemm_bf16x6); ASX_GEMM_F16X3 / asx_set_option(gemm_f16x3)"},
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
                # the level-change convs run six bf16 products per multiply-add on the 16-bit pipe since round 6 (csrc/kernels_updown6.h): price them there
                ud6 = (k == "down" and eng.option("conv_down_bf16x6") > 0 and eng.option("gemm_bf16x6") > 0) or \
                      (k == "up" and eng.option("conv_up_bf16x6") > 0 and eng.option("gemm_bf16x6") > 0)
                mfma_frac = tf * 6.0 / PEAK_BF16_MFMA_TFLOPS if ud6 else tf / PEAK_FP32_MFMA_TFLOPS
                if ud6 and mfma_frac >= gb / 8000.0:
                    stages[k] = {"bound": "mfma", "achieved": round(tf * 6.0, 1), "unit": "TFLOP/s", "peak": PEAK_BF16_MFMA_TFLOPS, "frac": round(mfma_frac, 4),
                                 "dtype": "bf16 x 6 products (fp32-exact split operands)", "fp32_equivalent": round(tf, 2)}
                elif ud6:
                    stages[k] = {"bound": "hbm", "achieved": round(gb, 1), "unit": "GB/s", "frac": round(gb / 8000.0, 4), "frac_vs_copy": round(gb / 6290.0, 4),
                                 "mfma": {"executed_tflops_16bit": round(tf * 6.0, 1), "frac": round(mfma_frac, 4),
                                          "dtype": "bf16 x 6 products (fp32-exact split operands)", "fp32_equivalent": round(tf, 2)}}
                elif tf / PEAK_FP32_MFMA_TFLOPS >= gb / 8000.0:   # the roof the class sits closer to is the one that binds it
                    if k == "conv3x3" and eng.option("winograd") > 0 and w6c > 0:
                        # two kernels share the class since round 5: conv_wino3_kernel (fp32 MFMA) below `w6c` channels -- its executed rate
                        # and fraction are `roofline.achieved / frac` -- and conv_wino6_kernel (bf16 x 6) from there up; per level: roofline.per_level
                        stages[k] = {"bound": "mfma", "achieved": roofline["achieved"], "unit": "TFLOP/s", "frac": roofline["frac"], "peak": roofline["peak"],
                                     "frac_is": "the launches of the class's dominant kernel only (roofline.kernel; executed MFMA FLOPs over that pipe's peak); every level "
                                                "with the kernel that ran it: roofline.per_level",
                                     "algorithmic_achieved": round(tf, 2)}
                    elif k == "conv3x3" and eng.option("winograd") > 0:   # executed MFMA rate (4/9 of the algorithmic one, see roofline)
                        stages[k] = {"bound": "mfma", "achieved": round(tf * 4.0 / 9.0, 2), "unit": "TFLOP/s",
                                     "frac": round(tf * 4.0 / 9.0 / PEAK_FP32_MFMA_TFLOPS, 4), "algorithmic_achieved": round(tf, 2)}
                    elif k == "tdf" and eng.option("gemm_bf16x6") > 0:
                        # csrc/kernels_gemm3.h: six bf16 MFMA products per fp32 multiply-add on exactly split operands.  achieved /
                        # frac = EXECUTED bf16 FLOPs (6 x the GEMM's) against the dense bf16 peak; fp32_equivalent = the GEMM's own
                        # FLOPs over the same time (what the fp32-MFMA kernel would have to reach: its peak is 157.3)
                        # gemm_f16x3 (round 5, default): three fp16 products instead -- HALF the executed FLOPs for the same GEMM, so
                        # `frac` (executed / peak) falls where the kernel got faster; fp32_equivalent is the comparable figure
                        h3 = eng.option("gemm_f16x3") > 0
                        npr = 3.0 if h3 else 6.0
                        stages[k] = {"bound": "mfma", "achieved": round(tf * npr, 1), "unit": "TFLOP/s", "peak": PEAK_BF16_MFMA_TFLOPS,
                                     "frac": round(tf * npr / PEAK_BF16_MFMA_TFLOPS, 4),
                                     "dtype": ("fp16 x 3 products (block-scaled two-way split operands)" if h3
                                               else "bf16 x 6 products (fp32-exact split operands)"),
                                     "fp32_equivalent": round(tf, 2), "fp32_equivalent_over_fp32_peak": round(tf / PEAK_FP32_MFMA_TFLOPS, 4)}
                    else:
                        stages[k] = {"bound": "mfma", "achieved": round(tf, 2), "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4)}
                else:
                    # frac: of the 8 TB/s spec; frac_vs_copy: of the 6.29 TB/s a device copy reaches
                    stages[k] = {"bound": "hbm", "achieved": round(gb, 1), "unit": "GB/s", "frac": round(gb / 8000.0, 4),
                                 "frac_vs_copy": round(gb / 6290.0, 4)}
            res["stage_roofline"] = stages
        if world == 1 and (args.siblings or args.file_level):
            eng.close()
        if world == 1 and args.file_level:
            try:
                res["file_level"] = file_level_line(args, sd)
            except Exception as e:                  # never take the headline line down
                res["file_level"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.siblings:
            res["siblings"] = sibling_lines(args)
        emit(res)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def file_level_line(args, sd):
    """File-to-files rate of the plugin surface (SURVEY.md 8d "separately reported"; the reference logs it at separator.py:1016,
    1043): ``MDXSeparator.separate(song.wav)`` -> two PCM16 stem files, through ``install()``'s class, on a 4-minute PCM16 WAV
    in tmpfs.  Wall clock of whole calls (after one warm-up call), with the per-phase breakdown the class records when
    ``asx_profile_file`` is set (each phase fenced by a stream synchronise, so the phases sum to the wall time), and the same
    call with the device-resident path switched off (ASX_FILE_FASTPATH=0: host decode, host stems, re-upload for int16) as
    the A/B.  Not part of `value`."""
    import logging
    import shutil
    import tempfile
    import audio_separator_amd as A
    from audio_separator_amd import audio_io
    from audio_separator_amd.architectures.mdx_separator import MDXSeparator
    from workload import synth as O
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="asx_file_level_", dir=base)
    try:
        n = int(SR * args.seconds)
        mix = O.synth_mix(n, seed=0)
        wav = os.path.join(tmp, "song.wav")
        audio_io.write_wav(wav, np.ascontiguousarray(mix.T), SR, "PCM_16")
        log = logging.getLogger("bench.file_level")
        log.setLevel(logging.ERROR)
        common = {"logger": log, "log_level": logging.ERROR, "torch_device": "cuda:0", "torch_device_cpu": "cpu", "torch_device_mps": None,
                  "onnx_execution_provider": ["ROCMExecutionProvider"], "model_name": "UVR-MDX-NET-Inst_HQ_3", "model_path": None,
                  "model_data": {"compensate": 1.022, "mdx_dim_f_set": 3072, "mdx_dim_t_set": 8, "mdx_n_fft_scale_set": 6144,
                                 "primary_stem": "Instrumental"},
                  "output_format": "WAV", "output_bitrate": None, "output_dir": os.path.join(tmp, "out"),
                  "normalization_threshold": 0.9, "amplification_threshold": 0.0, "output_single_stem": None, "invert_using_spec": False,
                  "sample_rate": SR, "use_soundfile": False, "asx_state_dict": sd, "asx_net_config": A.NetConfig(), "asx_profile_file": True}
        arch = {"hop_length": 1024, "segment_size": 256, "overlap": 0.25, "batch_size": 1, "enable_denoise": False}
        sep = MDXSeparator(common, arch)

        def run(calls):
            walls, phases = [], {}
            for _ in range(calls):
                t0 = time.perf_counter()
                files = sep.separate(wav)
                walls.append(time.perf_counter() - t0)
                for k, v in sep.file_timings.items():
                    phases[k] = phases.get(k, 0.0) + v
                sep.clear_gpu_cache()
                sep.clear_file_specific_paths()
            return files, walls, {k: v / calls for k, v in phases.items()}

        run(1)                                              # warm-up: workspaces, pinned staging, page cache
        files, walls, phases = run(3)
        wall = sum(walls) / len(walls)
        sizes = [os.path.getsize(os.path.join(common["output_dir"], f)) for f in files]
        os.environ["ASX_FILE_FASTPATH"] = "0"
        try:
            run(1)
            _, walls_h, _ = run(2)
        finally:
            os.environ.pop("ASX_FILE_FASTPATH", None)
        wall_h = sum(walls_h) / len(walls_h)
        sep.engine.close()
        ph = {k: round(v * 1e3, 2) for k, v in phases.items()}
        return {"what": "MDXSeparator.separate(4-min PCM16 WAV on tmpfs) -> 2 PCM16 stem files (Instrumental, Vocals), plugin class of install()",
                "rtf": round(args.seconds / wall, 1), "wall_ms": round(wall * 1e3, 2), "calls": len(walls),
                "phases_ms": ph, "phases_sum_ms": round(sum(ph.values()), 2),
                "unaccounted_ms": round(wall * 1e3 - sum(ph.values()), 2),
                "files": files, "file_bytes": sizes,
                "host_path": {"what": "same call with ASX_FILE_FASTPATH=0 (round-2 path: host decode, float stems to the host, "
                                      "re-upload per stem for the int16 pass)",
                              "rtf": round(args.seconds / wall_h, 1), "wall_ms": round(wall_h * 1e3, 2)}}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def sibling_lines(args):
    """Short runs of the sibling segment loops (BASELINE configs 0, 2, 3 + hdemucs_mmi) after the timed region, so that
    their rates are observed by whoever runs bench.py and not only claimed: 1 warm-up + 2 steps of a 4-minute song each,
    roofline fraction of the dominant kernel class from the in-engine hipEvent profile.  Not part of `value`."""
    import types
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_siblings as BS
    out = {}
    a = types.SimpleNamespace(seconds=float(SONG_SECONDS), steps=2, warmup=1, cpu=0)
    for name, fn in (("htdemucs", BS.run_htdemucs), ("bs_roformer", BS.run_roformer), ("vr", BS.run_vr), ("hdemucs_mmi", BS.run_hdemucs)):
        try:
            r = fn(a)
            out[name] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "workload": r["config"]["workload"],
                         "roofline": {k: r["roofline"][k] for k in ("kernel", "achieved", "unit", "frac", "share_of_step_ms",
                                                                     "mfma_bound_launches", "hbm_bound_launches", "stage_roofline")
                                      if k in r["roofline"]},
                         "net_tflops_per_s": r["config"].get("net_tflops_per_s")}
        except Exception as e:                      # a sibling must never take the headline line down
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return out


def dry_run(args, world, rank):
    """Launcher / rendezvous / collective plumbing on CPU (gloo), through the SAME FilesPipeline (double-buffered stems, one
    asynchronous gather per step, buffer reuse only after its gather completed) as the GPU run; the engine is a stand-in
    whose output encodes (rank, song, step), and rank 0 checks that every gathered buffer holds exactly its step's data."""
    import torch
    import torch.distributed as dist
    from audio_separator_amd.sharding import FilesPipeline
    use_dist = world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.mode == "chunks":
        return dry_run_chunks(args, world, rank, use_dist)
    n = 4096
    S = args.songs_per_rank
    mixes = [torch.full((2, n), float(100 * rank + s)) for s in range(S)]
    calls = [0]

    def demix(mix, out):                      # stand-in hot path: out = mix + 10000 * (step + 1); S calls per step
        step = calls[0] // S
        calls[0] += 1
        out.copy_(mix).add_(10000.0 * (step + 1))

    seen, bad = [], []

    def on_gathered(step, slabs):
        seen.append(step)
        for r, slab in enumerate(slabs):
            for s in range(S):
                want = 100.0 * r + s + 10000.0 * (step + 1)
                if not bool((slab[s] == want).all()):
                    bad.append((step, r, s, float(slab[s].flatten()[0]), want))

    pipe = FilesPipeline(demix, mixes, world, rank, use_dist, overlap=not args.no_overlap, on_gathered=on_gathered)
    total = args.warmup + args.steps
    for k in range(total):
        pipe.step(k)
    pipe.drain()
    ok = True
    if use_dist:
        dist.barrier()
        if rank == 0:
            ok = not bad and seen == list(range(total))
    if rank == 0:
        emit({"metric": METRIC, "value": None, "unit": "audio-s/wall-s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry": True, "mode": args.mode, "gather_ok": ok,
                          "gathers_checked": len(seen), "gather_mismatches": bad[:4],
                          "rccl": {"world_size": dist.get_world_size() if use_dist else 1, "backend": "gloo" if use_dist else None,
                                   "launcher": "self" if os.environ.get("ASX_BENCH_LAUNCHED") == "1" else
                                   ("external" if "WORLD_SIZE" in os.environ else None),
                                   "gather_bytes_per_step": pipe.gather_bytes_per_step}})
    if use_dist:
        dist.destroy_process_group()


def dry_run_chunks(args, world, rank, use_dist):
    """--mode chunks on CPU: the SAME sharded_demix (chunk ranges, seam chunks to the right neighbour, local fold of the own sample range,
    one gather of [2, N / G] slabs, ShardWorkspace reused over the steps) as the GPU run, over gloo, with a stand-in engine: 55 chunks of
    4096 samples at stride 3072, each the song's own samples under a strictly positive window, folded by a per-sample gather with the
    analytic divider -- so the folded song must come back equal to the input on rank 0, whichever rank computed which chunk.  The
    workspace poisons (NaN) every chunk slot a rank does not hold: a fold that read a foreign chunk would show."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from audio_separator_amd.sharding import ShardWorkspace, sharded_demix, partition_chunks

    class StandIn:
        local_fold = True
        C, STEP, NK = 4096, 3072, 55

        def __init__(self):
            self.w = (0.25 + torch.hann_window(self.C, periodic=False, dtype=torch.float64)).to(torch.float32)
            self.calls = 0

        def plan(self, n):
            return {"chunk_size": self.C, "n_chunks": self.NK, "step": self.STEP, "padded_len": (self.NK - 1) * self.STEP + self.C, "trim": 0}

        def demix_chunks(self, mix, n, k0, k1, out):
            self.calls += 1
            pad = torch.zeros((2, (self.NK - 1) * self.STEP + self.C), dtype=torch.float32)
            pad[:, :n] = mix
            for k in range(k0, k1):
                out[k - k0].copy_(pad[:, k * self.STEP:k * self.STEP + self.C] * self.w)

        def finalize(self, chunks, n, out):
            L = (self.NK - 1) * self.STEP + self.C
            acc = torch.zeros((2, L), dtype=torch.float64)
            div = torch.zeros(L, dtype=torch.float64)
            for k in range(self.NK):                  # a per-sample gather in chunk order: foreign (NaN) chunks land only outside the rank's range
                acc[:, k * self.STEP:k * self.STEP + self.C] += chunks[k].to(torch.float64)
                div[k * self.STEP:k * self.STEP + self.C] += self.w.to(torch.float64)
            out.copy_((acc / div)[:, :n].to(torch.float32))

    ad = StandIn()
    n = (ad.NK - 1) * ad.STEP + ad.C - 1000
    g = torch.Generator().manual_seed(0)
    mix = torch.randn((2, n), generator=g, dtype=torch.float32)         # the same song on every rank
    ws = ShardWorkspace(poison=True)
    total = args.warmup + args.steps
    worst, ok = 0.0, True
    for k in range(total):
        out = sharded_demix(ad, mix, workspace=ws)
        if rank == 0:
            ok = ok and out is not None and bool(torch.isfinite(out).all())
            worst = max(worst, float((out - mix).abs().max())) if out is not None else float("inf")
        else:
            ok = ok and out is None
    if use_dist:
        dist.barrier()
    ranges = partition_chunks(ad.NK, world)
    if rank == 0:
        emit({"metric": METRIC, "value": None, "unit": "audio-s/wall-s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "dry": True, "mode": "chunks", "fold_ok": bool(ok and worst < 1e-5), "fold_max_abs_err": worst, "calls": total,
              "demix_calls_rank0": ad.calls, "chunk_ranges": ranges, "scaling": "strong",
              "rccl": {"world_size": dist.get_world_size() if use_dist else 1, "backend": "gloo" if use_dist else None,
                       "launcher": "self" if os.environ.get("ASX_BENCH_LAUNCHED") == "1" else ("external" if "WORLD_SIZE" in os.environ else None),
                       "gather_bytes_per_step": (world - 1) * 2 * ad.C * 4 + int(2 * n * 4 * (world - 1) / world)}})
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
