#!/usr/bin/env python3
"""Throughput of the sibling segment loops (BASELINE.json configs 0, 2, 3 and MDX23C) in bench.py's JSON schema.

    python tools/bench_siblings.py [--workloads vr,htdemucs,hdemucs,roformer,mdx23c] [--seconds 240] [--steps 3] [--cpu 1]

One JSON line per workload: whole-song RTF with the input resident in HBM, the roofline fraction of the kernel class
that dominates the step (in-engine hipEvent profile), and the CPU oracle timed on a bounded sample beside it.
bench.py (the driver's contract) stays on the headline MDX metric; these lines are kept under profiles/.
"""
import argparse
import json
import os
import sys
import time
from fractions import Fraction

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_separator_amd as A  # noqa: E402

SR = 44100
PEAK = 157.3
PEAK_BF16 = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md)
VR_MP = {"bins": 768, "unstable_bins": 7, "reduction_bins": 668, "sr": 44100, "pre_filter_start": 740, "pre_filter_stop": 768,
         "band": {1: {"sr": 11025, "hl": 128, "n_fft": 1024, "crop_start": 0, "crop_stop": 186, "lpf_start": 37, "lpf_stop": 73, "res_type": "polyphase"},
                  2: {"sr": 11025, "hl": 128, "n_fft": 512, "crop_start": 4, "crop_stop": 185, "hpf_start": 36, "hpf_stop": 18, "lpf_start": 93, "lpf_stop": 185, "res_type": "polyphase"},
                  3: {"sr": 22050, "hl": 256, "n_fft": 512, "crop_start": 46, "crop_stop": 186, "hpf_start": 93, "hpf_stop": 46, "lpf_start": 164, "lpf_stop": 186, "res_type": "polyphase"},
                  4: {"sr": 44100, "hl": 512, "n_fft": 768, "crop_start": 121, "crop_stop": 382, "hpf_start": 138, "hpf_stop": 123, "res_type": "sinc_medium"}}}


def synth(n, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / SR
    x = sum(rng.uniform(0.02, 0.1) * np.sin(2 * np.pi * rng.uniform(60, 8000) * t + rng.uniform(0, 6.28)) for _ in range(8))
    return (np.stack([x, 0.8 * x]) + 0.1 * rng.standard_normal((2, n))).astype(np.float32)


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def timed(step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def dominant(eng, step, names, x6_classes=("tdf",)):
    """Roofline of the kernel class that takes most of the step (class sums, as in round 2) PLUS a per-launch split of that class:
    a class can mix MFMA-bound and HBM-bound launches (Demucs "conv": 3x3 rewrites vs DConv / 1x1), so the launches are sorted
    by arithmetic intensity -- >= 40 flop/B (the fp32-MFMA / achievable-HBM ridge is ~25-30) count as MFMA-bound and are priced
    against 157.3 TFLOP/s, the rest against 8 TB/s."""
    eng.profile_enable(True)
    step()
    prof = eng.profile_read()
    recs = eng.profile_launches_ex()
    eng.profile_enable(False)
    k, v = max(((k, v) for k, v in prof.items() if v["flops"] > 0), key=lambda kv: kv[1]["ms"])
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
    # Every launch record says how many 16-bit MFMA products per multiply-add its kernel executed (ABI 7): 6 = bf16 x 6 (exactly split operands),
    # 3 = fp16 x 3 (block-scaled two-part operands, the default), 0 = an fp32-MFMA / VALU kernel.  A class's roofline is EXECUTED work over the
    # peak of the pipe that executed it: the split-operand launches' algorithmic FLOPs x their product count against the dense 16-bit peak, the
    # fp32 launches' own FLOPs against the fp32-MFMA peak -- never a flat factor over the class (a "x 6" label over fp16 x 3 launches
    # doubled four fractions).  `fp32_equivalent` keeps the algorithmic rate.

    def mfma_roof(rs):
        """rs: launch records (cls, ms, flops, bytes, nprod) of one group"""
        ms = sum(r[1] for r in rs)
        lo = [r for r in rs if r[4] > 0]
        fp = [r for r in rs if r[4] == 0]
        out = {}
        if lo:
            ex = sum(r[2] * r[4] for r in lo)
            t = sum(r[1] for r in lo)
            npr = sorted(set(r[4] for r in lo))
            out = {"achieved": round(ex / t / 1e9, 1), "peak": PEAK_BF16, "unit": "TFLOP/s", "frac": round(ex / t / 1e9 / PEAK_BF16, 4),
                   "dtype": " + ".join("fp16 x 3 products (block-scaled two-part operands)" if n == 3 else f"bf16 x {n} products (fp32-exact split operands)" for n in npr),
                   "launches_16bit": len(lo), "ms_16bit": round(t, 2),
                   "fp32_equivalent": round(sum(r[2] for r in lo) / t / 1e9, 2)}
            if fp:
                t2 = sum(r[1] for r in fp)
                out["fp32_launches"] = {"launches": len(fp), "ms": round(t2, 2), "achieved": round(sum(r[2] for r in fp) / t2 / 1e9, 2), "peak": PEAK,
                                        "frac": round(sum(r[2] for r in fp) / t2 / 1e9 / PEAK, 4)}
            return out
        a = sum(r[2] for r in fp) / max(ms, 1e-9) / 1e9
        return {"achieved": round(a, 2), "peak": PEAK, "unit": "TFLOP/s", "frac": round(a / PEAK, 4)}
    roof = dict({"kernel": names.get(k, k), "bound": "mfma"}, **mfma_roof([r for r in recs if r[0] == k]))
    roof.update({"traffic": None, "launches": v["launches"], "share_of_step_ms": round(v["ms"], 2)})
    mf = [r for r in recs if r[0] == k and r[3] > 0 and r[2] / r[3] >= 40.0]
    hb = [r for r in recs if r[0] == k and not (r[3] > 0 and r[2] / r[3] >= 40.0)]
    if mf:
        ms, fl = sum(r[1] for r in mf), sum(r[2] for r in mf)
        roof["mfma_bound_launches"] = dict({"launches": len(mf), "ms": round(ms, 2)}, **mfma_roof(mf))
    if hb:
        ms, by = sum(r[1] for r in hb), sum(r[3] for r in hb)
        roof["hbm_bound_launches"] = {"launches": len(hb), "ms": round(ms, 2), "achieved": round(by / ms / 1e6, 1), "unit": "GB/s",
                                      "frac": round(by / ms / 1e6 / 8000.0, 4)}
    stages = {}
    for kk, vv in prof.items():
        if not vv["launches"] or vv["ms"] <= 0:
            continue
        t, g = vv["flops"] / (vv["ms"] * 1e-3) / 1e12, vv["bytes"] / (vv["ms"] * 1e-3) / 1e9
        stages[names.get(kk, kk)] = (dict({"bound": "mfma"}, **mfma_roof([r for r in recs if r[0] == kk]))
                                     if t / PEAK >= g / 8000.0 else
                                     {"bound": "hbm", "achieved": round(g, 1), "unit": "GB/s", "frac": round(g / 8000.0, 4)})
    roof["stage_roofline"] = stages
    return roof, {names.get(k, k): round(v["ms"], 2) for k, v in prof.items() if v["launches"]}


def line(workload, secs, dt, steps, warmup, roof, kms, cpu, extra):
    return {"metric": "audio-sec separated / wall-sec (RTF)", "value": round(secs / dt, 2), "unit": "audio-s/wall-s", "n_gpus": 1,
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": dict({"workload": workload}, **extra),
            "roofline": roof, "cpu_baseline": cpu, "kernel_ms": kms}


def run_vr(args):
    from oracle import vr_oracle as V
    arch = 123821
    sd = V.make_vr_state(arch, 0)
    dm = A.VRDemixer({"model_params": VR_MP, "primary_stem_name": "Instrumental", "torch_device": 0},
                     {"window_size": 512, "batch_size": 8, "aggression": 5}, state_dict=sd, nn_arch_size=arch)   # engine default: <= 32 patches per pass, evened out
    eng = dm.engine
    n = int(SR * args.seconds)
    wave = synth(n)
    T, n_out = eng.vr_plan(n)
    dw = torch.from_numpy(wave).cuda()
    p = torch.empty((2, n_out), dtype=torch.float32, device="cuda")
    s = torch.empty_like(p)
    st = torch.cuda.current_stream().cuda_stream
    step = lambda: eng.vr_separate_dev(dw.data_ptr(), n, p.data_ptr(), s.data_ptr(), 0.05, 186, stream=st)  # noqa: E731
    dt = timed(step, args.steps, args.warmup)
    roof, kms = dominant(eng, step, {"conv3x3": "tdf3_kernel<GATHER> / hg_kernel / gg_kernel (3x3 / 1x1 convs)",
                                     "down": "tdf3_kernel<GATHER> / gg_kernel (stride-2 convs)"}, x6_classes=("conv3x3", "down"))
    cpu = None
    if args.cpu:
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        cs = 10.0
        w = wave[:, : int(SR * cs)]
        t0 = time.perf_counter()
        op, osec = V.vr_separate(w, sd, arch, V.ModelParams(VR_MP), window_size=512, batch_size=4, aggression=5)
        cdt = time.perf_counter() - t0
        gp, _ = dm.separate_stems(w)
        cpu = {"value": round(cs / cdt, 3), "unit": "audio-s/wall-s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{cs:g} s clip (BASELINE config 0), numpy/scipy/torch-CPU oracle, {cdt:.1f} s wall",
               "parity_rel_rms": rel_rms(gp, op)}
    return line("VR arch, 4band_44100 layout, HP-size CascadedASPPNet (31.65 M params, 2_HP-UVR shape), window 512, synthetic weights",
                args.seconds, dt, args.steps, args.warmup, roof, kms, cpu,
                {"windows": T // 256 + 1, "gflop_per_window": round(eng.vr_flops() / 1e9, 1)})


def run_htdemucs(args):
    from oracle import demucs_oracle as D
    oc = D.HTConfig()
    sd = D.make_ht_state(oc, 0)
    hc = A.HTConfig(segment=Fraction(39, 5))
    eng = A.Engine(A.MDXConfig(n_fft=4096, hop_length=1024, dim_f=2048, segment_size=8))
    eng.load_ht(hc, sd)
    n = int(SR * args.seconds)
    mixh = synth(n)
    mix = torch.from_numpy(mixh).cuda()
    out = torch.empty((4, 2, n), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    offs = [11025, 3000]
    step = lambda: eng.ht_demix_dev(mix.data_ptr(), n, out.data_ptr(), shifts=2, offsets=offs, flags=3, stream=st)  # noqa: E731
    dt = timed(step, args.steps, args.warmup)
    roof, kms = dominant(eng, step, {"conv3x3": "tdf3_kernel<GATHER> (rewrite 3x3 GLU) + gg_kernel (DConv / 1x1)",
                                     "tdf": "row GEMM tdf3_kernel / tdf2_kernel (transformer linears)",
                                     "conv1x1": "mha6_kernel<3> (attention)", "down": "tdf3_kernel<GATHER> / gg_kernel (k8/s4 convs)",
                                     "up": "gg_kernel (transposed convs)"}, x6_classes=("conv3x3", "tdf", "conv1x1", "down"))
    TL = hc.segment_samples
    nseg = sum(len(range(0, n + 22050 - o, int(0.75 * TL))) for o in offs)
    cpu = None
    if args.cpu:
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        cs = 12.0
        m = mixh[:, : int(SR * cs)]
        t0 = time.perf_counter()
        want = D.demix_demucs(m, sd, oc, shifts=1, overlap=0.25, offsets=[11025])
        cdt = time.perf_counter() - t0
        got = eng.ht_demix(m, shifts=1, offsets=[11025], overlap=0.25, standardize=True, swap01=True)
        cpu = {"value": round(cs / cdt / 2, 3), "unit": "audio-s/wall-s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{cs:g} s with shifts=1 ({cdt:.1f} s wall), halved to the shifts=2 rate of the GPU line; torch-CPU oracle",
               "parity_rel_rms": rel_rms(got, want)}
    return line("htdemucs layout (26.9 M params, 4 sources, nfft 4096, 5-layer cross transformer, 7.8-s segments), shifts=2, overlap 0.25, synthetic weights",
                args.seconds, dt, args.steps, args.warmup, roof, kms, cpu,
                {"segment_forwards": nseg, "gflop_per_segment": round(eng.ht_flops() / 1e9, 1),
                 "net_tflops_per_s": round(eng.ht_flops() * nseg / dt / 1e12, 1)})


def run_roformer(args):
    from oracle import roformer_oracle as R
    cfg = R.RoformerConfig(freqs_per_bands=R.DEFAULT_FREQS_PER_BANDS)
    sd = R.make_roformer_state(cfg, 0)
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0, "secondary_stem_name": "other"}, {"overlap": 8},
                       state_dict=sd, max_batch=16)   # chunks per net pass: 8 / 16 / 31 -> 151.4 / 154.4 / 154.6x (profiles/r05_sibling_batch_sweep.txt)
    eng = dm.engine
    n = int(SR * args.seconds)
    mixh = synth(n)
    mix = torch.from_numpy(mixh).cuda()
    out = torch.empty((2, 2, n), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    C = 441 * 800
    step = lambda: eng.rof_demix_dev(mix.data_ptr(), n, C, out.data_ptr(), stream=st)  # noqa: E731
    dt = timed(step, args.steps, args.warmup)
    roof, kms = dominant(eng, step, {"tdf": "row GEMM tdf3_kernel / tdf2_kernel (linears)", "conv1x1": "attention6_kernel"},
                         x6_classes=("tdf", "conv1x1"))
    nch = len(R.roformer_plan(n, cfg, 8)[2])
    cpu = None
    if args.cpu:
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        m = mixh[:, :C]
        t0 = time.perf_counter()
        want = R.roformer_forward(m[None], sd, cfg)
        cdt = time.perf_counter() - t0
        got = eng.rof_forward(m[None])
        cpu = {"value": round(C / SR / cdt, 3), "unit": "audio-s/wall-s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"one 8-s chunk forward, torch-CPU oracle, {cdt:.1f} s wall", "parity_rel_rms": rel_rms(got, want)}
    return line("BS-Roformer ep_317 layout (159.8 M params, dim 512, depth 12, 62 bands, 8-s chunks, step = chunk), synthetic weights",
                args.seconds, dt, args.steps, args.warmup, roof, kms, cpu,
                {"chunks": nch, "tflop_per_chunk": round(eng.rof_flops(1) / 1e12, 2),
                 "net_tflops_per_s": round(eng.rof_flops(nch) / dt / 1e12, 1)})


def run_mdx23c(args):
    from oracle import mdxc_oracle as M
    cfg = M.V3Config()
    sd = M.make_v3_state(cfg, 0)
    ov = 2
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0}, {"overlap": ov}, state_dict=sd, max_batch=8)
    eng = dm.engine
    n = int(SR * args.seconds)
    mixh = synth(n)
    mix = torch.from_numpy(mixh).cuda()
    out = torch.empty((2, 2, n), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    step = lambda: eng.mdxc_demix_dev(mix.data_ptr(), n, ov, out.data_ptr(), stream=st)  # noqa: E731
    dt = timed(step, args.steps, args.warmup)
    roof, kms = dominant(eng, step, {"conv3x3": "conv_dma_kernel (3x3 convs)"})
    plan = eng.mdxc_plan(n, ov)
    cpu = None
    if args.cpu:
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        Cn = plan["chunk_size"]
        m = mixh[:, :Cn]
        t0 = time.perf_counter()
        want = M.v3_forward(m[None], sd, cfg)
        cdt = time.perf_counter() - t0
        got = eng.v3_forward(m[None])
        cpu = {"value": round(Cn / SR / cdt / ov, 3), "unit": "audio-s/wall-s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"one chunk forward ({cdt:.1f} s wall) / overlap {ov}; torch-CPU oracle", "parity_rel_rms": rel_rms(got, want)}
    return line(f"MDX23C layout (112 M params, n_fft 8192, 4 subbands, 5 scales), overlap {ov}, synthetic weights", args.seconds, dt,
                args.steps, args.warmup, roof, kms, cpu,
                {"chunks": plan["n_chunks"], "tflop_per_chunk": round(eng.v3_flops(1) / 1e12, 2),
                 "net_tflops_per_s": round(eng.v3_flops(plan["n_chunks"]) / dt / 1e12, 1)})


def run_htdemucs_ft(args):
    """htdemucs_ft layout: a BagOfModels of FOUR htdemucs-size members (one per source in the published bag; here per-source
    weights like its YAML) through DemucsDemixer's device path -- resident engines, combine on the device."""
    from oracle import demucs_oracle as D
    oc = D.HTConfig()
    hc = A.HTConfig(segment=Fraction(39, 5))
    models = [(hc, D.make_ht_state(oc, seed)) for seed in range(4)]
    weights = [[1.0 if k == i else 0.0 for k in range(4)] for i in range(4)]     # htdemucs_ft.yaml: member i carries source i
    dm = A.DemucsDemixer({"torch_device": 0}, {"shifts": 2, "overlap": 0.25}, models=models, weights=weights)
    n = int(SR * args.seconds)
    mix = torch.from_numpy(synth(n)).cuda()
    out = torch.empty((4, 2, n), dtype=torch.float32, device="cuda")
    offs = [[11025, 3000]] * 4
    step = lambda: dm.bag_demix_dev(mix, out, offs)  # noqa: E731
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    first = time.perf_counter() - t0            # includes the four weight commits (once per process, not per file)
    dt = timed(step, args.steps, args.warmup)
    dm._load(0)
    single = lambda: dm.engine.ht_demix_dev(mix.data_ptr(), n, out.data_ptr(), shifts=2, offsets=offs[0], flags=3,  # noqa: E731
                                            stream=torch.cuda.current_stream().cuda_stream)
    ds = timed(single, args.steps, args.warmup)
    res = {"metric": "audio-sec separated / wall-sec (RTF)", "value": round(args.seconds / dt, 2), "unit": "audio-s/wall-s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": "htdemucs_ft layout: BagOfModels of 4 htdemucs-size members, shifts=2, overlap 0.25, 4-min song, device resident",
                      "single_member_ms": round(ds * 1e3, 2), "bag_over_single": round(dt / ds, 3),
                      "first_call_ms_with_weight_commits": round(first * 1e3, 1)},
           "roofline": None, "cpu_baseline": None}
    dm.close()
    return res


def run_hdemucs(args):
    from oracle import hdemucs_oracle as H
    oc = H.HDConfig(segment=44)
    sd = H.make_hd_state(oc, 0)
    hc = A.HDConfig(segment=44)
    eng = A.Engine(A.MDXConfig(n_fft=4096, hop_length=1024, dim_f=2048, segment_size=8))
    eng.load_hd(hc, sd)
    n = int(SR * args.seconds)
    mixh = synth(n)
    mix = torch.from_numpy(mixh).cuda()
    out = torch.empty((4, 2, n), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    offs = [11025, 3000]
    step = lambda: eng.hd_demix_dev(mix.data_ptr(), n, out.data_ptr(), shifts=2, offsets=offs, flags=3, stream=st)  # noqa: E731
    dt = timed(step, args.steps, args.warmup)
    roof, kms = dominant(eng, step, {"conv3x3": "tdf3_kernel<GATHER> (rewrite 3x3 GLU) + gg_kernel (DConv / 1x1)",
                                     "tdf": "row GEMM tdf3_kernel / tdf2_kernel (LSTM input / LocalState projections)",
                                     "conv1x1": "hd_lstm_step_kernel + hd_local_attn_kernel", "down": "tdf3_kernel<GATHER> / gg_kernel (k8/s4 convs)",
                                     "up": "gg_kernel (transposed convs)"}, x6_classes=("conv3x3", "tdf", "down"))
    TL = hc.segment_samples
    lens = []
    for o in offs:
        vl = n + 22050 - o
        lens += [min(vl - k, TL) for k in range(0, vl, int(0.75 * TL))]
    flops = sum(eng.hd_flops(l) for l in lens)
    cpu = None
    if args.cpu:
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        cs = 12.0
        m = mixh[:, : int(SR * cs)]
        t0 = time.perf_counter()
        want = H.demix_hdemucs(m, sd, oc, shifts=1, overlap=0.25, offsets=[11025])
        cdt = time.perf_counter() - t0
        got = eng.hd_demix(m, shifts=1, offsets=[11025], overlap=0.25, standardize=True, swap01=True)
        cpu = {"value": round(cs / cdt / 2, 3), "unit": "audio-s/wall-s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{cs:g} s with shifts=1 ({cdt:.1f} s wall), halved to the shifts=2 rate of the GPU line; torch-CPU oracle",
               "parity_rel_rms": rel_rms(got, want)}
    return line("hdemucs_mmi layout (Demucs v3: 83.6 M params, 4 sources, nfft 4096, depth 6, BLSTM + LocalState on the two inner levels, 44-s chunks), "
                "shifts=2, overlap 0.25, synthetic weights", args.seconds, dt, args.steps, args.warmup, roof, kms, cpu,
                {"chunk_forwards": len(lens), "gflop_per_song": round(flops / 1e9, 1), "net_tflops_per_s": round(flops / dt / 1e12, 1)})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="vr,htdemucs,roformer,mdx23c")
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu", type=int, default=1)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    fns = {"vr": run_vr, "htdemucs": run_htdemucs, "hdemucs": run_hdemucs, "roformer": run_roformer, "mdx23c": run_mdx23c,
           "htdemucs_ft": run_htdemucs_ft}
    for w in args.workloads.split(","):
        print(json.dumps(fns[w](args)), flush=True)


if __name__ == "__main__":
    main()
