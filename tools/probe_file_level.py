#!/usr/bin/env python3
"""File-to-files rate of the Demucs plugin class (htdemucs layout, shifts 2): ``DemucsSeparator.separate(song.wav)`` -> four PCM16
stem files on tmpfs, device-resident path against the generic one (ASX_FILE_FASTPATH=0).  The MDX equivalent is bench.py's
``file_level`` key.  One JSON line.

    python tools/probe_file_level.py [--seconds 240]
"""
import argparse
import json
import logging
import os
import random
import shutil
import sys
import tempfile
import time
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_separator_amd as A  # noqa: E402
from audio_separator_amd import audio_io  # noqa: E402
from audio_separator_amd.architectures.demucs_separator import DemucsSeparator  # noqa: E402
from oracle import demucs_oracle as D  # noqa: E402
from oracle import mdx_oracle as O  # noqa: E402

SR = 44100


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    args = ap.parse_args()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="asx_file_level_", dir=base)
    try:
        n = int(SR * args.seconds)
        wav = os.path.join(tmp, "song.wav")
        audio_io.write_wav(wav, np.ascontiguousarray(O.synth_mix(n, seed=0).T), SR, "PCM_16")
        oc = D.HTConfig()
        log = logging.getLogger("probe.file_level")
        log.setLevel(logging.ERROR)
        common = {"logger": log, "log_level": logging.ERROR, "torch_device": "cuda:0", "torch_device_cpu": "cpu", "torch_device_mps": None,
                  "onnx_execution_provider": ["ROCMExecutionProvider"], "model_name": "htdemucs", "model_path": None, "model_data": {},
                  "output_format": "WAV", "output_bitrate": None, "output_dir": os.path.join(tmp, "out"), "normalization_threshold": 0.9,
                  "amplification_threshold": 0.0, "output_single_stem": None, "invert_using_spec": False, "sample_rate": SR,
                  "use_soundfile": False, "asx_profile_file": True,
                  "asx_models": [(A.HTConfig(segment=Fraction(39, 5)), D.make_ht_state(oc, 0))]}
        arch = {"segment_size": "Default", "shifts": 2, "overlap": 0.25, "segments_enabled": True}
        sep = DemucsSeparator(common, arch)

        def run(calls):
            walls, phases = [], {}
            for _ in range(calls):
                random.seed(7)
                t0 = time.perf_counter()
                files = sep.separate(wav)
                walls.append(time.perf_counter() - t0)
                for k, v in sep.file_timings.items():
                    phases[k] = phases.get(k, 0.0) + v
                sep.clear_gpu_cache()
                sep.clear_file_specific_paths()
            return files, sum(walls) / len(walls), {k: round(v / calls * 1e3, 2) for k, v in phases.items()}

        run(1)
        files, wall, phases = run(3)
        os.environ["ASX_FILE_FASTPATH"] = "0"
        run(1)
        _, wall_h, phases_h = run(2)
        os.environ.pop("ASX_FILE_FASTPATH")
        print(json.dumps({"what": "DemucsSeparator.separate(4-min PCM16 WAV on tmpfs) -> 4 PCM16 stem files, htdemucs layout, shifts 2",
                          "rtf": round(args.seconds / wall, 1), "wall_ms": round(wall * 1e3, 2), "phases_ms": phases,
                          "phases_sum_ms": round(sum(phases.values()), 2), "files": files,
                          "host_path": {"rtf": round(args.seconds / wall_h, 1), "wall_ms": round(wall_h * 1e3, 2), "phases_ms": phases_h}}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
