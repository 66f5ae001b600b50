"""The three soundfile calls the reference's bit-depth unit tests make, on audio_io's RIFF/WAVE reader / writer (soundfile itself
is not installed in this image)."""
import types

import numpy as np

from audio_separator_amd import audio_io


def write(path, data, samplerate, subtype="PCM_16"):
    audio_io.write_wav(str(path), np.asarray(data), int(samplerate), subtype)


def info(path):
    i = audio_io.wav_info(str(path))          # (audio_io.info would come back here: this module IS `soundfile` in that process)
    return types.SimpleNamespace(samplerate=i["samplerate"], channels=i["channels"], frames=i["frames"], subtype=i["subtype"])


def read(path, dtype="float64", always_2d=False):
    data, sr = audio_io.read_wav(str(path))
    data = np.asarray(data, dtype=dtype)
    if always_2d and data.ndim == 1:
        data = data[:, None]
    return data, sr
