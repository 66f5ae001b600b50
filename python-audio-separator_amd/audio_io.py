"""File edges of the plugin surface: decode and encode stay on the host ("boundary, stays host").

The reference decodes with ``librosa.load`` and encodes with pydub (ffmpeg) or soundfile
(common_separator.py:217-282, 284-461).  When those packages are installed they are used here too, unchanged.
When they are not (this image ships none of them), RIFF/WAVE files -- integer PCM 8/16/24/32 and IEEE float
32/64, plain or WAVE_FORMAT_EXTENSIBLE -- are read and written by the small codec below so that
``separate(path) -> [file names]`` works end to end; every other container then raises with the name of the
missing package.  Nothing here touches the GPU: the sample arithmetic of the writer (normalise, * 32767,
int16 cast, interleave) is ``asx_pcm16`` (include/asx.h), called by CommonSeparator.
"""
from __future__ import annotations

import os
import struct

import numpy as np

_FMT_PCM, _FMT_FLOAT, _FMT_EXT = 1, 3, 0xFFFE


class AudioIOError(RuntimeError):
    pass


def _optional(name):
    try:
        return __import__(name)
    except Exception:
        return None


def _riff_chunks(f):
    head = f.read(12)
    if len(head) < 12 or head[:4] not in (b"RIFF", b"RF64") or head[8:12] != b"WAVE":
        raise AudioIOError("not a RIFF/WAVE file")
    while True:
        h = f.read(8)
        if len(h) < 8:
            return
        cid, size = h[:4], struct.unpack("<I", h[4:])[0]
        pos = f.tell()
        yield cid, size, pos
        f.seek(pos + size + (size & 1))


def wav_info(path: str) -> dict:
    """{samplerate, channels, frames, subtype} with soundfile's subtype names (PCM_16, PCM_24, PCM_32, PCM_U8, FLOAT, DOUBLE)."""
    with open(path, "rb") as f:
        fmt = None
        for cid, size, pos in _riff_chunks(f):
            if cid == b"fmt ":
                raw = f.read(min(size, 40))
                tag, ch, sr, _, align, bits = struct.unpack("<HHIIHH", raw[:16])
                if tag == _FMT_EXT and len(raw) >= 26:
                    tag = struct.unpack("<H", raw[24:26])[0]
                fmt = (tag, ch, sr, align, bits)
            elif cid == b"data":
                if fmt is None:
                    raise AudioIOError("data chunk before fmt chunk")
                tag, ch, sr, align, bits = fmt
                size = min(size, os.path.getsize(path) - pos)
                if tag == _FMT_PCM and bits in (8, 16, 24, 32):
                    subtype = {8: "PCM_U8", 16: "PCM_16", 24: "PCM_24", 32: "PCM_32"}[bits]
                elif tag == _FMT_FLOAT and bits in (32, 64):
                    subtype = "FLOAT" if bits == 32 else "DOUBLE"
                else:
                    raise AudioIOError(f"unsupported WAVE encoding (format tag {tag}, {bits} bits)")
                return {"samplerate": sr, "channels": ch, "frames": size // max(align, 1), "subtype": subtype,
                        "data_offset": pos, "data_bytes": size, "bits": bits}
    raise AudioIOError("no data chunk")


def read_wav(path: str):
    """(float32 [channels, frames], samplerate); integer PCM is scaled by 2^-(bits-1) like libsndfile / audioread."""
    info = wav_info(path)
    ch, bits = info["channels"], info["bits"]
    with open(path, "rb") as f:
        f.seek(info["data_offset"])
        raw = f.read(info["frames"] * ch * bits // 8)
    st = info["subtype"]
    if st == "PCM_16":
        x = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
    elif st == "PCM_24":
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
        v = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = (v.astype(np.float64) / 8388608.0).astype(np.float32)
    elif st == "PCM_32":
        x = (np.frombuffer(raw, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif st == "PCM_U8":
        x = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
    elif st == "FLOAT":
        x = np.frombuffer(raw, "<f4").astype(np.float32)
    else:
        x = np.frombuffer(raw, "<f8").astype(np.float32)
    return np.ascontiguousarray(x.reshape(-1, ch).T), info["samplerate"]


def write_wav(path: str, data: np.ndarray, samplerate: int, subtype: str = "PCM_16"):
    """data: [frames, channels] (or [frames]); int16 input is widened bit-exactly (what ffmpeg's s16 -> s32 / pcm_s24le
    conversion does, common_separator.py:375-391), float input is rounded to the target width with clipping."""
    a = np.asarray(data)
    if a.ndim == 1:
        a = a[:, None]
    frames, ch = a.shape
    if subtype in ("FLOAT", "DOUBLE"):
        body = a.astype("<f4" if subtype == "FLOAT" else "<f8").tobytes()
        tag, bits = _FMT_FLOAT, 32 if subtype == "FLOAT" else 64
    else:
        bits = {"PCM_16": 16, "PCM_24": 24, "PCM_32": 32}.get(subtype)
        if bits is None:
            raise AudioIOError(f"unsupported WAV subtype {subtype}")
        if a.dtype == np.int16 and bits == 16:
            v = None                                   # already the file's sample format: no widening pass
        elif a.dtype == np.int16:
            v = a.astype(np.int32) << (bits - 16)
        else:
            full = float(2 ** (bits - 1) - 1)
            v = np.clip(np.rint(a.astype(np.float64) * full), -full - 1, full).astype(np.int64)
        if v is None:
            body = memoryview(np.ascontiguousarray(a.astype("<i2", copy=False))).cast("B")
        elif bits == 16:
            body = v.astype("<i2").tobytes()
        elif bits == 32:
            body = v.astype("<i4").tobytes()
        else:
            u = (v & 0xFFFFFF).astype(np.uint32).reshape(-1)
            body = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes()
        tag = _FMT_PCM
    align = ch * bits // 8
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, tag, ch, samplerate, samplerate * align, align, bits))
        f.write(b"data" + struct.pack("<I", len(body)))
        f.write(body)


def info(path: str) -> dict:
    """soundfile.info's fields the plugin reads (subtype; common_separator.py:233-250, vr_separator.py:137-156)."""
    sf = _optional("soundfile")
    if sf is not None:
        i = sf.info(path)
        return {"samplerate": i.samplerate, "channels": i.channels, "frames": i.frames, "subtype": i.subtype}
    return wav_info(path)


def duration(path: str) -> float:
    try:
        i = info(path)
        return i["frames"] / float(i["samplerate"])
    except Exception:
        return 0.0


def load(path: str, sr: int | None = 44100, mono: bool = False, res_type: str | None = None):
    """librosa.load(path, sr=sr, mono=mono) (common_separator.py:252): float32 [channels, n] (or [n] for a mono file)."""
    librosa = _optional("librosa")
    if librosa is not None and hasattr(librosa, "load"):
        kw = {"res_type": res_type} if res_type else {}
        return librosa.load(path, mono=mono, sr=sr, **kw)
    try:
        x, file_sr = read_wav(path)
    except AudioIOError as e:
        raise AudioIOError(f"{path}: {e}; install librosa (+ soundfile / audioread) to decode other containers") from e
    if sr is not None and file_sr != sr:
        raise AudioIOError(f"{path} is sampled at {file_sr} Hz and librosa (which resamples to {sr} Hz on load) is not installed")
    if x.shape[0] == 1:
        x = x[0]
    elif mono:
        x = x.mean(0)
    return x, file_sr
