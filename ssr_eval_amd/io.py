"""Host-side audio file I/O (SURVEY 8(f) N2 - out of the accelerated path).

The reference decodes with librosa.load (+ resampy) and shells out to ``sox`` (ssr_eval/eval.py:133-134,
242).  Neither exists in the build image; this module decodes with ``soundfile`` when importable and with
the standard-library ``wave`` module otherwise (PCM .wav only) and changes the sampling rate with the
polyphase kernel (K7).  Resampling filters therefore differ from sox / resampy: end-to-end numbers on
real VCTK files are comparable, not bit-identical ("Out of scope").
"""
import os
import wave

import numpy as np


def read_audio(path):
    """-> (float32 mono [n], sample_rate)."""
    try:
        import soundfile as sf  # optional
        x, sr = sf.read(path, dtype="float32", always_2d=True)
        return np.ascontiguousarray(x.mean(axis=1), dtype=np.float32), int(sr)
    except ImportError:
        pass
    if not path.lower().endswith(".wav"):
        raise RuntimeError("decoding %s needs the `soundfile` package (only PCM .wav is readable without it)"
                           % os.path.basename(path))
    with wave.open(path, "rb") as f:
        sr, nch, sw, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, "<i4").astype(np.float32) / 2147483648.0
    elif sw == 3:
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
    elif sw == 1:
        x = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise RuntimeError("unsupported sample width %d" % sw)
    return np.ascontiguousarray(x.reshape(-1, nch).mean(axis=1), dtype=np.float32), int(sr)


def write_wav(path, x, sr):
    x = np.clip(np.asarray(x, np.float32), -1.0, 1.0 - 1.0 / 32768)
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(sr))
        f.writeframes((x * 32768.0).astype("<i2").tobytes())


def load_audio(path, sr=None):
    """Decode and (if sr is given and differs) resample with the polyphase kernel."""
    x, file_sr = read_audio(path)
    if sr is None or int(sr) == file_sr:
        return x
    from . import backend as B
    return B.resample_poly([x], int(sr), file_sr)[0].cpu().numpy()
