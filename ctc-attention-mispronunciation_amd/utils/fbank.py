"""WAV -> normalised log-mel features on the GPU: the feature step in front of the hot path.

The reference shells out to prebuilt Kaldi binaries for this (``AA/infer.py:567-574``):

    compute-fbank-feats --config=conf/fbank.conf scp:wav.scp ark:- | apply-cmvn --norm-vars=true data/global_fbank_cmvn.txt ark:- ark:- | copy-feats ark:- ark,scp:fbank.ark,fbank.scp

Here: ``compute_fbank_feats`` (HIP kernel behind ``mdd_fbank``), ``read_cmvn_stats`` / ``cmvn_scale_offset`` (Kaldi text
matrix, ``AA/data/global_fbank_cmvn.txt``), and ``write_ark_scp`` / ``read_ark`` for the Kaldi binary float-matrix wire
format the data loader reads (``kaldiio.load_mat``, ``AA/utils/data_loader.py:129``).  Kaldi's dither (random, default
1.0) is not applied.  Parity with Kaldi is unpinned: there is no Kaldi output here to compare with.
"""
import ctypes as C
import struct
import wave

import numpy as np
import torch

from .. import _lib

SAMPLE_RATE = 16000
NUM_COLS = 81


def read_wav(path):
    """16-bit mono PCM WAV -> (float32 samples on the int16 scale, sample rate), as Kaldi's WaveData holds them."""
    with wave.open(path, "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError("%s: only 16-bit PCM is supported" % path)
        x = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, w.getnchannels())
        return x[:, 0].astype(np.float32), w.getframerate()


def read_cmvn_stats(path):
    """Kaldi text matrix ``[ sums.. count \\n sumsq.. 0 ]`` -> float64 array [2, D+1]."""
    txt = open(path).read().replace("[", " ").replace("]", " ")
    rows = [r.split() for r in txt.strip().split("\n") if r.split()]
    return np.array([[float(v) for v in r] for r in rows], dtype=np.float64)


def cmvn_scale_offset(stats, norm_vars=True):
    """(scale, offset) float32 vectors of ``apply-cmvn`` with global stats: out = feat * scale + offset."""
    D = stats.shape[1] - 1
    count = stats[0, D]
    mean = stats[0, :D] / count
    scale = 1.0 / np.sqrt(np.maximum(stats[1, :D] / count - mean * mean, 1e-20)) if norm_vars else np.ones(D)
    return scale.astype(np.float32), (-mean * scale).astype(np.float32)


def compute_fbank_feats(samples, sample_rate=SAMPLE_RATE, cmvn=None, device=None):
    """[num_frames, 81] float32 CUDA tensor (column 0 log energy, 1..80 log mel) for one utterance.

    ``samples``: 1-D array / tensor on the int16 scale; ``cmvn``: None or the (scale, offset) pair of
    ``cmvn_scale_offset`` -- the normalisation is then fused into the kernel's store.
    """
    _lib.require_gpu()
    if sample_rate != SAMPLE_RATE:
        raise ValueError("compute_fbank_feats expects %d Hz audio (resample first, as AA/infer.py:486-516 does)" % SAMPLE_RATE)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    x = torch.as_tensor(np.asarray(samples, dtype=np.float32) if not torch.is_tensor(samples) else samples).to(dev, torch.float32).contiguous()
    n = _lib.lib().mdd_fbank_num_frames(x.numel())
    out = torch.empty((n, NUM_COLS), dtype=torch.float32, device=dev)
    sc = of = None
    if cmvn is not None:
        sc = torch.as_tensor(cmvn[0]).to(dev, torch.float32).contiguous()
        of = torch.as_tensor(cmvn[1]).to(dev, torch.float32).contiguous()
        if sc.numel() != NUM_COLS or of.numel() != NUM_COLS:
            raise ValueError("cmvn vectors must have %d entries" % NUM_COLS)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().mdd_fbank(C.c_void_p(x.data_ptr()), x.numel(), C.c_void_p(sc.data_ptr()) if sc is not None else None,
                                        C.c_void_p(of.data_ptr()) if of is not None else None, C.c_void_p(out.data_ptr()),
                                        _lib.current_stream_ptr()))
    return out


def write_ark_scp(ark_path, scp_path, feats):
    """Kaldi binary archive of float matrices (``<key> \\0B FM \\4 rows \\4 cols data``) plus its scp index.

    ``feats``: dict utt_id -> [T, D] float32 array / tensor (insertion order kept)."""
    with open(ark_path, "wb") as ark, open(scp_path, "w") as scp:
        for key, m in feats.items():
            a = np.ascontiguousarray(m.detach().cpu().numpy() if torch.is_tensor(m) else m, dtype="<f4")
            ark.write(key.encode() + b" ")
            scp.write("%s %s:%d\n" % (key, ark_path, ark.tell()))
            ark.write(b"\0BFM " + b"\x04" + struct.pack("<i", a.shape[0]) + b"\x04" + struct.pack("<i", a.shape[1]))
            ark.write(a.tobytes())


def read_ark(ark_path):
    """dict utt_id -> float32 [T, D] of a binary float-matrix archive written by Kaldi's copy-feats or ``write_ark_scp``."""
    out = {}
    data = open(ark_path, "rb").read()
    pos = 0
    while pos < len(data):
        sp = data.index(b" ", pos)
        key = data[pos:sp].decode()
        pos = sp + 1
        if data[pos:pos + 6] != b"\0BFM \x04"[:6] or data[pos + 5:pos + 6] != b"\x04":
            raise ValueError("%s: entry %r is not an uncompressed binary float matrix" % (ark_path, key))
        rows = struct.unpack_from("<i", data, pos + 6)[0]
        if data[pos + 10:pos + 11] != b"\x04":
            raise ValueError("%s: malformed header at %r" % (ark_path, key))
        cols = struct.unpack_from("<i", data, pos + 11)[0]
        pos += 15
        out[key] = np.frombuffer(data, dtype="<f4", count=rows * cols, offset=pos).reshape(rows, cols).copy()
        pos += 4 * rows * cols
    return out


def load_mat(spec):
    """One matrix by its scp entry ``<ark path>:<byte offset>`` (what ``kaldiio.load_mat`` does for the reference's data
    loader, AA/utils/data_loader.py:130): uncompressed binary float matrices only."""
    path, _, off = spec.rpartition(":")
    if not path:
        raise ValueError("load_mat expects '<ark>:<offset>', got %r" % spec)
    with open(path, "rb") as f:
        f.seek(int(off))
        head = f.read(15)
        if head[:6] != b"\0BFM \x04"[:6] or head[5:6] != b"\x04" or head[10:11] != b"\x04":
            raise ValueError("%s: not an uncompressed binary float matrix" % spec)
        rows, cols = struct.unpack_from("<i", head, 6)[0], struct.unpack_from("<i", head, 11)[0]
        return np.frombuffer(f.read(4 * rows * cols), dtype="<f4").reshape(rows, cols).copy()
