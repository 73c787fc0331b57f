"""Detector / recogniser parameters as upstream-named state-dicts (numpy fp32).

easyocr downloads ``craft_mlt_25k.pth`` and ``english_g2.pth`` on first use
(``easyocr.Reader.__init__``, reached from ``pipeline_demo/extractor/enhanced_extractor.py:153``).
This backend never downloads.  It takes weights from

* real checkpoints in a directory (``load_checkpoint_dir``; key names / shapes are the
  upstream ones, optional ``module.`` prefix), or
* ``synthetic_*_state``: seeded random parameters of the exact architecture, used by tests
  and by ``bench.py`` (there is no network for checkpoints), or
* ``designed_craft_state``: the same random detector with a hand-built "ink detector"
  sub-network routed through a few channels, so that synthetic pages produce word-level
  region/affinity maps with wide decision margins (every other channel stays random and
  is computed at full cost -- the FLOPs do not depend on the weight values).
"""
from __future__ import annotations

import os

import numpy as np

# (key prefix, Cout, Cin, k, has_bias, bn prefix or None)
_VGG = [
    ("basenet.slice1.0", 64, 3, 3, "basenet.slice1.1"), ("basenet.slice1.3", 64, 64, 3, "basenet.slice1.4"),
    ("basenet.slice1.7", 128, 64, 3, "basenet.slice1.8"), ("basenet.slice1.10", 128, 128, 3, "basenet.slice1.11"),
    ("basenet.slice2.14", 256, 128, 3, "basenet.slice2.15"), ("basenet.slice2.17", 256, 256, 3, "basenet.slice2.18"),
    ("basenet.slice3.20", 256, 256, 3, "basenet.slice3.21"), ("basenet.slice3.24", 512, 256, 3, "basenet.slice3.25"),
    ("basenet.slice3.27", 512, 512, 3, "basenet.slice3.28"), ("basenet.slice4.30", 512, 512, 3, "basenet.slice4.31"),
    ("basenet.slice4.34", 512, 512, 3, "basenet.slice4.35"), ("basenet.slice4.37", 512, 512, 3, "basenet.slice4.38"),
    ("basenet.slice5.1", 1024, 512, 3, None), ("basenet.slice5.2", 1024, 1024, 1, None),
    ("upconv1.conv.0", 512, 1536, 1, "upconv1.conv.1"), ("upconv1.conv.3", 256, 512, 3, "upconv1.conv.4"),
    ("upconv2.conv.0", 256, 768, 1, "upconv2.conv.1"), ("upconv2.conv.3", 128, 256, 3, "upconv2.conv.4"),
    ("upconv3.conv.0", 128, 384, 1, "upconv3.conv.1"), ("upconv3.conv.3", 64, 128, 3, "upconv3.conv.4"),
    ("upconv4.conv.0", 64, 192, 1, "upconv4.conv.1"), ("upconv4.conv.3", 32, 64, 3, "upconv4.conv.4"),
    ("conv_cls.0", 32, 32, 3, None), ("conv_cls.2", 32, 32, 3, None), ("conv_cls.4", 16, 32, 3, None),
    ("conv_cls.6", 16, 16, 1, None), ("conv_cls.8", 2, 16, 1, None),
]


def _conv(rng, sd, name, cout, cin, k, bias=True, gain=np.sqrt(2.0)):
    std = gain / np.sqrt(cin * k * k)
    sd[name + ".weight"] = (rng.standard_normal((cout, cin, k, k)) * std).astype(np.float32)
    if bias:
        sd[name + ".bias"] = (rng.standard_normal(cout) * 0.02).astype(np.float32)


def _bn(rng, sd, name, c):
    sd[name + ".weight"] = (1.0 + 0.1 * rng.standard_normal(c)).astype(np.float32)
    sd[name + ".bias"] = (0.05 * rng.standard_normal(c)).astype(np.float32)
    sd[name + ".running_mean"] = (0.05 * rng.standard_normal(c)).astype(np.float32)
    sd[name + ".running_var"] = (1.0 + 0.1 * np.abs(rng.standard_normal(c))).astype(np.float32)
    sd[name + ".num_batches_tracked"] = np.zeros((), dtype=np.int64)


def synthetic_craft_state(seed: int = 0) -> dict:
    """Seeded random CRAFT (craft_mlt_25k architecture, 20.77 M parameters)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, cout, cin, k, bn in _VGG:
        _conv(rng, sd, name, cout, cin, k)
        if bn:
            _bn(rng, sd, bn, cout)
    return sd


def _signal(sd, conv, bn, out_ch, taps, bias):
    """Make output channel ``out_ch`` of ``conv`` a fixed function: taps = {in_ch: 3x3 (or 1x1) array}."""
    w = sd[conv + ".weight"]
    w[out_ch] = 0.0
    for ic, t in taps.items():
        w[out_ch, ic] = np.asarray(t, dtype=np.float32).reshape(w.shape[2:])
    sd[conv + ".bias"][out_ch] = bias
    if bn:
        sd[bn + ".weight"][out_ch] = 1.0
        sd[bn + ".bias"][out_ch] = 0.0
        sd[bn + ".running_mean"][out_ch] = 0.0
        sd[bn + ".running_var"][out_ch] = 1.0 - 1e-5


def designed_craft_state(seed: int = 0) -> dict:
    """Random CRAFT + an ink-detector sub-network (see module docstring).

    region = 0.25 * B, affinity = 0.073 * L where, at half resolution, I = 2x2 max-pooled binary ink,
    B = 3x3 box sum of I and L = B filtered twice with [1,1,1] horizontally.  Dark-on-light pages only.
    B and L are integers on binary ink, so both maps live on a lattice: region 0.25 k (thresholds 0.4 / 0.7 fall between 0.25 | 0.5 and
    0.5 | 0.75), affinity 0.073 k (0.4 falls between 0.365 | 0.438).  Round 2 used 0.08, whose fifth multiple IS the link threshold:
    ~100 pixels per page sat exactly on `link > 0.4`, where the fp32 oracle itself decides by its last bit (tools/flip_report.py).
    Ink is a BAND of grey levels (full response for green <= 16 .. 96, none at pure black): the zero canvas that
    ``resize_aspect_ratio`` pads a page with (a 13-pixel stripe beside a 2480x3504 scan) is not text, as for a trained CRAFT.
    """
    sd = synthetic_craft_state(seed)
    c = [[0, 0, 0], [0, 1, 0], [0, 0, 0]]          # centre tap
    box = [[1, 1, 1], [1, 1, 1], [1, 1, 1]]
    hor = [[0, 0, 0], [1, 1, 1], [0, 0, 0]]
    ctr = lambda v: [[0, 0, 0], [0, v, 0], [0, 0, 0]]
    # u = (128 - p) / 32 from the normalised green channel x = (p - 116.28) / 57.12
    a, b = -57.12 / 32.0, (128.0 - 116.28) / 32.0
    _signal(sd, "basenet.slice1.0", "basenet.slice1.1", 0, {1: ctr(a)}, b)            # relu(u)
    _signal(sd, "basenet.slice1.0", "basenet.slice1.1", 1, {1: ctr(a)}, b - 1.0)      # relu(u - 1)
    _signal(sd, "basenet.slice1.0", "basenet.slice1.1", 2, {1: ctr(4 * a)}, 4 * b - 14.0)   # relu(4 (u - 3.5)): rises from green = 16 down
    _signal(sd, "basenet.slice1.0", "basenet.slice1.1", 3, {1: ctr(4 * a)}, 4 * b - 15.0)   # relu(4 (u - 3.5) - 1)
    # I = clip(u, 0, 1) - clip(4 (u - 3.5), 0, 1): 1 on ink (green 16 .. 96), 0 on paper AND on pure black (u = 4)
    _signal(sd, "basenet.slice1.3", "basenet.slice1.4", 0, {0: c, 1: ctr(-1.0), 2: ctr(-1.0), 3: c}, 0.0)
    _signal(sd, "basenet.slice1.7", "basenet.slice1.8", 0, {0: box}, 0.0)              # B
    _signal(sd, "basenet.slice1.10", "basenet.slice1.11", 0, {0: c}, 0.0)              # skip s1 channel 0 = B
    _signal(sd, "upconv4.conv.0", "upconv4.conv.1", 0, {64: [[1.0]]}, 0.0)             # cat([up3(64), s1(128)])
    _signal(sd, "upconv4.conv.3", "upconv4.conv.4", 0, {0: c}, 0.0)
    _signal(sd, "conv_cls.0", None, 0, {0: c}, 0.0)
    _signal(sd, "conv_cls.0", None, 1, {0: hor}, 0.0)
    _signal(sd, "conv_cls.2", None, 0, {0: c}, 0.0)
    _signal(sd, "conv_cls.2", None, 1, {1: hor}, 0.0)
    _signal(sd, "conv_cls.4", None, 0, {0: c}, 0.0)
    _signal(sd, "conv_cls.4", None, 1, {1: c}, 0.0)
    _signal(sd, "conv_cls.6", None, 0, {0: [[1.0]]}, 0.0)
    _signal(sd, "conv_cls.6", None, 1, {1: [[1.0]]}, 0.0)
    _signal(sd, "conv_cls.8", None, 0, {0: [[0.25]]}, 0.0)
    _signal(sd, "conv_cls.8", None, 1, {1: [[0.073]]}, 0.0)
    return sd


def synthetic_crnn_state(seed: int = 0, logit_gain: float = 120.0, ih_gain: float = 4.0, hh_gain: float = 1.0,
                         lin_gain: float = 4.0) -> dict:
    """Seeded random CRNN (english_g2 architecture: 1x64xW -> T x 97).

    PyTorch-default initialisation makes a random CRNN emit one constant class with ~1/97
    confidence, which would send every box through the contrast-retry pass.  The gains widen the
    input / linear / prediction weights so that, like a trained model on a printed line, the arg-max changes every
    few time steps (~50 characters on a 600-pixel line) and the confidence lands around 0.25-0.85 (above
    ``contrast_ths`` = 0.1, so the contrast-retry pass stays the exception it is in practice).
    The recurrent weights keep PyTorch's scale (``hh_gain`` 1): a wider W_hh makes the random recurrence
    chaotic, i.e. an amplifier of rounding noise that no trained recogniser is.
    """
    rng = np.random.default_rng(seed + 1000)
    sd = {}
    fe = "FeatureExtraction.ConvNet."
    _conv(rng, sd, fe + "0", 32, 1, 3)
    _conv(rng, sd, fe + "3", 64, 32, 3)
    _conv(rng, sd, fe + "6", 128, 64, 3)
    _conv(rng, sd, fe + "8", 128, 128, 3)
    _conv(rng, sd, fe + "11", 256, 128, 3, bias=False)
    _bn(rng, sd, fe + "12", 256)
    _conv(rng, sd, fe + "14", 256, 256, 3, bias=False)
    _bn(rng, sd, fe + "15", 256)
    _conv(rng, sd, fe + "18", 256, 256, 2)
    k = 1.0 / np.sqrt(256.0)
    for l in range(2):
        sm = f"SequenceModeling.{l}."
        for sfx in ("", "_reverse"):
            sd[sm + "rnn.weight_ih_l0" + sfx] = (rng.uniform(-k, k, (1024, 256)) * ih_gain).astype(np.float32)
            sd[sm + "rnn.weight_hh_l0" + sfx] = (rng.uniform(-k, k, (1024, 256)) * hh_gain).astype(np.float32)
            sd[sm + "rnn.bias_ih_l0" + sfx] = rng.uniform(-k, k, 1024).astype(np.float32)
            sd[sm + "rnn.bias_hh_l0" + sfx] = rng.uniform(-k, k, 1024).astype(np.float32)
        kl = 1.0 / np.sqrt(512.0)
        sd[sm + "linear.weight"] = (rng.uniform(-kl, kl, (256, 512)) * lin_gain).astype(np.float32)
        sd[sm + "linear.bias"] = rng.uniform(-kl, kl, 256).astype(np.float32)
    sd["Prediction.weight"] = (rng.uniform(-k, k, (97, 256)) * logit_gain).astype(np.float32)
    sd["Prediction.bias"] = (rng.uniform(-k, k, 97) * logit_gain).astype(np.float32)
    return sd


def load_checkpoint(path: str) -> dict:
    """A real upstream ``.pth`` -> {name: numpy fp32}.  torch is only the unpickler here."""
    import torch

    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def load_npz_state(path: str, restore_fp32: bool = True) -> dict:
    """A state-dict stored as ``.npz`` (upstream key names, any float dtype) -> {name: numpy fp32}.

    ``restore_fp32``: a tensor STORED in float16 (tests/golden/crnn_synth_fp16.npz keeps the repository small) would make every fp32
    weight exactly fp16-representable, so the fp16 / split-fp16 weight rounding of the MFMA paths could never show against the fp32
    oracle.  Such tensors get a seeded perturbation uniform within +-0.49 fp16 ulp of each value (seed = CRC of the tensor name: the same
    fp32 state in every process): the oracle then holds genuine fp32 weights, and packing them to fp16 commits the 2^-12 relative
    rounding error a real fp32 checkpoint (english_g2.pth) would see.  Tensors stored in fp32 are returned as they are."""
    import zlib

    out = {}
    with np.load(path) as z:
        for k in z.files:
            a = z[k]
            v = np.ascontiguousarray(a, dtype=np.float32)
            if restore_fp32 and a.dtype == np.float16 and v.size:
                rng = np.random.default_rng(zlib.crc32(k.encode()))
                ulp = np.spacing(np.abs(a)).astype(np.float32)                        # fp16 ulp at each stored value
                v = (v + rng.uniform(-0.49, 0.49, v.shape).astype(np.float32) * ulp).astype(np.float32)
            out[k] = v
    return out


def load_checkpoint_dir(directory: str):
    """(craft_state, crnn_state) from ``craft_mlt_25k.pth`` + ``english_g2.pth``; raises FileNotFoundError."""
    det = os.path.join(directory, "craft_mlt_25k.pth")
    rec = os.path.join(directory, "english_g2.pth")
    for p in (det, rec):
        if not os.path.exists(p):
            raise FileNotFoundError(f"{p} not found (this backend never downloads model files)")
    return load_checkpoint(det), load_checkpoint(rec)


def to_descs(state: dict):
    """state-dict -> (ctypes array of bbocr_tensor_desc, keep-alive list)."""
    import ctypes as C

    from ._lib import bbocr_tensor_desc

    items = [(k, v) for k, v in state.items() if not k.endswith("num_batches_tracked")]
    arr = (bbocr_tensor_desc * len(items))()
    keep = []
    for i, (k, v) in enumerate(items):
        a = np.ascontiguousarray(v, dtype=np.float32)
        name = k.encode()
        keep += [a, name]
        arr[i].name = name
        arr[i].ndim = min(a.ndim, 4)
        shape = list(a.shape)[:4] + [1] * (4 - min(a.ndim, 4))
        if a.ndim == 0:
            shape = [1, 1, 1, 1]
            arr[i].ndim = 1
        for j in range(4):
            arr[i].shape[j] = shape[j]
        arr[i].data = a.ctypes.data_as(C.POINTER(C.c_float))
    return arr, keep
