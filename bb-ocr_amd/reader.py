"""``Reader``: the host-side mirror of ``easyocr.Reader`` for the one path BB-OCR uses.

Reference call sites this class stands in for (same names, argument meaning, return
shape and error behaviour):
  * ``easyocr.Reader(["en"], gpu=use_gpu)``        pipeline_demo/extractor/enhanced_extractor.py:153
  * ``reader.readtext(path, paragraph=False, batch_size=1, workers=0)``          ...:520
    consumed as ``" ".join(result[1] for result in results)`` (:521), any exception is
    caught by the caller and turns into empty text (:529-531)
  * ``for (bbox, text, prob) in reader.readtext(path)``   pipeline_components/img_to_json/
    ocr_testing/ocr_engines/test_easyocr.py:23,50
Upstream semantics followed: easyocr==1.7.2 ``easyocr/easyocr.py::Reader.{__init__,detect,
recognize,readtext,readtext_batched}``.

Everything numeric happens in libbbocr.so (HIP kernels) through the C ABI in
``include/bbocr.h``; torch is used only to hold device memory.  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
import io
import os
import threading

import numpy as np

from . import _lib
from . import weights as _weights

# english_g2 character list (easyocr/config.py); index 0 of the class axis is the CTC blank
_SYMBOLS = "0123456789!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~ €"
CHARSET = _SYMBOLS + "ABCDEFGHIJKLMNOPQRSTUVWXYZ" + "abcdefghijklmnopqrstuvwxyz"
CHARACTER = ["[blank]"] + list(CHARSET)
# byte per class index (the CTC blank never reaches the decoded text): lets _collect build all texts with ONE latin-1 decode (a memcpy-grade
# operation; the utf-32 decode it replaces cost 0.45 ms per 100 k characters).  The one non-latin-1 character of the english_g2 charset,
# the euro sign, travels as byte 0x80 and is put back afterwards.
_EURO_BYTE = 0x80
_CLASS_BYTES = np.array([ord("?")] + [(_EURO_BYTE if c == "\u20ac" else ord(c)) for c in CHARSET], dtype=np.uint8)
assert all(ord(c) < 0x80 or c == "\u20ac" for c in CHARSET)

_DET_KW = ("min_size", "text_threshold", "low_text", "link_threshold", "canvas_size", "mag_ratio", "slope_ths", "ycenter_ths",
           "height_ths", "width_ths", "add_margin")


def _gray_bgr2gray(a: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(a, COLOR_BGR2GRAY) on the channels as given: OpenCV 4's 15-bit fixed point (R 9798, G 19235, B 3735,
    round to nearest) -- pinned by the reference's stored pre-processing outputs (tests/golden/legacy_preprocess)."""
    a = a.astype(np.int32)
    return ((a[..., 2] * 9798 + a[..., 1] * 19235 + a[..., 0] * 3735 + (1 << 14)) >> 15).astype(np.uint8)


_DECODE_POOL = None


def _decode_pool():
    global _DECODE_POOL
    if _DECODE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor

        _DECODE_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix="bbocr-decode")     # helpers of concurrent single-page callers
    return _DECODE_POOL


def decode_file(source, parallel=False):
    """What upstream's path branch holds after ``cv2.imread(path, IMREAD_GRAYSCALE)`` + ``loadImage(path)`` (skimage -> RGB):
    ``(rgb uint8 HWC, gray uint8 HW)``.  ONE stated rule for the gray plane, by container:
      * JPEG: libjpeg decodes straight to its Y plane (``out_color_space = JCS_GRAYSCALE``) -- PIL's ``draft('L')`` asks
        libjpeg for exactly that, so no RGB -> gray formula is involved;
      * single-channel files (PNG/TIFF/... mode L, 1): the stored samples;
      * every other colour file (OpenCV's PNG reader: ``png_set_rgb_to_gray(1, 0.299, 0.587)``): libpng's truncating
        15-bit sum ``(9797 R + 19234 G + 3737 B) >> 15``, grey pixels unchanged.
    ``source`` is a path or a bytes object holding the file."""
    from PIL import Image

    def _open():
        return Image.open(io.BytesIO(source)) if isinstance(source, (bytes, bytearray)) else Image.open(os.path.expanduser(str(source)))

    pil = _open()
    if pil.format in ("JPEG", "MPO") and pil.mode in ("RGB", "YCbCr"):
        # two libjpeg passes over the same file (RGB, and the Y plane alone).  `parallel` (the single-page call Reader.readtext(path),
        # the reference's own pattern): side by side on two threads -- PIL releases the GIL while it decodes, so the page costs one decode
        # time instead of two (8.9 -> 4.9 ms for 1280x960).  Callers that already decode on a thread pool (extractor_batch) keep it serial.
        def _y_plane():
            y = _open()
            y.draft("L", y.size)
            return np.ascontiguousarray(y.convert("L"))

        fut = _decode_pool().submit(_y_plane) if parallel else None
        rgb = np.ascontiguousarray(pil.convert("RGB"))
        grey = fut.result() if fut is not None else _y_plane()
        if grey.shape != rgb.shape[:2]:                      # draft() may not scale; keep the rule total
            grey = np.ascontiguousarray(pil.convert("L"))
        return rgb, grey
    rgb = np.ascontiguousarray(pil.convert("RGB"))
    if pil.mode in ("L", "1"):
        return rgb, np.ascontiguousarray(pil.convert("L"))
    a = rgb.astype(np.int32)
    return rgb, ((a[..., 0] * 9797 + a[..., 1] * 19234 + a[..., 2] * 3737) >> 15).astype(np.uint8)


def _pil_pixels_zero_copy(pil):
    """``uint8 [H,W,4]`` view of a loaded 3-band PIL image's own storage (4 bytes per pixel), or None when this Pillow / pyarrow cannot
    export it (older Pillow, image held in several memory blocks, pyarrow absent).  The view keeps the image alive."""
    try:
        import pyarrow as pa

        pil.load()
        arr = pa.array(pil)                                          # Image.__arrow_c_array__: fixed_size_list<uint8>[4], no copy
        v = arr.values.to_numpy(zero_copy_only=True)
        W, H = pil.size
        return v.reshape(H, W, 4) if v.size == H * W * 4 and v.dtype == np.uint8 else None
    except Exception:
        return None


def decode_file_ycc(source, padded=False):
    """A YCbCr-coded JPEG (JFIF, or Adobe marker with transform 1) decoded ONCE into libjpeg's YCbCr triples, ``uint8 [H,W,3]``; ``None``
    for every other file (callers fall back to ``decode_file``).  The two planes upstream's path branch reads are both functions of this
    one decode: the Y channel is ``cv2.imread(path, IMREAD_GRAYSCALE)``'s plane, and the RGB image is libjpeg's pointwise
    ``ycc_rgb_convert`` of the triple, which the device applies (``bbocr_op_ycc_to_rgb``) -- half the host's decode work of
    ``decode_file`` (8.9 -> 4.6 ms of one core for a 1280x960 page) and no second pass over the file.  ``source``: path or bytes.

    ``padded=True`` (decode pools): the result may be ``uint8 [H,W,4]`` -- Pillow's own pixel storage (Y Cb Cr x), exported without a copy
    through the Arrow C data interface (Pillow >= 11.2 + pyarrow) -- instead of the tight ``[H,W,3]`` that ``tobytes`` assembles while holding
    the interpreter lock (1.3 ms per page: with eight decode threads that serialised copy was what bounded the pool)."""
    from PIL import Image

    try:
        pil = Image.open(io.BytesIO(source)) if isinstance(source, (bytes, bytearray)) else Image.open(os.path.expanduser(str(source)))
        if pil.format not in ("JPEG", "MPO") or pil.mode != "RGB":
            return None                                              # greyscale / CMYK / YCCK files and other containers
        if "jfif" not in pil.info and pil.info.get("adobe_transform") != 1:
            return None                                              # may be RGB-coded (libjpeg guesses from the component ids): not taken
        size = pil.size
        pil.draft("YCbCr", size)
        if pil.mode != "YCbCr" or pil.size != size:
            return None
        if padded:
            v = _pil_pixels_zero_copy(pil)
            if v is not None:
                return v
        ycc = np.asarray(pil)
        if ycc.ndim != 3 or ycc.shape[2] != 3 or ycc.dtype != np.uint8:
            return None
        return np.ascontiguousarray(ycc)
    except Exception:
        return None


def reformat_input(image, device_gray=False, parallel_decode=False):
    """easyocr/utils.py::reformat_input -> (RGB uint8 HWC, gray uint8 HW); decode is host work (PIL).

    ``device_gray=True`` returns ``None`` for the gray plane wherever upstream derives it from the colour array with
    ``cv2.cvtColor(BGR2GRAY)``: the device computes exactly that (``gray_kernel``), which saves ~2 ms of numpy per 1280x960 page."""
    from PIL import Image

    _gray = (lambda a: None) if device_gray else _gray_bgr2gray

    if isinstance(image, (str, os.PathLike)):
        return decode_file(image, parallel=parallel_decode)
    if isinstance(image, (bytes, bytearray)):
        pil = Image.open(io.BytesIO(bytes(image)))
        img = np.ascontiguousarray(pil.convert("RGB"))
        return img, _gray(img)
    if isinstance(image, np.ndarray):
        if image.dtype != np.uint8:
            raise ValueError("Invalid input type. numpy input must be uint8")
        if image.ndim == 2:
            return np.ascontiguousarray(np.repeat(image[:, :, None], 3, axis=2)), np.ascontiguousarray(image)
        if image.ndim == 3 and image.shape[2] == 1:
            g = np.ascontiguousarray(image[:, :, 0])
            return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2)), g
        if image.ndim == 3 and image.shape[2] == 3:
            return np.ascontiguousarray(image), _gray(image)
        if image.ndim == 3 and image.shape[2] == 4:
            img = np.ascontiguousarray(image[:, :, :3][:, :, ::-1])
            return img, _gray(img)
    elif hasattr(image, "convert"):
        arr = np.asarray(image.convert("RGB"))
        return np.ascontiguousarray(arr[:, :, ::-1]), _gray_bgr2gray(arr)     # (gray of the UN-flipped array: not what the device would derive)
    raise ValueError("Invalid input type. Supporting format = string(file path or url), bytes, numpy array")


def format_output(result, output_format="standard", paragraph=False, detail=1):
    """Tail of easyocr.Reader.readtext: ``detail == 0`` (already text only) wins, then 'dict' / 'json' re-shape each item; the confidence key
    is spelled 'confident' upstream."""
    if detail == 0 or output_format == "standard":
        return result
    if output_format == "dict":
        if paragraph:
            return [{"boxes": item[0], "text": item[1]} for item in result]
        return [{"boxes": item[0], "text": item[1], "confident": item[2]} for item in result]
    if output_format == "json":
        import json

        if paragraph:
            return [json.dumps({"boxes": [list(map(int, lst)) for lst in item[0]], "text": item[1]}, ensure_ascii=False) for item in result]
        return [json.dumps({"boxes": [list(map(int, lst)) for lst in item[0]], "text": item[1], "confident": item[2]}, ensure_ascii=False)
                for item in result]
    raise NotImplementedError(f"output_format {output_format!r}")


def ignore_mask(character, lang_char, allowlist=None, blocklist=None):
    """easyocr.Reader.recognize's ``ignore_char`` rule as the 128-bit class mask of ``bbocr_params.ignore_mask``: with an allowlist
    every character outside it, else the blocklist, else the characters of the model that are not in the language list (none for
    ``['en']`` + english_g2).  Class index = position in ``character`` (0 is the CTC blank and is never ignored)."""
    if allowlist:
        ignore = set(character[1:]) - set(allowlist)
    elif blocklist:
        ignore = set(blocklist)
    else:
        ignore = set(character[1:]) - set(lang_char)
    words = [0, 0, 0, 0]
    for i, ch in enumerate(character):
        if i and ch in ignore:
            words[i >> 5] |= 1 << (i & 31)
    return words


def get_paragraph(raw_result, x_ths=1, y_ths=0.5, mode="ltr"):
    """easyocr/utils.py::get_paragraph: greedy clustering of result boxes into paragraphs (a box joins the current group when one
    of its x extremes and one of its y extremes fall inside the group's extent grown by x_ths / y_ths mean heights), then reading
    order inside a group: repeatedly the left-most (ltr) box among those within 0.4 mean heights of the top-most centre.
    Returns ``[[box, text]]`` (no confidence), like upstream."""
    box_group = []
    for box in raw_result:
        all_x = [int(coord[0]) for coord in box[0]]
        all_y = [int(coord[1]) for coord in box[0]]
        min_x, max_x, min_y, max_y = min(all_x), max(all_x), min(all_y), max(all_y)
        box_group.append([box[1], min_x, max_x, min_y, max_y, max_y - min_y, 0.5 * (min_y + max_y), 0])
    current_group = 1
    while len([b for b in box_group if b[7] == 0]) > 0:
        box_group0 = [b for b in box_group if b[7] == 0]
        if len([b for b in box_group if b[7] == current_group]) == 0:
            box_group0[0][7] = current_group
        else:
            cur = [b for b in box_group if b[7] == current_group]
            mean_height = float(np.mean([b[5] for b in cur]))
            min_gx = min(b[1] for b in cur) - x_ths * mean_height
            max_gx = max(b[2] for b in cur) + x_ths * mean_height
            min_gy = min(b[3] for b in cur) - y_ths * mean_height
            max_gy = max(b[4] for b in cur) + y_ths * mean_height
            add_box = False
            for b in box_group0:
                same_h = (min_gx <= b[1] <= max_gx) or (min_gx <= b[2] <= max_gx)
                same_v = (min_gy <= b[3] <= max_gy) or (min_gy <= b[4] <= max_gy)
                if same_h and same_v:
                    b[7] = current_group
                    add_box = True
                    break
            if not add_box:
                current_group += 1
    result = []
    for i in sorted(set(b[7] for b in box_group)):
        cur = [b for b in box_group if b[7] == i]
        mean_height = float(np.mean([b[5] for b in cur]))
        min_gx, max_gx = min(b[1] for b in cur), max(b[2] for b in cur)
        min_gy, max_gy = min(b[3] for b in cur), max(b[4] for b in cur)
        text = ""
        while len(cur) > 0:
            highest = min(b[6] for b in cur)
            candidates = [b for b in cur if b[6] < highest + 0.4 * mean_height]
            if mode == "ltr":
                key = min(b[1] for b in candidates)
                best = [b for b in candidates if b[1] == key][-1]
            else:
                key = max(b[2] for b in candidates)
                best = [b for b in candidates if b[2] == key][-1]
            text += " " + best[0]
            cur.remove(best)
        result.append([[[min_gx, min_gy], [max_gx, min_gy], [max_gx, max_gy], [min_gx, max_gy]], text[1:]])
    return result


def auto_host_threads(local_world=None, cpus=None):
    """``bbocr_config::host_threads`` for this process: 0 (the library sizes its pools from the process's own CPU share) unless several
    ranks share the node un-pinned (torchrun sets LOCAL_WORLD_SIZE): then this rank's share of the CPUs it may run on, at most 16."""
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    if local_world <= 1:
        return 0
    if cpus is None:
        cpus = len(os.sched_getaffinity(0))
    return max(1, min(16, cpus // local_world))


def _as_tensor(torch, a):
    """CPU tensor sharing ``a``'s memory, as the SOURCE of a copy.  Arrays decoded by PIL are read-only views of a bytes object; torch only
    warns about those because a write through the tensor would be undefined -- nothing here writes, so the warning is silenced for this
    one call instead of paying a 3.7-MB copy per page to make the array writable."""
    if a.flags.writeable:
        return torch.from_numpy(a)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        return torch.from_numpy(a)


class Reader:
    """Drop-in for ``easyocr.Reader`` (English ``english_g2`` recogniser + CRAFT detector) on one MI355X."""

    def __init__(self, lang_list, gpu=True, model_storage_directory=None, user_network_directory=None,
                 detect_network="craft", recog_network="standard", download_enabled=True, detector=True, recognizer=True,
                 verbose=True, quantize=True, cudnn_benchmark=False, weights=None, device_index=None, det_sub_batch=0,
                 rec_max_cols=0, precision=None, call_slots=0, host_threads=None, **_ignored):
        import torch

        if list(lang_list) != ["en"]:
            raise ValueError(f"{lang_list} is not supported: this backend ships the English (english_g2) model only")
        if detect_network != "craft":
            raise ValueError("only detect_network='craft' is implemented")
        if not torch.cuda.is_available():
            raise RuntimeError("bb_ocr_amd.Reader needs a HIP device (MI355X); there is no CPU path")
        self._torch = torch
        self.device_index = torch.cuda.current_device() if device_index is None else int(device_index)
        self.device = f"cuda:{self.device_index}"
        self._lib = _lib.load()
        if precision is None:       # the reference constructs Reader(["en"], gpu=...) (enhanced_extractor.py:153): the mode comes from the environment
            # default "fp16": the cheapest mode whose boxes AND strings equalled the fp32 CPU path's on everything measured -- 2,051 boxes
            # of synthetic pages, 471 of dense A4 scans, 110 on the reference's seven real images.  "mixed" (bf16
            # detector) is 1.6 % faster and equally exact on binary-ink pages, but flips threshold decisions on continuous-tone images
            precision = os.environ.get("BBOCR_PRECISION", "fp16").strip().lower()
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")      # see bbocr_config::precision (include/bbocr.h)
        self.precision = precision
        if host_threads is None:
            host_threads = auto_host_threads()
        self.host_threads = int(host_threads)
        cfg = _lib.bbocr_config(device=self.device_index, det_sub_batch=int(det_sub_batch), rec_max_cols=int(rec_max_cols),
                                precision=_lib.PRECISIONS[precision], call_slots=int(call_slots), host_threads=self.host_threads)
        h = C.c_void_p()
        rc = self._lib.bbocr_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"bbocr_create failed with status {rc}")
        self._h = h
        self.character = CHARACTER
        self.lang_char = list(CHARSET)
        if isinstance(weights, str) and weights == "empty":
            # multi-GPU receiver: the packed plans are laid out without values and filled by import_weights_blob (dist.broadcast_packed)
            for which, wanted in ((0, detector), (1, recognizer)):
                if wanted:
                    self._check(self._lib.bbocr_alloc_weights(self._h, which))
            return
        craft_state, crnn_state = self._resolve_weights(weights, model_storage_directory)
        if detector:
            self._load(0, craft_state)
        if recognizer:
            self._load(1, crnn_state)

    # -- weights -------------------------------------------------------------------
    @staticmethod
    def _resolve_weights(weights, model_storage_directory):
        if isinstance(weights, tuple):
            return weights
        if weights == "synthetic" or os.environ.get("BBOCR_SYNTHETIC_WEIGHTS", "") == "1":
            return _weights.designed_craft_state(0), _weights.synthetic_crnn_state(0)
        directory = model_storage_directory or os.environ.get("BBOCR_WEIGHTS_DIR") or os.path.expanduser("~/.EasyOCR/model")
        return _weights.load_checkpoint_dir(directory)

    def _load(self, which, state):
        arr, keep = _weights.to_descs(state)
        self._check(self._lib.bbocr_load_weights(self._h, which, arr, len(arr)))
        del keep

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.bbocr_last_error(self._h)
            raise RuntimeError(f"libbbocr status {rc}: {msg.decode(errors='replace') if msg else ''}")

    # -- packed weights as one device blob (multi-GPU broadcast, include/bbocr.h) -------
    def weights_blob_size(self):
        n = C.c_size_t()
        self._check(self._lib.bbocr_weights_blob_size(self._h, C.byref(n)))
        return int(n.value)

    def export_weights_blob(self):
        """-> uint8 device tensor holding every packed weight block of this context (BN folded, element type rounded, MFMA order)."""
        blob = self._torch.empty(self.weights_blob_size(), dtype=self._torch.uint8, device=self.device)
        self._check(self._lib.bbocr_weights_export(self._h, C.c_void_p(blob.data_ptr()), blob.numel()))
        return blob

    def import_weights_blob(self, blob):
        """Fill a ``Reader(weights="empty")`` from another rank's blob (same precision, same networks)."""
        self._dev_u8(blob, "weight blob", 1, (self.weights_blob_size(),))
        self._check(self._lib.bbocr_weights_import(self._h, C.c_void_p(blob.data_ptr()), blob.numel()))

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.bbocr_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------
    def _params(self, kw):
        p = _lib.bbocr_params()
        self._lib.bbocr_default_params(C.byref(p))
        for k in _DET_KW + ("contrast_ths", "adjust_contrast"):
            if k in kw and kw[k] is not None:
                setattr(p, k, kw[k])
        for i, w in enumerate(ignore_mask(self.character, self.lang_char, kw.get("allowlist"), kw.get("blocklist"))):
            p.ignore_mask[i] = w
        decoder = kw.get("decoder", "greedy")
        self._unsupported(decoder, None, None, None, False, "standard")
        rotation = list(kw.get("rotation_info") or [])
        if len(rotation) > 3 or any(a not in (90, 180, 270) for a in rotation):
            raise ValueError("rotation_info: up to three angles out of 90, 180, 270")
        for i, a in enumerate(rotation):
            p.rotation_info[i] = int(a)
        if decoder == "beamsearch":          # BBOCR_DECODER_BEAMSEARCH
            p.decoder, p.beam_width = 1, int(kw.get("beamWidth", 5))
        return p

    def _to_dev(self, arr):
        """Host array -> device tensor; a LIST of equal-shape arrays -> one device tensor ``[len, ...]`` filled page by page (no host-side
        ``np.stack``: for 64 decoded pages that copy is 236 MB on one thread, the largest serial cost of a decode-bound caller)."""
        torch = self._torch
        if isinstance(arr, (list, tuple)):
            if not arr:
                raise ValueError("empty page list")
            pages = [np.ascontiguousarray(a) for a in arr]           # (kept alive until the call below has returned)
            first = pages[0]
            if any(a.dtype != np.uint8 or a.shape != first.shape for a in pages):
                raise ValueError("pages of one batch must be uint8 arrays of one shape")
            t = torch.empty((len(pages),) + first.shape, dtype=torch.uint8, device=self.device)
            torch.cuda.current_stream(self.device_index).synchronize()
            ptrs = (C.c_void_p * len(pages))(*[a.ctypes.data for a in pages])
            # ONE call for the whole batch: the interpreter lock is released once, not once per page (with a decode pool running, every
            # re-acquisition can wait for a thread that holds it)
            self._check(self._lib.bbocr_upload_pages(self._h, ptrs, len(pages), first.nbytes, C.c_void_p(t.data_ptr())))
            return t
        arr = np.ascontiguousarray(arr)
        t = _as_tensor(torch, arr).to(self.device)
        torch.cuda.current_stream(self.device_index).synchronize()
        return t

    def _dev_u8(self, t, name, ndim, shape=None):
        """A tensor about to cross the C ABI as a raw pointer: anything but a contiguous uint8 tensor of the expected shape on THIS
        context's device would be an out-of-bounds device access (a GPU fault kills the process; the reference relies on exceptions,
        enhanced_extractor.py:529-531) -> ValueError before the ctypes call."""
        torch = self._torch
        if not isinstance(t, torch.Tensor):
            raise ValueError(f"{name}: expected a torch uint8 device tensor, got {type(t).__name__}")
        if not t.is_cuda or t.device.index != self.device_index:
            raise ValueError(f"{name}: tensor lives on {t.device}, this Reader runs on {self.device}")
        if t.dtype != torch.uint8:
            raise ValueError(f"{name}: dtype must be uint8, got {t.dtype}")
        if t.ndim != ndim or (shape is not None and tuple(t.shape) != tuple(shape)) or min(t.shape) <= 0:
            raise ValueError(f"{name}: bad shape {tuple(t.shape)}" + (f", expected {tuple(shape)}" if shape is not None else f" ({ndim}-D expected)"))
        if not t.is_contiguous():
            raise ValueError(f"{name}: tensor must be contiguous")
        return t

    def _collect(self, res_p, detail=1):
        """bbocr_result -> per-page [(bbox, text, conf)] (upstream's list shape); numpy views instead of per-element ctypes access."""
        r = res_p.contents
        out = []
        try:
            B = r.n_images
            box_off = np.ctypeslib.as_array(r.box_off, shape=(B + 1,)).tolist()
            nb = box_off[-1]
            items = []
            if nb:
                quads = np.ctypeslib.as_array(r.quads, shape=(nb, 8))
                text_off = np.ctypeslib.as_array(r.text_off, shape=(nb + 1,)).tolist()
                conf = np.ctypeslib.as_array(r.conf, shape=(nb,)).tolist()
                nt = text_off[-1]
                # every box's text in ONE decode, then plain str slices (a per-box join over numpy objects cost 3 ms per 64 pages); a euro sign
                # is one character before and after the substitution, so the offsets stay valid
                chars = _CLASS_BYTES[np.ctypeslib.as_array(r.text_idx, shape=(max(nt, 1),))[:nt]].tobytes().decode("latin-1") if nt else ""
                if "\x80" in chars:
                    chars = chars.replace("\x80", "\u20ac")
                boxes = quads.astype(np.int64).reshape(nb, 4, 2).tolist()          # horizontal boxes: python ints, like upstream
                free = np.flatnonzero(np.ctypeslib.as_array(r.is_free, shape=(nb,)))
                if free.size:                                                        # free boxes keep their float corners
                    for i, q in zip(free.tolist(), quads[free].reshape(-1, 4, 2).tolist()):
                        boxes[i] = q
                items = list(zip(boxes, [chars[a:b] for a, b in zip(text_off[:-1], text_off[1:])], conf))
            out = [items[box_off[b]:box_off[b + 1]] for b in range(B)]
        finally:
            self._lib.bbocr_free_result(res_p)
        if detail == 0:
            return [[item[1] for item in page] for page in out]
        return out

    def stage_times(self):
        ms = (C.c_float * 8)()
        self._check(self._lib.bbocr_stage_times(self._h, ms, 8))
        keys = ("detector_net", "ccl_device", "box_geometry_host", "crops", "recognizer_net", "ctc", "contrast_retry", "total")
        return dict(zip(keys, [float(v) for v in ms]))

    def set_profiling(self, on: bool):
        """Time every conv_mfma launch with HIP events on the library's stream (bench.py roofline leg)."""
        self._check(self._lib.bbocr_set_profiling(self._h, int(on)))      # 0 off, 1/True detector launches, 2 also the recogniser's

    def conv_profile(self, group: int):
        """-> (sum of launch ms, sum of algorithmic flops, launches) for group 0 (detector) / 1 (recogniser)."""
        ms, fl, n = C.c_double(), C.c_double(), C.c_longlong()
        self._check(self._lib.bbocr_conv_profile(self._h, group, C.byref(ms), C.byref(fl), C.byref(n)))
        return ms.value, fl.value, n.value

    # -- easyocr surface ---------------------------------------------------------------
    def readtext_device(self, rgb_dev, gray_dev=None, **kw):
        """Batch entry for pages already resident in HBM: uint8 torch tensors [B,H,W,3] (+ optional [B,H,W])."""
        self._dev_u8(rgb_dev, "rgb", 4)
        B, H, W, ch = rgb_dev.shape
        if ch != 3:
            raise ValueError(f"rgb: last dimension must be 3, got {ch}")
        if gray_dev is not None:
            self._dev_u8(gray_dev, "gray", 3, (B, H, W))
        p = self._params(kw)
        res = C.POINTER(_lib.bbocr_result)()
        gp = C.c_void_p(gray_dev.data_ptr()) if gray_dev is not None else C.c_void_p(None)
        self._check(self._lib.bbocr_readtext_batch(self._h, C.c_void_p(rgb_dev.data_ptr()), gp, B, H, W, C.byref(p), C.byref(res)))
        if kw.get("paragraph"):
            # Reader.readtext tail: get_paragraph on the raw result, then detail == 0 keeps the text only
            pages = [get_paragraph(page, x_ths=kw.get("x_ths", 1.0), y_ths=kw.get("y_ths", 0.5), mode="ltr") for page in self._collect(res, 1)]
            return [[item[1] for item in page] for page in pages] if kw.get("detail", 1) == 0 else pages
        return self._collect(res, kw.get("detail", 1))

    def pages_from_ycc(self, ycc_dev):
        """Device tensor ``uint8 [B,H,W,3]`` of libjpeg YCbCr triples (``decode_file_ycc``) -> ``(rgb_dev, gray_dev)``: the RGB pages and Y
        planes ``readtext_device`` takes, computed on the card by libjpeg's own integer colour conversion."""
        self._dev_u8(ycc_dev, "ycc", 4)
        B, H, W, ch = ycc_dev.shape
        if ch not in (3, 4):                                         # 4: Pillow's padded pixels (decode_file_ycc(padded=True))
            raise ValueError(f"ycc: last dimension must be 3 or 4, got {ch}")
        torch = self._torch
        rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device=ycc_dev.device)
        gray = torch.empty((B, H, W), dtype=torch.uint8, device=ycc_dev.device)
        torch.cuda.current_stream(self.device_index).synchronize()       # the library runs on its own stream
        self._check(self._lib.bbocr_op_ycc_to_rgb(self._h, C.c_void_p(ycc_dev.data_ptr()), B * H * W, ch, C.c_void_p(rgb.data_ptr()),
                                                  C.c_void_p(gray.data_ptr())))
        return rgb, gray

    def readtext_ycc_arrays(self, ycc, **kw):
        """Host array ``uint8 [B,H,W,3]`` of once-decoded JPEG pages (``decode_file_ycc``) -> per-page results, identical to
        ``readtext_arrays(rgb, gray)`` of the same files decoded twice."""
        if not isinstance(ycc, (list, tuple)):               # a list of [H,W,3] pages is uploaded page by page (Reader._to_dev)
            ycc = np.asarray(ycc)
            if ycc.dtype != np.uint8 or ycc.ndim != 4 or ycc.shape[3] not in (3, 4):
                raise ValueError("readtext_ycc_arrays expects uint8 [B,H,W,3] (or [B,H,W,4]: Pillow's padded pixels)")
        rgb, gray = self.pages_from_ycc(self._to_dev(ycc))
        return self.readtext_device(rgb, gray, **kw)

    def readtext_stream(self, batches, in_flight=2, **kw):
        """Device page batches in, per-batch results out, IN ORDER, with up to ``in_flight`` calls running on this Reader at once.

        The reference shares one Reader between ``ThreadPoolExecutor`` workers (batch_processor_enhanced.py:215, default 2); libbbocr gives
        each concurrent call its own call slot (``bbocr_config::call_slots``), so batch k+1's detector is on the card while batch k's host
        thread finishes box geometry, CTC read-back and result marshalling.  ``batches`` yields uint8 ``[B,H,W,3]`` device tensors or
        ``(rgb, gray)`` pairs (it may produce them lazily, e.g. H2D copies: it is drained one batch ahead of the workers).  Results are those
        of calling ``readtext_device`` batch by batch."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        in_flight = max(1, int(in_flight))
        with ThreadPoolExecutor(max_workers=in_flight, thread_name_prefix="bbocr-call") as ex:
            pending = deque()
            for b in batches:
                rgb, gray = b if isinstance(b, tuple) else (b, None)
                pending.append(ex.submit(self.readtext_device, rgb, gray, **kw))
                if len(pending) > in_flight:                 # one batch queued behind the running ones: a worker never waits for the producer
                    yield pending.popleft().result()
            while pending:
                yield pending.popleft().result()

    def readtext(self, image, decoder="greedy", beamWidth=5, batch_size=1, workers=0, allowlist=None, blocklist=None, detail=1,
                 rotation_info=None, paragraph=False, min_size=20, contrast_ths=0.1, adjust_contrast=0.5, filter_ths=0.003,
                 text_threshold=0.7, low_text=0.4, link_threshold=0.4, canvas_size=2560, mag_ratio=1.0, slope_ths=0.1, ycenter_ths=0.5,
                 height_ths=0.5, width_ths=0.5, y_ths=0.5, x_ths=1.0, add_margin=0.1, threshold=0.2, bbox_min_score=0.2, bbox_min_size=3,
                 max_candidates=0, output_format="standard"):
        """``image -> [(bbox, text, confidence)]`` exactly as ``easyocr.Reader.readtext`` shapes it."""
        self._unsupported(decoder, allowlist, blocklist, rotation_info, paragraph, output_format)
        # one page per call is latency-bound: RGB and the Y plane decoded side by side on two threads (4.9 ms) beat the single YCbCr decode
        # (decode_file_ycc, ~5.6 ms on one core) that the throughput-bound callers (extractor_batch) take
        img, grey = reformat_input(image, device_gray=True, parallel_decode=True)
        kw = dict(min_size=min_size, contrast_ths=contrast_ths, adjust_contrast=adjust_contrast, text_threshold=text_threshold,
                  low_text=low_text, link_threshold=link_threshold, canvas_size=canvas_size, mag_ratio=mag_ratio, slope_ths=slope_ths,
                  ycenter_ths=ycenter_ths, height_ths=height_ths, width_ths=width_ths, add_margin=add_margin, detail=detail,
                  allowlist=allowlist, blocklist=blocklist, paragraph=paragraph, x_ths=x_ths, y_ths=y_ths, decoder=decoder,
                  beamWidth=beamWidth, rotation_info=rotation_info)
        result = self.readtext_device(self._to_dev(img[None]), self._to_dev(grey[None]) if grey is not None else None, **kw)[0]
        return format_output(result, output_format, paragraph, detail)

    def readtext_batched(self, image, n_width=None, n_height=None, **kw):
        """List (or 4-D array) of pages -> list of per-page results.  Equal-size pages share one device batch."""
        self._unsupported(kw.get("decoder", "greedy"), kw.get("allowlist"), kw.get("blocklist"), kw.get("rotation_info"),
                          kw.get("paragraph", False), kw.get("output_format", "standard"))
        output_format = kw.pop("output_format", "standard")
        for k in ("batch_size", "workers", "filter_ths", "threshold", "bbox_min_score", "bbox_min_size", "max_candidates"):
            kw.pop(k, None)
        pages = [reformat_input(im) for im in image]
        if n_width is not None and n_height is not None:
            from PIL import Image

            pages = [(np.asarray(Image.fromarray(a).resize((n_width, n_height), Image.BILINEAR)),
                      np.asarray(Image.fromarray(g).resize((n_width, n_height), Image.BILINEAR))) for a, g in pages]
        out = [None] * len(pages)
        by_shape = {}
        for i, (a, g) in enumerate(pages):
            by_shape.setdefault(a.shape, []).append(i)
        max_pages = int(os.environ.get("BBOCR_MAX_DEVICE_BATCH", "64"))     # pages per device batch; larger groups stream, two batches in flight
        for _, idxs in by_shape.items():
            chunks = [idxs[k:k + max_pages] for k in range(0, len(idxs), max_pages)]
            # (lists: the pages reach the card in one upload call per batch, without a host-side np.stack)
            feed = ((self._to_dev([pages[i][0] for i in ch]), self._to_dev([pages[i][1] for i in ch])) for ch in chunks)
            for ch, res in zip(chunks, self.readtext_stream(feed, **kw)):
                for i, r in zip(ch, res):
                    out[i] = format_output(r, output_format, kw.get("paragraph", False), kw.get("detail", 1))
        return out

    def readtext_files(self, paths, max_batch=64, decode_workers=None, **kw):
        """Image FILES in, ``Reader.readtext(path)``'s result per file out (same order), through the decode pool / upload stage / two device
        batches in flight of ``extractor_batch.read_files``: JPEG files are decoded once (YCbCr triples), pages of one size share device
        batches.  A file that cannot be decoded or read maps to ``[]``."""
        from .extractor_batch import read_files

        paths = list(paths)
        res = read_files(self, paths, None, max_batch, decode_workers, **kw)
        return [res[i] for i in range(len(paths))]

    def readtext_arrays(self, rgb, gray=None, **kw):
        """Host arrays ``uint8 [B,H,W,3]`` (+ optional ``[B,H,W]`` gray planes, else derived on device) -> per-page results."""
        if isinstance(rgb, (list, tuple)):                   # lists of [H,W,3] (+ [H,W]) pages: uploaded page by page (Reader._to_dev)
            if gray is not None and len(gray) != len(rgb):
                raise ValueError("readtext_arrays: one gray plane per page")
            return self.readtext_device(self._to_dev(rgb), self._to_dev(gray) if gray is not None else None, **kw)
        rgb = np.asarray(rgb)
        if rgb.dtype != np.uint8 or rgb.ndim != 4 or rgb.shape[3] != 3:
            raise ValueError("readtext_arrays expects uint8 [B,H,W,3]")
        if gray is not None and (np.asarray(gray).dtype != np.uint8 or np.asarray(gray).shape != rgb.shape[:3]):
            raise ValueError("readtext_arrays: gray must be uint8 [B,H,W] matching rgb")
        return self.readtext_device(self._to_dev(rgb), self._to_dev(np.asarray(gray)) if gray is not None else None, **kw)

    def detect(self, img, min_size=20, text_threshold=0.7, low_text=0.4, link_threshold=0.4, canvas_size=2560, mag_ratio=1.0,
               slope_ths=0.1, ycenter_ths=0.5, height_ths=0.5, width_ths=0.5, add_margin=0.1, reformat=True, **_ignored):
        """``Reader.detect`` -> (horizontal_list_agg, free_list_agg) for one page."""
        if reformat:
            img, _ = reformat_input(img)
        kw = dict(min_size=min_size, text_threshold=text_threshold, low_text=low_text, link_threshold=link_threshold,
                  canvas_size=canvas_size, mag_ratio=mag_ratio, slope_ths=slope_ths, ycenter_ths=ycenter_ths, height_ths=height_ths,
                  width_ths=width_ths, add_margin=add_margin)
        heat, ratio = self.heatmap_device(self._to_dev(img[None]), **kw)
        h, f, _ = self.boxes_from_heatmap(heat, ratio, **kw)
        return [h[0]], [f[0]]

    def recognize(self, img_cv_grey, horizontal_list=None, free_list=None, decoder="greedy", beamWidth=5, detail=1, paragraph=False,
                  contrast_ths=0.1, adjust_contrast=0.5, reformat=True, rotation_info=None, allowlist=None, blocklist=None, y_ths=0.5, x_ths=1.0,
                  output_format="standard", **_ignored):
        """``Reader.recognize`` for explicit boxes of one gray page (``paragraph=True``: ``get_paragraph`` on the result, like upstream)."""
        self._unsupported(decoder, None, None, None, False, output_format)
        if paragraph:
            raw = self.recognize(img_cv_grey, horizontal_list, free_list, decoder=decoder, beamWidth=beamWidth, detail=1, paragraph=False,
                                 contrast_ths=contrast_ths, adjust_contrast=adjust_contrast, reformat=reformat, rotation_info=rotation_info,
                                 allowlist=allowlist, blocklist=blocklist)
            para = get_paragraph(raw, x_ths=x_ths, y_ths=y_ths, mode="ltr")
            return [item[1] for item in para] if detail == 0 else format_output(para, output_format, True, detail)
        if reformat:
            _, img_cv_grey = reformat_input(img_cv_grey)
        H, W = img_cv_grey.shape
        if horizontal_list is None and free_list is None:
            horizontal_list, free_list = [[0, W, 0, H]], []
        res = self.recognize_device(self._to_dev(img_cv_grey[None]), [horizontal_list or []], [free_list or []],
                                    contrast_ths=contrast_ths, adjust_contrast=adjust_contrast, detail=detail, decoder=decoder,
                                    beamWidth=beamWidth, rotation_info=rotation_info, allowlist=allowlist, blocklist=blocklist)[0]
        return format_output(res, output_format, False, detail)

    # -- stage-level entry points (tests, bench) --------------------------------------------
    def detect_dims(self, H, W, canvas_size=2560, mag_ratio=1.0):
        v = [C.c_int() for _ in range(4)]
        ratio = C.c_double()
        rc = self._lib.bbocr_detect_dims(H, W, canvas_size, float(mag_ratio), *[C.byref(x) for x in v], C.byref(ratio))
        if rc != 0:
            raise ValueError("bad page size")
        return v[0].value, v[1].value, v[2].value, v[3].value, ratio.value

    def heatmap_device(self, rgb_dev, **kw):
        """uint8 [B,H,W,3] device tensor -> (fp32 [B,h,w,2] device tensor, ratio)."""
        self._dev_u8(rgb_dev, "rgb", 4)
        B, H, W, ch = rgb_dev.shape
        if ch != 3:
            raise ValueError(f"rgb: last dimension must be 3, got {ch}")
        _, _, h, w, ratio = self.detect_dims(H, W, kw.get("canvas_size", 2560), kw.get("mag_ratio", 1.0))
        heat = self._torch.empty((B, h, w, 2), dtype=self._torch.float32, device=self.device)
        p = self._params(kw)
        self._check(self._lib.bbocr_detect(self._h, C.c_void_p(rgb_dev.data_ptr()), B, H, W, C.byref(p), C.c_void_p(heat.data_ptr())))
        return heat, ratio

    def boxes_from_heatmap(self, heat_dev, ratio, **kw):
        """fp32 [B,h,w,2] device tensor -> (horizontal lists, free lists, polys) per page."""
        torch = self._torch
        if (not isinstance(heat_dev, torch.Tensor) or not heat_dev.is_cuda or heat_dev.device.index != self.device_index
                or heat_dev.dtype != torch.float32 or heat_dev.ndim != 4 or heat_dev.shape[3] != 2 or not heat_dev.is_contiguous()):
            raise ValueError(f"heat-map: expected a contiguous float32 [B,h,w,2] tensor on {self.device}")
        B, h, w, _ = heat_dev.shape
        p = self._params(kw)
        bl = C.POINTER(_lib.bbocr_boxlist)()
        self._check(self._lib.bbocr_boxes(self._h, C.c_void_p(heat_dev.data_ptr()), B, h, w, float(ratio), C.byref(p), C.byref(bl)))
        try:
            b = bl.contents
            hori = [[[b.hori[i * 4 + k] for k in range(4)] for i in range(b.hori_off[n], b.hori_off[n + 1])] for n in range(B)]
            free = [[[[b.free_q[i * 8 + 2 * k], b.free_q[i * 8 + 2 * k + 1]] for k in range(4)]
                     for i in range(b.free_off[n], b.free_off[n + 1])] for n in range(B)]
            polys = [[[b.polys[i * 8 + k] for k in range(8)] for i in range(b.poly_off[n], b.poly_off[n + 1])] for n in range(B)]
        finally:
            self._lib.bbocr_free_boxlist(bl)
        return hori, free, polys

    def recognize_device(self, gray_dev, horizontal_lists, free_lists, **kw):
        self._dev_u8(gray_dev, "gray", 3)
        B, H, W = gray_dev.shape
        if len(horizontal_lists) != B or len(free_lists) != B:
            raise ValueError("one horizontal list and one free list per page")
        n_h = [len(x) for x in horizontal_lists]
        n_f = [len(x) for x in free_lists]
        hoff = (C.c_int * (B + 1))(*np.concatenate([[0], np.cumsum(n_h)]).astype(int).tolist())
        foff = (C.c_int * (B + 1))(*np.concatenate([[0], np.cumsum(n_f)]).astype(int).tolist())
        poff = (C.c_int * (B + 1))()
        hflat = [int(v) for page in horizontal_lists for box in page for v in box]
        fflat = [float(v) for page in free_lists for box in page for pt in box for v in pt]
        harr = (C.c_int * max(1, len(hflat)))(*hflat)
        farr = (C.c_double * max(1, len(fflat)))(*fflat)
        parr = (C.c_int * 8)()
        bl = _lib.bbocr_boxlist(n_images=B, poly_off=poff, polys=parr, hori_off=hoff, hori=harr, free_off=foff, free_q=farr)
        p = self._params(kw)
        res = C.POINTER(_lib.bbocr_result)()
        self._check(self._lib.bbocr_recognize(self._h, C.c_void_p(gray_dev.data_ptr()), B, H, W, C.byref(bl), C.byref(p), C.byref(res)))
        return self._collect(res, kw.get("detail", 1))

    @staticmethod
    def _unsupported(decoder, allowlist, blocklist, rotation_info, paragraph, output_format):
        if decoder not in ("greedy", "beamsearch"):
            raise NotImplementedError("decoder='wordbeamsearch' needs easyocr's dictionary files, which are not available offline; "
                                      "'greedy' (the reference's call) and 'beamsearch' are implemented")
        if output_format not in ("standard", "dict", "json"):
            raise NotImplementedError("output_format 'free_merge' is not implemented "
                                      "(the reference calls readtext(path, paragraph=False, batch_size=1, workers=0))")
