"""ctypes binding of libbbocr.so (include/bbocr.h) -- the stub a BB-OCR maintainer would add.

There is no fallback: if the shared library (hand-written HIP kernels for gfx950) is not
built, importing this module raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C bb-ocr_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BBOCR_LIB_PATH") or os.path.join(_HERE, "libbbocr.so")   # override: A/B runs of two builds on one box


class bbocr_config(C.Structure):
    _fields_ = [("device", C.c_int), ("det_sub_batch", C.c_int), ("rec_max_cols", C.c_int), ("precision", C.c_int), ("call_slots", C.c_int),
                ("host_threads", C.c_int), ("reserved", C.c_int * 2)]


PRECISIONS = {"bf16": 0, "fp16": 1, "exact": 2, "mixed": 3, "exact_rec": 4}


class bbocr_tensor_desc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ndim", C.c_int), ("shape", C.c_int64 * 4), ("data", C.POINTER(C.c_float))]


class bbocr_params(C.Structure):
    _fields_ = [
        ("text_threshold", C.c_double), ("low_text", C.c_double), ("link_threshold", C.c_double), ("mag_ratio", C.c_double),
        ("slope_ths", C.c_double), ("ycenter_ths", C.c_double), ("height_ths", C.c_double), ("width_ths", C.c_double),
        ("add_margin", C.c_double), ("contrast_ths", C.c_double), ("adjust_contrast", C.c_double),
        ("canvas_size", C.c_int), ("min_size", C.c_int), ("ignore_mask", C.c_uint * 4), ("decoder", C.c_int), ("beam_width", C.c_int), ("rotation_info", C.c_int * 4),
    ]


class bbocr_preproc_params(C.Structure):
    _fields_ = [("scale", C.c_double), ("blur_sigma", C.c_double), ("contrast", C.c_double), ("brightness", C.c_double),
                ("clahe_clip", C.c_double), ("unsharp_radius", C.c_double), ("unsharp_percent", C.c_int), ("unsharp_threshold", C.c_int),
                ("reserved", C.c_int * 4)]


class bbocr_boxlist(C.Structure):
    _fields_ = [
        ("n_images", C.c_int), ("poly_off", C.POINTER(C.c_int)), ("polys", C.POINTER(C.c_int)),
        ("hori_off", C.POINTER(C.c_int)), ("hori", C.POINTER(C.c_int)),
        ("free_off", C.POINTER(C.c_int)), ("free_q", C.POINTER(C.c_double)),
    ]


class bbocr_result(C.Structure):
    _fields_ = [
        ("n_images", C.c_int), ("box_off", C.POINTER(C.c_int)), ("quads", C.POINTER(C.c_double)),
        ("is_free", C.POINTER(C.c_int)), ("text_off", C.POINTER(C.c_int)), ("text_idx", C.POINTER(C.c_int)),
        ("conf", C.POINTER(C.c_double)),
    ]


# every symbol include/bbocr.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
PROTOTYPES = {
    "bbocr_create": (C.c_int, [C.POINTER(bbocr_config), C.POINTER(_vp)]),
    "bbocr_destroy": (None, [_vp]),
    "bbocr_last_error": (C.c_char_p, [_vp]),
    "bbocr_default_params": (None, [C.POINTER(bbocr_params)]),
    "bbocr_load_weights": (C.c_int, [_vp, C.c_int, C.POINTER(bbocr_tensor_desc), C.c_int]),
    "bbocr_alloc_weights": (C.c_int, [_vp, C.c_int]),
    "bbocr_weights_blob_size": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "bbocr_weights_export": (C.c_int, [_vp, _vp, C.c_size_t]),
    "bbocr_weights_import": (C.c_int, [_vp, _vp, C.c_size_t]),
    "bbocr_dist_unique_id": (C.c_int, [_vp]),
    "bbocr_dist_init": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "bbocr_dist_finalize": (C.c_int, [_vp]),
    "bbocr_bcast_weights": (C.c_int, [_vp, C.c_int]),
    "bbocr_scatter_images": (C.c_int, [_vp, _vp, C.c_longlong, C.c_size_t, C.c_int, _vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "bbocr_gather_results": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "bbocr_result_pack": (C.c_int, [C.POINTER(bbocr_result), C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "bbocr_result_unpack": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.POINTER(bbocr_result))]),
    "bbocr_free_bytes": (None, [_vp]),
    "bbocr_detect_dims": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "bbocr_detect": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(bbocr_params), _vp]),
    "bbocr_boxes": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(bbocr_params), C.POINTER(C.POINTER(bbocr_boxlist))]),
    "bbocr_recognize": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(bbocr_boxlist), C.POINTER(bbocr_params),
                                  C.POINTER(C.POINTER(bbocr_result))]),
    "bbocr_readtext_batch": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(bbocr_params), C.POINTER(C.POINTER(bbocr_result))]),
    "bbocr_free_boxlist": (None, [C.POINTER(bbocr_boxlist)]),
    "bbocr_free_result": (None, [C.POINTER(bbocr_result)]),
    "bbocr_stage_times": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_int]),
    "bbocr_set_profiling": (C.c_int, [_vp, C.c_int]),
    "bbocr_conv_profile": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "bbocr_host_cpu_share": (C.c_int, []),
    "bbocr_host_component_polys": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]),
    "bbocr_host_group_boxes": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(bbocr_params), C.POINTER(C.POINTER(bbocr_boxlist))]),
    "bbocr_host_ctc_beam": (C.c_int, [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bbocr_op_conv2d": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    "bbocr_crnn_logits": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp]),
    "bbocr_op_ctc": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                               C.POINTER(C.c_uint), C.c_int]),
    "bbocr_op_resize_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int]),
    "bbocr_op_ycc_to_rgb": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp, _vp]),
    "bbocr_upload_pages": (C.c_int, [_vp, C.POINTER(C.c_void_p), C.c_int, C.c_size_t, _vp]),
    "bbocr_op_crops": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_float,
                                 _vp, C.POINTER(C.c_int), C.c_int]),
    "bbocr_preprocess_book_cover": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bbocr_preproc_defaults": (None, [C.POINTER(bbocr_preproc_params), C.c_int]),
    "bbocr_preprocess_chain": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(bbocr_preproc_params), _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bbocr_op_preprocess_stage": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_double]),
}

_lib = None


def load():
    """Load libbbocr.so and attach prototypes.  Raises RuntimeError if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X backend has no CPU fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).")
    # torch wheels bundle their own libamdhip64; libbbocr.so needs the same SONAME.  Whichever is loaded first serves both, and a
    # process with TWO HIP runtimes cannot create streams (bbocr_create fails): load torch's first so that there is only one.
    try:
        import torch  # noqa: F401
    except Exception:       # a torch-free host still gets the ABI (system ROCm runtime)
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libbbocr.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
