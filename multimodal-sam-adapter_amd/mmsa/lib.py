"""ctypes binding of libmmsa_hip.so (the C ABI declared in include/mmsa.h).

There is NO fallback: if the shared library is missing the import of this module raises, and every op
raises RuntimeError when the library reports an error.  Build it with
`python multimodal-sam-adapter_amd/build.py` (hipcc --offload-arch=gfx950)."""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMSA_LIB: an alternative build of the SAME library (tools/build_variant.sh: one source rebuilt with -D flags, e.g. -DMMSA_DEBUG_KNOBS for
# the timing ablations) -- an A/B aid of the tools under tools/; the tests and bench.py use the in-tree library.
LIB_PATH = os.environ.get("MMSA_LIB") or os.path.join(_HERE, "libmmsa_hip.so")

if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f"{LIB_PATH} not found: the MI355X HIP library is required (no CPU/PyTorch fallback exists). "
        "Build it with: python multimodal-sam-adapter_amd/build.py")

# torch must be imported first: it loads the HIP runtime it was built against, and libmmsa_hip.so's
# libamdhip64.so.7 dependency then binds to that same runtime instance (one process, one HIP runtime, so the
# library sees torch's device context, allocations and streams).
import torch  # noqa: E402,F401

_lib = ctypes.CDLL(LIB_PATH)

P = c_void_p
I = c_int
L = c_long
F = c_float

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "mmsa_version": [],
    "mmsa_last_error": [],
    "mmsa_debug_poison_lds": [ctypes.c_uint, P],
    "mmsa_event_create": [POINTER(c_void_p)],
    "mmsa_event_record": [P, P],
    "mmsa_event_elapsed_ms": [P, P, POINTER(c_float)],
    "mmsa_event_destroy": [P],
    "mmsa_ms_deform_attn_forward": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "mmsa_ms_deform_attn_backward": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "mmsa_msda_fused": [P, P, P, P, L, P, P, L, P, L, I, I, I, I, I, I, I, I, P, P],
    "mmsa_msda_fused_planes": [P, L, I, P, P, P, L, P, P, L, P, L, I, I, I, I, I, I, I, I, P, P],
    "mmsa_gemm_split3": [P, P, L, L, P, L, P, L, P, P, L, L, I, F, P, L, L, P, L, L, I, I, I, I, I, F, I, I, I, I, I, I, I, P, P, P, I, P, P],
    "mmsa_convnext_mlp_fused": [P, L, L, P, L, P, L, P, P, P, P, L, L, I, I, I, I, I, P, P],
    "mmsa_rowstats_finalize": [P, I, I, I, F, P, P],
    "mmsa_zero_bytes": [P, ctypes.c_size_t, P],
    "mmsa_split_planes": [P, L, I, I, I, P, I, P, P],
    "mmsa_attention": [P, L, P, P, P, L, I, I, I, I, I, I, F, P],
    "mmsa_attention_planes": [P, L, P, P, P, L, I, I, I, I, I, I, F, I, I, P, P],
    "mmsa_relpos_bias": [P, L, P, P, P, I, I, I, I, I, I, P],
    "mmsa_relpos_bias_planes": [P, L, P, P, P, I, I, I, I, I, I, P],
    "mmsa_layernorm_rows": [P, L, P, P, F, P, L, P, L, P, L, I, I, I, I, I, I, L, L, I, I, P, P],
    "mmsa_colstats": [P, L, L, P, I, I, I, P, I, P],
    "mmsa_ffrm_finalize": [P, I, I, I, F, F, P, P, P, P, P, P, P, P],
    "mmsa_lnhw_apply": [P, L, P, P, P, P, P, P, L, I, I, I, P],
    "mmsa_dwconv_nhwc": [P, L, L, P, P, P, L, L, P, L, L, I, I, I, I, I, I, I, I, P, P, P],
    "mmsa_dwpair_gate": [P, L, P, P, L, P, L, I, I, I, I, P],
    "mmsa_gconv_nhwc": [P, L, P, P, P, L, I, I, I, I, I, I, I, I, P],
    "mmsa_im2col_nchw": [P, I, I, I, I, I, I, I, P, I, P],
    "mmsa_gram_tn": [P, L, P, L, L, P, I, I, I, I, P, L, P],
    "mmsa_gram_tn_scratch_bytes": [I, I, I],
    "mmsa_chanattn_build": [P, P, L, P, L, P, P, P, I, I, I, I, P],
    "mmsa_gffm_build": [P, P, P, I, I, I, P],
    "mmsa_gelu_gate": [P, L, P, L, L, I, P],
    "mmsa_pool_hw": [P, L, P, L, I, I, I, I, P],
    "mmsa_ca_apply": [P, L, P, L, P, L, P, L, I, I, I, I, P],
    "mmsa_tail_fuse": [P, L, L, P, L, P, P, P, P, L, I, I, I, I, I, I, P],
    "mmsa_global_attention_planes": [P, L, P, P, P, L, I, I, I, I, I, F, I, I, P, P],
    "mmsa_window_attention_planes": [P, L, P, P, P, P, L, I, I, I, I, I, I, F, I, I, P, P],
    "mmsa_nchw_to_planes": [P, L, P, L, I, I, L, P],
    "mmsa_head_fuse": [P, P, I, I, P, I, I, P, I, I, L, P, P, P, L, P, L, I, I, I, I, I, P],
    "mmsa_tokens_to_nchw": [P, L, P, I, L, I, P],
    "mmsa_bilinear_accum_nchw": [P, L, I, I, I, I, P, I, I, I, I, I, I, P, I, P],
    "mmsa_div_count_nchw": [P, P, I, I, L, P],
    "mmsa_argmax_nchw": [P, P, I, I, L, P],
    "mmsa_crop_batch_nchw": [P, I, I, I, I, P, I, P, I, I, P],
    "mmsa_slide_argmax": [P, I, I, I, I, P, P, I, I, I, I, I, P, P],
}
_RESTYPES = {"mmsa_last_error": c_char_p, "mmsa_gram_tn_scratch_bytes": c_long}

for _name, _args in SIGNATURES.items():
    _fn = getattr(_lib, _name)  # AttributeError here = header/library mismatch
    _fn.argtypes = _args
    _fn.restype = _RESTYPES.get(_name, c_int)


# The C ABI is not self-describing: a library built from another tree (MMSA_LIB variants, a stale in-tree .so) may export every symbol and still take
# different argument lists.  include/mmsa.h MMSA_ABI_VERSION is bumped with every such change; this binding was written for:
ABI_VERSION = 102
if _lib.mmsa_version() != ABI_VERSION:
    raise RuntimeError(f"{LIB_PATH}: ABI version {_lib.mmsa_version()} but mmsa/lib.py binds version {ABI_VERSION} (include/mmsa.h MMSA_ABI_VERSION): "
                       "rebuild with python multimodal-sam-adapter_amd/build.py")


def last_error() -> str:
    return _lib.mmsa_last_error().decode()


POISON_LDS = os.environ.get("MMSA_DEBUG_POISON_LDS") == "1"   # testing aid: NaN-fill every CU's LDS before every launch


def call(name, *args):
    """Invoke an entry point; raise RuntimeError(mmsa_last_error()) on a non-zero return code."""
    if POISON_LDS and not name.startswith(("mmsa_event", "mmsa_debug")) and args:
        _lib.mmsa_debug_poison_lds(0x7FC07FC0, args[-1])   # the stream is the last argument of every launching entry point
    rc = getattr(_lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {last_error()}")


def version() -> int:
    return _lib.mmsa_version()


raw = _lib
