"""ctypes binding of the C-ABI in ``include/obman_hip.h`` (the drop-in boundary).

There is NO fallback: if ``csrc/libobman_hip.so`` is missing or stale-incompatible the import of
any op fails loudly with the build command.
"""
import ctypes
import os

from .build import LIB

_c_void_p, _c_int, _c_long, _c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
ABI_VERSION = 7

# name -> (restype, argtypes); 'p' pointer, 'i' int, 'l' long, 'f' float
_SIGNATURES = {
    "obman_abi_version": (_c_int, ""),
    "obman_pairmin_ws_bytes": (_c_long, "iii"),
    "obman_pairmin_fwd": (_c_int, "ppiiipppp" "plp"),
    "obman_pairmin_bwd": (_c_int, "ppiiipppppp" "p"),
    "obman_chamfer_fwd": (_c_int, "ppiiipppppp" "plplp"),
    "obman_chamfer_sync_bytes": (_c_long, "iii"),
    "obman_chamfer_bwd": (_c_int, "ppiiipppppp" "p"),
    "obman_mesh_contains_fwd": (_c_int, "ppp" "iiii" "pp"),
    "obman_mesh_contains_groups_fwd": (_c_int, "ppp" "iiii" "i" "pp"),
    "obman_mesh_contains_bruteforce_fwd": (_c_int, "ppp" "iiii" "i" "pp"),
    "obman_contact_fwd": (_c_int, "ppppp" "iii" "pp" "ii" "ifif" "ppppp" "p"),
    "obman_contact_bwd": (_c_int, "pppppppp" "iii" "ifif" "i" "pp" "p"),
    "obman_pointgen_ws_floats": (_c_long, "pi"),
    "obman_pointgen_fwd": (_c_int, "pppp"),
    "obman_pointgen_bwd": (_c_int, "pppppp"),
    "obman_edge_loss_fwd": (_c_int, "pp" "iii" "pp" "p"),
    "obman_edge_loss_bwd": (_c_int, "pp" "iii" "ppp" "p"),
    "obman_laplacian_fwd": (_c_int, "pppp" "ii" "ppp" "p"),
    "obman_laplacian_bwd": (_c_int, "ppppp" "ii" "pp" "p"),
    "obman_bnact_ws_floats": (_c_long, "li"),
    "obman_bnact_fwd": (_c_int, "pppppp" "li" "iffi" "ppp" "p"),
    "obman_bnact_bwd": (_c_int, "ppppp" "li" "iii" "ppppp" "p"),
    "obman_bnact_bwd2": (_c_int, "pppppp" "li" "iii" "ppppp" "p"),
    "obman_bnpool_fwd": (_c_int, "ppppp" "iiii" "iff" "ppp" "p"),
    "obman_bnpool_bwd": (_c_int, "ppppp" "iiii" "i" "pppp" "p"),
    "obman_bnact_fwd_bf16": (_c_int, "pppppp" "li" "iffi" "ppp" "p"),
    "obman_bnact_bwd_bf16": (_c_int, "ppppp" "li" "iii" "ppppp" "p"),
    "obman_bnact_bwd2_bf16": (_c_int, "pppppp" "li" "iii" "ppppp" "p"),
    "obman_bnpool_fwd_bf16": (_c_int, "ppppp" "iiii" "iff" "pppp" "p"),
    "obman_bnpool_bwd_bf16": (_c_int, "ppppp" "iiii" "i" "pppp" "p"),
    "obman_imgstream_ws_bytes": (_c_long, "iii"),
    "obman_imgstream_fwd": (_c_int, "piiip" "ii" "iii" "pp" "pp" "p"),
    "obman_adam_step": (_c_int, "pi" "ddddd" "p"),
    "obman_bf16_shadow": (_c_int, "ppl" "p"),
    "obman_affine_points_fwd": (_c_int, "ppp" "ii" "p" "p"),
    "obman_affine_points_ws_floats": (_c_long, "i"),
    "obman_affine_points_bwd": (_c_int, "ppp" "ii" "pppp" "p"),
    "obman_mse_terms_ws_floats": (_c_long, ""),
    "obman_mse_terms_fwd": (_c_int, "pi" "pp" "p"),
    "obman_mse_terms_bwd": (_c_int, "pi" "p" "p"),
    "obman_gt_object_stats": (_c_int, "p" "ii" "ppp" "p"),
    "obman_prof_enable": (_c_int, "i"),
    "obman_prof_summary": (_c_int, "ipp"),
    "obman_mano_model_floats": (_c_int, ""),
    "obman_mano_state_floats": (_c_int, ""),
    "obman_mano_lbs_fwd": (_c_int, "ppppp" "iiiii" "ppp" "p"),
    "obman_mano_bwd_scratch_floats": (_c_int, "i"),
    "obman_mano_lbs_bwd": (_c_int, "pppppp" "iiiii" "ppp" "p"),
}
_KIND = {"p": _c_void_p, "i": _c_int, "l": _c_long, "f": _c_float, "d": ctypes.c_double}
_lib = None
_fp = ctypes.c_void_p


class PointGenParams(ctypes.Structure):  # obman_pointgen_params
    _fields_ = [("B", _c_int), ("N", _c_int), ("C1", _c_int), ("training", _c_int),
                ("eps", _c_float), ("momentum", _c_float), ("out_factor", _c_float), ("mfma_bf16", _c_int),
                ("grid_per_sample", _c_int), ("reserved_", _c_int),
                ("grid", _fp), ("feat", _fp),
                ("w1", _fp), ("b1", _fp), ("w2", _fp), ("b2", _fp), ("w3", _fp), ("b3", _fp), ("w4", _fp), ("b4", _fp),
                ("bn_w", _fp * 3), ("bn_b", _fp * 3), ("bn_rm", _fp * 3), ("bn_rv", _fp * 3)]


class PointGenGrads(ctypes.Structure):  # obman_pointgen_grads
    _fields_ = [("w1", _fp), ("b1", _fp), ("w2", _fp), ("b2", _fp), ("w3", _fp), ("b3", _fp), ("w4", _fp), ("b4", _fp),
                ("bn_w", _fp * 3), ("bn_b", _fp * 3), ("feat", _fp)]


class AdamTensor(ctypes.Structure):  # obman_adam_tensor
    _fields_ = [("p", _fp), ("g", _fp), ("m", _fp), ("v", _fp), ("shadow_bf16", _fp), ("step", _fp), ("n", _c_long)]


class MseTerm(ctypes.Structure):  # obman_mse_term
    _fields_ = [("pred", _fp), ("target", _fp), ("grad", _fp), ("n", _c_long)]


class ImgParams(ctypes.Structure):  # obman_img_params: 24 x 32-bit words per sample
    _fields_ = [("src_h", ctypes.c_int32), ("src_w", ctypes.c_int32), ("flip", ctypes.c_int32), ("A", ctypes.c_int32 * 6),
                ("blur_r", ctypes.c_int32), ("blur_ww", ctypes.c_uint32), ("blur_fw", ctypes.c_uint32),
                ("n_ops", ctypes.c_int32), ("op", ctypes.c_int32 * 4), ("factor", _c_float * 4),
                ("hue_shift", ctypes.c_int32), ("reserved", ctypes.c_int32 * 2)]


assert ctypes.sizeof(ImgParams) == 96


class ObmanHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise ObmanHipError(
            "HIP kernel library %s not built - run `python -m obman_train_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU/PyTorch fallback." % LIB
        )
    handle = ctypes.CDLL(LIB)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = [_KIND[a] for a in args]
    got = handle.obman_abi_version()
    if got != ABI_VERSION:
        raise ObmanHipError("libobman_hip.so ABI %d != expected %d - rebuild" % (got, ABI_VERSION))
    if os.environ.get("OBMAN_TRACE_LAUNCH", "0") not in ("", "0"):
        _trace_launches(handle)
    _lib = handle
    return _lib


def _trace_launches(handle):
    """Debug mode (OBMAN_TRACE_LAUNCH=1, used by tools/efence): every kernel launcher writes its name to stderr before it
    runs and waits for the device afterwards, so a GPU memory fault is attributed to the launch that caused it."""
    import sys

    import torch

    def wrap(name, fn):
        def traced(*args):
            sys.stderr.write("[obman-launch] %s\n" % name)
            sys.stderr.flush()
            status = fn(*args)
            torch.cuda.synchronize()
            return status

        return traced

    for name, (res, args) in _SIGNATURES.items():
        if res is _c_int and args.endswith("p") and len(args) > 3:  # launchers: int status, last argument = the stream
            setattr(handle, name, wrap(name, getattr(handle, name)))


def check(status, what):
    if status != 0:
        kind = "bad argument" if status < 0 else "hipError_t"
        raise ObmanHipError("%s failed: %s %d" % (what, kind, status))


def declared_symbols(header_path):
    """Names of every ``obman_*`` function declared in the C header (used by the loader test)."""
    import re

    with open(header_path) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(obman_[a-z0-9_]+)\s*\(", text)))


def prof_enable(on=True):
    check(lib().obman_prof_enable(1 if on else 0), "obman_prof_enable")


def prof_summary(kernel_id):
    """-> (total_ms, launches) of one kernel id since the last prof_enable()."""
    tot, n = ctypes.c_double(0.0), ctypes.c_long(0)
    check(lib().obman_prof_summary(int(kernel_id), ctypes.addressof(tot), ctypes.addressof(n)), "obman_prof_summary")
    return tot.value, n.value
