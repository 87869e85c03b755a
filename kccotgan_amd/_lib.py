"""ctypes binding of the C-ABI library ``csrc/libkccot.so`` (declared in ``include/kccot.h``).

The HIP library is the product: there is NO CPU fallback anywhere in this package.  If the
shared object is missing or a call is made without a GPU tensor the import / call fails loudly.
PyTorch is used for device memory, streams and autograd plumbing only.
"""
import ctypes
import os

import torch  # must be imported first: the library then binds to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# KCCOT_LIB_PATH: load another build of the SAME library (tools only: csrc/libkccot_diag.so, the diagnostic twin with
# in-kernel stamps and timing-experiment kernel variants; never set by the package or the tests)
LIB_PATH = os.environ.get("KCCOT_LIB_PATH") or os.path.join(_HERE, "csrc", "libkccot.so")

EINVAL, EUNSUPPORTED, EWORKSPACE, EABORTED = -1, -2, -3, -4

COST_SAME, COST_FORCE_DIRECT, COST_FORCE_MFMA, COST_PARTIAL_ONLY = 1, 2, 4, 8
COST_GRAM_SUMS_ONLY, COST_FROM_GRAM_SUMS = 16, 32
STOP_COUNT, STOP_INDEX = 0, 1
SMOOTH_T, SMOOTH_H, SMOOTH_W, SMOOTH_NO_DIVIDE, SMOOTH_EXTERNAL_MAX = 1, 2, 4, 16, 32
SMOOTH_STATS_ONLY, SMOOTH_EXTERNAL_STATS = 64, 128

_c = ctypes
_fp = _c.c_void_p      # device pointers travel as plain addresses
_i, _i64, _f, _u, _sz = _c.c_int, _c.c_int64, _c.c_float, _c.c_uint, _c.c_size_t

# name -> (restype, argtypes); mirrors include/kccot.h one to one
SIGNATURES = {
    "kccot_version": (_i, []),
    "kccot_last_error": (_c.c_char_p, []),
    "kccot_set_option": (_i, [_c.c_char_p, _i]),
    "kccot_get_option": (_i, [_c.c_char_p, _c.POINTER(_i)]),
    "kccot_option_count": (_i, []),
    "kccot_option_name": (_c.c_char_p, [_i]),
    "kccot_pairwise_cost_workspace_bytes": (_sz, [_i, _i, _i64]),
    "kccot_pairwise_cost_f32": (_i, [_fp, _fp, _i, _i, _i64, _f, _fp, _fp, _fp, _fp, _i, _i, _u, _fp, _fp, _sz, _fp]),
    "kccot_pairwise_cost3_workspace_bytes": (_sz, [_i, _i64]),
    "kccot_pairwise_cost3_f32": (_i, [_fp, _fp, _i, _i64, _f, _fp, _fp, _fp, _fp, _i, _i, _u, _fp, _fp, _sz, _fp]),
    "kccot_pairwise_cost3_gram_sums_span": (_i, [_i, _i64, _c.POINTER(_sz), _c.POINTER(_sz)]),
    "kccot_pairwise_cost3_rows_workspace_bytes": (_sz, [_i, _i, _i64]),
    "kccot_pairwise_cost3_rows_f32": (_i, [_fp, _fp, _i, _i64, _f, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _sz, _fp]),
    "kccot_pairwise_cost3_rows_gram_supported": (_i, [_i, _i, _i64]),
    "kccot_pairwise_cost3_rows_gram_workspace_bytes": (_sz, [_i, _i, _i64]),
    "kccot_row_norms_workspace_bytes": (_sz, [_i]),
    "kccot_row_norms_f64": (_i, [_fp, _fp, _i, _i64, _fp, _fp, _sz, _fp]),
    "kccot_pairwise_cost3_rows_gram_f32": (_i, [_fp, _fp, _i, _i64, _f, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _sz, _fp]),
    "kccot_pairwise_cost3_rows_gram_sums_count": (_sz, [_i, _i]),
    "kccot_pairwise_cost3_rows_gram_sums_f64": (_i, [_fp, _fp, _i, _i64, _i, _i, _fp, _i, _fp, _sz, _fp]),
    "kccot_pairwise_cost3_rows_gram_from_sums_f32": (_i, [_fp, _i, _f, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp]),
    "kccot_pairwise_cost3_bwd_workspace_bytes": (_sz, [_i, _i64]),
    "kccot_pairwise_cost3_bwd_f32": (_i, [_fp, _fp, _fp, _i, _i64, _f, _fp, _fp, _fp, _fp, _i, _i,
                                          _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "kccot_pairwise_cost3_bwd_rows_f32": (_i, [_fp, _fp, _fp, _i, _i64, _f, _fp, _fp, _fp, _fp, _i, _i, _i, _i,
                                               _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "kccot_pairwise_cost_bwd_workspace_bytes": (_sz, [_i, _i]),
    "kccot_pairwise_cost_bwd_f32": (_i, [_fp, _fp, _fp, _i, _i, _i64, _f, _fp, _fp, _i, _i, _u,
                                         _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "kccot_sinkhorn_workspace_bytes": (_sz, [_i, _i]),
    "kccot_sinkhorn_fwd_f32": (_i, [_fp, _i, _i, _f, _i, _i, _f, _i, _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "kccot_sinkhorn_status": (_i, [_fp, _i, _fp]),
    "kccot_sinkhorn_bwd_f32": (_i, [_fp, _fp, _fp, _fp, _i, _i, _f, _i, _fp, _fp, _fp, _sz, _fp]),
    "kccot_sinkhorn_divergence_fwd_f32": (_i, [_fp, _i, _f, _i, _i, _f, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "kccot_sinkhorn_divergence_bwd_f32": (_i, [_fp, _fp, _fp, _fp, _i, _f, _i, _fp, _fp, _fp, _sz, _fp]),
    "kccot_sinkhorn_loss_workspace_bytes": (_sz, [_i, _i64]),
    "kccot_sinkhorn_loss_fwd_f32": (_i, [_fp, _fp, _i, _i64, _f, _fp, _fp, _fp, _fp, _i, _i, _f, _i, _i, _f, _u,
                                         _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "kccot_sinkhorn_loss_bwd_f32": (_i, [_fp, _fp, _fp, _i, _i64, _f, _fp, _fp, _fp, _fp, _i, _i, _f, _i,
                                         _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "kccot_sinkhorn_fused_eligible": (_i, [_i, _i]),
    "kccot_sinkhorn_divergence_fused_f32": (_i, [_fp, _i, _f, _i, _i, _f, _fp, _fp, _fp, _fp, _fp, _fp]),
    "kccot_pairwise_cost3_bwd_scaled_f32": (_i, [_fp, _fp, _fp, _fp, _i, _i64, _f, _fp, _fp, _fp, _fp, _i, _i,
                                                 _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "kccot_sinkhorn_loss_fused_fwd_f32": (_i, [_fp, _fp, _i, _i64, _f, _fp, _fp, _fp, _fp, _i, _i, _f, _i, _i, _f, _u,
                                               _fp, _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "kccot_sinkhorn_loss_fused_bwd_f32": (_i, [_fp, _fp, _fp, _fp, _i, _i64, _f, _fp, _fp, _fp, _fp, _i, _i,
                                               _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "kccot_mixed_divergence_fwd_f32": (_i, [_fp, _fp, _fp]),
    "kccot_mixed_divergence_bwd_f32": (_i, [_fp, _fp, _fp]),
    "kccot_martingale_fwd_f32": (_i, [_fp, _i, _i, _i, _f, _f, _fp, _fp]),
    "kccot_martingale_bwd_f32": (_i, [_fp, _i, _i, _i, _f, _f, _fp, _fp, _fp]),
    "kccot_rbf_mmd_f32": (_i, [_fp, _i, _f, _fp, _fp, _fp]),
    "kccot_rbf_mmd_bwd_f32": (_i, [_fp, _i, _f, _fp, _fp, _fp]),
    "kccot_smooth_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "kccot_smooth_fwd_f32": (_i, [_fp, _i, _i, _i, _i, _i, _f, _i, _u, _fp, _fp, _fp, _sz, _fp]),
    "kccot_channel_layernorm_chunks": (_i, [_i, _i, _i]),
    "kccot_channel_layernorm_fwd_f32": (_i, [_fp, _fp, _fp, _i, _i, _i, _f, _fp, _fp, _fp, _fp]),
    "kccot_channel_layernorm_bwd_f32": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp, _fp, _fp]),
    "kccot_convlstm_cell_fwd_f32": (_i, [_fp, _fp, _fp, _i, _i, _i, _fp, _fp, _fp]),
    "kccot_convlstm_cell_bwd_f32": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp, _fp, _fp]),
    "kccot_smooth_bwd_f32": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _i, _u, _fp, _fp, _sz, _fp]),
    "kccot_smooth_bwd_sharded_f32": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _i, _u, _fp, _fp, _sz, _fp]),
}


class KccotError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "kccotgan_amd: %s is missing -- build it with `make -C kccotgan_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of step
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


# ---- debugging aid: KCCOT_DEBUG_CANARY=1 surrounds every buffer the wrappers hand to the library
# (outputs and workspace) with sentinel-filled guard zones and verifies them after every call.
_CANARY = os.environ.get("KCCOT_DEBUG_CANARY") == "1"
_GUARD = 4096            # elements on either side
_SENT = 1234567.0
_guards = []


def empty(shape, dtype=torch.float32, device=None):
    if not _CANARY:
        return torch.empty(shape, dtype=dtype, device=device)
    n = 1
    for d in shape:
        n *= int(d)
    buf = torch.full((n + 2 * _GUARD,), _SENT, dtype=dtype, device=device)
    _guards.append((buf, n))
    del _guards[:-400]
    return buf[_GUARD:_GUARD + n].view(tuple(shape))


def empty_like(t):
    return empty(tuple(t.shape), t.dtype, t.device)


def _verify_guards(what):
    if torch.cuda.is_current_stream_capturing():
        return                      # no synchronisation inside a hipGraph capture: the next eager call verifies
    torch.cuda.synchronize()
    for buf, n in _guards:
        lo, hi = buf[:_GUARD], buf[_GUARD + n:]
        if not (bool((lo == _SENT).all()) and bool((hi == _SENT).all())):
            bad_lo = int((lo != _SENT).sum())
            bad_hi = int((hi != _SENT).sum())
            raise KccotError("guard zone overwritten after %s: buffer of %d elements (%s), %d bad below, %d bad above"
                             % (what, n, buf.dtype, bad_lo, bad_hi))


def check(rc, what):
    if _CANARY:
        _verify_guards(what)
    if rc == 0:
        return
    msg = lib.kccot_last_error().decode("utf-8", "replace")
    if rc == EINVAL:
        raise ValueError("%s: %s" % (what, msg))
    if rc == EUNSUPPORTED:
        raise NotImplementedError("%s: %s" % (what, msg))
    raise KccotError("%s failed (code %d): %s" % (what, rc, msg))


def set_option(name, value):
    """kccot_set_option (include/kccot.h lists the options).  Process-wide."""
    check(lib.kccot_set_option(name.encode(), int(value)), "set_option(%s)" % name)


def get_option(name):
    v = _i(0)
    check(lib.kccot_get_option(name.encode(), ctypes.byref(v)), "get_option(%s)" % name)
    return v.value


def option_names():
    return [lib.kccot_option_name(k).decode() for k in range(lib.kccot_option_count())]


class options:
    """``with options(sinkhorn_shortcut=0, gram_f32=1): ...`` -- set, run, restore (tests, bench, A/B tools)."""

    def __init__(self, **kv):
        self.kv = kv
        self.old = {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def require_gpu(t):
    if not t.is_cuda:
        raise KccotError("kccotgan_amd needs tensors on a ROCm device (got %s); there is no CPU path" % t.device)


def ptr(t, dtypes=(torch.float32, torch.int32)):
    """Device address of a tensor (None -> NULL).  Refuses anything the kernels cannot read: the ABI's `float*` / `int*`
    parameters take fp32 / int32 tensors only -- a float64 tensor here would be reinterpreted, not converted."""
    if t is None:
        return None
    if not t.is_cuda:
        raise KccotError("kccotgan_amd needs tensors on a ROCm device (got %s); there is no CPU path" % t.device)
    if t.dtype not in dtypes:
        raise TypeError("kccotgan_amd kernel argument must be %s (got %s)" % (" / ".join(str(d) for d in dtypes), t.dtype))
    if not t.is_contiguous():
        raise ValueError("kccotgan_amd kernels need contiguous tensors")
    return t.data_ptr()


def ptr_f64(t):
    """Device address of a float64 tensor: ONLY for the `double*` parameters of the ABI (fp64 Gram sums and row norms of
    the sharded path: kccot_row_norms_f64, kccot_pairwise_cost3_rows_gram*_f64/_from_sums_f32)."""
    return ptr(t, (torch.float64,))


def stream_of(t):
    return torch.cuda.current_stream(t.device).cuda_stream


_ws_cache = {}


def workspace(nbytes, ref):
    """A per-(device, stream) scratch buffer, grown on demand.  Calls issued on one stream are
    ordered, so successive kernels may reuse it."""
    if nbytes <= 0:
        return None, 0
    key = (ref.device, stream_of(ref))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if _CANARY:   # float32 view so that the guard sentinel is exact
            buf = empty(((int(nbytes) + 3) // 4,), torch.float32, ref.device)
        else:
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=ref.device)
        _ws_cache[key] = buf
    return buf.data_ptr(), int(nbytes) if _CANARY else buf.numel()
