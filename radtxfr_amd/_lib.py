"""ctypes binding of libradtxfr_hip.so (include/radtxfr_hip.h). No torch types cross this boundary:
device pointers are plain integers (tensor.data_ptr()), sizes are int64, the stream is a void*.

The library is REQUIRED: there is no CPU fallback. load() raises if the .so is missing or a
declared symbol is absent.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RADTXFR_LIB: developer override to A/B-test another build of the same ABI (tools/time_c3.py)
LIB_PATH = os.environ.get("RADTXFR_LIB") or os.path.join(_HERE, "libradtxfr_hip.so")


class RtxGrid(C.Structure):
    """struct rtx_grid: np.linspace(xmin, xmax, n_total), shard [offset, offset+n)."""
    _fields_ = [("xmin", C.c_double), ("xmax", C.c_double), ("step", C.c_double),
                ("n_total", C.c_int64), ("offset", C.c_int64), ("n", C.c_int64)]


_vp, _i64, _i32, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
_pd = C.POINTER(C.c_double)
_gp = C.POINTER(RtxGrid)

# name -> (restype, argtypes); must list every symbol include/radtxfr_hip.h declares
PROTOTYPES = {
    "rtx_version": (_i32, []),
    "rtx_last_error": (C.c_char_p, []),
    "rtx_device_info": (_i32, [C.c_char_p, _i32, C.POINTER(_i32)]),
    "rtx_lines_create": (_i32, [_i64, _i32] + [_vp] * 10 + [_vp, C.POINTER(_vp)]),
    "rtx_lines_free": (_i32, [_vp]),
    "rtx_lines_count": (_i64, [_vp]),
    "rtx_prep_create": (_i32, [_vp, _i32, _i64, C.POINTER(_vp)]),
    "rtx_prep_free": (_i32, [_vp]),
    "rtx_line_prep": (_i32, [_vp, _vp, _gp, _i32, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _vp]),
    "rtx_line_prep_profile": (_i32, [_vp, _vp, _gp, _i32, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _i32, _vp]),
    "rtx_prep_split_bound": (_i64, [_vp]),
    "rtx_voigt_sum": (_i32, [_vp, _gp, _i32, _vp, _vp, _i64, _vp]),
    "rtx_sdvoigt_sum": (_i32, [_vp, _gp, _i32, _vp, _vp, _i64, _vp]),
    "rtx_lines_set_sd": (_i32, [_vp, _vp, _vp]),
    "rtx_lines_set_deltap_self": (_i32, [_vp, _vp]),
    "rtx_voigt_tile_points": (_i32, []),
    "rtx_planck": (_i32, [_gp, _vp, _i64, _vp, _i64, _i32, _vp, _vp]),
    "rtx_tud": (_i32, [_vp, _i64, _gp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "rtx_tud_gtable_size": (_i32, []),
    "rtx_tud_gtable": (_i32, [_i32, _vp, _vp]),
    "rtx_compute_tud": (_i32, [_vp, _vp, _gp, _i32, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _dbl, _i32, _vp, _i32, _vp,
                                _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "rtx_apparent_radiance": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "rtx_ils": (_i32, [_i32, _gp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "rtx_interp_knots": (_i32, [_gp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "rtx_band_moments": (_i32, [_i32, _gp, _vp, _vp, _vp, _dbl, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rtx_band_mix": (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _vp, _i64, _vp, _vp]),
    "rtx_band_basis_moments": (_i32, [_i32, _gp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _dbl,
                                      _vp, _vp, _vp, _vp, _vp, _vp]),
    "rtx_band_mix_stacked": (_i32, [_vp, _vp, _i32, _i32, _i64, _vp, _i64, _vp, _vp]),
    "rtx_pixel_cube": (_i32, [_i32, _i32, _vp, _vp, _dbl, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "rtx_fir_reflect": (_i32, [_vp, _i32, _i64, _i32, _i64, _vp, _i32, _i32, _vp, _i64, _vp]),
    "rtx_cubic_resample": (_i32, [_vp, _i64, _i32, _i64, _dbl, _dbl, _vp, _i64, _vp, _i64, _vp]),
    "rtx_cubic_end": (_i32, [_vp, _i64, _i32, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _i64, _vp]),
    "rtx_cubic_resample_unchecked": (_i32, [_vp, _i64, _i32, _i64, _dbl, _dbl, _vp, _i64, _vp, _i64, _vp]),
    "rtx_brightness_temperature": (_i32, [_vp, _i64, _vp, _i64, _i32, _dbl, _vp, _vp]),
    "rtx_bt2l": (_i32, [_vp, _i64, _vp, _i64, _i32, _dbl, _vp, _vp]),
    "rtx_comm_init_all": (_i32, [_i32, _vp, _i32, C.POINTER(_vp)]),
    "rtx_comm_backend": (_i32, [_vp]),
    "rtx_allgather": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "rtx_comm_destroy": (_i32, [_vp]),
}

_lib = None


class RtxError(RuntimeError):
    pass


def load():
    """dlopen the HIP library (once). Fails loudly: the product path has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtxError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C radtxfr_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch first: it carries its own HIP runtime, and the library must bind to THAT copy (the one that owns the tensors'
    # device memory and streams). Loaded the other way round -- the library before torch, as a build() followed by a smoke()
    # in one process does -- the process ends up with two runtimes and the library's sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RtxError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.rtx_version() // 100 != 1:
        raise RtxError(f"libradtxfr_hip ABI version {lib.rtx_version()} is not 1.x")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RtxError((load().rtx_last_error() or b"unknown error").decode())


def make_grid(xmin, xmax, n_total, offset=0, n=None):
    """rtx_grid for np.linspace(xmin, xmax, n_total): `step` is the value NumPy computes."""
    n_total = int(n_total)
    step = (float(xmax) - float(xmin)) / (n_total - 1) if n_total > 1 else 0.0
    return RtxGrid(float(xmin), float(xmax), step, n_total, int(offset), int(n_total - offset if n is None else n))
