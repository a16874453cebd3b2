"""ctypes binding of libmrx.so (include/mrx.h).

The library is the product: there is no CPU fallback.  Importing this module
only loads the shared object (that works on a machine without a GPU, so the
CPU test tier can check the exported symbols); creating a :class:`Context`
needs a gfx950 device and raises :class:`MrxError` otherwise.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MRX_LIB_PATH: another build of the same sources (kernel A/B timing on one box, scripts/ab.sh)
LIB_PATH = os.environ.get("MRX_LIB_PATH") or os.path.join(_HERE, "libmrx.so")

MRX_OK = 0
FLAG_SCREEN_OOB = 1
FLAG_TABLE_OOB = 2
FLAG_NAN = 4
FLAG_HANDOVER = 8
OPT_POINTING_CHAIN = 0
OPT_AXIS_LITERAL = 1
OPT_SAMPLE_TIMES = 2
OPT_SAMPLE_CHUNK = 3
OPT_UPSAMPLE_GROUPS = 4
OPT_SAMPLE_WGS_PER_CU = 6
OPT_NOISE_GENERIC = 5
OPT_WRITER_PER_TILE = 7
OPT_NOISE_LANES = 8
OPT_SCREEN_STOCKHAM = 9
OPT_SYNTH_WGS_PER_CU = 10
OPT_SYNTH_TILE_ORDER = 11
OPT_GAUSS_ACCUM = 12
OPT_SYNTH_ACQUIRE = 13

_STATUS = {
    0: "MRX_OK",
    -1: "MRX_ERR_INVALID",
    -2: "MRX_ERR_HIP",
    -3: "MRX_ERR_NO_DEVICE",
    -4: "MRX_ERR_UNSUPPORTED",
    -5: "MRX_ERR_ALLOC",
}


class MrxError(RuntimeError):
    """A libmrx call returned a negative status."""

    def __init__(self, code: int, message: str):
        self.code = code
        super().__init__(f"{_STATUS.get(code, code)}: {message}")


class MrxLayer(C.Structure):
    """``mrx_layer`` (include/mrx.h)."""

    _fields_ = [
        ("d_values", C.c_void_p),
        ("d_axis_e", C.c_void_p),
        ("d_axis_c", C.c_void_p),
        ("d_off_e", C.c_void_p),
        ("d_off_c", C.c_void_p),
        ("n_e", C.c_int32),
        ("n_c", C.c_int32),
        ("h", C.c_double),
        ("r00", C.c_double),
        ("r10", C.c_double),
        ("r01", C.c_double),
        ("r11", C.c_double),
        ("e0", C.c_double),
        ("de", C.c_double),
        ("c0", C.c_double),
        ("dc", C.c_double),
        ("pwv_rms", C.c_float),
        ("reserved", C.c_int32),
    ]


class MrxBandTable(C.Structure):
    """``mrx_band_table`` (include/mrx.h)."""

    _fields_ = [
        ("d_values", C.c_void_p),
        ("d_axis_pwv", C.c_void_p),
        ("d_axis_el", C.c_void_p),
        ("n_pwv", C.c_int32),
        ("n_el", C.c_int32),
        ("w_t", C.c_float),
        ("t_oob", C.c_int32),
        ("d_cubic", C.c_void_p),
    ]


class MrxScreenDesc(C.Structure):
    """``mrx_screen_desc`` (include/mrx.h)."""

    _fields_ = [
        ("d_out", C.c_void_p),
        ("stream", C.c_uint32),
        ("out_ny", C.c_int32),
        ("out_nx", C.c_int32),
        ("periodic_beam", C.c_int32),
        ("ld_out", C.c_size_t),
        ("dy", C.c_double),
        ("dx", C.c_double),
        ("r0", C.c_double),
        ("nu", C.c_double),
        ("sigma_y", C.c_double),
        ("sigma_x", C.c_double),
        ("d_amp", C.c_void_p),
    ]


class MrxSkyMap(C.Structure):
    """``mrx_sky_map`` (include/mrx.h)."""

    _fields_ = [
        ("d_values", C.c_void_p),
        ("n_channels", C.c_int32),
        ("n_stokes", C.c_int32),
        ("n_eta", C.c_int32),
        ("n_xi", C.c_int32),
        ("eta0", C.c_double),
        ("deta", C.c_double),
        ("xi0", C.c_double),
        ("dxi", C.c_double),
        ("center_phi", C.c_double),
        ("center_theta", C.c_double),
        ("bilinear", C.c_int32),
        ("reserved", C.c_int32),
    ]


class MrxMapCal(C.Structure):
    """``mrx_map_cal`` (include/mrx.h)."""

    _fields_ = [
        ("d_table", C.c_void_p),
        ("d_axis_pwv", C.c_void_p),
        ("d_axis_el", C.c_void_p),
        ("n_pwv", C.c_int32),
        ("n_el", C.c_int32),
        ("d_pwv", C.c_void_p),
        ("Ta", C.c_int32),
        ("steps_per_tile", C.c_int32),
        ("ta0", C.c_double),
        ("dta", C.c_double),
        ("d_t", C.c_void_p),
        ("d_scalar", C.c_void_p),
    ]


# name -> (restype, argtypes); the single list the symbol test walks
_vp, _i, _d, _sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
SIGNATURES = {
    "mrx_version": (_i, []),
    "mrx_streams_concurrent": (_i, [_vp, _vp, C.POINTER(_i)]),
    "mrx_init": (_i, [_i, C.POINTER(_vp)]),
    "mrx_destroy": (_i, [_vp]),
    "mrx_set_stream": (_i, [_vp, _vp]),
    "mrx_synchronize": (_i, [_vp]),
    "mrx_set_option": (_i, [_vp, _i, _i]),
    "mrx_last_error": (C.c_char_p, [_vp]),
    "mrx_device_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_sz), C.c_char_p, _i]),
    "mrx_timer_start": (_i, [_vp]),
    "mrx_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "mrx_atm_plan_create": (_i, [_vp, C.POINTER(MrxLayer), _i, C.POINTER(MrxBandTable), _i, _i, C.POINTER(_vp)]),
    "mrx_atm_plan_info": (_i, [_vp, _vp, C.POINTER(_i), C.POINTER(_i)]),
    "mrx_atm_plan_destroy": (_i, [_vp, _vp]),
    "mrx_atm_sample": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _d, _vp, _vp, _vp]),
    "mrx_clear_flags": (_i, [_vp, _vp]),
    "mrx_read_flags": (_i, [_vp, _vp, C.POINTER(C.c_uint32)]),
    "mrx_spline_prepare": (_i, [_vp, _vp, _i, _i, _vp]),
    "mrx_spline_upsample": (_i, [_vp, _vp, _i, _i, _d, _d, _vp, _i, _vp, _vp, _vp, _sz]),
    "mrx_spline_upsample_fused": (_i, [_vp, _vp, _i, _i, _d, _d, _vp, _i, _vp, _vp, _vp, _sz]),
    "mrx_atm_synthesize_block_rows": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "mrx_atm_synthesize": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _d, _vp, _i, _i, _vp, _d, _d, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "mrx_atm_synthesize_krj": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _d, _vp, _i, _i, _vp, _d, _d, _vp, _i, _vp, _vp, _vp, _sz,
                                    _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _sz, _vp]),
    "mrx_spline_upsample_krj": (_i, [_vp, _vp, _i, _i, _d, _d, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz]),
    "mrx_coarse_to_krj": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mrx_coarse_to_krj_keep_tail": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _sz]),
    "mrx_tod_to_krj": (_i, [_vp, _vp, _sz, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    "mrx_tod_from_krj": (_i, [_vp, _vp, _sz, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    "mrx_pointing_broadcast": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _sz]),
    "mrx_linear_upsample": (_i, [_vp, _vp, _i, _i, _d, _d, _vp, _i, _vp, _sz]),
    "mrx_gauss_smooth2d": (_i, [_vp, _vp, _vp, _vp, _i, _i, _d, _d, _d]),
    "mrx_resample_columns": (_i, [_vp, _vp, _i, _i, _sz, _vp, _vp, _vp, _i, _vp, _sz]),
    "mrx_map_smooth": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _d, _d]),
    "mrx_screen_generate": (_i, [_vp, C.c_uint64, C.c_uint32, _i, _i, _d, _d, _d, _d, _vp, _vp]),
    "mrx_comm_unique_id": (_i, [_vp, _vp]),
    "mrx_comm_create": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "mrx_comm_wrap": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "mrx_comm_destroy": (_i, [_vp, _vp]),
    "mrx_allgather_tod": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "mrx_allgather_tod_p2p": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "mrx_exchange_screens": (_i, [_vp, _vp, _vp, _vp, _i]),
    "mrx_screen_work_floats": (_i, [_i, _i, _i, C.POINTER(_sz)]),
    "mrx_screen_generate_batch": (_i, [_vp, C.c_uint64, _i, _i, C.POINTER(MrxScreenDesc), _i, _vp, _sz]),
    "mrx_screen3d_work_floats": (_i, [_i, _i, _i, _i, C.POINTER(_sz)]),
    "mrx_screen_generate_3d": (_i, [_vp, C.c_uint64, C.c_uint32, _i, _i, _i, _d, _d, _d, _d, _d, C.POINTER(_d), C.POINTER(_d),
                                    C.POINTER(MrxScreenDesc), _i, _vp, _sz, _vp]),
    "mrx_screen_amp_floats": (_i, [_i, _i, _i, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "mrx_screen_amplitudes": (_i, [_vp, _i, _i, _i, _d, _d, _d, _d, C.POINTER(_d), C.POINTER(_d), _i, _d, _d, _d, _vp, _vp, _sz]),
    "mrx_screen_psd_sum": (_i, [_vp, _i, _i, _d, _d, _d, _d, C.POINTER(_d)]),
    "mrx_map_sample": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _sz]),
    "mrx_map_sample_krj": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz]),
    "mrx_bin_map": (_i, [_vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "mrx_bin_map_work_bytes": (_i, [_vp, _i, _i, _vp, _vp]),
    "mrx_bin_map_bucketed": (_i, [_vp, _vp, _vp, _sz, _vp, _sz, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz]),
    "mrx_tod_detrend_window": (_i, [_vp, _vp, _sz, _i, _i, _i, _vp, _vp]),
    "mrx_sosfilt_chunk": (_i, []),
    "mrx_sosfilt_work_doubles": (_i, [_i, _i, _i, C.POINTER(_sz)]),
    "mrx_sosfilt": (_i, [_vp, C.POINTER(_d), _i, _vp, _vp, _sz, _i, _i, _i, _vp, _sz, _vp]),
    "mrx_fft_rows": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "mrx_noise_period": (_i, [_i, C.POINTER(_i), C.POINTER(_i)]),
    "mrx_noise_work_floats": (_i, [_i, _i, _i, C.POINTER(_sz)]),
    "mrx_noise_generate": (_i, [_vp, C.c_uint64, _i, _i, _i, _d, _d, _d, _vp, _i, _vp, _vp, _sz, _d, _vp, _sz, _i, _vp, _sz]),
    "mrx_noise_generate_krj": (_i, [_vp, C.c_uint64, _i, _i, _i, _d, _d, _d, _vp, _i, _vp, _vp, _sz, _d, _vp, _sz, _vp, _sz,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _i, _i]),
    "mrx_philox_normal": (_i, [_vp, C.c_uint64, C.c_uint32, _sz, _vp]),
    "mrx_philox_raw": (_i, [_vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
}

_lib = None


def load():
    """Load libmrx.so; raise ImportError with the build recipe if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C maria_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "maria_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def philox4x32(seed: int, counter) -> tuple:
    """Host evaluation of the library's Philox-4x32-10 (no GPU needed)."""
    lib = load()
    out = (C.c_uint32 * 4)()
    rc = lib.mrx_philox_raw(None, seed, *[int(c) & 0xFFFFFFFF for c in counter], out)
    if rc != MRX_OK:
        raise MrxError(rc, "mrx_philox_raw")
    return tuple(int(v) for v in out)


class Context:
    """One ``mrx_ctx``: a device, a stream, and checked calls."""

    def __init__(self, device: int = 0, stream=None):
        self.lib = load()
        handle = _vp()
        rc = self.lib.mrx_init(int(device), C.byref(handle))
        if rc != MRX_OK:
            raise MrxError(rc, f"mrx_init(device={device}) found no usable gfx950 device")
        self.handle = handle
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mrx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def call(self, name: str, *args):
        rc = getattr(self.lib, name)(self.handle, *args)
        if rc != MRX_OK:
            raise MrxError(rc, self.lib.mrx_last_error(self.handle).decode(errors="replace"))

    def set_stream(self, stream):
        """`stream`: a raw hipStream_t (int), a torch.cuda.Stream, or None."""
        raw = getattr(stream, "cuda_stream", stream)
        self.call("mrx_set_stream", _vp(int(raw) if raw else None))

    def synchronize(self):
        self.call("mrx_synchronize")

    def set_option(self, option: int, value: int):
        self.call("mrx_set_option", int(option), int(value))
        self.__dict__.setdefault("_options", {})[int(option)] = int(value)

    def streams_concurrent(self, other) -> bool:
        """mrx_streams_concurrent: does ``other`` (a torch stream) run beside this context's stream, or do the two
        share a hardware queue and take turns?"""
        flag = _i()
        self.call("mrx_streams_concurrent", _vp(other.cuda_stream), C.byref(flag))
        return bool(flag.value)

    def get_option(self, option: int) -> int:
        """The value this handle last set (0, the library default, if never set)."""
        return self.__dict__.get("_options", {}).get(int(option), 0)

    def device_info(self) -> dict:
        n_cu, lds, hbm = _i(), _i(), _sz()
        name = C.create_string_buffer(128)
        self.call("mrx_device_info", C.byref(n_cu), C.byref(lds), C.byref(hbm), name, 128)
        return {"n_cu": n_cu.value, "lds_bytes_per_cu": lds.value, "hbm_bytes": hbm.value, "name": name.value.decode()}

    def timer_start(self):
        self.call("mrx_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self.call("mrx_timer_stop", C.byref(ms))
        return float(ms.value)


def ptr(t) -> _vp:
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return _vp(None)
    return _vp(t.data_ptr())
