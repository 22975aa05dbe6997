"""gnsscorr -- ctypes binding of libgnsscorr.so (include/gnsscorr.h).

The product is the C-ABI HIP library; this module only marshals numpy arrays
and raw device pointers (e.g. ``torch.Tensor.data_ptr()``) into it for tests
and the benchmark.  There is no CPU fallback: every compute entry point needs
the HIP library and a GPU, and raises ``GnsscorrError`` otherwise.

Class and method names follow the reference
(``Cpu_Multicorrelator_Real_Codes`` -> :class:`HipMulticorrelatorRealCodes`,
``pcps_acquisition`` -> :class:`PcpsAcquisition`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNSSCORR_LIB", os.path.join(os.path.dirname(_HERE), "libgnsscorr.so"))

GC_OK, GC_ERR_INVALID, GC_ERR_NO_DEVICE, GC_ERR_HIP, GC_ERR_STATE = range(5)
GC_MAX_TAPS = 8
GC_IQ_F32, GC_IQ_I16, GC_IQ_I8 = range(3)


class GnsscorrError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("gnsscorr status %d: %s" % (status, msg))
        self.status = status


class EpochParams(C.Structure):
    """gc_epoch_params (include/gnsscorr.h)."""
    _fields_ = [
        ("sample_offset", C.c_uint64),
        ("phase0_re", C.c_float), ("phase0_im", C.c_float),
        ("phase_inc_re", C.c_float), ("phase_inc_im", C.c_float),
        ("phase_rate_re", C.c_float), ("phase_rate_im", C.c_float),
        ("rem_code_phase_chips", C.c_float),
        ("code_phase_step_chips", C.c_float),
        ("code_phase_rate_step_chips", C.c_float),
        ("n_samples", C.c_int32),
    ]


EPOCH_DTYPE = np.dtype([
    ("sample_offset", np.uint64),
    ("phase0_re", np.float32), ("phase0_im", np.float32),
    ("phase_inc_re", np.float32), ("phase_inc_im", np.float32),
    ("phase_rate_re", np.float32), ("phase_rate_im", np.float32),
    ("rem_code_phase_chips", np.float32),
    ("code_phase_step_chips", np.float32),
    ("code_phase_rate_step_chips", np.float32),
    ("n_samples", np.int32),
], align=True)
assert EPOCH_DTYPE.itemsize == C.sizeof(EpochParams) == 48


class AcqConf(C.Structure):
    """gc_acq_conf: the Acq_Conf fields pcps_acquisition reads."""
    _fields_ = [
        ("fs_in", C.c_int64),
        ("sampled_ms", C.c_uint32),
        ("ms_per_code", C.c_uint32),
        ("samples_per_ms", C.c_float),
        ("samples_per_code", C.c_float),
        ("samples_per_chip", C.c_uint32),
        ("doppler_max", C.c_uint32),
        ("doppler_step", C.c_uint32),
        ("max_dwells", C.c_uint32),
        ("bit_transition_flag", C.c_int32),
        ("use_CFAR_algorithm_flag", C.c_int32),
        ("num_doppler_bins_override", C.c_uint32),
        ("make_2_steps", C.c_int32),
        ("num_doppler_bins_step2", C.c_uint32),
        ("doppler_step2", C.c_float),
    ]


class AcqResult(C.Structure):
    """gc_acq_result."""
    _fields_ = [
        ("indext", C.c_uint32),
        ("doppler_hz", C.c_int32),
        ("doppler_index", C.c_uint32),
        ("test_statistics", C.c_float),
        ("mag", C.c_float),
        ("input_power", C.c_float),
        ("second_peak", C.c_float),
        ("second_peak_full_row", C.c_float),
        ("acq_delay_samples", C.c_double),
        ("acq_doppler_hz", C.c_double),
    ]


class LoopConf(C.Structure):
    """gc_loop_conf."""
    _fields_ = [
        ("fs_in", C.c_double), ("signal_carrier_freq_hz", C.c_double), ("code_chip_rate_hz", C.c_double),
        ("code_period_s", C.c_double), ("carrier_lock_th", C.c_double), ("acq_delay_samples", C.c_double),
        ("acq_doppler_hz", C.c_double), ("acq_samplestamp_samples", C.c_uint64), ("sample_counter", C.c_uint64),
        ("code_length_chips", C.c_uint32), ("code_samples_per_chip", C.c_uint32), ("vector_length", C.c_uint32),
        ("pull_in_time_s", C.c_uint32), ("veml", C.c_int32), ("pll_filter_order", C.c_int32), ("dll_filter_order", C.c_int32),
        ("enable_fll_pull_in", C.c_int32), ("enable_fll_steady_state", C.c_int32), ("cn0_samples", C.c_int32),
        ("cn0_min", C.c_int32), ("max_lock_fail", C.c_int32), ("pll_bw_hz", C.c_float), ("dll_bw_hz", C.c_float),
        ("fll_bw_hz", C.c_float), ("early_late_space_chips", C.c_float), ("very_early_late_space_chips", C.c_float),
        ("high_dyn_smoother_length", C.c_uint32),
    ]


LOOP_RECORD_DTYPE = np.dtype([
    ("corr", np.float32, (10,)), ("carrier_doppler_hz", np.float32), ("code_freq_chips", np.float32),
    ("carr_phase_error_hz", np.float32), ("carr_error_filt_hz", np.float32), ("code_error_chips", np.float32),
    ("code_error_filt_chips", np.float32), ("cn0_db_hz", np.float32), ("carrier_lock_test", np.float32),
    ("sample_counter", np.uint64), ("acc_carrier_phase_rad", np.float64), ("rem_code_phase_samples", np.float64),
    ("state", np.int32), ("valid", np.int32), ("current_prn_length_samples", np.int32), ("extend_count", np.int32),
    ("accu", np.float32, (10,)), ("prompt_data", np.float32, (2,)), ("integrating", np.int32), ("reserved", np.int32),
], align=True)


class LoopSyncConf(C.Structure):
    """gc_loop_sync_conf: symbol synchronisation, extended integration and pilot tracking of one channel."""
    _fields_ = [
        ("extend_correlation_symbols", C.c_int32), ("track_pilot", C.c_int32), ("symbols_per_bit", C.c_int32),
        ("secondary_code_length", C.c_int32), ("preamble_length_symbols", C.c_int32), ("bit_sync_min_time_s", C.c_float),
        ("pll_bw_narrow_hz", C.c_float), ("dll_bw_narrow_hz", C.c_float), ("early_late_space_narrow_chips", C.c_float),
        ("very_early_late_space_narrow_chips", C.c_float), ("secondary_code", C.c_char * 128), ("preamble_symbols", C.c_int8 * 192),
    ]

    @classmethod
    def make(cls, extend_correlation_symbols=1, track_pilot=False, symbols_per_bit=1, secondary_code="", preamble_symbols=(),
            bit_sync_min_time_s=10.0, pll_bw_narrow_hz=0.0, dll_bw_narrow_hz=0.0, early_late_space_narrow_chips=0.0,
            very_early_late_space_narrow_chips=0.0):
        y = cls()
        y.extend_correlation_symbols, y.track_pilot, y.symbols_per_bit = int(extend_correlation_symbols), int(bool(track_pilot)), int(symbols_per_bit)
        y.secondary_code_length, y.preamble_length_symbols = len(secondary_code), len(preamble_symbols)
        y.bit_sync_min_time_s = bit_sync_min_time_s
        y.pll_bw_narrow_hz, y.dll_bw_narrow_hz = pll_bw_narrow_hz, dll_bw_narrow_hz
        y.early_late_space_narrow_chips, y.very_early_late_space_narrow_chips = early_late_space_narrow_chips, very_early_late_space_narrow_chips
        if len(secondary_code) > 128 or len(preamble_symbols) > 192:
            raise ValueError("secondary code / preamble too long")
        y.secondary_code = secondary_code.encode()
        for i, v in enumerate(preamble_symbols):
            y.preamble_symbols[i] = int(v)
        return y


assert LOOP_RECORD_DTYPE.itemsize == 168 and C.sizeof(LoopConf) == 144 and C.sizeof(LoopSyncConf) == 360


# every symbol include/gnsscorr.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_fp = C.POINTER(C.c_float)
_i16p = C.POINTER(C.c_int16)
API = {
    "gc_last_error": (C.c_char_p, []),
    "gc_version": (C.c_char_p, []),
    "gc_abi_check": (C.c_int, [C.c_size_t] * 6),
    "gc_device_count": (C.c_int, []),
    "gc_build_has_experiments": (C.c_int, []),
    "gc_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "gc_ctx_destroy": (C.c_int, [_vp]),
    "gc_ctx_synchronize": (C.c_int, [_vp]),
    "gc_correlator_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "gc_correlator_destroy": (C.c_int, [_vp]),
    "gc_correlator_set_high_dynamics_resampler": (C.c_int, [_vp, C.c_int]),
    "gc_correlator_init": (C.c_int, [_vp, C.c_int, C.c_int]),
    "gc_correlator_set_local_code_and_taps": (C.c_int, [_vp, C.c_int, _fp, _fp]),
    "gc_correlator_set_input_output_vectors": (C.c_int, [_vp, _fp, _fp]),
    "gc_correlator_carrier_wipeoff_multicorrelator_resampler": (C.c_int, [_vp] + [C.c_float] * 6 + [C.c_int]),
    "gc_correlator_carrier_wipeoff_multicorrelator_resampler_6": (C.c_int, [_vp] + [C.c_float] * 5 + [C.c_int]),
    "gc_correlator_free": (C.c_int, [_vp]),
    "gc_correlator_batch_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "gc_ctx_register_host_buffer": (C.c_int, [_vp, _vp, C.c_size_t]),
    "gc_ctx_unregister_host_buffer": (C.c_int, [_vp, _vp]),
    "gc_correlator_set_local_code_and_taps_complex": (C.c_int, [_vp, C.c_int, _fp, _fp]),
    "gc_correlator_carrier_wipeoff_multicorrelator_resampler_5": (C.c_int, [_vp] + [C.c_float] * 4 + [C.c_int]),
    "gc_correlator_set_local_code_and_taps_16sc": (C.c_int, [_vp, C.c_int, _i16p, _fp]),
    "gc_correlator_set_input_output_vectors_16sc": (C.c_int, [_vp, _i16p, _i16p]),
    "gc_epoch_params_fill": (None, [C.POINTER(EpochParams), C.c_uint64] + [C.c_float] * 6 + [C.c_int]),
    "gc_stream_create": (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_uint32, C.POINTER(_vp)]),
    "gc_stream_destroy": (C.c_int, [_vp]),
    "gc_stream_push": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "gc_stream_push_pinned": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "gc_stream_broadcast_pinned": (C.c_int, [C.POINTER(_vp), C.c_int, _vp, C.c_uint64]),
    "gc_stream_info": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gc_stream_synchronize": (C.c_int, [_vp]),
    "gc_trk_loop_set_input_format": (C.c_int, [_vp, C.c_int]),
    "gc_trk_loop_set_input_stream": (C.c_int, [_vp, C.c_int, _vp]),
    "gc_trk_batch_set_input_stream": (C.c_int, [_vp, C.c_int, _vp]),
    "gc_trk_batch_set_read_floor": (C.c_int, [_vp, C.c_uint64]),
    "gc_acq_set_frequency_offset": (C.c_int, [_vp, C.c_int64]),
    "gc_acq_dwell_stream": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "gc_trk_batch_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "gc_trk_batch_destroy": (C.c_int, [_vp]),
    "gc_trk_batch_set_code": (C.c_int, [_vp, C.c_int, _fp, C.c_int, _fp]),
    "gc_trk_batch_set_shifts": (C.c_int, [_vp, C.c_int, _fp]),
    "gc_trk_batch_set_complex_codes": (C.c_int, [_vp, C.c_int]),
    "gc_trk_batch_set_code_complex": (C.c_int, [_vp, C.c_int, _fp, C.c_int, _fp]),
    "gc_trk_batch_set_16sc": (C.c_int, [_vp, C.c_int]),
    "gc_trk_batch_set_code_16sc": (C.c_int, [_vp, C.c_int, _i16p, C.c_int, _fp]),
    "gc_trk_batch_set_input_format": (C.c_int, [_vp, C.c_int]),
    "gc_trk_batch_set_input_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64]),
    "gc_trk_batch_run_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "gc_trk_batch_run": (C.c_int, [_vp, C.c_int, _vp, _fp]),
    "gc_trk_batch_set_nominal_length": (C.c_int, [_vp, C.c_int]),
    "gc_trk_batch_set_slices": (C.c_int, [_vp, C.c_int]),
    "gc_trk_loop_create": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "gc_trk_loop_destroy": (C.c_int, [_vp]),
    "gc_trk_loop_set_input_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64]),
    "gc_trk_loop_set_sync": (C.c_int, [_vp, C.c_int, C.POINTER(LoopSyncConf), _fp, C.c_int]),
    "gc_loop_sync_for_signal": (C.c_int, [C.c_char, C.c_char_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(LoopSyncConf)]),
    "gc_trk_loop_start": (C.c_int, [_vp, C.c_int, C.POINTER(LoopConf), _fp, C.c_int]),
    "gc_trk_loop_stop": (C.c_int, [_vp, C.c_int]),
    "gc_trk_loop_run_dev": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "gc_trk_loop_set_geometry": (C.c_int, [_vp, C.c_int, C.c_int]),
    "gc_trk_loop_run": (C.c_int, [_vp, C.c_int, _vp]),
    "gc_gps_l1_ca_code_gen_float": (C.c_int, [_fp, C.c_int32, C.c_uint32]),
    "gc_gps_l1_ca_code_gen_complex_sampled": (C.c_int, [_fp, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_int32)]),
    "gc_glonass_l1_ca_code_gen_float": (C.c_int, [_fp, C.c_uint32]),
    "gc_glonass_l1_ca_code_gen_complex_sampled": (C.c_int, [_fp, C.c_int32, C.c_uint32, C.POINTER(C.c_int32)]),
    "gc_beidou_b1i_code_gen_float": (C.c_int, [_fp, C.c_int32, C.c_uint32]),
    "gc_beidou_b1i_code_gen_complex_sampled": (C.c_int, [_fp, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_int32)]),
    "gc_galileo_e1_code_gen_sinboc11_float": (C.c_int, [_fp, C.c_char_p, C.c_uint32]),
    "gc_galileo_e1_code_gen_complex_sampled": (C.c_int, [_fp, C.c_char_p, C.c_int32, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_int32)]),
    "gc_gps_l2c_m_code_gen_float": (C.c_int, [_fp, C.c_uint32]),
    "gc_gps_l2c_m_code_gen_complex_sampled": (C.c_int, [_fp, C.c_uint32, C.c_int32, C.POINTER(C.c_int32)]),
    "gc_gps_l5i_code_gen_float": (C.c_int, [_fp, C.c_uint32]),
    "gc_gps_l5q_code_gen_float": (C.c_int, [_fp, C.c_uint32]),
    "gc_gps_l5i_code_gen_complex_sampled": (C.c_int, [_fp, C.c_uint32, C.c_int32, C.POINTER(C.c_int32)]),
    "gc_gps_l5q_code_gen_complex_sampled": (C.c_int, [_fp, C.c_uint32, C.c_int32, C.POINTER(C.c_int32)]),
    "gc_beidou_b3i_code_gen_float": (C.c_int, [_fp, C.c_int32, C.c_uint32]),
    "gc_beidou_b3i_code_gen_complex_sampled": (C.c_int, [_fp, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_int32)]),
    "gc_galileo_e5_a_code_gen_complex_primary": (C.c_int, [_fp, C.c_int32, C.c_char_p]),
    "gc_galileo_e5_a_code_gen_complex_sampled": (C.c_int, [_fp, C.c_char_p, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_int32)]),
    "gc_secondary_code": (C.c_int, [C.c_char_p, C.c_uint32, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]),
    "gc_acq_create": (C.c_int, [_vp, C.POINTER(AcqConf), C.c_int, C.POINTER(_vp)]),
    "gc_acq_destroy": (C.c_int, [_vp]),
    "gc_acq_fft_size": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "gc_acq_set_local_code": (C.c_int, [_vp, C.c_int, _fp]),
    "gc_acq_reset": (C.c_int, [_vp]),
    "gc_acq_set_step_two": (C.c_int, [_vp, C.c_int, C.c_float]),
    "gc_acq_set_input_format": (C.c_int, [_vp, C.c_int]),
    "gc_acq_dwell_dev": (C.c_int, [_vp, _vp, C.POINTER(AcqResult), _vp]),
    "gc_acq_dwell": (C.c_int, [_vp, _fp, C.POINTER(AcqResult)]),
    "gc_acq_dwell_enqueue": (C.c_int, [_vp, _vp, _vp]),
    "gc_acq_fetch_results": (C.c_int, [_vp, C.POINTER(AcqResult), _vp]),
    "gc_acq_flush": (C.c_int, [_vp, _vp]),
    "gc_acq_get_grid": (C.c_int, [_vp, C.c_int, _fp]),
    "gc_acq_peek": (C.c_int, [_vp, C.c_int, C.c_int, _fp]),
}

_lib = None


def load_library():
    """Loads libgnsscorr.so (built by gnss-sdr-1_amd/csrc/Makefile).  Raises
    when it is missing: the HIP library IS the product."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GnsscorrError(GC_ERR_NO_DEVICE, "%s not built (run __graft_entry__.build())" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own
        # libamdhip64.so.7.  When torch is going to share the process (device
        # buffers for tests/bench) it must be loaded first, so that this library
        # binds to the same runtime instead of /opt/rocm's copy.
        if os.environ.get("GNSSCORR_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in API.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        # the ctypes mirrors above against the library that was actually loaded
        st = lib.gc_abi_check(C.sizeof(EpochParams), C.sizeof(LoopConf), LOOP_RECORD_DTYPE.itemsize, C.sizeof(LoopSyncConf), C.sizeof(AcqConf), C.sizeof(AcqResult))
        if st != 0:
            raise GnsscorrError(st, lib.gc_last_error().decode() + " -- rebuild with __graft_entry__.build()")
        _lib = lib
    return _lib


def _check(st):
    if st != GC_OK:
        raise GnsscorrError(st, load_library().gc_last_error().decode())


def _f32p(a):
    return a.ctypes.data_as(_fp)


def device_count():
    return load_library().gc_device_count()


def epoch_params(sample_offset, rem_carr_phase_rad, carr_phase_step_rad, rem_code_phase_chips,
        code_phase_step_chips, n_samples, carr_phase_rate_step_rad=0.0, code_phase_rate_step_chips=0.0):
    """gc_epoch_params_fill: the reference's scalar arguments -> kernel arguments
    (cpu_multicorrelator_real_codes.cc:141-149)."""
    p = EpochParams()
    load_library().gc_epoch_params_fill(C.byref(p), int(sample_offset), rem_carr_phase_rad, carr_phase_step_rad,
        carr_phase_rate_step_rad, rem_code_phase_chips, code_phase_step_chips, code_phase_rate_step_chips, int(n_samples))
    return p


def epoch_params_array(records):
    """list (or nested list [ch][epoch]) of EpochParams -> numpy structured array."""
    flat = []
    for r in records:
        if isinstance(r, (list, tuple)):
            flat.extend(r)
        else:
            flat.append(r)
    out = np.zeros(len(flat), EPOCH_DTYPE)
    for i, p in enumerate(flat):
        out[i] = np.frombuffer(bytes(p), EPOCH_DTYPE)[0]
    return out


# ---- PRN replica generators (host side; names follow the reference's free functions) ----
def gps_l1_ca_code_gen_float(prn, chip_shift=0):
    d = np.zeros(1023, np.float32)
    _check(load_library().gc_gps_l1_ca_code_gen_float(_f32p(d), prn, chip_shift))
    return d


def gps_l1_ca_code_gen_complex_sampled(prn, fs, chip_shift=0):
    d = np.zeros(int(fs // 1000) + 8, np.complex64)
    n = C.c_int32()
    _check(load_library().gc_gps_l1_ca_code_gen_complex_sampled(d.view(np.float32).ctypes.data_as(_fp), prn, fs, chip_shift, C.byref(n)))
    return d[:n.value].copy()


def glonass_l1_ca_code_gen_float(chip_shift=0):
    d = np.zeros(511, np.float32)
    _check(load_library().gc_glonass_l1_ca_code_gen_float(_f32p(d), chip_shift))
    return d


def glonass_l1_ca_code_gen_complex_sampled(fs, chip_shift=0):
    d = np.zeros(int(fs // 1000) + 8, np.complex64)
    n = C.c_int32()
    _check(load_library().gc_glonass_l1_ca_code_gen_complex_sampled(d.view(np.float32).ctypes.data_as(_fp), fs, chip_shift, C.byref(n)))
    return d[:n.value].copy()


def beidou_b1i_code_gen_float(prn, chip_shift=0):
    d = np.zeros(2046, np.float32)
    _check(load_library().gc_beidou_b1i_code_gen_float(_f32p(d), prn, chip_shift))
    return d


def beidou_b1i_code_gen_complex_sampled(prn, fs, chip_shift=0):
    d = np.zeros(int(fs // 1000) + 8, np.complex64)
    n = C.c_int32()
    _check(load_library().gc_beidou_b1i_code_gen_complex_sampled(d.view(np.float32).ctypes.data_as(_fp), prn, fs, chip_shift, C.byref(n)))
    return d[:n.value].copy()


def _chips10230(fn, *args):
    d = np.zeros(10230, np.float32)
    _check(getattr(load_library(), fn)(_f32p(d), *args))
    return d


def _sampled10230(fn, fs, code_rate_hz, *args):
    d = np.zeros(int(fs / (code_rate_hz / 10230.0)) + 8, np.complex64)
    n = C.c_int32()
    _check(getattr(load_library(), fn)(d.view(np.float32).ctypes.data_as(_fp), *args, C.byref(n)))
    return d[:n.value].copy()


def gps_l2c_m_code_gen_float(prn):
    return _chips10230("gc_gps_l2c_m_code_gen_float", prn)


def gps_l2c_m_code_gen_complex_sampled(prn, fs):
    return _sampled10230("gc_gps_l2c_m_code_gen_complex_sampled", fs, 0.5115e6, prn, fs)


def gps_l5i_code_gen_float(prn):
    return _chips10230("gc_gps_l5i_code_gen_float", prn)


def gps_l5q_code_gen_float(prn):
    return _chips10230("gc_gps_l5q_code_gen_float", prn)


def gps_l5i_code_gen_complex_sampled(prn, fs):
    return _sampled10230("gc_gps_l5i_code_gen_complex_sampled", fs, 10.23e6, prn, fs)


def gps_l5q_code_gen_complex_sampled(prn, fs):
    return _sampled10230("gc_gps_l5q_code_gen_complex_sampled", fs, 10.23e6, prn, fs)


def beidou_b3i_code_gen_float(prn, chip_shift=0):
    return _chips10230("gc_beidou_b3i_code_gen_float", prn, chip_shift)


def beidou_b3i_code_gen_complex_sampled(prn, fs, chip_shift=0):
    return _sampled10230("gc_beidou_b3i_code_gen_complex_sampled", fs, 10.23e6, prn, fs, chip_shift)


def galileo_e5_a_code_gen_complex_primary(prn, signal):
    d = np.zeros(10230, np.complex64)
    _check(load_library().gc_galileo_e5_a_code_gen_complex_primary(d.view(np.float32).ctypes.data_as(_fp), prn, signal.encode()))
    return d


def galileo_e5_a_code_gen_complex_sampled(signal, prn, fs, chip_shift=0):
    return _sampled10230("gc_galileo_e5_a_code_gen_complex_sampled", fs, 10.23e6, signal.encode(), prn, fs, chip_shift)


def loop_sync_for_signal(system, signal, prn, track_pilot=False, extend_correlation_symbols=1):
    """gc_loop_sync_for_signal: the LoopSyncConf the block's constructor would derive for this signal and satellite."""
    y = LoopSyncConf()
    _check(load_library().gc_loop_sync_for_signal(system.encode(), signal.encode(), prn, int(bool(track_pilot)), extend_correlation_symbols, C.byref(y)))
    return y


def secondary_code(signal, prn=0):
    """Secondary (overlay) code of a signal as a '0'/'1' string: "1C", "B1", "B3", "L5I", "L5Q", "5I", "5Q" (per PRN)."""
    buf = C.create_string_buffer(128)
    n = C.c_int32()
    _check(load_library().gc_secondary_code(signal.encode(), prn, buf, 128, C.byref(n)))
    return buf.value.decode()


def galileo_e1_code_gen_sinboc11_float(signal, prn):
    d = np.zeros(8184, np.float32)
    _check(load_library().gc_galileo_e1_code_gen_sinboc11_float(_f32p(d), signal.encode(), prn))
    return d


def galileo_e1_code_gen_complex_sampled(signal, cboc, prn, fs, chip_shift=0):
    d = np.zeros(int(fs * 0.004) + 8, np.complex64)
    n = C.c_int32()
    _check(load_library().gc_galileo_e1_code_gen_complex_sampled(d.view(np.float32).ctypes.data_as(_fp), signal.encode(), int(cboc), prn, fs,
        chip_shift, C.byref(n)))
    return d[:n.value].copy()


class Context:
    """gc_ctx: one per GPU."""

    def __init__(self, device=0):
        self._h = _vp()
        _check(load_library().gc_ctx_create(device, C.byref(self._h)))
        self.device = device

    def synchronize(self):
        _check(load_library().gc_ctx_synchronize(self._h))

    def register_host_buffer(self, array):
        """gc_ctx_register_host_buffer on a numpy array (the caller keeps it alive until unregister_host_buffer)."""
        _check(load_library().gc_ctx_register_host_buffer(self._h, _vp(array.ctypes.data), array.nbytes))

    def unregister_host_buffer(self, array):
        _check(load_library().gc_ctx_unregister_host_buffer(self._h, _vp(array.ctypes.data)))

    def correlator_batch_stats(self):
        """(launches, calls served, calls that shared their window's upload, largest batch) of the level-1 epoch batcher."""
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
        _check(load_library().gc_correlator_batch_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def close(self):
        if self._h:
            load_library().gc_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IqStream:
    """HBM ring of one RF stream (gc_stream_*): push host blocks once, every channel / acquisition of the
    stream reads them by absolute sample number."""
    _DTYPES = {GC_IQ_F32: (np.complex64, 1), GC_IQ_I16: (np.int16, 2), GC_IQ_I8: (np.int8, 2)}

    def __init__(self, ctx, capacity_samples, max_window_samples, iq_format=GC_IQ_F32):
        self._ctx = ctx
        self.iq_format = iq_format
        self._h = _vp()
        _check(load_library().gc_stream_create(ctx._h, int(iq_format), int(capacity_samples), int(max_window_samples), C.byref(self._h)))

    def push(self, block):
        """block: complex64 [n] (GC_IQ_F32) or int16 / int8 [n, 2].  Returns the absolute index of its first sample."""
        dt, per = self._DTYPES[self.iq_format]
        block = np.ascontiguousarray(block, dt)
        n = block.size // per
        first = C.c_uint64(0)
        _check(load_library().gc_stream_push(self._h, block.ctypes.data_as(_vp), n, C.byref(first)))
        return int(first.value)

    def push_pinned(self, host_ptr, n_samples):
        """host_ptr: address of page-locked host memory (e.g. a pinned torch tensor's data_ptr()); no staging copy."""
        first = C.c_uint64(0)
        _check(load_library().gc_stream_push_pinned(self._h, _vp(host_ptr), int(n_samples), C.byref(first)))
        return int(first.value)

    def info(self):
        o, h, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(load_library().gc_stream_info(self._h, C.byref(o), C.byref(h), C.byref(c)))
        return int(o.value), int(h.value), int(c.value)

    def synchronize(self):
        _check(load_library().gc_stream_synchronize(self._h))

    def close(self):
        if self._h:
            load_library().gc_stream_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipMulticorrelatorRealCodes:
    """Image of Cpu_Multicorrelator_Real_Codes
    (src/algorithms/tracking/libs/cpu_multicorrelator_real_codes.h:45-69): same
    methods, same argument meaning, pointers retained not copied, every method
    returns True."""

    def __init__(self, ctx):
        self._ctx = ctx
        self._h = _vp()
        _check(load_library().gc_correlator_create(ctx._h, C.byref(self._h)))
        self._keep = {}

    def set_high_dynamics_resampler(self, use_high_dynamics_resampler):
        _check(load_library().gc_correlator_set_high_dynamics_resampler(self._h, int(bool(use_high_dynamics_resampler))))

    def init(self, max_signal_length_samples, n_correlators):
        _check(load_library().gc_correlator_init(self._h, max_signal_length_samples, n_correlators))
        return True

    def set_local_code_and_taps(self, code_length_chips, local_code_in, shifts_chips):
        assert local_code_in.dtype == np.float32 and shifts_chips.dtype == np.float32
        self._keep["code"] = local_code_in
        self._keep["shifts"] = shifts_chips
        _check(load_library().gc_correlator_set_local_code_and_taps(self._h, code_length_chips, _f32p(local_code_in), _f32p(shifts_chips)))
        return True

    def set_input_output_vectors(self, corr_out, sig_in):
        assert corr_out.dtype == np.complex64 and sig_in.dtype == np.complex64
        self._keep["out"] = corr_out
        self._keep["in"] = sig_in
        _check(load_library().gc_correlator_set_input_output_vectors(self._h,
            corr_out.view(np.float32).ctypes.data_as(_fp), sig_in.view(np.float32).ctypes.data_as(_fp)))
        return True

    def Carrier_wipeoff_multicorrelator_resampler(self, rem_carrier_phase_in_rad, phase_step_rad, *rest):
        """7-argument form (…, phase_rate_step_rad, rem_code_phase_chips, code_phase_step_chips,
        code_phase_rate_step_chips, signal_length_samples) or the 6-argument overload without
        phase_rate_step_rad, as in the reference (.cc:129-170)."""
        lib = load_library()
        if len(rest) == 5:
            rate, rem_code, code_step, code_rate, n = rest
            _check(lib.gc_correlator_carrier_wipeoff_multicorrelator_resampler(self._h, rem_carrier_phase_in_rad,
                phase_step_rad, rate, rem_code, code_step, code_rate, int(n)))
        elif len(rest) == 4:
            rem_code, code_step, code_rate, n = rest
            _check(lib.gc_correlator_carrier_wipeoff_multicorrelator_resampler_6(self._h, rem_carrier_phase_in_rad,
                phase_step_rad, rem_code, code_step, code_rate, int(n)))
        else:
            raise TypeError("expected 6 or 7 arguments")
        return True

    def free(self):
        _check(load_library().gc_correlator_free(self._h))
        return True

    def close(self):
        if self._h:
            load_library().gc_correlator_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipMulticorrelator(HipMulticorrelatorRealCodes):
    """Image of Cpu_Multicorrelator (complex chips;
    src/algorithms/tracking/libs/cpu_multicorrelator.h:46-64): init, set_local_code_and_taps,
    set_input_output_vectors, the 5-argument Carrier_wipeoff_multicorrelator_resampler, free."""

    def set_high_dynamics_resampler(self, use_high_dynamics_resampler):
        raise AttributeError("Cpu_Multicorrelator has no high-dynamics resampler")

    def set_local_code_and_taps(self, code_length_chips, local_code_in, shifts_chips):
        assert local_code_in.dtype == np.complex64 and shifts_chips.dtype == np.float32
        self._keep["code"] = local_code_in
        self._keep["shifts"] = shifts_chips
        _check(load_library().gc_correlator_set_local_code_and_taps_complex(self._h, code_length_chips,
            local_code_in.view(np.float32).ctypes.data_as(_fp), _f32p(shifts_chips)))
        return True

    def Carrier_wipeoff_multicorrelator_resampler(self, rem_carrier_phase_in_rad, phase_step_rad, rem_code_phase_chips,
            code_phase_step_chips, signal_length_samples):
        _check(load_library().gc_correlator_carrier_wipeoff_multicorrelator_resampler_5(self._h, rem_carrier_phase_in_rad,
            phase_step_rad, rem_code_phase_chips, code_phase_step_chips, int(signal_length_samples)))
        return True


class HipMulticorrelator16sc(HipMulticorrelatorRealCodes):
    """Image of Cpu_Multicorrelator_16sc (src/algorithms/tracking/libs/cpu_multicorrelator_16sc.h:44-67):
    lv_16sc_t chips, input and output as int16 arrays of shape (n, 2)."""

    def set_high_dynamics_resampler(self, use_high_dynamics_resampler):
        raise AttributeError("Cpu_Multicorrelator_16sc has no high-dynamics resampler")

    def set_local_code_and_taps(self, code_length_chips, local_code_in, shifts_chips):
        assert local_code_in.dtype == np.int16 and local_code_in.shape[-1] == 2 and shifts_chips.dtype == np.float32
        self._keep["code"] = local_code_in
        self._keep["shifts"] = shifts_chips
        _check(load_library().gc_correlator_set_local_code_and_taps_16sc(self._h, code_length_chips,
            local_code_in.ctypes.data_as(_i16p), _f32p(shifts_chips)))
        return True

    def set_input_output_vectors(self, corr_out, sig_in):
        assert corr_out.dtype == np.int16 and sig_in.dtype == np.int16
        self._keep["out"] = corr_out
        self._keep["in"] = sig_in
        _check(load_library().gc_correlator_set_input_output_vectors_16sc(self._h, corr_out.ctypes.data_as(_i16p), sig_in.ctypes.data_as(_i16p)))
        return True

    def Carrier_wipeoff_multicorrelator_resampler(self, rem_carrier_phase_in_rad, phase_step_rad, rem_code_phase_chips,
            code_phase_step_chips, signal_length_samples):
        _check(load_library().gc_correlator_carrier_wipeoff_multicorrelator_resampler_5(self._h, rem_carrier_phase_in_rad,
            phase_step_rad, rem_code_phase_chips, code_phase_step_chips, int(signal_length_samples)))
        return True


def stream_broadcast_pinned(streams, host_ptr, n_samples):
    """gc_stream_broadcast_pinned: one page-locked block into the ring of every GPU."""
    arr = (_vp * len(streams))(*[s._h for s in streams])
    _check(load_library().gc_stream_broadcast_pinned(arr, len(streams), _vp(host_ptr), int(n_samples)))


class TrackingBatch:
    """gc_trk_batch: all channels of one GPU, many epochs per launch."""

    def __init__(self, ctx, n_channels, n_taps, max_code_length, high_dyn=False):
        self._ctx = ctx
        self.n_channels, self.n_taps = n_channels, n_taps
        self._sc16 = False
        self._h = _vp()
        _check(load_library().gc_trk_batch_create(ctx._h, n_channels, n_taps, max_code_length, int(high_dyn), C.byref(self._h)))

    def set_code(self, ch, code, shifts_chips):
        code = np.ascontiguousarray(code, np.float32)
        shifts = np.ascontiguousarray(shifts_chips, np.float32)
        assert shifts.size == self.n_taps
        _check(load_library().gc_trk_batch_set_code(self._h, ch, _f32p(code), code.size, _f32p(shifts)))

    def set_complex_codes(self, on=True):
        _check(load_library().gc_trk_batch_set_complex_codes(self._h, int(bool(on))))

    def set_code_complex(self, ch, code, shifts_chips):
        code = np.ascontiguousarray(code, np.complex64)
        shifts = np.ascontiguousarray(shifts_chips, np.float32)
        assert shifts.size == self.n_taps
        _check(load_library().gc_trk_batch_set_code_complex(self._h, ch, code.view(np.float32).ctypes.data_as(_fp), code.size, _f32p(shifts)))

    def set_16sc(self, on=True):
        _check(load_library().gc_trk_batch_set_16sc(self._h, int(bool(on))))
        self._sc16 = bool(on)

    def set_code_16sc(self, ch, code, shifts_chips):
        code = np.ascontiguousarray(code, np.int16)
        shifts = np.ascontiguousarray(shifts_chips, np.float32)
        assert shifts.size == self.n_taps and code.ndim == 2 and code.shape[1] == 2
        _check(load_library().gc_trk_batch_set_code_16sc(self._h, ch, code.ctypes.data_as(_i16p), code.shape[0], _f32p(shifts)))

    def set_shifts(self, ch, shifts_chips):
        shifts = np.ascontiguousarray(shifts_chips, np.float32)
        assert shifts.size == self.n_taps
        _check(load_library().gc_trk_batch_set_shifts(self._h, ch, _f32p(shifts)))

    def set_input_format(self, iq_format):
        _check(load_library().gc_trk_batch_set_input_format(self._h, int(iq_format)))

    def set_input_dev(self, ch, dev_ptr, n_samples):
        _check(load_library().gc_trk_batch_set_input_dev(self._h, ch, _vp(dev_ptr), int(n_samples)))

    def set_input_stream(self, ch, stream):
        _check(load_library().gc_trk_batch_set_input_stream(self._h, ch, stream._h))
        self._streams = getattr(self, "_streams", {})
        self._streams[ch] = stream  # keep the ring alive

    def set_read_floor(self, oldest_index_read):
        _check(load_library().gc_trk_batch_set_read_floor(self._h, int(oldest_index_read)))

    def set_nominal_length(self, n):
        _check(load_library().gc_trk_batch_set_nominal_length(self._h, int(n)))

    def set_slices(self, n):
        _check(load_library().gc_trk_batch_set_slices(self._h, int(n)))

    def run_dev(self, n_epochs, dev_params_ptr, dev_out_ptr, stream=None):
        _check(load_library().gc_trk_batch_run_dev(self._h, n_epochs, _vp(dev_params_ptr), _vp(dev_out_ptr), _vp(stream or 0)))

    def run(self, n_epochs, params):
        """params: structured array (EPOCH_DTYPE) of n_channels*n_epochs records, channel-major.
        Returns complex64 [n_channels, n_epochs, n_taps] (int16 [n_channels, n_epochs, n_taps, 2] in 16-bit mode)."""
        params = np.ascontiguousarray(params, EPOCH_DTYPE)
        assert params.size == self.n_channels * n_epochs
        if self._sc16:
            out = np.zeros((self.n_channels, n_epochs, self.n_taps, 2), np.int16)
            _check(load_library().gc_trk_batch_run(self._h, n_epochs, params.ctypes.data_as(_vp), C.cast(out.ctypes.data, _fp)))
            return out
        out = np.zeros((self.n_channels, n_epochs, self.n_taps), np.complex64)
        _check(load_library().gc_trk_batch_run(self._h, n_epochs, params.ctypes.data_as(_vp), out.view(np.float32).ctypes.data_as(_fp)))
        return out

    def close(self):
        if self._h:
            load_library().gc_trk_batch_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TrackingLoop:
    """gc_trk_loop: closed-loop DLL/PLL tracking on the device (correlations + loop maths per epoch in one launch)."""

    def __init__(self, ctx, n_channels, max_code_length):
        self._ctx = ctx
        self.n_channels = n_channels
        self._h = _vp()
        _check(load_library().gc_trk_loop_create(ctx._h, n_channels, max_code_length, C.byref(self._h)))

    def set_input_dev(self, ch, dev_ptr, n_samples):
        _check(load_library().gc_trk_loop_set_input_dev(self._h, ch, _vp(dev_ptr), int(n_samples)))

    def set_input_format(self, iq_format):
        _check(load_library().gc_trk_loop_set_input_format(self._h, int(iq_format)))

    def set_input_stream(self, ch, stream):
        _check(load_library().gc_trk_loop_set_input_stream(self._h, ch, stream._h))

    def set_sync(self, ch, sync, data_code=None):
        """Installs (or, with sync=None, removes) the LoopSyncConf of a channel; data_code is the data component's
        replica for pilot tracking.  Takes effect at the next start()."""
        if sync is None:
            _check(load_library().gc_trk_loop_set_sync(self._h, ch, None, None, 0))
            return
        if data_code is None:
            _check(load_library().gc_trk_loop_set_sync(self._h, ch, C.byref(sync), None, 0))
        else:
            data_code = np.ascontiguousarray(data_code, np.float32)
            _check(load_library().gc_trk_loop_set_sync(self._h, ch, C.byref(sync), _f32p(data_code), data_code.size))

    def start(self, ch, conf, code):
        code = np.ascontiguousarray(code, np.float32)
        _check(load_library().gc_trk_loop_start(self._h, ch, C.byref(conf), _f32p(code), code.size))

    def stop(self, ch):
        _check(load_library().gc_trk_loop_stop(self._h, ch))

    def set_geometry(self, threads_per_workgroup=0, slices_per_channel=0):
        """0 = automatic.  threads: 256 / 512 / 1024 (persistent kernel); slices: workgroups per channel-period (1: persistent
        one-workgroup-per-channel kernel; >= 2: one launch per code period, the last slice runs the loop maths)."""
        _check(load_library().gc_trk_loop_set_geometry(self._h, int(threads_per_workgroup), int(slices_per_channel)))

    def run(self, n_epochs):
        """Returns a structured array [n_channels, n_epochs] of LOOP_RECORD_DTYPE."""
        out = np.zeros((self.n_channels, n_epochs), LOOP_RECORD_DTYPE)
        _check(load_library().gc_trk_loop_run(self._h, n_epochs, out.ctypes.data_as(_vp)))
        return out

    def run_dev(self, n_epochs, dev_records_ptr, stream=None):
        _check(load_library().gc_trk_loop_run_dev(self._h, n_epochs, _vp(dev_records_ptr), _vp(stream or 0)))

    def close(self):
        if self._h:
            load_library().gc_trk_loop_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PcpsAcquisition:
    """gc_acq: pcps_acquisition (pcps_acquisition.cc) for n_sats satellites that
    search the same input block."""

    def __init__(self, ctx, n_sats, fs_in, sampled_ms, ms_per_code, samples_per_ms, samples_per_code, samples_per_chip,
            doppler_max, doppler_step, max_dwells=1, bit_transition_flag=False, use_cfar=True, num_doppler_bins_override=0,
            make_2_steps=False, num_doppler_bins_step2=4, doppler_step2=125.0):
        self._ctx = ctx
        self.n_sats = n_sats
        conf = AcqConf(int(fs_in), sampled_ms, ms_per_code, samples_per_ms, samples_per_code, samples_per_chip,
            doppler_max, doppler_step, max_dwells, int(bit_transition_flag), int(use_cfar), num_doppler_bins_override,
            int(make_2_steps), num_doppler_bins_step2, doppler_step2)
        self.conf = conf
        self._h = _vp()
        _check(load_library().gc_acq_create(ctx._h, C.byref(conf), n_sats, C.byref(self._h)))
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(load_library().gc_acq_fft_size(self._h, C.byref(a), C.byref(b), C.byref(c)))
        self.fft_size, self.consumed_samples, self.num_doppler_bins = a.value, b.value, c.value

    def set_local_code(self, sat, code):
        code = np.ascontiguousarray(code, np.complex64)
        need = self.fft_size // 2 if self.conf.bit_transition_flag else self.consumed_samples
        assert code.size >= need, (code.size, need)
        _check(load_library().gc_acq_set_local_code(self._h, sat, code.view(np.float32).ctypes.data_as(_fp)))

    def reset(self):
        _check(load_library().gc_acq_reset(self._h))

    def set_step_two(self, enable, doppler_center_hz=0.0):
        _check(load_library().gc_acq_set_step_two(self._h, int(enable), float(doppler_center_hz)))
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(load_library().gc_acq_fft_size(self._h, C.byref(a), C.byref(b), C.byref(c)))
        self.num_doppler_bins = c.value

    def dwell(self, iq):
        iq = np.ascontiguousarray(iq, np.complex64)
        assert iq.size >= self.consumed_samples
        res = (AcqResult * self.n_sats)()
        _check(load_library().gc_acq_dwell(self._h, iq.view(np.float32).ctypes.data_as(_fp), res))
        return list(res)

    def set_frequency_offset(self, offset_hz):
        _check(load_library().gc_acq_set_frequency_offset(self._h, int(offset_hz)))

    def dwell_stream(self, stream, first_index):
        res = (AcqResult * self.n_sats)()
        _check(load_library().gc_acq_dwell_stream(self._h, stream._h, int(first_index), C.cast(res, _vp)))
        return list(res)

    def set_input_format(self, iq_format):
        _check(load_library().gc_acq_set_input_format(self._h, int(iq_format)))

    def dwell_dev(self, dev_ptr, stream=None):
        res = (AcqResult * self.n_sats)()
        _check(load_library().gc_acq_dwell_dev(self._h, _vp(dev_ptr), res, _vp(stream or 0)))
        return list(res)

    def dwell_enqueue(self, dev_ptr, stream=None):
        _check(load_library().gc_acq_dwell_enqueue(self._h, _vp(dev_ptr), _vp(stream or 0)))

    def flush(self, stream=None):
        """Enqueues held-back inverse passes and the last dwell's statistics kernel; no copy, no synchronisation."""
        _check(load_library().gc_acq_flush(self._h, _vp(stream or 0)))

    def fetch_results(self, stream=None):
        res = (AcqResult * self.n_sats)()
        _check(load_library().gc_acq_fetch_results(self._h, res, _vp(stream or 0)))
        return list(res)

    def grid(self, sat):
        g = np.zeros((self.num_doppler_bins, self.fft_size), np.float32)
        _check(load_library().gc_acq_get_grid(self._h, sat, _f32p(g)))
        return g

    PEEK_WIPEOFF, PEEK_SPECTRUM, PEEK_CODE, PEEK_ROW_MAX = 0, 1, 2, 3

    def peek(self, what, index):
        """gc_acq_peek: a device-resident intermediate in natural order (complex64[fft_size], or float32[num_doppler_bins, 2]
        = (row maximum, its index) for PEEK_ROW_MAX)."""
        if what == self.PEEK_ROW_MAX:
            out = np.zeros((self.num_doppler_bins, 2), np.float32)
        else:
            out = np.zeros(2 * self.fft_size, np.float32)
        _check(load_library().gc_acq_peek(self._h, int(what), int(index), _f32p(out)))
        return out if what == self.PEEK_ROW_MAX else out.view(np.complex64)

    def close(self):
        if self._h:
            load_library().gc_acq_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
