"""ctypes binding of libconsenrich_amd.so (the C ABI in include/consenrich_amd.h).

There is deliberately NO fallback: if the shared library is missing, or no MI355X is visible when a compute entry
point is called, an exception is raised.  (The oracle under /oracle is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libconsenrich_amd.so")

FP = C.POINTER(C.c_float)
DP = C.POINTER(C.c_double)
I64P = C.POINTER(C.c_int64)

# flag bits (include/consenrich_amd.h)
USE_LAMBDA, USE_KAPPA, USE_QSCALE, USE_APN, RETURN_NLL, NLL_IN_D = (1 << i for i in range(6))
(ARR_D, ARR_XF, ARR_PF, ARR_PNOISE, ARR_XS, ARR_PS, ARR_LAG, ARR_RESID, ARR_LAMBDA, ARR_KAPPA, ARR_QSCALE,
 ARR_SUMGAIN0, ARR_SUMGAIN1, ARR_EFFQ_LEVEL, ARR_EFFQ_TREND, ARR_MUNCTRACE, ARR_BACKGROUND, ARR_BACKGROUND_NEXT,
 ARR_COUNT) = range(19)
BG_OK, BG_NO_SUPPORT, BG_BAD_PIVOT, BG_UNRELIABLE, BG_NONFINITE = range(5)
BG_INIT_FROM_CURRENT, BG_ZERO_STATE = 1, 2
EXPORT_FORWARD, EXPORT_SMOOTH, EXPORT_RESID, EXPORT_MULT = 1, 2, 4, 8


class Model(C.Structure):
    _fields_ = [
        ("state_dim", C.c_int32), ("reserved_", C.c_int32),
        ("F", C.c_double * 4), ("Q0", C.c_double * 4),
        ("state_init", C.c_double), ("state_covar_init", C.c_double), ("pad", C.c_double),
        ("w_min", C.c_double), ("w_max", C.c_double), ("k_min", C.c_double), ("k_max", C.c_double),
        ("apn_min_q", C.c_double), ("apn_max_q", C.c_double), ("apn_thresh", C.c_double),
        ("apn_scale", C.c_double), ("apn_pc", C.c_double),
    ]


class FwdIO(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n", C.c_int64), ("data", FP), ("munc", FP), ("lam", FP), ("kappa", FP), ("qscale", FP),
        ("flags", C.c_uint32), ("reserved_", C.c_uint32), ("D", FP), ("xf", FP), ("Pf", FP), ("pnoise", FP),
    ]


class FwdOut(C.Structure):
    _fields_ = [("sum_d", C.c_double), ("sum_nll", C.c_double)]


class EcmCfg(C.Structure):
    _fields_ = [
        ("max_iters", C.c_int64), ("inner_iters", C.c_int64), ("rtol", C.c_double), ("nu", C.c_double),
        ("use_lambda", C.c_int32), ("use_kappa", C.c_int32), ("use_apn", C.c_int32), ("reserved_", C.c_int32),
    ]


class EcmOut(C.Structure):
    _fields_ = [
        ("iters_done", C.c_int64), ("final_nll", C.c_double), ("initial_nll", C.c_double),
        ("abs_rel_change", C.c_double), ("rel_improvement", C.c_double), ("stable_iters", C.c_int64),
        ("nll_increase_count", C.c_int64), ("converged", C.c_int32), ("skipped", C.c_int32),
        ("has_initial_nll", C.c_int32), ("reserved_", C.c_int32),
    ]


class BgCfg(C.Structure):
    _fields_ = [
        ("lam_first", C.c_double), ("lam", C.c_double), ("negative_penalty_multiplier", C.c_double),
        ("zero_center", C.c_int32), ("use_nonnegative", C.c_int32), ("use_lambda", C.c_int32),
        ("use_initial", C.c_int32), ("max_passes", C.c_int32), ("block_len", C.c_int32),
    ]


class BgOut(C.Structure):
    _fields_ = [
        ("support", C.c_int64), ("weight_sum", C.c_double), ("weight_scale", C.c_double),
        ("roundoff_index", C.c_double), ("shift_rms", C.c_double), ("proposal_rms", C.c_double),
        ("reference_rms", C.c_double), ("bad_index", C.c_int64),
        ("bad_value", C.c_double), ("passes", C.c_int32), ("status", C.c_int32),
    ]


class QseedSampleCfg(C.Structure):
    _fields_ = [("precision_cap_quantile", C.c_double), ("precision_cap_multiplier", C.c_double),
                ("max_transition_samples", C.c_int64), ("precision_sample_cap", C.c_int64),
                ("signal_panel_size", C.c_int64)]


class QseedSampleDiag(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("pair_count", "sampled_pair_count", "precision_sample_count", "scan_count",
                                         "candidate_count", "selected_count")] + \
               [("capped_mode", C.c_int32), ("reserved", C.c_int32), ("precision_cap", C.c_double),
                ("precision_cap_fraction", C.c_double), ("transition_sample_fraction", C.c_double)]


class QseedPostCfg(C.Structure):
    _fields_ = [("q_floor", C.c_double), ("q_cap", C.c_double), ("robust_t_nu", C.c_double),
                ("q_seed_prior_level", C.c_double), ("min_transitions", C.c_int64), ("prior_log_sd", C.c_double),
                ("default_t_nu", C.c_double), ("grid_size", C.c_int64)]


class QseedPost(C.Structure):
    _fields_ = [("transition_count", C.c_int64), ("ok", C.c_int32), ("reserved", C.c_int32)] + \
               [(k, C.c_double) for k in ("effective_transition_count", "median_sampling_variance", "prior_level",
                                          "posterior_mode", "posterior_median", "posterior_q05", "posterior_q95",
                                          "transition_q90")]


class QseedCfg(C.Structure):
    _fields_ = [("sample", QseedSampleCfg), ("pad", C.c_double), ("min_q", C.c_double), ("max_q", C.c_double),
                ("delta_f", C.c_double), ("robust_t_nu", C.c_double), ("q_seed_prior_level", C.c_double),
                ("min_transitions", C.c_int64), ("prior_log_sd", C.c_double), ("default_t_nu", C.c_double),
                ("grid_size", C.c_int64), ("state_dim", C.c_int32), ("reserved", C.c_int32)]


class QseedOut(C.Structure):
    _fields_ = [("q_level", C.c_double), ("q_trend", C.c_double), ("level_pre_clamp", C.c_double),
                ("trend_pre_clamp", C.c_double), ("source", C.c_int32), ("reason", C.c_int32),
                ("sample", QseedSampleDiag), ("post", QseedPost)]


class ObjectiveCfg(C.Structure):
    _fields_ = [("nu", C.c_double), ("lam_first", C.c_double), ("lam", C.c_double),
                ("negative_penalty_multiplier", C.c_double), ("pad", C.c_double), ("use_lambda_penalty", C.c_int32),
                ("use_kappa_penalty", C.c_int32), ("use_lambda_weights", C.c_int32), ("use_nonnegative", C.c_int32)]


class ObjectiveTerms(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("robust_observation_penalty", "robust_process_penalty",
                                          "first_difference_penalty", "second_difference_penalty", "negative_penalty",
                                          "weight_median")] + [("effective_observation_count", C.c_int64)]


class BwSummary(C.Structure):
    _fields_ = [("bases_covered", C.c_int64), ("non_finite", C.c_int64), ("min_val", C.c_double), ("max_val", C.c_double),
                ("sum_data", C.c_double), ("sum_squares", C.c_double)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double)]


class RunStats(C.Structure):
    _fields_ = [
        ("blocks", C.c_int64), ("fix_launches", C.c_int64), ("reruns_p", C.c_int64), ("reruns_x", C.c_int64),
        ("reruns_b", C.c_int64), ("block_len", C.c_int32), ("warm_p", C.c_int32), ("warm_x", C.c_int32),
        ("warm_b", C.c_int32), ("x_tol_ulps", C.c_int32), ("pipeline_redos", C.c_int32),
        ("local_repairs", C.c_int64), ("ws_warm_f", C.c_int32), ("ws_warm_b", C.c_int32), ("sb_bailouts", C.c_int64), ("tail_groups", C.c_int64),
        ("nat_first_use_off_main", C.c_int64),
    ]


# every symbol include/consenrich_amd.h declares: (restype, argtypes)
SYMBOLS = {
    "csr_last_error": (C.c_char_p, []),
    "csr_abi_version": (C.c_int, []),
    "csr_build_id": (C.c_char_p, []),
    "csr_device_count": (C.c_int, []),
    "csr_create": (C.c_void_p, [C.c_int]),
    "csr_destroy": (None, [C.c_void_p]),
    "csr_set_tuning": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "csr_set_validation": (C.c_int, [C.c_void_p, C.c_int32]),
    "csr_synchronize": (C.c_int, [C.c_void_p]),
    "csr_batch_configure": (C.c_int, [C.c_void_p, C.POINTER(Model), C.c_int64, C.c_int32, I64P]),
    "csr_batch_set_model": (C.c_int, [C.c_void_p, C.POINTER(Model)]),
    "csr_batch_upload": (C.c_int, [C.c_void_p, C.c_int32, FP, FP]),
    "csr_batch_upload_multipliers": (C.c_int, [C.c_void_p, C.c_int32, FP, FP, FP]),
    "csr_batch_synthesize": (C.c_int, [C.c_void_p, C.c_uint64]),
    "csr_batch_stats": (C.c_int, [C.c_void_p]),
    "csr_batch_forward": (C.c_int, [C.c_void_p, C.c_uint32, DP, DP]),
    "csr_batch_backward": (C.c_int, [C.c_void_p]),
    "csr_batch_forward_backward": (C.c_int, [C.c_void_p, C.c_uint32, DP, DP]),
    "csr_batch_sums": (C.c_int, [C.c_void_p, DP, DP]),
    "csr_batch_diagnostics": (C.c_int, [C.c_void_p, C.c_uint32]),
    "csr_batch_background_update": (C.c_int, [C.c_void_p, C.POINTER(BgCfg), C.POINTER(BgOut)]),
    "csr_batch_background_apply": (C.c_int, [C.c_void_p, C.c_char_p]),
    "csr_batch_set_background": (C.c_int, [C.c_void_p, C.c_int32, FP]),
    "csr_format_bedgraph": (C.c_int64, [C.c_char_p, C.c_int64, I64P, I64P, C.c_int64, C.c_int64, C.c_int64, FP, C.c_int32,
                                        C.c_char_p, C.c_int64]),
    "csr_batch_format_bedgraph": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_int64,
                                              C.c_int64, C.c_int64, C.c_char_p, C.c_int64]),
    "csr_batch_bigwig_sections": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_int64,
                                              C.c_int64, C.c_int64, C.c_int32, C.c_char_p, C.c_int64, C.POINTER(BwSummary)]),
    "csr_batch_bigwig_zoom": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_int64,
                                          C.c_int64, C.c_int64, C.c_int64, C.c_char_p, C.c_int64]),
    "csr_observation_total_information": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(C.c_uint8), DP,
                                                    C.c_double, C.c_double, DP]),
    "csr_fold_mask_and_information": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int32), I64P, I64P,
                                                C.c_int64, C.c_void_p, C.c_int32, C.POINTER(C.c_uint8), DP, DP, C.c_double,
                                                C.c_double, C.POINTER(C.c_uint8), DP, DP, DP, DP]),
    "csr_batch_make_fold": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_int32), I64P, I64P,
                                      C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_float, DP, DP, DP]),
    "csr_solve_background": (C.c_int, [C.c_int32, I64P, DP, DP, C.c_double, C.c_double, C.c_int32, C.c_int32, DP, I64P,
                                       DP]),
    "csr_background_weighted_stats": (C.c_int, [C.c_int64, C.c_int64, FP, FP, DP, DP, I64P]),
    "csr_output_diagnostics": (C.c_int, [C.POINTER(Model), C.c_int64, C.c_int64, FP, FP, FP, FP, FP, FP, FP, FP, FP, FP,
                                         FP]),
    "csr_batch_ecm": (C.c_int, [C.c_void_p, C.POINTER(EcmCfg), C.c_uint32, C.POINTER(EcmOut), DP]),
    "csr_batch_ecm_masked": (C.c_int, [C.c_void_p, C.POINTER(EcmCfg), C.c_uint32, C.c_char_p, C.POINTER(EcmOut), DP]),
    "csr_batch_export": (C.c_int, [C.c_void_p, C.c_uint32]),
    "csr_batch_download": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "csr_batch_device_array": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), I64P]),
    "csr_batch_chain_offset": (C.c_int64, [C.c_void_p, C.c_int32]),
    "csr_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "csr_profile_read": (C.c_int, [C.c_void_p, C.POINTER(KernelTime), C.c_int32, C.POINTER(C.c_int32)]),
    "csr_get_run_stats": (C.c_int, [C.c_void_p, C.POINTER(RunStats)]),
    "csr_forward_pass": (C.c_int, [C.POINTER(Model), C.POINTER(FwdIO), C.POINTER(FwdOut)]),
    "csr_backward_pass": (C.c_int, [C.POINTER(Model), C.c_int64, C.c_int64, FP, FP, FP, FP, FP, FP, FP, C.c_int64, FP]),
    "csr_fixed_background_ecm": (C.c_int, [C.POINTER(Model), C.POINTER(EcmCfg), C.c_int64, C.c_int64, FP, FP, FP, FP,
                                           FP, FP, FP, FP, FP, DP, C.POINTER(EcmOut)]),
    "csr_batch_download_inputs": (C.c_int, [C.c_void_p, C.c_int32, FP, FP]),
    "csr_batch_set_chain_q": (C.c_int, [C.c_void_p, DP]),
    "csr_batch_step": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, DP, DP]),
    "csr_batch_step_forward": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, DP, DP]),
    "csr_batch_objective_terms": (C.c_int, [C.c_void_p, C.POINTER(ObjectiveCfg), C.POINTER(ObjectiveTerms)]),
    "csr_batch_phase_tracks": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, DP, DP, C.POINTER(C.c_int32)]),
    "csr_batch_gain_summary": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, DP]),
    "csr_batch_forward_masked": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, DP, DP]),
    "csr_qseed_same_track": (C.c_int, [C.c_int64, C.c_int64, DP, DP, C.POINTER(C.c_uint8), C.POINTER(QseedSampleCfg), DP, DP,
                                       DP, I64P, C.POINTER(QseedSampleDiag)]),
    "csr_qseed_pooled": (C.c_int, [C.c_int64, C.c_int64, DP, DP, C.POINTER(C.c_uint8), DP, DP, DP, I64P]),
    "csr_qseed_posterior": (C.c_int, [C.c_int64, DP, DP, DP, C.POINTER(QseedPostCfg), C.POINTER(QseedPost)]),
    "csr_batch_qseed": (C.c_int, [C.c_void_p, C.POINTER(QseedCfg), C.POINTER(QseedOut)]),
    "csr_comm_unique_id": (C.c_int, [C.c_char_p]),
    "csr_comm_create": (C.c_void_p, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32]),
    "csr_comm_destroy": (None, [C.c_void_p]),
    "csr_comm_world": (C.c_int, [C.c_void_p]),
    "csr_comm_rank": (C.c_int, [C.c_void_p]),
    "csr_comm_allreduce_max": (C.c_int, [C.c_void_p, DP]),
    "csr_comm_allreduce_sum": (C.c_int, [C.c_void_p, DP]),
    "csr_comm_barrier": (C.c_int, [C.c_void_p]),
    "csr_batch_gather_tracks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, FP]),
    "csr_expected_transition_residual_sums": (C.c_int, [C.c_int32, C.c_int64, DP, DP, DP, DP, DP, DP, I64P]),
}

_lib = None


class ConsenrichAMDError(RuntimeError):
    pass


def lib():
    """Load the shared library; raise if it has not been built (python -m consenrich_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ConsenrichAMDError(
                f"{LIB_PATH} is missing: build it with `python -m consenrich_amd.build` (hipcc, gfx950). "
                "consenrich_amd has no CPU fallback."
            )
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)       # AttributeError if the ABI and the binding disagree
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def build_id() -> str:
    """What the loaded library was built from: "abi N src <hash16> <flags>" (csr_build_id)."""
    return lib().csr_build_id().decode("ascii", "replace")


def last_error() -> str:
    msg = lib().csr_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int) -> None:
    if rc != 0:
        raise ConsenrichAMDError(last_error() or f"consenrich_amd call failed (rc={rc})")


def device_count() -> int:
    return int(lib().csr_device_count())


def require_gpu() -> None:
    if device_count() <= 0:
        raise ConsenrichAMDError("no MI355X/HIP device visible: consenrich_amd has no CPU fallback")


def fp(a):
    return None if a is None else a.ctypes.data_as(FP)


def dp(a):
    return None if a is None else a.ctypes.data_as(DP)
