"""ctypes binding of ``libscarlet_amd.so`` (C ABI in ``include/scarlet_amd.h``).

There is no CPU fallback: if the library is missing or no GPU is visible the
calls raise.  ``import torch`` happens first on purpose -- torch ships its own
HIP runtime / rocFFT with the same SONAMEs as /opt/rocm, and loading it first
makes this library resolve against the copies that are already in the process
(one HIP runtime per process).
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCARLET_AMD_LIB: another build of the same library (A/B comparisons, tools/ab_compare.py)
LIB_PATH = os.environ.get("SCARLET_AMD_LIB") or os.path.join(_HERE, "libscarlet_amd.so")

PROX_MONOTONIC = 1
PROX_SYMMETRY = 2
PROX_POSITIVE = 4
PROX_CENTER_ON = 8
PROX_NORM_MAX = 16
PROX_NORM_SUM = 32
PROX_L1 = 64
PROX_L0 = 128
PROX_FIT_CENTER = 256  # MonotonicityConstraint(fit_center_radius=1): 9 consecutive plans
PROX_BG_THRESH = 512   # scarlet.lite background threshold (replaces positivity)
PROX_L_RELATIVE = 1024  # L0/L1 threshold x step of the proximal sub-iteration
PROX_MONO_MASK = 2048  # MonotonicityConstraint(use_mask=True)
SCHEME_AMSGRAD, SCHEME_FISTA = 0, 1
COMPONENT_POINT_SOURCE = 1 << 16  # PointSource: morphology = model PSF at a free centre
COMPONENT_SHIFTING = 1 << 17  # image morphology moved by a free Fourier shift
COMPONENT_FIXED_SED = 1 << 18
COMPONENT_FIXED_MORPH = 1 << 19
PROX_EXTENDED_SOURCE = PROX_MONOTONIC | PROX_POSITIVE | PROX_CENTER_ON | PROX_NORM_MAX

ERR_ARITHMETIC = -4

c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_u8p = ctypes.POINTER(ctypes.c_uint8)


class BatchDesc(ctypes.Structure):
    _fields_ = [
        (name, ctypes.c_int32)
        for name in (
            "n_blends", "C", "H", "W", "n_components", "kernel_h", "kernel_w",
            "kernel_bands", "kernel_per_blend", "fft_h", "fft_w", "max_iter", "conv_path",
        )
    ]


class Components(ctypes.Structure):
    _fields_ = [
        ("blend", c_i32p), ("origin_y", c_i32p), ("origin_x", c_i32p),
        ("box_h", c_i32p), ("box_w", c_i32p), ("sed", c_f32p), ("morph", c_f32p),
        ("sed_min_step", c_f32p), ("sed_rel_step", c_f32p), ("morph_step", c_f32p),
        ("prox_flags", c_i32p), ("sweep_plan", c_i32p), ("min_gradient", c_f32p),
        ("l_thresh", c_f32p), ("morph_rel_step", c_f32p),
        ("center", c_f64p), ("psf_sigma", c_f32p), ("shift_step", c_f32p),
        ("center_floor", c_f32p), ("bg_level", c_f32p), ("fista_step", c_f32p),
        ("sym_strength", c_f32p), ("pos_floor", c_f32p), ("chain_repeat", c_i32p),
        ("shift_rel_step", c_f32p), ("psf_beta", c_f32p),
    ]


# name -> (restype, argtypes); every symbol include/scarlet_amd.h declares
SYMBOLS = {
    "smi_last_error": (ctypes.c_char_p, []),
    "smi_device_count": (ctypes.c_int, []),
    "smi_version": (ctypes.c_char_p, []),
    "smi_prox_weighted_monotonic_f32": (
        ctypes.c_int,
        [c_f32p, c_f32p, c_i32p, ctypes.c_int32, c_i32p, ctypes.c_int32, ctypes.c_int32,
         ctypes.c_float],
    ),
    "smi_prox_weighted_monotonic_f64": (
        ctypes.c_int,
        [c_f64p, c_f64p, c_i32p, ctypes.c_int32, c_i32p, ctypes.c_int32, ctypes.c_int32,
         ctypes.c_double],
    ),
    "smi_prox_weighted_monotonic_many_f32": (
        ctypes.c_int,
        [ctypes.c_int32, c_f32p, ctypes.c_int32, c_f32p, c_i32p, ctypes.c_int32, c_i32p,
         ctypes.c_int32, ctypes.c_float],
    ),
    "smi_prox_weighted_monotonic_many_f64": (
        ctypes.c_int,
        [ctypes.c_int32, c_f64p, ctypes.c_int32, c_f64p, c_i32p, ctypes.c_int32, c_i32p,
         ctypes.c_int32, ctypes.c_double],
    ),
    "smi_apply_filter_f32": (
        ctypes.c_int,
        [c_f32p, ctypes.c_int32, ctypes.c_int32, c_f32p, ctypes.c_int32, c_i32p, c_i32p,
         c_i32p, c_i32p, c_f32p],
    ),
    "smi_apply_filter_f64": (
        ctypes.c_int,
        [c_f64p, ctypes.c_int32, ctypes.c_int32, c_f64p, ctypes.c_int32, c_i32p, c_i32p,
         c_i32p, c_i32p, c_f64p],
    ),
    "smi_get_valid_monotonic_pixels_f32": (
        ctypes.c_int,
        [ctypes.c_int32, ctypes.c_int32, c_f32p, ctypes.c_int32, ctypes.c_int32, c_u8p, c_u8p,
         ctypes.c_double, c_i32p, ctypes.c_double],
    ),
    "smi_get_valid_monotonic_pixels_f64": (
        ctypes.c_int,
        [ctypes.c_int32, ctypes.c_int32, c_f64p, ctypes.c_int32, ctypes.c_int32, c_u8p, c_u8p,
         ctypes.c_double, c_i32p, ctypes.c_double],
    ),
    "smi_linear_interpolate_invalid_pixels_f32": (
        ctypes.c_int,
        [c_i32p, c_i32p, ctypes.c_int32, c_u8p, c_f32p, ctypes.c_int32, ctypes.c_int32, c_u8p,
         ctypes.c_double, ctypes.c_int32, c_i32p],
    ),
    "smi_linear_interpolate_invalid_pixels_f64": (
        ctypes.c_int,
        [c_i32p, c_i32p, ctypes.c_int32, c_u8p, c_f64p, ctypes.c_int32, ctypes.c_int32, c_u8p,
         ctypes.c_double, ctypes.c_int32, c_i32p],
    ),
    "smi_resampler_create": (
        ctypes.c_int,
        [c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
         ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)],
    ),
    "smi_resampler_render": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p]),
    "smi_resampler_time": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32,
                                          ctypes.POINTER(ctypes.c_double)]),
    "smi_resampler_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "smi_resampler_get_path": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "smi_resampler_set_path": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "smi_batch_attach_lowres": (
        ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_i32p, c_f32p, c_f32p, ctypes.c_double]
    ),
    "smi_batch_get_lowres_rendered": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, c_f32p]),
    "smi_batch_create": (
        ctypes.c_int,
        [ctypes.POINTER(BatchDesc), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)],
    ),
    "smi_batch_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "smi_batch_add_sweep_plan": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, c_f64p, c_i32p, c_i32p,
         ctypes.c_int32],
    ),
    "smi_sweep_ring_plan": (
        ctypes.c_int,
        [ctypes.c_int32, ctypes.c_int32, c_f64p, c_i32p, c_i32p, ctypes.c_int32, c_i32p, c_f32p,
         ctypes.POINTER(ctypes.c_uint16), ctypes.c_int64],
    ),
    "smi_batch_set_observation": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p]),
    "smi_batch_set_observation_device": (
        ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    ),
    "smi_batch_set_kernel": (ctypes.c_int, [ctypes.c_void_p, c_f32p]),
    "smi_batch_set_components": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Components)]),
    "smi_batch_set_moments": (ctypes.c_int, [ctypes.c_void_p] + [c_f32p] * 6),
    "smi_batch_get_moments": (ctypes.c_int, [ctypes.c_void_p] + [c_f32p] * 6),
    "smi_batch_get_parameters": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p]),
    "smi_batch_set_parameters": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p]),
    "smi_batch_get_centers": (ctypes.c_int, [ctypes.c_void_p] + [c_f64p] * 5),
    "smi_batch_set_center_moments": (ctypes.c_int, [ctypes.c_void_p] + [c_f64p] * 3),
    "smi_batch_set_centers": (ctypes.c_int, [ctypes.c_void_p, c_f64p]),
    "smi_batch_get_model_morphology": (ctypes.c_int, [ctypes.c_void_p, c_f32p]),
    "smi_batch_set_scheme": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "smi_batch_get_fista_state": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f64p]),
    "smi_batch_set_fista_state": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f64p]),
    "smi_batch_set_log_norm": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "smi_batch_add_observation": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p]),
    "smi_batch_add_loss_constant": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "smi_batch_set_previous_loss": (ctypes.c_int, [ctypes.c_void_p, c_f64p]),
    "smi_batch_set_sub_ranges": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "smi_batch_get_sub_ranges": (ctypes.c_int, [ctypes.c_void_p, c_i32p]),
    "smi_batch_set_inline_render": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "smi_batch_set_optimizer": (
        ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float]
    ),
    "smi_batch_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "smi_batch_forward": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, c_f64p]),
    "smi_batch_gradient": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p]),
    "smi_batch_step": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_int32,
         ctypes.c_int32, ctypes.c_int32],
    ),
    "smi_batch_status": (ctypes.c_int, [ctypes.c_void_p, c_i32p, c_i32p]),
    "smi_batch_get_states": (ctypes.c_int, [ctypes.c_void_p, c_i32p]),
    "smi_batch_fit": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_int32, ctypes.c_int32,
         ctypes.c_int32, c_i32p],
    ),
    "smi_batch_get_loss": (ctypes.c_int, [ctypes.c_void_p, c_f64p, ctypes.c_int32, c_i32p]),
    "smi_batch_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "smi_batch_set_kernel_shift": (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_int32, ctypes.c_int32,
                                                  c_i32p, c_f64p, c_f64p, ctypes.c_double]),
    "smi_batch_get_kernel_shift": (ctypes.c_int, [ctypes.c_void_p, c_f64p, c_f64p, c_f64p, c_f32p]),
    "smi_batch_set_kernel_shift_relative_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double]),
    "smi_batch_save_state": (ctypes.c_int, [ctypes.c_void_p]),
    "smi_batch_restore_state": (ctypes.c_int, [ctypes.c_void_p]),
    "smi_batch_enable_timing": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "smi_batch_get_timing": (ctypes.c_int, [ctypes.c_void_p, c_f64p, ctypes.c_int32]),
    "smi_batch_fft_shape": (ctypes.c_int, [ctypes.c_void_p, c_i32p, c_i32p]),
    "smi_batch_conv_path_used": (ctypes.c_int, [ctypes.c_void_p, c_i32p]),
    "smi_set_hw_queues": (ctypes.c_int, [ctypes.c_int32]),
    "smi_observation_uploads": (ctypes.c_int64, []),
    "smi_batch_resize_test": (ctypes.c_int, [ctypes.c_void_p, c_i32p, c_f64p]),
    "smi_batch_get_component_states": (ctypes.c_int, [ctypes.c_void_p, c_i32p, ctypes.c_int32, c_f32p]),
    "smi_batch_update_components": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Components), c_i32p, c_f32p]),
    "smi_batch_set_states": (ctypes.c_int, [ctypes.c_void_p, c_i32p]),
    "smi_batch_set_iteration_base": (ctypes.c_int, [ctypes.c_void_p, c_i32p]),
    "smi_batch_set_pause_at": (ctypes.c_int, [ctypes.c_void_p, c_i32p]),
    "smi_batch_get_converged": (ctypes.c_int, [ctypes.c_void_p, c_i32p]),
    "smi_batch_set_round": (ctypes.c_int, [ctypes.c_void_p, c_i32p, c_i32p, c_i32p]),
    "smi_batch_get_round": (ctypes.c_int, [ctypes.c_void_p, c_i32p, c_i32p, c_i32p]),
    "smi_batch_get_progress": (ctypes.c_int, [ctypes.c_void_p, c_i32p, c_i32p]),
}

_lib = None


class ScarletAmdError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ScarletAmdError(
                "{} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C scarlet_amd/csrc` (there is no CPU fallback)".format(
                    LIB_PATH
                )
            )
        try:
            import torch  # noqa: F401  (see module docstring)
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            if not hasattr(lib, name) and os.environ.get("SCARLET_AMD_LIB"):
                continue  # an older build under A/B comparison
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
        lib.smi_set_hw_queues(_effective_hw_queues())
    return _lib


# ---------------------------------------------------------------------------
# Hardware queues.  Ranges of blends are stepped on streams of their own
# (smi_batch_set_sub_ranges); HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4) and streams that share a queue run one after the other.  Eight queues let a
# small batch -- one GPU's shard of a multi-GPU job -- run four ranges side by side (128
# blends: 540 k -> 615 k blend-iterations/s).  The HIP runtime reads the variable once, when it
# starts, and nothing reports the number it took.  Importing the package therefore ASKS for
# eight queues -- it exports GPU_MAX_HW_QUEUES=8 -- when the caller's environment says nothing
# and the runtime has not started yet (a user's 128 .. 767-blend batch then runs four ranges
# without knowing about ``configure``); a value the caller exported is never touched,
# ``SCARLET_AMD_HW_QUEUES=keep`` leaves the environment alone, ``configure(hw_queues=n)``
# sets another number on request.  The library is told the number known to be in effect
# (4 when the runtime was already up).
# ---------------------------------------------------------------------------
_hw_queues = None  # set by configure()
_warned = False


def _hip_started():
    import sys

    torch = sys.modules.get("torch")  # (not imported yet: it has not started anything)
    return bool(torch is not None and torch.cuda.is_initialized())


if ("GPU_MAX_HW_QUEUES" not in os.environ and os.environ.get("SCARLET_AMD_HW_QUEUES") != "keep"
        and not _hip_started()):
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
# what the environment said when this module was imported: a value exported later, after the
# HIP runtime has started, is not what the runtime read (it would make the library assume
# queues that do not exist)
_env_at_import = os.environ.get("GPU_MAX_HW_QUEUES")


def _effective_hw_queues():
    if _hw_queues is not None:
        return _hw_queues
    # exported by the caller's environment (the shell, a launcher): in effect from the start.
    # A value that appeared after the import counts only while the runtime has not started.
    value = os.environ.get("GPU_MAX_HW_QUEUES") if not _hip_started() else _env_at_import
    try:
        return max(1, int(value or "4"))
    except ValueError:
        return 4


def configure(hw_queues=None):
    """Process-wide settings; call before anything touches the GPU.

    hw_queues: number of hardware queues the HIP runtime should map streams onto (sets
        ``GPU_MAX_HW_QUEUES`` for this process and its children).  With 8, batches of 128 to
        767 blends run four ranges of blends side by side instead of three.  When the HIP
        runtime has already started (``torch.cuda`` initialised) the setting cannot take
        effect any more: a warning is issued once, nothing is changed and the library keeps
        three ranges.
    Returns the number of hardware queues the library assumes."""
    global _hw_queues, _warned
    if hw_queues is not None:
        hw_queues = int(hw_queues)
        if os.environ.get("GPU_MAX_HW_QUEUES") == str(hw_queues):
            _hw_queues = hw_queues  # already what the runtime reads / has read
        elif _hip_started():
            if not _warned:
                import warnings

                warnings.warn(
                    "scarlet_amd.configure(hw_queues=%d): the HIP runtime has already started "
                    "with GPU_MAX_HW_QUEUES=%s; small batches keep three ranges of blends. "
                    "Call configure() before the first use of torch.cuda, or export "
                    "GPU_MAX_HW_QUEUES=%d." % (hw_queues, os.environ.get("GPU_MAX_HW_QUEUES", "unset (4)"),
                                               hw_queues), RuntimeWarning, stacklevel=2)
                _warned = True
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = str(hw_queues)
            _hw_queues = hw_queues
        if _lib is not None:
            _lib.smi_set_hw_queues(_effective_hw_queues())
    return _effective_hw_queues()


def check(status):
    """Turn a negative smi_status into an exception (ArithmeticError for
    non-finite parameters, as the reference's Model.check_parameters)."""
    if status < 0:
        msg = load().smi_last_error().decode()
        if status == ERR_ARITHMETIC:
            raise ArithmeticError(msg)
        raise ScarletAmdError("libscarlet_amd: {} (status {})".format(msg, status))
    return status


def ptr(arr, ctype):
    if arr is None:
        return None
    return arr.ctypes.data_as(ctypes.POINTER(ctype))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)
