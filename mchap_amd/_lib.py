"""ctypes loader for libmchap_hip.so (the C ABI declared in include/mchap_hip.h).

There is no CPU fallback: if the library is missing, or no MI355X is visible when a compute
entry point is called, an exception is raised.
"""
import ctypes as C
import os
import threading
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# MCHAP_HIP_LIB: an alternative build of the same library (profiling variants made by `make stats` / `make phases`)
SO = os.environ.get("MCHAP_HIP_LIB") or os.path.join(CSRC, "libmchap_hip.so")
# The parity suite's library: the same C ABI plus the two earlier sampler designs (kernels 1 and 4) and the narrower
# instantiations of the phased sampler.  Loaded instead of SO while MCHAP_HIP_TEST_KERNELS is set (tests only).
TEST_SO = os.path.join(CSRC, "libmchap_hip_test.so")

MAX_TEMPS = 16
MAX_PLOIDY = 8          # the fast de novo samplers and the device posterior summary (include/mchap_hip.h MCHAP_MAX_PLOIDY)
MAX_PLOIDY_GENERAL = 15  # the general de novo sampler, the exact caller and the call sampler (MCHAP_MAX_PLOIDY_DENOVO)
MAX_ALLELE = 8
MAX_READS = 4096

OK = 0
ERR_NAN_LLK = -1
ERR_BAD_ARG = -2
ERR_BREAKS = -3
ERR_LIMIT = -4
ERR_HIP = -5
ERR_NO_DEVICE = -6

UNIT_OK = 0
UNIT_ALL_FIXED = 1
UNIT_NAN_LLK = 2
UNIT_BREAKS = 3
UNIT_BAD_INITIAL = 4


class DenovoTuning(C.Structure):
    """mchap_denovo_tuning: measurement / test knobs (0 = default); results never depend on them."""

    _fields_ = [
        ("cache_slots", C.c_int32),
        ("flags", C.c_int32),
        ("spec_group", C.c_int32),
        ("pipe_first", C.c_int32),
        ("pipe_resume", C.c_int32),
        ("pipe_rounds", C.c_int32),
        ("pipe_max", C.c_int32),
        ("pipe_parts", C.c_int32),
        ("prep_lds_limit", C.c_int32),
        ("pipe_stop", C.c_int32),
        ("reserved", C.c_int32 * 6),
    ]


# environment variable -> field of DenovoTuning.  Read by the Python mirror (DenovoMCMC._cfg) when a fit is set up, so that
# the test-suite and tools/ can sweep them; the library itself reads no environment.
TUNING_ENV = {
    "MCHAP_HIP_CACHE_SLOTS": "cache_slots", "MCHAP_HIP_FLAGS": "flags", "MCHAP_HIP_GROUP": "spec_group",
    "MCHAP_HIP_PIPE_FIRST": "pipe_first", "MCHAP_HIP_PIPE_RESUME": "pipe_resume", "MCHAP_HIP_PIPE_MAX": "pipe_max",
    "MCHAP_HIP_PIPE_PARTS": "pipe_parts", "MCHAP_HIP_PREP_LDS": "prep_lds_limit", "MCHAP_HIP_PIPE_STOP": "pipe_stop",
}


def tuning_from_env():
    """A DenovoTuning filled from the MCHAP_HIP_* variables that are set, or None."""
    t = DenovoTuning()
    used = False
    for env, field in TUNING_ENV.items():
        v = os.environ.get(env)
        if v is not None and v != "":
            setattr(t, field, int(v))
            used = True
    v = os.environ.get("MCHAP_HIP_ROUNDS")  # resume rounds: the field holds rounds + 1 (0 = default)
    if v is not None and v != "":
        t.pipe_rounds = int(v) + 1
        used = True
    if os.environ.get("MCHAP_HIP_NO_BP_CACHE"):
        t.flags |= 16
        used = True
    v = os.environ.get("MCHAP_HIP_PIPE_GROUP")  # lanes per chain of the phased sampler: test library only
    if v is not None and v != "":
        t.reserved[0] = int(v)
        used = True
    return t if used else None


_epoch_lock = threading.Lock()
_epoch = [0]


def next_cache_epoch():
    """The number a fit tags its likelihood-cache entries with (mchap_denovo_cfg.cache_epoch): ONE sequence for the whole process,
    whichever copy of the library runs the fit -- the parity suite loads libmchap_hip.so and libmchap_hip_test.so side by side, and
    torch's allocator hands the same workspace block to both.  0 after 2^31 - 2 fits: the library then clears the tables."""
    with _epoch_lock:
        _epoch[0] += 1
        return _epoch[0] if _epoch[0] < (1 << 31) - 1 else 0


class DenovoCfg(C.Structure):
    _fields_ = [
        ("steps", C.c_int32),
        ("chains", C.c_int32),
        ("n_temps", C.c_int32),
        ("n_intervals", C.c_int32),
        ("temperatures", C.c_double * MAX_TEMPS),
        ("fix_homozygous", C.c_double),
        ("p_recomb", C.c_double),
        ("p_partial_dosage", C.c_double),
        ("p_dosage", C.c_double),
        ("seed", C.c_uint64),
        ("break_table", C.c_void_p),
        ("max_pos", C.c_int32),
        ("llk_cache", C.c_int32),
        ("kernel", C.c_int32),
        ("cache_epoch", C.c_int32),
        ("tuning", C.POINTER(DenovoTuning)),
        ("timer", C.c_void_p),
    ]


UNIT_DTYPE = np.dtype(
    [
        ("reads_off", "<i8"),
        ("counts_off", "<i8"),
        ("nalleles_off", "<i8"),
        ("initial_off", "<i8"),
        ("trace_off", "<i8"),
        ("llk_off", "<i8"),
        ("fixed_off", "<i8"),
        ("n_reads", "<i4"),
        ("n_pos", "<i4"),
        ("max_allele", "<i4"),
        ("ploidy", "<i4"),
        ("inbreeding", "<f8"),
        ("stream_id", "<u8"),
        ("initial_n_het", "<i4"),
        ("reserved", "<i4"),
    ],
    align=True,
)
assert UNIT_DTYPE.itemsize == 96


class ExactOut(C.Structure):
    """mchap_exact_out: optional device outputs of mchap_exact_call_batch_device."""

    _fields_ = [(n, C.c_void_p) for n in (
        "mode_alleles", "mode_llk", "mode_prob", "support_prob", "freqs", "occur", "llks", "llks64", "posteriors",
        "arr_mode_alleles", "arr_mode_prob", "arr_support_prob", "arr_freqs", "arr_counts", "arr_occur")]


class MchapLibraryError(RuntimeError):
    pass


def build(force=False):
    """Compile libmchap_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j%d" % max(1, min(16, os.cpu_count() or 1))]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return SO


_libs = {}


def lib():
    """The loaded library: libmchap_hip.so, or the parity suite's libmchap_hip_test.so while MCHAP_HIP_TEST_KERNELS is set."""
    path = TEST_SO if os.environ.get("MCHAP_HIP_TEST_KERNELS") else SO
    L = _libs.get(path)
    if L is None:
        if not os.path.exists(path):
            raise MchapLibraryError(
                "%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C mchap_amd/csrc`. There is no CPU fallback." % path
            )
        # PyTorch-ROCm bundles its own libamdhip64.so.7; whichever HIP runtime is loaded first serves the whole
        # process (same soname).  Load torch's first when torch is installed so that tensors handed to the
        # *_device entry points and this library share one runtime.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is optional for the host-pointer entry points
            pass
        L = C.CDLL(path)
        L.mchap_version.restype = C.c_char_p
        L.mchap_last_error.restype = C.c_char_p
        L.mchap_denovo_lds_bytes.restype = C.c_int64
        L.mchap_denovo_workspace_bytes.restype = C.c_int64
        L.mchap_exact_workspace_bytes.restype = C.c_int64
        L.mchap_exact_workspace_bytes_cached.restype = C.c_int64
        L.mchap_call_mcmc_workspace_bytes.restype = C.c_int64
        L.mchap_call_mcmc_workspace_bytes_for.restype = C.c_int64
        L.mchap_timer_ms.restype = C.c_double
        L.mchap_bam_count.restype = C.c_int64
        L.mchap_bam_count.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.mchap_bam_columns.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_char_p, C.c_int] + [C.c_void_p] * 16
        L.mchap_timer_ms.argtypes = [C.c_void_p]
        L.mchap_timer_destroy.argtypes = [C.c_void_p]
        _libs[path] = L
    return L


class SamplerTimer:
    """A pair of HIP events owned by the caller (mchap_timer_create): set `cfg.timer = timer.handle` and the fit brackets
    its sampler launches with them on its own stream; `ms()` waits for the second event."""

    def __init__(self):
        self.L = lib()
        h = C.c_void_p()
        check(self.L.mchap_timer_create(C.byref(h)))
        self.handle = h.value

    def ms(self):
        return float(self.L.mchap_timer_ms(C.c_void_p(self.handle)))

    def close(self):
        if self.handle:
            self.L.mchap_timer_destroy(C.c_void_p(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sampler_name(cfg, units_host):
    """Name of the sampler kernel(s) a fit of this batch dispatches to."""
    buf = C.create_string_buffer(160)
    check(lib().mchap_denovo_sampler_name(C.byref(cfg), len(units_host), ptr(units_host), buf, 160))
    return buf.value.decode()


EXPORTS = [
    "mchap_bam_count",
    "mchap_bam_columns",
    "mchap_denovo_fit_batch_device",
    "mchap_denovo_fit_batch_calls_device",
    "mchap_denovo_fit_batch",
    "mchap_log_likelihood_batch",
    "mchap_trace_posterior_batch_device",
    "mchap_trace_posterior_max_states",
    "mchap_trace_posterior_listed_device",
    "mchap_trace_incongruence_listed_device",
    "mchap_trace_incongruence_batch_device",
    "mchap_trace_posterior_batch_wph_device",
    "mchap_trace_posterior_max_states_wph",
    "mchap_trace_posterior_listed_wph_device",
    "mchap_trace_incongruence_batch_wph_device",
    "mchap_trace_incongruence_listed_wph_device",
    "mchap_call_incongruence_batch_device",
    "mchap_call_incongruence_listed_device",
    "mchap_exact_genotype_likelihoods",
    "mchap_exact_genotype_posteriors",
    "mchap_exact_posterior_mode_batch",
    "mchap_exact_workspace_bytes",
    "mchap_exact_workspace_bytes_cached",
    "mchap_exact_call_batch_device",
    "mchap_exact_posterior_summaries_batch_device",
    "mchap_exact_posterior_summaries",
    "mchap_call_mcmc_workspace_bytes",
    "mchap_call_mcmc_workspace_bytes_for",
    "mchap_call_mcmc_batch_device",
    "mchap_call_mcmc_batch",
    "mchap_version",
    "mchap_last_error",
    "mchap_device_count",
    "mchap_denovo_lds_bytes",
    "mchap_denovo_workspace_bytes",
    "mchap_timer_create",
    "mchap_timer_ms",
    "mchap_timer_destroy",
    "mchap_denovo_sampler_name",
    "mchap_read_log_batch",
    "mchap_read_log_product_batch",
    "mchap_wave_sum_batch",
    "mchap_denovo_trace_words_per_haplotype",
]


def last_error():
    return lib().mchap_last_error().decode()


def check(rc):
    """Map a library return code to the exception type the reference raises."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ERR_NAN_LLK:
        raise ValueError("Encountered log likelihood of nan")
    if rc == ERR_BREAKS:
        raise ValueError("breaks must be smaller then n")
    if rc == ERR_BAD_ARG:
        raise AssertionError(msg)
    if rc == ERR_LIMIT:
        raise NotImplementedError("mchap_hip: " + msg)
    raise MchapLibraryError("mchap_hip error %d: %s" % (rc, msg))


def ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)
