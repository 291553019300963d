"""ctypes binding of libwmix_amd.so (the only compute path; no fallback)."""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
# WMIX_AMD_LIB: another build of the same library (developer A/B runs of an experiment variant, tools_dev/); never a fallback
LIB_PATH = os.path.abspath(os.environ["WMIX_AMD_LIB"]) if os.environ.get("WMIX_AMD_LIB") else os.path.join(HERE, "libwmix_amd.so")
INCLUDE_DIR = os.path.join(os.path.dirname(HERE), "include")

# wmx_version() of the library these mirrors were written against (wmix_amd/csrc/wmx_core.hip)
EXPECTED_VERSION = 400

_lib = None


class WmxError(RuntimeError):
    pass


def lib():
    """Load libwmix_amd.so.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WmxError(
                "libwmix_amd.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C wmix_amd/csrc` (there is no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm brings its own copy of libamdhip64 and loads it by path.  Loaded AFTER torch, this
        # library's `libamdhip64.so.N` dependency binds to that copy (same SONAME) and both sides share devices, streams and pointers;
        # loaded BEFORE torch, /opt/rocm's copy comes in as a second runtime, and the one that initialises second finds no device
        # ("no ROCm-capable device is detected").  The mirrors here always work next to torch (device memory, streams), so torch goes
        # first.  A C host has no torch and none of this (examples/*.c).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        # a library older than these bindings (a stale build, a path given in WMIX_AMD_LIB) must say so, not die in ctypes
        # (round-5 ADVICE): both checks come before anything else is read from it
        if getattr(L, "wmx_build_info", None) is None or getattr(L, "wmx_version", None) is None:
            raise WmxError("%s is older than these bindings (no wmx_build_info / wmx_version): rebuild it with `make -C wmix_amd/csrc`" % LIB_PATH)
        L.wmx_version.restype = C.c_int
        if L.wmx_version() != EXPECTED_VERSION:
            raise WmxError("%s is wmx_version %d, these bindings expect %d: rebuild it with `make -C wmix_amd/csrc`"
                           % (LIB_PATH, L.wmx_version(), EXPECTED_VERSION))
        L.wmx_build_info.restype = C.c_char_p
        L.wmx_build_info.argtypes = []
        info = L.wmx_build_info().decode(errors="replace")
        if info != "default" and os.environ.get("WMIX_AMD_ALLOW_VARIANT_BUILD") != "1":
            raise WmxError("%s is not the product build: wmx_build_info() = %r (a developer variant; set WMIX_AMD_ALLOW_VARIANT_BUILD=1 to "
                           "load it on purpose)" % (LIB_PATH, info))
        _declare(L)
        _lib = L
    return _lib


def build_info():
    """wmx_build_info() of the loaded library: "default", or the developer flags of a variant build."""
    return lib().wmx_build_info().decode(errors="replace")


def check(rc, what=""):
    if rc != 0:
        msg = lib().wmx_last_error().decode(errors="replace")
        raise WmxError("%s failed (rc=%d): %s" % (what or "wmx call", rc, msg))


def declared_symbols():
    """Every function name declared in include/*.h (used by the ABI test)."""
    names = []
    for fn in sorted(os.listdir(INCLUDE_DIR)):
        if not fn.endswith(".h"):
            continue
        src = open(os.path.join(INCLUDE_DIR, fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}()]*\)\s*;", src):
            names.append(m.group(1))
    return sorted(set(names))


def _declare(L):
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    L.wmx_last_error.restype = C.c_char_p
    L.wmx_last_error.argtypes = []
    L.wmx_device_count.restype = i
    L.wmx_version.restype = i
    L.wmx_build_info.restype = C.c_char_p
    L.wmx_build_info.argtypes = []
    for name in ("wmx_g711_encode", "wmx_g711_decode"):
        f = getattr(L, name)
        f.restype = i
        f.argtypes = [i, vp, vp, sz, vp]
    L.wmx_ns_create.restype = i
    L.wmx_ns_create.argtypes = [C.POINTER(vp), i, i, i]
    L.wmx_ns_destroy.restype = i
    L.wmx_ns_destroy.argtypes = [vp]
    L.wmx_ns_packet_samples.restype = i
    L.wmx_ns_packet_samples.argtypes = [vp]
    L.wmx_ns_state_words.restype = i
    L.wmx_ns_state_words.argtypes = [vp]
    L.wmx_ns_export_state.restype = i
    L.wmx_ns_export_state.argtypes = [vp, i, vp, vp]
    L.wmx_ns_process.restype = i
    L.wmx_ns_process.argtypes = [vp, vp, vp, i, C.c_long, C.c_long, vp]
    L.wmx_nsx_create.restype = i
    L.wmx_nsx_create.argtypes = [C.POINTER(vp), i, i, i]
    L.wmx_nsx_destroy.restype = i
    L.wmx_nsx_destroy.argtypes = [vp]
    L.wmx_nsx_packet_samples.restype = i
    L.wmx_nsx_packet_samples.argtypes = [vp]
    L.wmx_nsx_state_bytes.restype = i
    L.wmx_nsx_state_bytes.argtypes = [vp]
    L.wmx_nsx_process.restype = i
    L.wmx_nsx_process.argtypes = [vp, vp, vp, i, C.c_long, C.c_long, vp]
    L.wmx_aecm_create.restype = i
    L.wmx_aecm_create.argtypes = [C.POINTER(vp), i, i, i, i]
    L.wmx_aecm_destroy.restype = i
    L.wmx_aecm_destroy.argtypes = [vp]
    L.wmx_aecm_packet_samples.restype = i
    L.wmx_aecm_packet_samples.argtypes = [vp]
    L.wmx_aecm_state_bytes.restype = i
    L.wmx_aecm_state_bytes.argtypes = [vp]
    L.wmx_aecm_run.restype = i
    L.wmx_aecm_run.argtypes = [vp, i, vp, C.c_long, vp, vp, i, C.c_long, C.c_long, i, vp]
    L.wmx_vad_create.restype = i
    L.wmx_vad_create.argtypes = [C.POINTER(vp), i, i, i, i]
    L.wmx_vad_destroy.restype = i
    L.wmx_vad_destroy.argtypes = [vp]
    L.wmx_vad_packet_samples.restype = i
    L.wmx_vad_packet_samples.argtypes = [vp]
    L.wmx_vad_process.restype = i
    L.wmx_vad_process.argtypes = [vp, vp, i, i, C.c_long, C.c_long, vp]
    L.wmx_agc_create.restype = i
    L.wmx_agc_create.argtypes = [C.POINTER(vp), i, i, i, i, i]
    L.wmx_agc_destroy.restype = i
    L.wmx_agc_destroy.argtypes = [vp]
    L.wmx_agc_set_gain.restype = i
    L.wmx_agc_set_gain.argtypes = [vp, i]
    L.wmx_chain_create_groups.restype = i
    L.wmx_chain_create_groups.argtypes = [C.POINTER(vp), i, i, i, i, i, C.c_uint, i, vp]
    L.wmx_chain_process_groups.restype = i
    L.wmx_chain_process_groups.argtypes = [vp, vp, C.c_long, C.c_long, vp, vp, i, C.c_long, C.c_long, vp, vp, vp, vp]
    L.wmx_pipe_create.restype = i
    L.wmx_pipe_create.argtypes = [C.POINTER(vp), i, i, i, i, C.c_uint]
    L.wmx_pipe_create_pcm.restype = i
    L.wmx_pipe_create_pcm.argtypes = [C.POINTER(vp), i, i, i, i, i, i, C.c_uint]
    L.wmx_pipe_create_pcm_calls.restype = i
    L.wmx_pipe_create_pcm_calls.argtypes = [C.POINTER(vp), i, i, i, i, i, i, C.c_uint]
    L.wmx_rt_create_pcm_calls.restype = i
    L.wmx_rt_create_pcm_calls.argtypes = [C.POINTER(vp), C.c_long, i, i, i, i, i, i, C.c_uint]
    L.wmx_pipe_destroy.restype = i
    L.wmx_pipe_destroy.argtypes = [vp]
    L.wmx_pipe_slots.restype = i
    L.wmx_pipe_slots.argtypes = [vp]
    L.wmx_pipe_datagram_bytes.restype = i
    L.wmx_pipe_datagram_bytes.argtypes = [vp]
    for name in ("wmx_pipe_in", "wmx_pipe_out", "wmx_pipe_far"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp, i]
    for name in ("wmx_pipe_chain", "wmx_pipe_senders"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp]
    L.wmx_pipe_submit.restype = i
    L.wmx_pipe_submit.argtypes = [vp, vp, C.POINTER(i), vp]
    L.wmx_pipe_wait.restype = i
    L.wmx_pipe_wait.argtypes = [vp, i]
    L.wmx_pipe_failed_steps.restype = C.c_long
    L.wmx_pipe_failed_steps.argtypes = [vp]
    L.wmx_rt_create_pcm.restype = i
    L.wmx_rt_create_pcm.argtypes = [C.POINTER(vp), C.c_long, i, i, i, i, i, i, C.c_uint]
    L.wmx_rt_create_rtp.restype = i
    L.wmx_rt_create_rtp.argtypes = [C.POINTER(vp), C.c_long, i, i, i, i, C.c_uint]
    L.wmx_rt_destroy.restype = i
    L.wmx_rt_destroy.argtypes = [vp]
    L.wmx_rt_set_compute_streams.restype = i
    L.wmx_rt_set_compute_streams.argtypes = [vp, i]
    L.wmx_rt_poll.restype = i
    L.wmx_rt_poll.argtypes = [vp]
    L.wmx_pipe_poll.restype = i
    L.wmx_pipe_poll.argtypes = [vp, i]
    L.wmx_rt_batches.restype = i
    L.wmx_rt_batches.argtypes = [vp]
    L.wmx_rt_batch_streams.restype = i
    L.wmx_rt_batch_streams.argtypes = [vp, i]
    L.wmx_rt_pipe.restype = vp
    L.wmx_rt_pipe.argtypes = [vp, i]
    L.wmx_rt_far.restype = vp
    L.wmx_rt_far.argtypes = [vp, i]
    L.wmx_rt_submit.restype = i
    L.wmx_rt_submit.argtypes = [vp, vp, C.POINTER(i), vp]
    L.wmx_rt_wait.restype = i
    L.wmx_rt_wait.argtypes = [vp]
    L.wmx_rt_tick.restype = i
    L.wmx_rt_tick.argtypes = [vp, vp, C.POINTER(i), vp]
    L.wmx_rt_step_resident.restype = i
    L.wmx_rt_step_resident.argtypes = [vp, vp, C.c_long, vp, vp, C.c_long, vp]
    L.wmx_pipe_step_resident.restype = i
    L.wmx_pipe_step_resident.argtypes = [vp, vp, C.c_long, vp, vp, C.c_long, vp]
    L.wmx_chain_set_stages.restype = i
    L.wmx_chain_set_stages.argtypes = [vp, C.c_uint, i]
    L.wmx_chain_stages.restype = C.c_uint
    L.wmx_chain_stages.argtypes = [vp]
    L.wmx_tick_set_stages.restype = i
    L.wmx_tick_set_stages.argtypes = [vp, C.c_uint, i]
    L.wmx_tick_create.restype = i
    L.wmx_tick_create.argtypes = [C.POINTER(vp), i, i, i, i, i, i, i, C.c_uint]
    L.wmx_tick_destroy.restype = i
    L.wmx_tick_destroy.argtypes = [vp]
    L.wmx_tick_play_ns.restype = i
    L.wmx_tick_play_ns.argtypes = [vp, i]
    L.wmx_tick_rw_test.restype = i
    L.wmx_tick_rw_test.argtypes = [vp, i]
    L.wmx_tick_set_play_correct.restype = i
    L.wmx_tick_set_play_correct.argtypes = [vp, C.c_uint32]
    L.wmx_tick_package_samples.restype = i
    L.wmx_tick_package_samples.argtypes = [vp]
    L.wmx_tick_load.restype = i
    L.wmx_tick_load.argtypes = [vp, vp, C.c_uint32, i, i, i, i, C.c_long, C.c_long, i, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), vp]
    L.wmx_tick_run.restype = i
    L.wmx_tick_run.argtypes = [vp, vp, C.c_long, vp, C.c_long, vp, C.c_long, C.c_uint32, C.POINTER(C.c_uint32), vp]
    L.wmx_tick_play.restype = i
    L.wmx_tick_play.argtypes = [vp, vp, C.c_long, vp]
    L.wmx_tick_record.restype = i
    L.wmx_tick_record.argtypes = [vp, vp, C.c_long, vp, C.c_long, C.c_uint32, C.POINTER(C.c_uint32), vp]
    for name in ("wmx_tick_mix", "wmx_tick_chain", "wmx_tick_fifo", "wmx_tick_far"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp]
    L.wmx_agc_set_gain_streams.restype = i
    L.wmx_agc_set_gain_streams.argtypes = [vp, vp, i, i, vp]
    L.wmx_agc_reset_streams_gain.restype = i
    L.wmx_agc_reset_streams_gain.argtypes = [vp, vp, i, i, vp]
    L.wmx_agc_stream_gain.restype = i
    L.wmx_agc_stream_gain.argtypes = [vp, i]
    L.wmx_chain_reset_streams_gain.restype = i
    L.wmx_chain_reset_streams_gain.argtypes = [vp, vp, i, i, i, vp]
    L.wmx_chain_set_agc_gain_streams.restype = i
    L.wmx_chain_set_agc_gain_streams.argtypes = [vp, vp, i, i, vp]
    L.wmx_agc_packet_samples.restype = i
    L.wmx_agc_packet_samples.argtypes = [vp]
    L.wmx_agc_gain_table.restype = i
    L.wmx_agc_gain_table.argtypes = [vp, vp]
    L.wmx_agc_process.restype = i
    L.wmx_agc_process.argtypes = [vp, vp, vp, i, C.c_long, C.c_long, vp]
    L.wmx_aec_create.restype = i
    L.wmx_aec_create.argtypes = [C.POINTER(vp), i, i, i, i]
    L.wmx_aec_destroy.restype = i
    L.wmx_aec_destroy.argtypes = [vp]
    L.wmx_aec_packet_samples.restype = i
    L.wmx_aec_packet_samples.argtypes = [vp]
    L.wmx_aec_state_words.restype = i
    L.wmx_aec_state_words.argtypes = [vp]
    L.wmx_aec_export_state.restype = i
    L.wmx_aec_export_state.argtypes = [vp, i, vp]
    L.wmx_aec_run.restype = i
    L.wmx_aec_run.argtypes = [vp, i, vp, C.c_long, vp, vp, i, C.c_long, C.c_long, i, vp]
    L.wmx_aec_create_groups.restype = i
    L.wmx_aec_create_groups.argtypes = [C.POINTER(vp), i, i, i, i, i, vp]
    L.wmx_aec_run_groups.restype = i
    L.wmx_aec_run_groups.argtypes = [vp, i, vp, C.c_long, C.c_long, vp, vp, i, C.c_long, C.c_long, i, vp]
    # per-stream lifetime (include/wmix_amd.h "per-stream lifetime inside a batch")
    for m in ("ns", "nsx", "agc", "vad"):
        f = getattr(L, "wmx_%s_reset_streams" % m)
        f.restype = i
        f.argtypes = [vp, vp, i, vp]
    for m in ("aec", "aecm", "chain"):
        f = getattr(L, "wmx_%s_reset_streams" % m)
        f.restype = i
        f.argtypes = [vp, vp, i, i, vp]
        f = getattr(L, "wmx_%s_reset_cohort" % m)
        f.restype = i
        f.argtypes = [vp, i, vp]
    for m in ("ns", "nsx", "agc", "vad", "aec", "aecm", "chain"):
        f = getattr(L, "wmx_%s_set_active" % m)
        f.restype = i
        f.argtypes = [vp, vp, vp]
    for m in ("aec", "aecm"):
        f = getattr(L, "wmx_%s_cohorts" % m)
        f.restype = i
        f.argtypes = [vp]
        f = getattr(L, "wmx_%s_run_cohorts" % m)
        f.restype = i
        f.argtypes = [vp, i, vp, C.c_long, C.c_long, vp, vp, i, C.c_long, C.c_long, vp, vp, vp, vp]
    # stream / cohort migration
    for m in ("ns", "nsx", "agc", "vad", "aec", "aecm", "chain"):
        f = getattr(L, "wmx_%s_stream_state_bytes" % m)
        f.restype = i
        f.argtypes = [vp]
        f = getattr(L, "wmx_%s_export_stream" % m)
        f.restype = i
        f.argtypes = [vp, i, vp]
        f = getattr(L, "wmx_%s_import_stream" % m)
        f.restype = i
        f.argtypes = [vp, i, vp] + ([i] if m in ("aec", "aecm", "chain") else [])
    for m in ("aec", "aecm"):
        f = getattr(L, "wmx_%s_cohort_state_bytes" % m)
        f.restype = i
        f.argtypes = [vp]
        f = getattr(L, "wmx_%s_export_cohort" % m)
        f.restype = i
        f.argtypes = [vp, i, vp]
        f = getattr(L, "wmx_%s_import_cohort" % m)
        f.restype = i
        f.argtypes = [vp, i, vp]
    for name in ("wmx_aec_add_cohort", "wmx_aecm_add_cohort", "wmx_chain_add_cohort"):
        f = getattr(L, name)
        f.restype = i
        f.argtypes = [vp, C.POINTER(i), vp]
    for name in ("wmx_aec_retire_cohort", "wmx_aecm_retire_cohort", "wmx_chain_retire_cohort"):
        f = getattr(L, name)
        f.restype = i
        f.argtypes = [vp, i]
    L.wmx_chain_cohorts.restype = i
    L.wmx_chain_cohorts.argtypes = [vp]
    if hasattr(L, "wmx_aec_coalesce"):  # (a WMIX_AMD_LIB variant built from an older tree, for A/B runs, may not have them)
        L.wmx_aec_live_cohorts.restype = i
        L.wmx_aec_live_cohorts.argtypes = [vp]
        L.wmx_aecm_live_cohorts.restype = i
        L.wmx_aecm_live_cohorts.argtypes = [vp]
        for name in ("wmx_aec_cohort_key", "wmx_aecm_cohort_key"):
            f = getattr(L, name)
            f.restype = i
            f.argtypes = [vp, i, vp]
        for name in ("wmx_aec_coalesce", "wmx_aecm_coalesce", "wmx_chain_coalesce"):
            f = getattr(L, name)
            f.restype = i
            f.argtypes = [vp, i, vp, vp, i, C.POINTER(i), vp]
    L.wmx_aecm_create_cohorts.restype = i
    L.wmx_aecm_create_cohorts.argtypes = [C.POINTER(vp), i, i, i, i, i]
    L.wmx_aec_set_timing.restype = i
    L.wmx_aec_set_timing.argtypes = [vp, i]
    L.wmx_aec_timing.restype = i
    L.wmx_aec_timing.argtypes = [vp, C.POINTER(i), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.wmx_aec_host_ctl.restype = i
    L.wmx_aec_host_ctl.argtypes = [vp, C.POINTER(C.c_long), C.POINTER(C.c_double)]
    L.wmx_chain_create.restype = i
    L.wmx_chain_create.argtypes = [C.POINTER(vp), i, i, i, i, i, C.c_uint, i]
    L.wmx_chain_destroy.restype = i
    L.wmx_chain_destroy.argtypes = [vp]
    L.wmx_chain_process.restype = i
    L.wmx_chain_process.argtypes = [vp, vp, C.c_long, vp, vp, i, C.c_long, C.c_long, vp, vp, vp, vp]
    for m in ("ns", "aec", "agc", "vad", "nsx", "aecm"):
        f = getattr(L, "wmx_chain_%s" % m)
        f.restype = vp
        f.argtypes = [vp]
    u32 = C.c_uint32
    L.wmx_pcm_zoom.restype = i
    L.wmx_pcm_zoom.argtypes = [i, i, vp, u32, i, i, vp, u32, C.c_long, C.c_long, i, C.POINTER(u32), vp]
    L.wmx_handle_device.restype = i
    L.wmx_handle_device.argtypes = [vp]
    L.wmx_mfft.restype = i
    L.wmx_mfft.argtypes = [i, i, C.c_uint, vp, vp, vp, vp, vp, vp, vp]
    L.wmx_mfft_stream.restype = i
    L.wmx_mfft_stream.argtypes = [i, vp, C.c_uint, vp, C.c_uint, vp, vp, vp]
    L.wmx_rtp_create.restype = i
    L.wmx_rtp_create.argtypes = [C.POINTER(vp), i, i]
    L.wmx_rtp_destroy.restype = i
    L.wmx_rtp_destroy.argtypes = [vp]
    L.wmx_rtp_egress.restype = i
    L.wmx_rtp_egress.argtypes = [vp, i, i, vp, C.c_uint32, C.c_long, i, i, vp, C.c_long, C.POINTER(C.c_uint32), vp]
    L.wmx_rtp_ingest.restype = i
    L.wmx_rtp_ingest.argtypes = [i, vp, C.c_long, vp, C.c_long, vp, vp, vp]
    L.wmx_rtp_export.restype = i
    L.wmx_rtp_export.argtypes = [vp, i, C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)]
    L.wmx_debug_fft.restype = i
    L.wmx_debug_fft.argtypes = [i, i, vp, vp, vp]
    L.wmx_debug_pow.restype = i
    L.wmx_debug_pow.argtypes = [vp, vp, vp, C.c_size_t]
    L.wmx_debug_ns_libm.restype = i
    L.wmx_debug_ns_libm.argtypes = [i, vp, vp, C.c_size_t]
    L.wmx_debug_pow_device.restype = i
    L.wmx_debug_pow_device.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.wmx_debug_div.restype = i
    L.wmx_debug_div.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    L.wmx_debug_div_host.restype = i
    L.wmx_debug_div_host.argtypes = [vp, vp, vp, C.c_size_t]
    L.wmx_pkgfifo_create.restype = i
    L.wmx_pkgfifo_create.argtypes = [C.POINTER(vp), i, i, i, i, i]
    L.wmx_pkgfifo_destroy.restype = i
    L.wmx_pkgfifo_destroy.argtypes = [vp]
    L.wmx_pkgfifo_add.restype = i
    L.wmx_pkgfifo_add.argtypes = [vp, vp, C.c_long, vp]
    L.wmx_pkgfifo_get.restype = i
    L.wmx_pkgfifo_get.argtypes = [vp, vp, C.c_long, i, vp]
    L.wmx_mix_create.restype = i
    L.wmx_mix_create.argtypes = [C.POINTER(vp), i, i, i]
    L.wmx_mix_destroy.restype = i
    L.wmx_mix_destroy.argtypes = [vp]
    L.wmx_mix_set.restype = i
    L.wmx_mix_set.argtypes = [vp, u32, u32, i]
    L.wmx_mix_set_play_correct.restype = i
    L.wmx_mix_set_play_correct.argtypes = [vp, u32]
    L.wmx_mix_ring_bytes.restype = i
    L.wmx_mix_ring_bytes.argtypes = [vp]
    L.wmx_mix_load.restype = i
    L.wmx_mix_load.argtypes = [vp, vp, u32, i, i, i, i, C.c_long, C.c_long, i, C.POINTER(u32), C.POINTER(u32), vp]
    L.wmx_mix_drain.restype = i
    L.wmx_mix_drain.argtypes = [vp, vp, u32, C.c_long, vp]
    L.wmx_mix_export.restype = i
    L.wmx_mix_export.argtypes = [vp, i, vp, C.POINTER(u32), C.POINTER(u32)]
    L.wmix_len_of_out.restype = u32
    L.wmix_len_of_out.argtypes = [C.c_uint8, C.c_uint16, u32, C.c_uint8, C.c_uint16]
    L.wmix_len_of_in.restype = u32
    L.wmix_len_of_in.argtypes = [C.c_uint8, C.c_uint16, C.c_uint8, C.c_uint16, u32]
    L.wmix_pcm_zoom.restype = u32
    L.wmix_pcm_zoom.argtypes = [C.c_uint8, C.c_uint16, vp, u32, C.c_uint8, C.c_uint16, vp]
    L.aec_init.restype = vp
    L.aec_init.argtypes = [i, i, i, vp]
    L.aec_setFrameFar.restype = i
    L.aec_setFrameFar.argtypes = [vp, vp, i]
    L.aec_process.restype = i
    L.aec_process.argtypes = [vp, vp, vp, i, i]
    L.aec_process2.restype = i
    L.aec_process2.argtypes = [vp, vp, vp, vp, i, i]
    L.aec_release.restype = None
    L.aec_release.argtypes = [vp]
    L.vad_init.restype = vp
    L.vad_init.argtypes = [i, i, i, vp]
    L.vad_process.restype = None
    L.vad_process.argtypes = [vp, vp, i]
    L.vad_release.restype = None
    L.vad_release.argtypes = [vp]
    L.agc_init.restype = vp
    L.agc_init.argtypes = [i, i, i, i, vp]
    L.agc_process.restype = i
    L.agc_process.argtypes = [vp, vp, vp, i]
    L.agc_addition.restype = None
    L.agc_addition.argtypes = [vp, C.c_ubyte]
    L.agc_release.restype = None
    L.agc_release.argtypes = [vp]
    L.ns_init.restype = vp
    L.ns_init.argtypes = [i, i, vp]
    L.ns_process.restype = None
    L.ns_process.argtypes = [vp, vp, vp, i]
    L.ns_release.restype = None
    L.ns_release.argtypes = [vp]
    for name in ("PCM2G711a", "PCM2G711u", "G711a2PCM", "G711u2PCM"):
        f = getattr(L, name)
        f.restype = i
        f.argtypes = [vp, vp, i, i]
    for name in ("g711a_encode", "g711u_encode", "g711a_decode", "g711u_decode"):
        f = getattr(L, name)
        f.restype = i
        f.argtypes = [vp, vp, i]
    for name in ("linear2alaw", "linear2ulaw"):
        f = getattr(L, name)
        f.restype = C.c_ubyte
        f.argtypes = [i]
