"""ctypes binding of llzlab_amd/libllzfilter_hip.so -- the C-ABI declared in include/*.h.

This is the only way Python reaches the kernels: plain pointers and sizes, no torch types.  There is no CPU
fallback: if the library (hipcc build, gfx950) is missing, loading fails loudly.
"""
import ctypes as C
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# LLZ_LIB: load another build of the same library (ablation builds under profiles/; never a fallback)
LIB_PATH = os.environ.get("LLZ_LIB") or os.path.join(HERE, "libllzfilter_hip.so")
INCLUDE_DIR = os.path.join(ROOT, "include")

BAD_HANDLE = C.c_ulong(-1).value
HAMMING, BLACKMAN, KAISER = 0, 1, 2
FIR_ALGO_AUTO, FIR_ALGO_TIME, FIR_ALGO_OVERLAP_SAVE, FIR_ALGO_TIME_MFMA, FIR_ALGO_OVERLAP_SAVE_2048 = 0, 1, 2, 3, 4
FIR_ALGO_OVERLAP_SAVE_4096 = 5
FIR_ALGO_OVERLAP_SAVE_8192 = 6
OVERLAP_HIGH, OVERLAP_LOW = 0, 1      # llz_asmodel.h: 3/4 and 1/2 overlap
MDCT_ORIGIN, MDCT_FFT, MDCT_FFT4 = 0, 1, 2
MDCT_SINE, MDCT_KBD = 0, 1
PCM_F32, PCM_I16, PCM_I16_FAST = 0, 1, 2

_lib = None


class LlzError(RuntimeError):
    pass


class WavInfo(C.Structure):                  # llz_wav_info (include/llz_pcm.h)
    _fields_ = [("format", C.c_int), ("channels", C.c_int), ("samplerate", C.c_long), ("bytes_per_sample", C.c_int),
                ("block_align", C.c_int), ("frames", C.c_long), ("data_offset", C.c_long)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _t in self._fields_}


def build(force=False, extra_hipflags=""):
    """Compile every HIP kernel for gfx950 and link the shared library in-tree (hipcc cross-compiles on CPU)."""
    cmd = ["make", "-s", "-C", os.path.join(HERE, "csrc"), "-j4"]
    if extra_hipflags:
        cmd.append("EXTRA_HIPFLAGS=" + extra_hipflags)
    if force:
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "csrc"), "clean"])
    subprocess.check_call(cmd)
    if not os.path.exists(LIB_PATH):
        raise LlzError("build did not produce " + LIB_PATH)


def declared_symbols():
    """Every function name declared in include/*.h (used by the export test and to bind lazily)."""
    names = []
    for fn in sorted(os.listdir(INCLUDE_DIR)):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(INCLUDE_DIR, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b(llz_[a-z0-9_]+)\s*\(", text):
            if m.group(1) not in names:
                names.append(m.group(1))
    return names


def lib():
    """The loaded library with argtypes set. Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LlzError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    ul, vp, i, d, lng = C.c_ulong, C.c_void_p, C.c_int, C.c_double, C.c_long
    dp = C.POINTER(C.c_double)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    # llz_hip.h
    sig("llz_hip_last_error", C.c_char_p)
    sig("llz_hip_device_count", i)
    sig("llz_hip_set_device", i, i)
    sig("llz_hip_get_device", i)
    sig("llz_hip_synchronize", i, vp)
    sig("llz_hip_tune", i, C.c_char_p, i)
    sig("llz_hip_tune_name", C.c_char_p, i)
    sig("llz_hip_malloc", vp, C.c_size_t)
    sig("llz_hip_free", None, vp)
    sig("llz_hip_upload", i, vp, vp, C.c_size_t)
    sig("llz_hip_download", i, vp, vp, C.c_size_t)
    sig("llz_hip_is_device_ptr", i, vp)
    sig("llz_hip_synth_f32", i, vp, i, lng, lng, C.c_uint, i, vp)
    sig("llz_hip_synth_i16", i, vp, i, lng, lng, C.c_uint, i, vp)
    sig("llz_hip_timer_new", vp)
    sig("llz_hip_timer_start", i, vp, vp)
    sig("llz_hip_timer_stop", i, vp, vp)
    sig("llz_hip_timer_ms", d, vp)
    sig("llz_hip_timer_free", None, vp)
    # llz_fir.h part 1
    sig("llz_fir_filter_lpf_init", ul, i, i, d, i)
    sig("llz_fir_filter_hpf_init", ul, i, i, d, i)
    sig("llz_fir_filter_bandpass_init", ul, i, i, d, d, i)
    sig("llz_fir_filter_bandstop_init", ul, i, i, d, d, i)
    sig("llz_fir_filter_uninit", None, ul)
    sig("llz_fir_filter", i, ul, dp, dp, i)
    sig("llz_fir_filter_flush", i, ul, dp)
    for n in ("llz_hamming", "llz_blackman", "llz_kaiser"):
        sig(n, i, dp, i)
    sig("llz_kaiser_beta", i, dp, i, d)
    sig("llz_kaiser_atten2beta", d, d)
    sig("llz_hamming_cof_num", i, d)
    sig("llz_blackman_cof_num", i, d)
    sig("llz_kaiser_cof_num", i, d, d)
    pdp = C.POINTER(dp)
    sig("llz_fir_lpf_cof", i, pdp, i, d, i)
    sig("llz_fir_hpf_cof", i, pdp, i, d, i)
    sig("llz_fir_bandpass_cof", i, pdp, i, d, d, i)
    sig("llz_fir_bandstop_cof", i, pdp, i, d, d, i)
    sig("llz_conv", d, dp, dp, i)
    # llz_fir.h part 2
    sig("llz_fir_filter_mc_init", ul, i, i, vp, i, i)
    sig("llz_fir_filter_mc_init_f64taps", ul, i, i, vp, i, i)
    sig("llz_fir_filter_mc_lpf_init", ul, i, i, i, d, i)
    sig("llz_fir_filter_mc_hpf_init", ul, i, i, i, d, i)
    sig("llz_fir_filter_mc_bandpass_init", ul, i, i, i, d, d, i)
    sig("llz_fir_filter_mc_bandstop_init", ul, i, i, i, d, d, i)
    sig("llz_fir_filter_mc_uninit", None, ul)
    sig("llz_fir_filter_mc", i, ul, vp, vp, i)
    sig("llz_fir_filter_mc_flush", i, ul, vp)
    sig("llz_fir_filter_mc_flt_len", i, ul)
    sig("llz_fir_filter_mc_algo", i, ul)
    sig("llz_fir_filter_mc_set_stream", i, ul, vp)
    # llz_iir.h
    sig("llz_iir_filter_init", ul, i, dp, i, dp)
    sig("llz_iir_filter_uninit", None, ul)
    sig("llz_iir_filter", i, ul, dp, dp, i)
    sig("llz_iir_filter_flush", i, ul, dp)
    sig("llz_iir_cascade_mc_init", ul, i, i, vp)
    sig("llz_iir_cascade_mc_uninit", None, ul)
    sig("llz_iir_cascade_mc", i, ul, vp, vp, i)
    sig("llz_iir_cascade_mc_set_stream", i, ul, vp)
    sig("llz_iir_cascade_mc_precision", i, ul)
    # llz_resample.h
    sig("llz_decimate_init", ul, i, d, i)
    sig("llz_decimate_uninit", None, ul)
    sig("llz_interp_init", ul, i, d, i)
    sig("llz_interp_uninit", None, ul)
    sig("llz_resample_filter_init", ul, i, i, d, i)
    sig("llz_resample_filter_uninit", None, ul)
    sig("llz_get_resample_framelen_bytes", i, ul)
    ip = C.POINTER(C.c_int)
    for n in ("llz_decimate", "llz_interp", "llz_resample"):
        sig(n, i, ul, vp, i, vp, ip)
    sig("llz_resample_mc_init", ul, i, i, i, d, i, i)
    sig("llz_resample_mc_uninit", None, ul)
    sig("llz_resample_mc_sub_len", i, ul)
    sig("llz_resample_mc_out_len", lng, ul, lng)
    sig("llz_resample_mc", lng, ul, vp, lng, vp)
    sig("llz_resample_mc_set_stream", i, ul, vp)
    sig("llz_resample_mc_get_matrix", i, ul, vp, i)
    sig("llz_resample_mc_set_matrix", i, ul, vp, i)
    # llz_fft.h / llz_fft_fixed.h
    sig("llz_fft_init", ul, i)
    sig("llz_fft_uninit", None, ul)
    sig("llz_fft", None, ul, dp)
    sig("llz_ifft", None, ul, dp)
    sig("llz_fft_batch_init", ul, i)
    sig("llz_fft_batch_uninit", None, ul)
    sig("llz_fft_batch", i, ul, vp, i)
    sig("llz_ifft_batch", i, ul, vp, i)
    sig("llz_fft_batch_set_stream", i, ul, vp)
    sig("llz_fft_fixed_init", ul, i)
    sig("llz_fft_fixed_uninit", None, ul)
    sig("llz_fft_fixed", None, ul, ip)
    sig("llz_ifft_fixed", None, ul, ip)
    sig("llz_fft_fixed_batch", i, ul, vp, i)
    sig("llz_ifft_fixed_batch", i, ul, vp, i)
    sig("llz_fft_fixed_set_stream", i, ul, vp)
    # llz_corr.h
    sig("llz_autocorr", None, dp, i, i, dp)
    sig("llz_crosscorr", None, dp, dp, i, i, dp)
    sig("llz_corr_cof", d, dp, dp, i)
    sig("llz_autocorr_fast_init", ul, i)
    sig("llz_autocorr_fast_uninit", None, ul)
    sig("llz_autocorr_fast", None, ul, dp, i, i, dp)
    sig("llz_autocorr_mc", i, vp, vp, i, i, i, vp)
    sig("llz_autocorr_fast_mc_init", ul, i, i)
    sig("llz_autocorr_fast_mc_uninit", None, ul)
    sig("llz_autocorr_fast_mc", i, ul, vp, vp, i)
    sig("llz_autocorr_fast_mc_set_stream", i, ul, vp)
    # llz_asmodel.h
    for n in ("llz_analysis_fft_init", "llz_synthesis_fft_init"):
        sig(n, ul, i, i, i)
    sig("llz_analysis_fft_uninit", None, ul)
    sig("llz_synthesis_fft_uninit", None, ul)
    sig("llz_analysis_fft", None, ul, dp, dp, dp)
    sig("llz_synthesis_fft", None, ul, dp, dp, dp)
    sig("llz_stft_mc_init", ul, i, i, i, i)
    sig("llz_stft_mc_uninit", None, ul)
    sig("llz_stft_mc_bins", i, ul)
    sig("llz_stft_mc_set_stream", i, ul, vp)
    sig("llz_stft_mc_analysis", i, ul, vp, vp, vp, i)
    sig("llz_stft_mc_synthesis", i, ul, vp, vp, vp, i)
    sig("llz_iir_mc_init", ul, i, i, dp, i, dp)
    sig("llz_iir_mc_uninit", None, ul)
    sig("llz_iir_mc", i, ul, vp, vp, i)
    sig("llz_iir_mc_flush", i, ul, vp)
    sig("llz_iir_mc_set_stream", i, ul, vp)
    sig("llz_mdct_frames_mc_init", ul, i, i, i)
    sig("llz_mdct_frames_mc_uninit", None, ul)
    sig("llz_mdct_frames_mc_set_stream", i, ul, vp)
    sig("llz_mdct_frames_mc_analysis", i, ul, vp, vp, i)
    sig("llz_mdct_frames_mc_synthesis", i, ul, vp, vp, i)
    for n in ("llz_analysis_mdct_init", "llz_synthesis_mdct_init"):
        sig(n, ul, i, i)
    sig("llz_analysis_mdct_uninit", None, ul)
    sig("llz_synthesis_mdct_uninit", None, ul)
    sig("llz_analysis_mdct", None, ul, dp, dp)
    sig("llz_synthesis_mdct", None, ul, dp, dp)
    # llz_mdct.h
    sig("llz_mdct_init", ul, i, i)
    sig("llz_mdct_uninit", None, ul)
    sig("llz_mdct", None, ul, dp, dp)
    sig("llz_imdct", None, ul, dp, dp)
    sig("llz_mdct_sine", i, dp, i)
    sig("llz_mdct_kbd", i, dp, i, d)
    sig("llz_mdct_batch_init", ul, i)
    sig("llz_mdct_batch_uninit", None, ul)
    sig("llz_mdct_batch_set_stream", i, ul, vp)
    sig("llz_mdct_batch", i, ul, vp, vp, i)
    sig("llz_imdct_batch", i, ul, vp, vp, i)
    # llz_mdct_fixed.h
    ip = C.POINTER(C.c_int)
    sig("llz_mdct_fixed_init", ul, i, i)
    sig("llz_mdct_fixed_uninit", None, ul)
    sig("llz_mdct_fixed", None, ul, ip, ip)
    sig("llz_imdct_fixed", None, ul, ip, ip)
    sig("llz_mdct_fixed_batch", i, ul, vp, vp, i)
    sig("llz_imdct_fixed_batch", i, ul, vp, vp, i)
    sig("llz_mdct_fixed_set_stream", i, ul, vp)
    sig("llz_mdct_fixed_len", i, ul)
    # llz_shard.h
    pp = C.POINTER(C.c_void_p)
    sig("llz_shard_range", i, i, i, i, ip, ip)
    sig("llz_fir_filter_mc_sharded_init", ul, i, i, vp, i, i, ip, i)
    sig("llz_fir_filter_mc_sharded", i, ul, pp, pp, i)
    sig("llz_fir_filter_mc_sharded_flush", i, ul, pp)
    sig("llz_iir_cascade_mc_sharded_init", ul, i, i, vp, ip, i)
    sig("llz_iir_cascade_mc_sharded", i, ul, pp, pp, i)
    sig("llz_resample_mc_sharded_init", ul, i, i, i, d, i, i, ip, i)
    sig("llz_resample_mc_sharded", lng, ul, pp, lng, pp)
    sig("llz_sharded_uninit", None, ul)
    sig("llz_sharded_count", i, ul)
    sig("llz_sharded_rccl_ranks", i, ul)
    sig("llz_sharded_shard", i, ul, i, ip, ip, ip)
    sig("llz_sharded_stream", vp, ul, i)
    sig("llz_sharded_sub", ul, ul, i)
    sig("llz_sharded_synchronize", i, ul)
    sig("llz_sharded_timer_start", i, ul)
    sig("llz_sharded_timer_stop", i, ul)
    sig("llz_sharded_timer_ms", d, ul, dp)
    # llz_pcm.h
    sig("llz_pcm_deinterleave_i16_f32", i, vp, vp, i, lng, C.c_float, vp)
    sig("llz_pcm_interleave_f32_i16", i, vp, vp, i, lng, C.c_float, vp)
    sig("llz_wav_parse", i, C.c_char_p, lng, vp)
    sig("llz_wav_write_header", i, vp, vp)
    sig("llz_wav_ingest_f32", lng, C.c_char_p, lng, vp, lng, vp, vp)
    _lib = L
    # measurement harness only: LLZ_TUNE="name=value,..." is applied HERE (Python), through the public override call;
    # the C library itself never reads the environment
    for item in filter(None, os.environ.get("LLZ_TUNE", "").split(",")):
        name, _, val = item.partition("=")
        if L.llz_hip_tune(name.strip().encode(), int(val)) != 0:
            raise LlzError("LLZ_TUNE: unknown override " + name)
    return L


def tune(name, value):
    """Override (value >= 0) or clear (value < 0) one of the library's own kernel-form choices: tests and A/B runs only."""
    if lib().llz_hip_tune(name.encode(), int(value)) != 0:
        raise LlzError("unknown override " + name)


class tuned:
    """with capi.tuned(iir_segs=4): ...  -- overrides set on entry and cleared on exit."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            tune(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            tune(k, -1)
        return False


def last_error():
    return lib().llz_hip_last_error().decode("utf-8", "replace")


def check(rc, what):
    """Negative return codes from the C ABI become exceptions carrying the library's own message."""
    if rc is None:
        return rc
    if isinstance(rc, int) and rc < 0:
        raise LlzError(f"{what} failed ({rc}): {last_error()}")
    return rc


def check_handle(h, what):
    if h == BAD_HANDLE or h == 0:
        raise LlzError(f"{what} failed: {last_error()}")
    return h
