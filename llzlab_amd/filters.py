"""Python mirror of the reference's handle interface (init / process / flush / uninit) over the C ABI.

Names and argument meaning follow reference libllzfilter/llz_{fir,iir,resample,fft,fft_fixed}.h; the *MC classes
are the multi-channel float32/int16 batch extension.  Device buffers are torch tensors (used for memory and
streams only: `tensor.data_ptr()` crosses the boundary as a plain pointer); numpy arrays are accepted as host
memory.  Nothing here computes: every sample goes through libllzfilter_hip.so.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import (BAD_HANDLE, BLACKMAN, FIR_ALGO_AUTO, FIR_ALGO_OVERLAP_SAVE, FIR_ALGO_OVERLAP_SAVE_2048, FIR_ALGO_OVERLAP_SAVE_4096, FIR_ALGO_OVERLAP_SAVE_8192, FIR_ALGO_TIME,  # noqa: F401
                   FIR_ALGO_TIME_MFMA, HAMMING, KAISER,
                   PCM_F32, PCM_I16, PCM_I16_FAST, LlzError, check, check_handle)


def _ptr(buf):
    """Plain address of a torch tensor (any device) or a numpy array; the library sorts host from device."""
    if isinstance(buf, np.ndarray):
        if not buf.flags["C_CONTIGUOUS"]:
            raise LlzError("numpy buffer must be C-contiguous")
        return buf.ctypes.data
    if hasattr(buf, "data_ptr"):
        if not buf.is_contiguous():
            raise LlzError("tensor must be contiguous")
        return buf.data_ptr()
    raise LlzError(f"unsupported buffer type {type(buf)}")


def _typed(buf, dtype, numel, what):
    """address of `buf` after checking that it holds exactly `numel` elements of `dtype` ("float32", "int16", ...): the C
    ABI takes plain pointers, so a wrong dtype or a short buffer would be read / written past its end"""
    name = str(buf.dtype).replace("torch.", "")
    size = buf.numel() if hasattr(buf, "numel") else buf.size
    if name != dtype or size != numel:
        raise LlzError(f"{what}: expected {numel} x {dtype}, got {size} x {name}")
    return _ptr(buf)


def _stream_ptr(stream):
    if stream is None:
        return None
    return getattr(stream, "cuda_stream", stream)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


_dp = C.POINTER(C.c_double)


# ------------------------------------------------------------------------------------------ design (host C)
def fir_design(kind, n, fc1, fc2=0.0, win=HAMMING):
    """kind: 'lpf' | 'hpf' | 'bandpass' | 'bandstop' -> float64 taps from llz_fir_*_cof (host C code)."""
    L = capi.lib()
    hp = _dp()
    if kind == "lpf":
        m = L.llz_fir_lpf_cof(C.byref(hp), n, fc1, win)
    elif kind == "hpf":
        m = L.llz_fir_hpf_cof(C.byref(hp), n, fc1, win)
    elif kind == "bandpass":
        m = L.llz_fir_bandpass_cof(C.byref(hp), n, fc1, fc2, win)
    elif kind == "bandstop":
        m = L.llz_fir_bandstop_cof(C.byref(hp), n, fc1, fc2, win)
    else:
        raise LlzError("unknown filter kind " + kind)
    if m < 1:
        raise LlzError("tap design failed")
    taps = np.ctypeslib.as_array(hp, shape=(m,)).copy()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(hp)                       # the caller owns *h (reference llz_fir.h:82-86)
    return taps


def window(win, n, beta=None):
    L = capi.lib()
    w = np.zeros(n)
    p = w.ctypes.data_as(_dp)
    if beta is not None:
        L.llz_kaiser_beta(p, n, beta)
    else:
        (L.llz_hamming, L.llz_blackman, L.llz_kaiser)[win](p, n)
    return w


# ------------------------------------------------------------------------------------------ FIR
class FirFilter:
    """Single channel, double, host buffers: llz_fir_filter_{lpf,hpf,bandpass,bandstop}_init & co."""

    def __init__(self, kind, frame_len, flt_len, fc1, fc2=0.0, win=HAMMING):
        L = self._L = capi.lib()
        if kind == "lpf":
            h = L.llz_fir_filter_lpf_init(frame_len, flt_len, fc1, win)
        elif kind == "hpf":
            h = L.llz_fir_filter_hpf_init(frame_len, flt_len, fc1, win)
        elif kind == "bandpass":
            h = L.llz_fir_filter_bandpass_init(frame_len, flt_len, fc1, fc2, win)
        elif kind == "bandstop":
            h = L.llz_fir_filter_bandstop_init(frame_len, flt_len, fc1, fc2, win)
        else:
            raise LlzError("unknown filter kind " + kind)
        self.handle = check_handle(h, "llz_fir_filter_*_init")
        self.frame_len = frame_len
        self.flt_len = flt_len if kind == "lpf" or flt_len & 1 else flt_len + 1

    def filter(self, x):
        x = _f64(x)
        y = np.zeros_like(x)
        check(self._L.llz_fir_filter(self.handle, x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), len(x)),
              "llz_fir_filter")
        return y

    def flush(self):
        y = np.zeros(max(self.flt_len - 1, 1))
        n = check(self._L.llz_fir_filter_flush(self.handle, y.ctypes.data_as(_dp)), "llz_fir_filter_flush")
        return y[:n]

    def close(self):
        if self.handle:
            self._L.llz_fir_filter_uninit(self.handle)
            self.handle = 0

    __del__ = close


class FirFilterMC:
    """channels x frame_len float32, planar; llz_fir_filter_mc_*."""

    def __init__(self, channels, frame_len, taps, algo=FIR_ALGO_AUTO, stream=None):
        self._L = capi.lib()
        taps = _f64(taps)
        self.handle = check_handle(
            self._L.llz_fir_filter_mc_init_f64taps(channels, frame_len, taps.ctypes.data, len(taps), algo),
            "llz_fir_filter_mc_init")
        self.channels, self.frame_len, self.flt_len = channels, frame_len, len(taps)
        self.algo = self._L.llz_fir_filter_mc_algo(self.handle)
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        check(self._L.llz_fir_filter_mc_set_stream(self.handle, _stream_ptr(stream)), "set_stream")

    def filter(self, x, out):
        """x, out: [channels, frame_len] float32 (torch device tensors or numpy). Returns out."""
        count = self.channels * self.frame_len
        check(self._L.llz_fir_filter_mc(self.handle, _typed(x, "float32", count, "FirFilterMC.filter x"),
                                        _typed(out, "float32", count, "FirFilterMC.filter out"), self.frame_len),
              "llz_fir_filter_mc")
        return out

    def flush(self, out):
        check(self._L.llz_fir_filter_mc_flush(self.handle, _typed(out, "float32", self.channels * (self.flt_len - 1),
                                                                  "FirFilterMC.flush out")), "llz_fir_filter_mc_flush")
        return out

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_fir_filter_mc_uninit(self.handle)
            self.handle = 0

    __del__ = close


# ------------------------------------------------------------------------------------------ IIR
class IirFilter:
    """Single channel direct form I, double, host buffers: llz_iir_filter_*."""

    def __init__(self, a, b):
        self._L = capi.lib()
        a = _f64(a)
        self.M = len(a) - 1
        if b is None:
            raise LlzError("pass b (zeros for the reference's NULL case)")
        b = _f64(b)
        self.N = len(b) - 1
        self.handle = check_handle(
            self._L.llz_iir_filter_init(self.M, a.ctypes.data_as(_dp), self.N, b.ctypes.data_as(_dp)),
            "llz_iir_filter_init")

    def filter(self, x):
        x = _f64(x)
        y = np.zeros_like(x)
        check(self._L.llz_iir_filter(self.handle, x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), len(x)),
              "llz_iir_filter")
        return y

    def flush(self):
        y = np.zeros(max(self.N, 1))
        n = check(self._L.llz_iir_filter_flush(self.handle, y.ctypes.data_as(_dp)), "llz_iir_filter_flush")
        return y[:n]

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_iir_filter_uninit(self.handle)
            self.handle = 0

    __del__ = close


class IirCascadeMC:
    """channels x n float32 through a fused cascade of biquads; coef rows {b0,b1,b2,a0,a1,a2}."""

    def __init__(self, channels, coef, stream=None):
        self._L = capi.lib()
        coef = _f64(coef).reshape(-1, 6)
        self.stages = coef.shape[0]
        self.channels = channels
        self.handle = check_handle(self._L.llz_iir_cascade_mc_init(channels, self.stages, coef.ctypes.data),
                                   "llz_iir_cascade_mc_init")
        self.precision = self._L.llz_iir_cascade_mc_precision(self.handle)      # 32 or 64 (arithmetic of the fast kernel)
        if stream is not None:
            check(self._L.llz_iir_cascade_mc_set_stream(self.handle, _stream_ptr(stream)), "set_stream")

    def filter(self, x, out):
        n = x.shape[-1]
        check(self._L.llz_iir_cascade_mc(self.handle, _typed(x, "float32", self.channels * n, "IirCascadeMC.filter x"),
                                         _typed(out, "float32", self.channels * n, "IirCascadeMC.filter out"), n),
              "llz_iir_cascade_mc")
        return out

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_iir_cascade_mc_uninit(self.handle)
            self.handle = 0

    __del__ = close


# ------------------------------------------------------------------------------------------ resample
class IirMC:
    """llz_iir_mc_* (include/llz_iir.h part 3): the general direct-form-I filter for many channels, float32 in / out."""

    def __init__(self, channels, a, b, stream=None):
        self._L = capi.lib()
        a, b = _f64(a), _f64(b)
        self.channels, self.M, self.N = channels, len(a) - 1, len(b) - 1
        self.handle = check_handle(self._L.llz_iir_mc_init(channels, self.M, a.ctypes.data_as(_dp), self.N, b.ctypes.data_as(_dp)),
                                   "llz_iir_mc_init")
        if stream is not None:
            check(self._L.llz_iir_mc_set_stream(self.handle, _stream_ptr(stream)), "llz_iir_mc_set_stream")

    def filter(self, x, out):
        n = int(x.shape[-1])
        check(self._L.llz_iir_mc(self.handle, _typed(x, "float32", self.channels * n, "x"), _typed(out, "float32", self.channels * n, "y"), n),
              "llz_iir_mc")
        return out

    def flush(self, out):
        """N more outputs per channel into out [channels][N] (N = 0: nothing to flush, out is not touched)"""
        if self.N == 0:
            return check(self._L.llz_iir_mc_flush(self.handle, _ptr(out)), "llz_iir_mc_flush")
        return check(self._L.llz_iir_mc_flush(self.handle, _typed(out, "float32", self.channels * self.N, "y")), "llz_iir_mc_flush")

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_iir_mc_uninit(self.handle)
            self.handle = 0

    __del__ = close


class _Resample1:
    """int16 single channel, host buffers: llz_{decimate,interp,resample}."""
    _init = _run = _uninit = None

    def _open(self, h):
        self.handle = check_handle(h, type(self).__name__ + " init")
        self.bytes_in = self._L.llz_get_resample_framelen_bytes(self.handle)

    def process(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        out = np.zeros(len(pcm) * 16 + 16, dtype=np.int16)
        ob = C.c_int(0)
        rc = self._run(self.handle, pcm.ctypes.data, 2 * len(pcm), out.ctypes.data, C.byref(ob))
        check(rc, type(self).__name__)
        return out[:ob.value // 2].copy()

    def close(self):
        if getattr(self, "handle", 0):
            self._uninit(self.handle)
            self.handle = 0

    __del__ = close


class Decimate(_Resample1):
    def __init__(self, M, gain=1.0, win=BLACKMAN):
        self._L = capi.lib()
        self._run, self._uninit = self._L.llz_decimate, self._L.llz_decimate_uninit
        self._open(self._L.llz_decimate_init(M, gain, win))


class Interp(_Resample1):
    def __init__(self, L_, gain=1.0, win=BLACKMAN):
        self._L = capi.lib()
        self._run, self._uninit = self._L.llz_interp, self._L.llz_interp_uninit
        self._open(self._L.llz_interp_init(L_, gain, win))


class Resample(_Resample1):
    def __init__(self, L_, M, gain=1.0, win=BLACKMAN):
        self._L = capi.lib()
        self._run, self._uninit = self._L.llz_resample, self._L.llz_resample_filter_uninit
        self._open(self._L.llz_resample_filter_init(L_, M, gain, win))


class ResampleMC:
    """channels x n_in -> channels x n_in*L/M; pcm_format PCM_F32 (float32), PCM_I16 (bit-exact int16) or PCM_I16_FAST
    (int16 within 1 LSB of the reference, matrix cores, decimators only)."""

    def __init__(self, channels, L_, M, gain=1.0, win=BLACKMAN, pcm_format=PCM_F32, stream=None):
        self._L = capi.lib()
        self.channels, self.L, self.M, self.fmt = channels, L_, M, pcm_format
        self.handle = check_handle(self._L.llz_resample_mc_init(channels, L_, M, gain, win, pcm_format),
                                   "llz_resample_mc_init")
        self.Q = self._L.llz_resample_mc_sub_len(self.handle)
        if stream is not None:
            check(self._L.llz_resample_mc_set_stream(self.handle, _stream_ptr(stream)), "set_stream")

    def out_len(self, n_in):
        return check(self._L.llz_resample_mc_out_len(self.handle, n_in), "llz_resample_mc_out_len")

    def matrix(self):
        m = np.zeros(self.L * self.Q)
        check(self._L.llz_resample_mc_get_matrix(self.handle, m.ctypes.data, m.size), "get_matrix")
        return m.reshape(self.L, self.Q)

    def set_matrix(self, m):
        m = _f64(m)
        check(self._L.llz_resample_mc_set_matrix(self.handle, m.ctypes.data, m.size), "set_matrix")

    def process(self, x, out):
        n_in = x.shape[-1]
        dt = "float32" if self.fmt == PCM_F32 else "int16"
        n_out = self.out_len(n_in)
        return check(self._L.llz_resample_mc(self.handle, _typed(x, dt, self.channels * n_in, "ResampleMC.process x"), n_in,
                                             _typed(out, dt, self.channels * n_out, "ResampleMC.process out")),
                     "llz_resample_mc")

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_resample_mc_uninit(self.handle)
            self.handle = 0

    __del__ = close


# ------------------------------------------------------------------------------------------ FFT
class Fft:
    """Reference API: complex128 host arrays, forward unscaled, inverse / N."""

    def __init__(self, size):
        self._L = capi.lib()
        self.size = size
        self.handle = check_handle(self._L.llz_fft_init(size), "llz_fft_init")

    def _run(self, fn, z):
        z = np.ascontiguousarray(z, dtype=np.complex128).copy()
        if len(z) != self.size:
            raise LlzError("length mismatch")
        fn(self.handle, z.view(np.float64).ctypes.data_as(_dp))
        return z

    def fft(self, z):
        return self._run(self._L.llz_fft, z)

    def ifft(self, z):
        return self._run(self._L.llz_ifft, z)

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_fft_uninit(self.handle)
            self.handle = 0

    __del__ = close


class FftBatch:
    """count x size complex64 transforms in place (device tensor of float32 pairs or numpy complex64)."""

    def __init__(self, size, stream=None):
        self._L = capi.lib()
        self.size = size
        self.handle = check_handle(self._L.llz_fft_batch_init(size), "llz_fft_batch_init")
        if stream is not None:
            check(self._L.llz_fft_batch_set_stream(self.handle, _stream_ptr(stream)), "set_stream")

    def fft(self, data, count):
        check(self._L.llz_fft_batch(self.handle, _ptr(data), count), "llz_fft_batch")
        return data

    def ifft(self, data, count):
        check(self._L.llz_ifft_batch(self.handle, _ptr(data), count), "llz_ifft_batch")
        return data

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_fft_batch_uninit(self.handle)
            self.handle = 0

    __del__ = close


class FftFixed:
    """int32 interleaved data, Q15 twiddles, bit-exact; single (host) and batched (device or host)."""

    def __init__(self, size, stream=None):
        self._L = capi.lib()
        self.size = size
        self.handle = check_handle(self._L.llz_fft_fixed_init(size), "llz_fft_fixed_init")
        if stream is not None:
            check(self._L.llz_fft_fixed_set_stream(self.handle, _stream_ptr(stream)), "set_stream")

    def fft(self, q):
        q = np.ascontiguousarray(q, dtype=np.int32).copy()
        self._L.llz_fft_fixed(self.handle, q.ctypes.data_as(C.POINTER(C.c_int)))
        return q

    def ifft(self, q):
        q = np.ascontiguousarray(q, dtype=np.int32).copy()
        self._L.llz_ifft_fixed(self.handle, q.ctypes.data_as(C.POINTER(C.c_int)))
        return q

    def fft_batch(self, data, count):
        check(self._L.llz_fft_fixed_batch(self.handle, _ptr(data), count), "llz_fft_fixed_batch")
        return data

    def ifft_batch(self, data, count):
        check(self._L.llz_ifft_fixed_batch(self.handle, _ptr(data), count), "llz_ifft_fixed_batch")
        return data

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_fft_fixed_uninit(self.handle)
            self.handle = 0

    __del__ = close


# ------------------------------------------------------------------------------------------ correlation
def autocorr(x, p):
    """llz_autocorr: host float64, exact."""
    x = _f64(x)
    r = np.zeros(p + 1)
    capi.lib().llz_autocorr(x.ctypes.data_as(_dp), len(x), p, r.ctypes.data_as(_dp))
    return r


def crosscorr(x, y, p):
    x, y = _f64(x), _f64(y)
    r = np.zeros(p + 1)
    capi.lib().llz_crosscorr(x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), len(x), p, r.ctypes.data_as(_dp))
    return r


def corr_cof(a, b):
    a, b = _f64(a), _f64(b)
    return capi.lib().llz_corr_cof(a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), len(a))


def autocorr_fast(x, p):
    """llz_autocorr_fast_{init,uninit} around one call: host float64, exact (the reference's definition)."""
    L = capi.lib()
    x = _f64(x)
    h = check_handle(L.llz_autocorr_fast_init(len(x)), "llz_autocorr_fast_init")
    r = np.zeros(p + 1)
    L.llz_autocorr_fast(h, x.ctypes.data_as(_dp), len(x), p, r.ctypes.data_as(_dp))
    L.llz_autocorr_fast_uninit(h)
    return r


def autocorr_mc(x, r, p, stream=None):
    """x: [frames, n] float32, r: [frames, p+1] float32 (device tensors or numpy)."""
    frames, n = x.shape
    check(capi.lib().llz_autocorr_mc(_ptr(x), _ptr(r), frames, n, p, _stream_ptr(stream)), "llz_autocorr_mc")
    return r


class AutocorrFastMC:
    def __init__(self, frames, n, stream=None):
        self._L = capi.lib()
        self.handle = check_handle(self._L.llz_autocorr_fast_mc_init(frames, n), "llz_autocorr_fast_mc_init")
        if stream is not None:
            check(self._L.llz_autocorr_fast_mc_set_stream(self.handle, _stream_ptr(stream)), "set_stream")

    def run(self, x, r, p):
        check(self._L.llz_autocorr_fast_mc(self.handle, _ptr(x), _ptr(r), p), "llz_autocorr_fast_mc")
        return r

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_autocorr_fast_mc_uninit(self.handle)
            self.handle = 0

    __del__ = close


# ------------------------------------------------------------------------------------------ windowed-FFT frames
class _FftFrames:
    """llz_analysis_fft_* / llz_synthesis_fft_* (llz_asmodel.h:36-42): one frame per call, host float64, exact."""
    _init = _uninit = None

    def __init__(self, overlap_hint, frame_len, win=HAMMING):
        self._L = capi.lib()
        self.handle = check_handle(getattr(self._L, self._init)(overlap_hint, frame_len, win), self._init)
        self.frame_len = frame_len
        self.fft_len = frame_len << (2 if overlap_hint == capi.OVERLAP_HIGH else 1)
        self.bins = self.fft_len // 2 + 1

    def close(self):
        if getattr(self, "handle", 0):
            getattr(self._L, self._uninit)(self.handle)
            self.handle = 0

    __del__ = close


class AnalysisFft(_FftFrames):
    _init, _uninit = "llz_analysis_fft_init", "llz_analysis_fft_uninit"

    def frame(self, x):
        x = _f64(x)
        re, im = np.zeros(self.bins), np.zeros(self.bins)
        self._L.llz_analysis_fft(self.handle, x.ctypes.data_as(_dp), re.ctypes.data_as(_dp), im.ctypes.data_as(_dp))
        return re, im


class SynthesisFft(_FftFrames):
    _init, _uninit = "llz_synthesis_fft_init", "llz_synthesis_fft_uninit"

    def frame(self, re, im):
        re, im = _f64(re), _f64(im)
        x = np.zeros(self.frame_len)
        self._L.llz_synthesis_fft(self.handle, re.ctypes.data_as(_dp), im.ctypes.data_as(_dp), x.ctypes.data_as(_dp))
        return x


class StftMC:
    """llz_stft_mc_*: many channels and frames per call, float32; x [channels, frames*frame_len],
    re/im [channels, frames, bins]; the handle carries both streams' state between calls."""

    def __init__(self, channels, overlap_hint, frame_len, win=HAMMING, stream=None):
        self._L = capi.lib()
        self.handle = check_handle(self._L.llz_stft_mc_init(channels, overlap_hint, frame_len, win), "llz_stft_mc_init")
        self.channels, self.frame_len = channels, frame_len
        self.bins = self._L.llz_stft_mc_bins(self.handle)
        if stream is not None:
            check(self._L.llz_stft_mc_set_stream(self.handle, _stream_ptr(stream)), "set_stream")

    def analysis(self, x, re, im):
        frames = x.shape[1] // self.frame_len
        check(self._L.llz_stft_mc_analysis(self.handle, _ptr(x), _ptr(re), _ptr(im), frames), "llz_stft_mc_analysis")
        return re, im

    def synthesis(self, re, im, x):
        frames = re.shape[1]
        check(self._L.llz_stft_mc_synthesis(self.handle, _ptr(re), _ptr(im), _ptr(x), frames), "llz_stft_mc_synthesis")
        return x

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_stft_mc_uninit(self.handle)
            self.handle = 0

    __del__ = close


class MdctFramesMC:
    """llz_mdct_frames_mc_*: windowed 50 %-overlap MDCT frames (the batch form of llz_analysis_mdct / llz_synthesis_mdct);
    x [channels, frames*frame_len] float32, X [channels, frames, frame_len]; the handle carries both streams' state."""

    def __init__(self, channels, frame_len, win=capi.MDCT_SINE, stream=None):
        self._L = capi.lib()
        self.handle = check_handle(self._L.llz_mdct_frames_mc_init(channels, frame_len, win), "llz_mdct_frames_mc_init")
        self.channels, self.frame_len = channels, frame_len
        if stream is not None:
            check(self._L.llz_mdct_frames_mc_set_stream(self.handle, _stream_ptr(stream)), "set_stream")

    def _frames(self, x, X):
        _typed(x, "float32", self.channels * X.shape[1] * self.frame_len, "x")
        _typed(X, "float32", self.channels * X.shape[1] * self.frame_len, "X")
        return X.shape[1]

    def analysis(self, x, X):
        check(self._L.llz_mdct_frames_mc_analysis(self.handle, _ptr(x), _ptr(X), self._frames(x, X)), "llz_mdct_frames_mc_analysis")
        return X

    def synthesis(self, X, x):
        check(self._L.llz_mdct_frames_mc_synthesis(self.handle, _ptr(X), _ptr(x), self._frames(x, X)), "llz_mdct_frames_mc_synthesis")
        return x

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_mdct_frames_mc_uninit(self.handle)
            self.handle = 0

    __del__ = close


# ------------------------------------------------------------------------------------------ MDCT
def mdct_window(win, n, alpha=6.0):
    w = np.zeros(n)
    if win == capi.MDCT_SINE:
        capi.lib().llz_mdct_sine(w.ctypes.data_as(_dp), n)
    else:
        capi.lib().llz_mdct_kbd(w.ctypes.data_as(_dp), n, alpha)
    return w


class Mdct:
    """llz_mdct_init / llz_mdct / llz_imdct (llz_mdct.h:37-41): one frame per call, host float64, exact."""

    def __init__(self, type_, length):
        self._L = capi.lib()
        self.handle = check_handle(self._L.llz_mdct_init(type_, length), "llz_mdct_init")
        self.length = length

    def forward(self, x):
        x = _f64(x)
        X = np.zeros(self.length // 2)
        self._L.llz_mdct(self.handle, x.ctypes.data_as(_dp), X.ctypes.data_as(_dp))
        return X

    def inverse(self, X):
        X = _f64(X)
        x = np.zeros(self.length)
        self._L.llz_imdct(self.handle, X.ctypes.data_as(_dp), x.ctypes.data_as(_dp))
        return x

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_mdct_uninit(self.handle)
            self.handle = 0

    __del__ = close


class MdctFixed:
    """llz_mdct_fixed_init / llz_mdct_fixed / llz_imdct_fixed (llz_mdct_fixed.h:24-28): host int32, bit-exact."""

    def __init__(self, type_, length):
        self._L = capi.lib()
        self.handle = check_handle(self._L.llz_mdct_fixed_init(type_, length), "llz_mdct_fixed_init")
        self.length = length

    def _run(self, fn, a, n_out):
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.zeros(n_out, dtype=np.int32)
        ip = C.POINTER(C.c_int)
        fn(self.handle, a.ctypes.data_as(ip), b.ctypes.data_as(ip))
        return b

    def forward(self, x):
        return self._run(self._L.llz_mdct_fixed, x, self.length // 2)

    def inverse(self, X):
        return self._run(self._L.llz_imdct_fixed, X, self.length)

    def _batch(self, fn, what, a, b, n_in, n_out):
        """a [count][n_in] -> b [count][n_out]: int32 numpy arrays (staged) or device tensors (in place, on the handle's stream)"""
        count = int(a.shape[0])
        assert tuple(a.shape) == (count, n_in) and tuple(b.shape) == (count, n_out)
        check(fn(self.handle, _ptr(a), _ptr(b), count), what)
        return b

    def forward_batch(self, x, X):
        return self._batch(self._L.llz_mdct_fixed_batch, "llz_mdct_fixed_batch", x, X, self.length, self.length // 2)

    def inverse_batch(self, X, x):
        return self._batch(self._L.llz_imdct_fixed_batch, "llz_imdct_fixed_batch", X, x, self.length // 2, self.length)

    def set_stream(self, stream):
        check(self._L.llz_mdct_fixed_set_stream(self.handle, C.c_void_p(stream.cuda_stream if stream is not None else 0)),
              "llz_mdct_fixed_set_stream")

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_mdct_fixed_uninit(self.handle)
            self.handle = 0

    __del__ = close


class _MdctFrames:
    _init = _uninit = _run = None

    def __init__(self, frame_len, win=0):
        self._L = capi.lib()
        self.handle = check_handle(getattr(self._L, self._init)(frame_len, win), self._init)
        self.frame_len = frame_len

    def frame(self, a):
        a = _f64(a)
        b = np.zeros(self.frame_len)
        getattr(self._L, self._run)(self.handle, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp))
        return b

    def close(self):
        if getattr(self, "handle", 0):
            getattr(self._L, self._uninit)(self.handle)
            self.handle = 0

    __del__ = close


class AnalysisMdct(_MdctFrames):
    _init, _uninit, _run = "llz_analysis_mdct_init", "llz_analysis_mdct_uninit", "llz_analysis_mdct"


class SynthesisMdct(_MdctFrames):
    _init, _uninit, _run = "llz_synthesis_mdct_init", "llz_synthesis_mdct_uninit", "llz_synthesis_mdct"


class MdctBatch:
    """llz_mdct_batch_*: many float32 frames per call, N/4-point-FFT algorithm; x [count, len], X [count, len/2]."""

    def __init__(self, length, stream=None):
        self._L = capi.lib()
        self.handle = check_handle(self._L.llz_mdct_batch_init(length), "llz_mdct_batch_init")
        self.length = length
        if stream is not None:
            check(self._L.llz_mdct_batch_set_stream(self.handle, _stream_ptr(stream)), "set_stream")

    def forward(self, x, X):
        check(self._L.llz_mdct_batch(self.handle, _ptr(x), _ptr(X), x.shape[0]), "llz_mdct_batch")
        return X

    def inverse(self, X, x):
        check(self._L.llz_imdct_batch(self.handle, _ptr(X), _ptr(x), X.shape[0]), "llz_imdct_batch")
        return x

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_mdct_batch_uninit(self.handle)
            self.handle = 0

    __del__ = close


# ------------------------------------------------------------------------------------------ PCM ingest / egress
def pcm_deinterleave(ileaved, planar, scale=1.0 / 32768.0, stream=None):
    """ileaved: [n, channels] int16 -> planar: [channels, n] float32."""
    n, ch = ileaved.shape
    check(capi.lib().llz_pcm_deinterleave_i16_f32(_ptr(ileaved), _ptr(planar), ch, n, scale, _stream_ptr(stream)),
          "llz_pcm_deinterleave_i16_f32")
    return planar


def pcm_interleave(planar, ileaved, scale=32768.0, stream=None):
    """planar: [channels, n] float32 -> ileaved: [n, channels] int16 (clamp, truncate toward zero)."""
    ch, n = planar.shape
    check(capi.lib().llz_pcm_interleave_f32_i16(_ptr(planar), _ptr(ileaved), ch, n, scale, _stream_ptr(stream)),
          "llz_pcm_interleave_f32_i16")
    return ileaved


# ------------------------------------------------------------------------------------------ synthetic PCM
def synth_f32(dst, seed, chan0=0, stream=None):
    """Fill a [channels, n] float32 device tensor with the counter-hash PCM of SURVEY.md 8(d)."""
    ch, n = dst.shape
    check(capi.lib().llz_hip_synth_f32(_ptr(dst), ch, n, dst.stride(0), seed, chan0, _stream_ptr(stream)),
          "llz_hip_synth_f32")
    return dst


def synth_i16(dst, seed, chan0=0, stream=None):
    ch, n = dst.shape
    check(capi.lib().llz_hip_synth_i16(_ptr(dst), ch, n, dst.stride(0), seed, chan0, _stream_ptr(stream)),
          "llz_hip_synth_i16")
    return dst


# ------------------------------------------------------------------------------------------ sharded handles (llz_shard.h)
class _Sharded:
    """One batch handle over several GPUs of a node, single process (include/llz_shard.h): contiguous channel ranges, one
    stream per shard, tables broadcast from the first device at init.  Buffers are passed per shard (lists)."""

    def _finish_init(self, handle, what):
        self.handle = check_handle(handle, what)
        L = self._L
        self.n_shards = check(L.llz_sharded_count(self.handle), "llz_sharded_count")
        self.shards = []
        for s in range(self.n_shards):
            dev, c0, cnt = C.c_int(), C.c_int(), C.c_int()
            check(L.llz_sharded_shard(self.handle, s, C.byref(dev), C.byref(c0), C.byref(cnt)), "llz_sharded_shard")
            self.shards.append((dev.value, c0.value, cnt.value))

    def _ptrs(self, bufs):
        if len(bufs) != self.n_shards:
            raise LlzError(f"{len(bufs)} buffers for {self.n_shards} shards")
        return (C.c_void_p * self.n_shards)(*[_ptr(b) for b in bufs])

    def split(self, tensor):
        """views of a [channels, ...] tensor / array, one per shard.  Host arrays are staged by each shard; a DEVICE tensor can
        only serve shards that live on its own GPU (the library refuses a buffer of another device, it never enables peer
        access), so with shards on several GPUs allocate per shard instead: see alloc()."""
        if hasattr(tensor, "is_cuda") and tensor.is_cuda:
            other = sorted({d for (d, _c0, _cnt) in self.shards if d != tensor.device.index})
            if other:
                raise LlzError(f"split(): tensor is on cuda:{tensor.device.index}, shards also live on GPU(s) {other}")
        return [tensor[c0:c0 + cnt] for (_d, c0, cnt) in self.shards]

    def alloc(self, per_channel, dtype):
        """one torch tensor [count, per_channel] per shard, each on its shard's GPU"""
        import torch
        return [torch.empty(cnt, per_channel, dtype=dtype, device=torch.device("cuda", d)) for (d, _c0, cnt) in self.shards]

    @property
    def rccl_ranks(self):
        return check(self._L.llz_sharded_rccl_ranks(self.handle), "llz_sharded_rccl_ranks")

    def synchronize(self):
        check(self._L.llz_sharded_synchronize(self.handle), "llz_sharded_synchronize")

    def timer_start(self):
        check(self._L.llz_sharded_timer_start(self.handle), "llz_sharded_timer_start")

    def timer_stop(self):
        check(self._L.llz_sharded_timer_stop(self.handle), "llz_sharded_timer_stop")

    def timer_ms(self):
        per = np.zeros(self.n_shards)
        ms = self._L.llz_sharded_timer_ms(self.handle, per.ctypes.data_as(_dp))
        if ms < 0:
            raise LlzError("llz_sharded_timer_ms: " + capi.last_error())
        return ms, per

    def close(self):
        if getattr(self, "handle", 0):
            self._L.llz_sharded_uninit(self.handle)
            self.handle = 0

    __del__ = close


def _devlist(devices):
    return (C.c_int * len(devices))(*[int(d) for d in devices])


class FirFilterMCSharded(_Sharded):
    def __init__(self, channels, frame_len, taps, devices, algo=FIR_ALGO_AUTO):
        self._L = capi.lib()
        taps32 = np.ascontiguousarray(_f64(taps).astype(np.float32))
        self.frame_len, self.flt_len = frame_len, len(taps32)
        self._finish_init(self._L.llz_fir_filter_mc_sharded_init(channels, frame_len, taps32.ctypes.data, len(taps32), algo,
                                                                 _devlist(devices), len(devices)),
                          "llz_fir_filter_mc_sharded_init")

    def filter(self, xs, outs):
        check(self._L.llz_fir_filter_mc_sharded(self.handle, self._ptrs(xs), self._ptrs(outs), self.frame_len),
              "llz_fir_filter_mc_sharded")
        return outs

    def flush(self, outs):
        return check(self._L.llz_fir_filter_mc_sharded_flush(self.handle, self._ptrs(outs)), "llz_fir_filter_mc_sharded_flush")


class IirCascadeMCSharded(_Sharded):
    def __init__(self, channels, coef, devices):
        self._L = capi.lib()
        coef = _f64(coef).reshape(-1, 6)
        self._finish_init(self._L.llz_iir_cascade_mc_sharded_init(channels, coef.shape[0], coef.ctypes.data,
                                                                  _devlist(devices), len(devices)),
                          "llz_iir_cascade_mc_sharded_init")

    def filter(self, xs, outs):
        n = xs[0].shape[-1]
        check(self._L.llz_iir_cascade_mc_sharded(self.handle, self._ptrs(xs), self._ptrs(outs), n),
              "llz_iir_cascade_mc_sharded")
        return outs


class ResampleMCSharded(_Sharded):
    def __init__(self, channels, L, M, gain, win, fmt, devices):
        self._L = capi.lib()
        self.L, self.M = L, M
        self._finish_init(self._L.llz_resample_mc_sharded_init(channels, L, M, gain, win, fmt, _devlist(devices),
                                                               len(devices)), "llz_resample_mc_sharded_init")

    def process(self, xs, outs):
        n_in = xs[0].shape[-1]
        return check(self._L.llz_resample_mc_sharded(self.handle, self._ptrs(xs), n_in, self._ptrs(outs)),
                     "llz_resample_mc_sharded")
