"""ctypes front-ends for the CPU checker.  TEST INFRASTRUCTURE ONLY.

Two back-ends with one scenario-level interface, so the same test body can run on either:

* ``Oracle``  -- oracle/liboracle.so, the restatement (orc_* symbols, oracle/llz_oracle.c)
* ``Ref``     -- oracle/_ref/libllzref.so, the reference's own C files compiled by oracle/Makefile
                 (llz_* symbols; reference libllzfilter/llz_{fir,iir,resample,fft,fft_fixed}.h)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libllzref.so")

HAMMING, BLACKMAN, KAISER = 0, 1, 2
LPF, HPF, BPF, BSF = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    """Compile liboracle.so (and _ref when the reference tree is mounted). Building the checker is not using it."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "llz_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    # rebuilt whenever the recipe is newer than the library (the list of reference files grows with the widening rows)
    if os.path.isdir("/root/reference/libllzfilter") and (
            force or not os.path.exists(REF_SO) or
            os.path.getmtime(REF_SO) < os.path.getmtime(os.path.join(HERE, "Makefile"))):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
    # the reference's example CLI, against its own sources and against the GPU library (needs the product .so)
    hip_so = os.path.join(os.path.dirname(HERE), "llzlab_amd", "libllzfilter_hip.so")
    cli = os.path.join(HERE, "_ref", "llz_resample_hip")
    if os.path.isdir("/root/reference/example/llz_resample") and os.path.exists(hip_so) and \
            (force or not os.path.exists(cli) or os.path.getmtime(cli) < os.path.getmtime(hip_so)):
        subprocess.check_call(["make", "-s", "-C", HERE, "cli"])


def have_ref():
    return os.path.exists(REF_SO)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, t=_dp):
    return a.ctypes.data_as(t)


class Oracle:
    """Restatement back-end."""
    name = "oracle"

    def __init__(self):
        build()
        L = self.lib = C.CDLL(ORACLE_SO)
        vp = C.c_void_p
        L.orc_fir_design.argtypes = [C.c_int, _dp, C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_conv.restype = C.c_double
        L.orc_conv.argtypes = [_dp, _dp, C.c_int]
        L.orc_kaiser_atten2beta.restype = C.c_double
        L.orc_kaiser_atten2beta.argtypes = [C.c_double]
        for n in ("orc_hamming_cof_num", "orc_blackman_cof_num"):
            getattr(L, n).argtypes = [C.c_double]
        L.orc_kaiser_cof_num.argtypes = [C.c_double, C.c_double]
        for n in ("orc_hamming", "orc_blackman", "orc_kaiser"):
            getattr(L, n).argtypes = [_dp, C.c_int]
        L.orc_kaiser_beta.argtypes = [_dp, C.c_int, C.c_double]
        L.orc_fir_new.restype = vp
        L.orc_fir_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_fir_new_taps.restype = vp
        L.orc_fir_new_taps.argtypes = [C.c_int, _dp, C.c_int]
        L.orc_fir_flt_len.argtypes = [vp]
        L.orc_fir_taps.restype = _dp
        L.orc_fir_taps.argtypes = [vp]
        L.orc_fir_run.argtypes = [vp, _dp, _dp, C.c_int]
        L.orc_fir_flush.argtypes = [vp, _dp]
        L.orc_fir_free.argtypes = [vp]
        L.orc_iir_new.restype = vp
        L.orc_iir_new.argtypes = [C.c_int, _dp, C.c_int, _dp]
        L.orc_iir_run.argtypes = [vp, _dp, _dp, C.c_int]
        L.orc_iir_flush.argtypes = [vp, _dp]
        L.orc_iir_free.argtypes = [vp]
        L.orc_rs_new.restype = vp
        L.orc_rs_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        for n in ("orc_rs_bytes_in", "orc_rs_bytes_out", "orc_rs_num_taps", "orc_rs_sub_len"):
            getattr(L, n).argtypes = [vp]
        L.orc_rs_matrix.restype = _dp
        L.orc_rs_matrix.argtypes = [vp]
        L.orc_rs_proto.restype = _dp
        L.orc_rs_proto.argtypes = [vp]
        L.orc_rs_run.argtypes = [vp, C.c_void_p, C.c_int, C.c_void_p, _ip]
        L.orc_rs_free.argtypes = [vp]
        for n in ("orc_fft_new", "orc_fftx_new"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [C.c_int]
        for n in ("orc_fft_fwd", "orc_fft_inv"):
            getattr(L, n).argtypes = [vp, _dp]
        for n in ("orc_fftx_fwd", "orc_fftx_inv"):
            getattr(L, n).argtypes = [vp, _ip]
        L.orc_fft_free.argtypes = [vp]
        L.orc_fftx_free.argtypes = [vp]
        L.orc_fftx_cos.restype = C.POINTER(C.c_short)
        L.orc_fftx_cos.argtypes = [vp]
        L.orc_fftx_sin.restype = C.POINTER(C.c_short)
        L.orc_fftx_sin.argtypes = [vp]
        L.orc_autocorr.argtypes = [_dp, C.c_int, C.c_int, _dp]
        L.orc_crosscorr.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp]
        L.orc_corr_cof.restype = C.c_double
        L.orc_corr_cof.argtypes = [_dp, _dp, C.c_int]
        L.orc_acf_new.restype = vp
        L.orc_acf_new.argtypes = [C.c_int]
        L.orc_acf_fft_len.argtypes = [vp]
        L.orc_acf_run.argtypes = [vp, _dp, C.c_int, C.c_int, _dp]
        L.orc_acf_free.argtypes = [vp]
        L.orc_stft_new.restype = vp
        L.orc_stft_new.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_stft_fft_len.argtypes = [vp]
        L.orc_stft_analysis.argtypes = [vp, _dp, _dp, _dp]
        L.orc_stft_synthesis.argtypes = [vp, _dp, _dp, _dp]
        L.orc_stft_free.argtypes = [vp]
        L.orc_mdct_sine.argtypes = [_dp, C.c_int]
        L.orc_mdct_kbd.argtypes = [_dp, C.c_int, C.c_double]
        L.orc_mdct_new.restype = vp
        L.orc_mdct_new.argtypes = [C.c_int, C.c_int]
        L.orc_mdct_length.argtypes = [vp]
        L.orc_mdct_fwd.argtypes = [vp, _dp, _dp]
        L.orc_mdct_inv.argtypes = [vp, _dp, _dp]
        L.orc_mdct_free.argtypes = [vp]
        L.orc_mdctx_new.restype = vp
        L.orc_mdctx_new.argtypes = [C.c_int, C.c_int]
        L.orc_mdctx_length.argtypes = [vp]
        L.orc_mdctx_fwd.argtypes = [vp, _ip, _ip]
        L.orc_mdctx_inv.argtypes = [vp, _ip, _ip]
        L.orc_mdctx_free.argtypes = [vp]
        L.orc_amdct_new.restype = vp
        L.orc_amdct_new.argtypes = [C.c_int, C.c_int]
        L.orc_amdct_analysis.argtypes = [vp, _dp, _dp]
        L.orc_amdct_synthesis.argtypes = [vp, _dp, _dp]
        L.orc_amdct_free.argtypes = [vp]
        L.orc_fir_batch_f32.argtypes = [C.c_void_p, _dp, C.c_int, C.c_long, _dp, C.c_int]
        L.orc_iir_cascade_batch_f32.argtypes = [C.c_void_p, _dp, C.c_int, C.c_long, _dp, C.c_int]
        L.orc_rs_batch_f32.restype = C.c_long
        L.orc_rs_batch_f32.argtypes = [C.c_void_p, _dp, C.c_int, C.c_long, C.c_int, C.c_int, C.c_double, C.c_int]
        L.orc_rs_batch_i16.restype = C.c_long
        L.orc_rs_batch_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_int, C.c_double, C.c_int]
        L.orc_synth_f32.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_uint, C.c_int]
        L.orc_synth_i16.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_uint, C.c_int]
        L.orc_pcm_deinterleave_i16_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_float]
        L.orc_pcm_interleave_f32_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_float]
        L.orc_wav_parse.argtypes = [C.c_char_p, C.c_long, C.POINTER(C.c_long)]
        L.orc_wav_header.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_long]

    # ---- design -------------------------------------------------------------------------------
    def window(self, win, n, beta=None):
        w = np.zeros(n)
        if beta is not None:
            self.lib.orc_kaiser_beta(_ptr(w), n, beta)
        else:
            (self.lib.orc_hamming, self.lib.orc_blackman, self.lib.orc_kaiser)[win](_ptr(w), n)
        return w

    def cof_num(self, win, ftrans, atten=90.0):
        if win == KAISER:
            return self.lib.orc_kaiser_cof_num(ftrans, atten)
        return (self.lib.orc_hamming_cof_num, self.lib.orc_blackman_cof_num)[win](ftrans)

    def atten2beta(self, atten):
        return self.lib.orc_kaiser_atten2beta(atten)

    def fir_design(self, kind, n, fc1, fc2=0.0, win=HAMMING):
        h = np.zeros(n + 1)
        m = self.lib.orc_fir_design(kind, _ptr(h), n, fc1, fc2, win)
        return h[:m].copy()

    def conv(self, x_hist, h):
        """x_hist: the last len(h) samples, oldest first; returns the dot product at the newest."""
        x = _f64(x_hist)
        h = _f64(h)
        newest = C.cast(C.c_void_p(x.ctypes.data + 8 * (len(h) - 1)), _dp)
        return self.lib.orc_conv(newest, _ptr(h), len(h))

    # ---- streaming FIR ------------------------------------------------------------------------
    def fir_stream(self, kind, frame_len, flt_len, fc1, fc2, win, x, flush=True):
        """Feed x in frames of frame_len (len(x) must be a multiple); returns (y, tail, taps)."""
        f = self.lib.orc_fir_new(kind, frame_len, flt_len, fc1, fc2, win)
        try:
            T = self.lib.orc_fir_flt_len(f)
            taps = np.ctypeslib.as_array(self.lib.orc_fir_taps(f), shape=(T,)).copy()
            x = _f64(x)
            y = np.zeros_like(x)
            for o in range(0, len(x), frame_len):
                xi = np.ascontiguousarray(x[o:o + frame_len])
                yo = np.zeros(frame_len)
                self.lib.orc_fir_run(f, _ptr(xi), _ptr(yo), frame_len)
                y[o:o + frame_len] = yo
            tail = np.zeros(max(T - 1, 1))
            if flush:
                self.lib.orc_fir_flush(f, _ptr(tail))
            return y, tail[:T - 1], taps
        finally:
            self.lib.orc_fir_free(f)

    def fir_taps_stream(self, h, frame_len, x):
        """Same state machine with caller-supplied taps (oracle only)."""
        h = _f64(h)
        f = self.lib.orc_fir_new_taps(frame_len, _ptr(h), len(h))
        x = _f64(x)
        y = np.zeros_like(x)
        for o in range(0, len(x), frame_len):
            xi = np.ascontiguousarray(x[o:o + frame_len])
            yo = np.zeros(frame_len)
            self.lib.orc_fir_run(f, _ptr(xi), _ptr(yo), frame_len)
            y[o:o + frame_len] = yo
        self.lib.orc_fir_free(f)
        return y

    # ---- IIR -----------------------------------------------------------------------------------
    def iir_stream(self, a, b, x, frame_len=None, flush=True):
        a = _f64(a)
        M = len(a) - 1
        if b is None:
            raise ValueError("pass b explicitly (use zeros for the reference's b==NULL case)")
        b = _f64(b)
        N = len(b) - 1
        f = self.lib.orc_iir_new(M, _ptr(a), N, _ptr(b))
        x = _f64(x)
        y = np.zeros_like(x)
        frame_len = frame_len or len(x)
        for o in range(0, len(x), frame_len):
            xi = np.ascontiguousarray(x[o:o + frame_len])
            yo = np.zeros(len(xi))
            self.lib.orc_iir_run(f, _ptr(xi), _ptr(yo), len(xi))
            y[o:o + len(xi)] = yo
        tail = np.zeros(max(N, 1))
        if flush:
            self.lib.orc_iir_flush(f, _ptr(tail))
        self.lib.orc_iir_free(f)
        return y, tail[:N]

    # ---- resample family -----------------------------------------------------------------------
    def rs_info(self, mode, L, M, gain, win):
        r = self.lib.orc_rs_new(mode, L, M, gain, win)
        if not r:
            return None
        rows = {0: M, 1: L, 2: L}[mode]
        cols = self.lib.orc_rs_sub_len(r)
        n = self.lib.orc_rs_num_taps(r)
        info = dict(bytes_in=self.lib.orc_rs_bytes_in(r), bytes_out=self.lib.orc_rs_bytes_out(r),
                    n=n, cols=cols,
                    matrix=np.ctypeslib.as_array(self.lib.orc_rs_matrix(r), shape=(rows, cols)).copy(),
                    proto=np.ctypeslib.as_array(self.lib.orc_rs_proto(r), shape=(n,)).copy())
        self.lib.orc_rs_free(r)
        return info

    def rs_stream(self, mode, L, M, gain, win, pcm, pad_tail=0):
        """pcm: int16 array whose length is a multiple of the frame; returns int16 output.
        (pad_tail is accepted for signature parity with Ref; the restatement always reads zeros there.)"""
        r = self.lib.orc_rs_new(mode, L, M, gain, win)
        if not r:
            return None
        nin = self.lib.orc_rs_bytes_in(r) // 2
        nout = self.lib.orc_rs_bytes_out(r) // 2
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        frames = len(pcm) // nin
        out = np.zeros(frames * nout, dtype=np.int16)
        ob = C.c_int(0)
        for f in range(frames):
            xi = np.ascontiguousarray(pcm[f * nin:(f + 1) * nin])
            yo = np.zeros(nout, dtype=np.int16)
            self.lib.orc_rs_run(r, xi.ctypes.data, 2 * nin, yo.ctypes.data, C.byref(ob))
            out[f * nout:(f + 1) * nout] = yo
        self.lib.orc_rs_free(r)
        return out

    # ---- FFTs ----------------------------------------------------------------------------------
    def fft(self, data, inverse=False):
        """data: complex128 array of power-of-two length; returns the transform (new array)."""
        z = np.ascontiguousarray(data, dtype=np.complex128).copy()
        f = self.lib.orc_fft_new(len(z))
        buf = z.view(np.float64)
        (self.lib.orc_fft_inv if inverse else self.lib.orc_fft_fwd)(f, _ptr(buf))
        self.lib.orc_fft_free(f)
        return z

    def fft_fixed(self, data, inverse=False):
        """data: int32 array [2*N] interleaved re,im; returns transformed copy."""
        z = np.ascontiguousarray(data, dtype=np.int32).copy()
        f = self.lib.orc_fftx_new(len(z) // 2)
        (self.lib.orc_fftx_inv if inverse else self.lib.orc_fftx_fwd)(f, _ptr(z, _ip))
        self.lib.orc_fftx_free(f)
        return z

    def fft_fixed_tables(self, n):
        f = self.lib.orc_fftx_new(n)
        c = np.ctypeslib.as_array(self.lib.orc_fftx_cos(f), shape=(n,)).copy()
        s = np.ctypeslib.as_array(self.lib.orc_fftx_sin(f), shape=(n,)).copy()
        self.lib.orc_fftx_free(f)
        return c, s

    # ---- correlation -----------------------------------------------------------------------------
    def autocorr(self, x, p):
        x = _f64(x)
        r = np.zeros(p + 1)
        self.lib.orc_autocorr(_ptr(x), len(x), p, _ptr(r))
        return r

    def crosscorr(self, x, y, p):
        x, y = _f64(x), _f64(y)
        r = np.zeros(p + 1)
        self.lib.orc_crosscorr(_ptr(x), _ptr(y), len(x), p, _ptr(r))
        return r

    def corr_cof(self, a, b):
        a, b = _f64(a), _f64(b)
        return self.lib.orc_corr_cof(_ptr(a), _ptr(b), len(a))

    def autocorr_fast(self, x, p):
        x = _f64(x)
        h = self.lib.orc_acf_new(len(x))
        r = np.zeros(p + 1)
        self.lib.orc_acf_run(h, _ptr(x), len(x), p, _ptr(r))
        self.lib.orc_acf_free(h)
        return r

    def stft_analysis(self, overlap_hint, frame_len, win, x):
        """stream x (a whole number of frames) through one analysis handle -> re, im of shape [frames, fft_len/2+1]"""
        x = _f64(x)
        h = self.lib.orc_stft_new(overlap_hint, frame_len, win)
        bins = self.lib.orc_stft_fft_len(h) // 2 + 1
        frames = len(x) // frame_len
        re, im = np.zeros((frames, bins)), np.zeros((frames, bins))
        for f in range(frames):
            xi = np.ascontiguousarray(x[f * frame_len:(f + 1) * frame_len])
            r, i = np.zeros(bins), np.zeros(bins)
            self.lib.orc_stft_analysis(h, _ptr(xi), _ptr(r), _ptr(i))
            re[f], im[f] = r, i
        self.lib.orc_stft_free(h)
        return re, im

    def stft_synthesis(self, overlap_hint, frame_len, win, re, im):
        re, im = np.ascontiguousarray(re, dtype=np.float64), np.ascontiguousarray(im, dtype=np.float64)
        h = self.lib.orc_stft_new(overlap_hint, frame_len, win)
        x = np.zeros(re.shape[0] * frame_len)
        for f in range(re.shape[0]):
            r, i, xo = np.ascontiguousarray(re[f]), np.ascontiguousarray(im[f]), np.zeros(frame_len)
            self.lib.orc_stft_synthesis(h, _ptr(r), _ptr(i), _ptr(xo))
            x[f * frame_len:(f + 1) * frame_len] = xo
        self.lib.orc_stft_free(h)
        return x

    def mdct_window(self, win, n, alpha=6.0):
        w = np.zeros(n)
        if win == 0:
            self.lib.orc_mdct_sine(_ptr(w), n)
        else:
            self.lib.orc_mdct_kbd(_ptr(w), n, alpha)
        return w

    def mdct(self, type_, x):
        x = _f64(x)
        h = self.lib.orc_mdct_new(type_, len(x))
        X = np.zeros(self.lib.orc_mdct_length(h) // 2)
        self.lib.orc_mdct_fwd(h, _ptr(x), _ptr(X))
        self.lib.orc_mdct_free(h)
        return X

    def imdct(self, type_, X):
        X = _f64(X)
        h = self.lib.orc_mdct_new(type_, 2 * len(X))
        x = np.zeros(2 * len(X))
        self.lib.orc_mdct_inv(h, _ptr(X), _ptr(x))
        self.lib.orc_mdct_free(h)
        return x

    def mdct_fixed(self, type_, x, inverse=False):
        x = np.ascontiguousarray(x, dtype=np.int32)
        n = 2 * len(x) if inverse else len(x)
        h = self.lib.orc_mdctx_new(type_, n)
        out = np.zeros(n if inverse else n // 2, dtype=np.int32)
        (self.lib.orc_mdctx_inv if inverse else self.lib.orc_mdctx_fwd)(h, x.ctypes.data_as(_ip), out.ctypes.data_as(_ip))
        self.lib.orc_mdctx_free(h)
        return out

    def mdct_frames(self, frame_len, win, x):
        """stream x through one analysis and one synthesis handle -> (coefficients [frames, frame_len], output)"""
        x = _f64(x)
        a, s = self.lib.orc_amdct_new(frame_len, win), self.lib.orc_amdct_new(frame_len, win)
        frames = len(x) // frame_len
        X, y = np.zeros((frames, frame_len)), np.zeros(frames * frame_len)
        for f in range(frames):
            xi, Xi, yi = np.ascontiguousarray(x[f * frame_len:(f + 1) * frame_len]), np.zeros(frame_len), np.zeros(frame_len)
            self.lib.orc_amdct_analysis(a, _ptr(xi), _ptr(Xi))
            self.lib.orc_amdct_synthesis(s, _ptr(Xi), _ptr(yi))
            X[f], y[f * frame_len:(f + 1) * frame_len] = Xi, yi
        self.lib.orc_amdct_free(a)
        self.lib.orc_amdct_free(s)
        return X, y

    # ---- batch drivers (oracle only) -----------------------------------------------------------
    def fir_batch_f32(self, x, h):
        x = np.ascontiguousarray(x, dtype=np.float32)
        h = _f64(h)
        Cn, n = x.shape
        out = np.zeros((Cn, n))
        self.lib.orc_fir_batch_f32(x.ctypes.data, _ptr(out), Cn, n, _ptr(h), len(h))
        return out

    def fir_batch_f32_mt(self, x, h, threads=None):
        """fir_batch_f32 with the channels spread over host threads (ctypes releases the GIL; channels are independent)"""
        import os
        from concurrent.futures import ThreadPoolExecutor
        x = np.ascontiguousarray(x, dtype=np.float32)
        threads = max(1, min(threads or (os.cpu_count() or 1), x.shape[0]))
        parts = np.array_split(np.arange(x.shape[0]), threads)
        with ThreadPoolExecutor(threads) as ex:
            outs = list(ex.map(lambda idx: self.fir_batch_f32(x[idx[0]:idx[-1] + 1], h), [q for q in parts if len(q)]))
        return np.concatenate(outs, axis=0)

    def iir_cascade_batch_f32(self, x, coef):
        x = np.ascontiguousarray(x, dtype=np.float32)
        coef = _f64(coef).reshape(-1, 6)
        Cn, n = x.shape
        out = np.zeros((Cn, n))
        self.lib.orc_iir_cascade_batch_f32(x.ctypes.data, _ptr(out), Cn, n, _ptr(coef), coef.shape[0])
        return out

    def rs_batch_f32(self, x, L, M, gain, win):
        x = np.ascontiguousarray(x, dtype=np.float32)
        Cn, n = x.shape
        out = np.zeros((Cn, (n * L) // M))
        r = self.lib.orc_rs_batch_f32(x.ctypes.data, _ptr(out), Cn, n, L, M, gain, win)
        assert r == out.shape[1]
        return out

    def rs_batch_i16(self, x, L, M, gain, win):
        x = np.ascontiguousarray(x, dtype=np.int16)
        Cn, n = x.shape
        info = self.rs_info(2, L, M, gain, win)
        nin = info["bytes_in"] // 2
        assert n % nin == 0
        out = np.zeros((Cn, (n // nin) * (info["bytes_out"] // 2)), dtype=np.int16)
        r = self.lib.orc_rs_batch_i16(x.ctypes.data, out.ctypes.data, Cn, n, L, M, gain, win)
        assert r == out.shape[1]
        return out

    def synth_f32(self, channels, n, seed, chan0=0):
        out = np.zeros((channels, n), dtype=np.float32)
        self.lib.orc_synth_f32(out.ctypes.data, channels, n, seed, chan0)
        return out

    def pcm_deinterleave(self, il, scale=1.0 / 32768.0):
        """il: [n][channels] int16 -> planar [channels][n] float32"""
        il = np.ascontiguousarray(il, dtype=np.int16)
        n, ch = il.shape
        out = np.zeros((ch, n), dtype=np.float32)
        self.lib.orc_pcm_deinterleave_i16_f32(il.ctypes.data, out.ctypes.data, ch, n, scale)
        return out

    def pcm_interleave(self, pl, scale=32768.0):
        pl = np.ascontiguousarray(pl, dtype=np.float32)
        ch, n = pl.shape
        out = np.zeros((n, ch), dtype=np.int16)
        self.lib.orc_pcm_interleave_f32_i16(pl.ctypes.data, out.ctypes.data, ch, n, scale)
        return out

    def wav_parse(self, data):
        out = (C.c_long * 7)()
        if self.lib.orc_wav_parse(bytes(data), len(data), out) != 0:
            return None
        return dict(zip(("format", "channels", "samplerate", "bytes_per_sample", "block_align", "frames", "data_offset"),
                        [int(v) for v in out]))

    def wav_header(self, channels, samplerate, bytes_per_sample, frames):
        h = np.zeros(44, dtype=np.uint8)
        self.lib.orc_wav_header(h.ctypes.data, channels, samplerate, bytes_per_sample, frames)
        return h.tobytes()

    def synth_i16(self, channels, n, seed, chan0=0):
        out = np.zeros((channels, n), dtype=np.int16)
        self.lib.orc_synth_i16(out.ctypes.data, channels, n, seed, chan0)
        return out


class _RefWavFmt(C.Structure):               # llz_wavfmt_t (libllzaudio/llz_wavfmt.h)
    _fields_ = [("format", C.c_ushort), ("channels", C.c_ushort), ("samplerate", C.c_ulong),
                ("bytes_per_sample", C.c_ushort), ("block_align", C.c_ushort), ("data_size", C.c_ulong)]


class Ref:
    """The reference's own compiled C files (llz_* API, handles are unsigned long)."""
    name = "ref"

    def wav_readheader(self, path):
        """llz_wavfmt_readheader on a file (the reference's own reader, FILE* from libc); returns the struct's fields and the
        file offset of the first data byte"""
        libc = C.CDLL(None)
        libc.fopen.restype = C.c_void_p
        libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        libc.fclose.argtypes = [C.c_void_p]
        libc.ftell.restype = C.c_long
        libc.ftell.argtypes = [C.c_void_p]
        self.lib.llz_wavfmt_readheader.restype = _RefWavFmt
        self.lib.llz_wavfmt_readheader.argtypes = [C.c_void_p]
        fp = libc.fopen(path.encode(), b"rb")
        assert fp
        f = self.lib.llz_wavfmt_readheader(fp)
        off = libc.ftell(fp)
        libc.fclose(fp)
        return {"format": f.format, "channels": f.channels, "samplerate": f.samplerate,
                "bytes_per_sample": f.bytes_per_sample, "block_align": f.block_align, "frames": f.data_size,
                "data_offset": off}

    def wav_writeheader(self, path, channels, samplerate, bytes_per_sample, frames):
        libc = C.CDLL(None)
        libc.fopen.restype = C.c_void_p
        libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        libc.fclose.argtypes = [C.c_void_p]
        self.lib.llz_wavfmt_writeheader.argtypes = [_RefWavFmt, C.c_void_p]
        self.lib.llz_wavfmt_writeheader.restype = None
        fmt = _RefWavFmt(1, channels, samplerate, bytes_per_sample, channels * bytes_per_sample, frames)
        fp = libc.fopen(path.encode(), b"wb")
        assert fp
        self.lib.llz_wavfmt_writeheader(fmt, fp)
        libc.fclose(fp)
        return open(path, "rb").read()

    def __init__(self):
        if not have_ref():
            build()
        if not have_ref():
            raise RuntimeError("oracle/_ref/libllzref.so not built (reference tree absent)")
        L = self.lib = C.CDLL(REF_SO)
        ul = C.c_ulong
        pdp = C.POINTER(_dp)
        for n in ("llz_fir_lpf_cof", "llz_fir_hpf_cof"):
            getattr(L, n).argtypes = [pdp, C.c_int, C.c_double, C.c_int]
        for n in ("llz_fir_bandpass_cof", "llz_fir_bandstop_cof"):
            getattr(L, n).argtypes = [pdp, C.c_int, C.c_double, C.c_double, C.c_int]
        L.llz_conv.restype = C.c_double
        L.llz_conv.argtypes = [_dp, _dp, C.c_int]
        L.llz_kaiser_atten2beta.restype = C.c_double
        L.llz_kaiser_atten2beta.argtypes = [C.c_double]
        for n in ("llz_hamming_cof_num", "llz_blackman_cof_num"):
            getattr(L, n).argtypes = [C.c_double]
        L.llz_kaiser_cof_num.argtypes = [C.c_double, C.c_double]
        for n in ("llz_hamming", "llz_blackman", "llz_kaiser"):
            getattr(L, n).argtypes = [_dp, C.c_int]
        L.llz_kaiser_beta.argtypes = [_dp, C.c_int, C.c_double]
        for n in ("llz_fir_filter_lpf_init", "llz_fir_filter_hpf_init"):
            getattr(L, n).restype = ul
            getattr(L, n).argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
        for n in ("llz_fir_filter_bandpass_init", "llz_fir_filter_bandstop_init"):
            getattr(L, n).restype = ul
            getattr(L, n).argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        L.llz_fir_filter.argtypes = [ul, _dp, _dp, C.c_int]
        L.llz_fir_filter_flush.argtypes = [ul, _dp]
        L.llz_fir_filter_uninit.argtypes = [ul]
        L.llz_iir_filter_init.restype = ul
        L.llz_iir_filter_init.argtypes = [C.c_int, _dp, C.c_int, _dp]
        L.llz_iir_filter.argtypes = [ul, _dp, _dp, C.c_int]
        L.llz_iir_filter_flush.argtypes = [ul, _dp]
        L.llz_iir_filter_uninit.argtypes = [ul]
        for n in ("llz_decimate_init", "llz_interp_init"):
            getattr(L, n).restype = ul
            getattr(L, n).argtypes = [C.c_int, C.c_double, C.c_int]
        L.llz_resample_filter_init.restype = ul
        L.llz_resample_filter_init.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
        for n in ("llz_decimate_uninit", "llz_interp_uninit", "llz_resample_filter_uninit"):
            getattr(L, n).argtypes = [ul]
        L.llz_get_resample_framelen_bytes.argtypes = [ul]
        for n in ("llz_decimate", "llz_interp", "llz_resample"):
            getattr(L, n).argtypes = [ul, C.c_void_p, C.c_int, C.c_void_p, _ip]
        for n in ("llz_fft_init", "llz_fft_fixed_init"):
            getattr(L, n).restype = ul
            getattr(L, n).argtypes = [C.c_int]
        for n in ("llz_fft", "llz_ifft"):
            getattr(L, n).argtypes = [ul, _dp]
        for n in ("llz_fft_fixed", "llz_ifft_fixed"):
            getattr(L, n).argtypes = [ul, _ip]
        L.llz_fft_uninit.argtypes = [ul]
        L.llz_fft_fixed_uninit.argtypes = [ul]
        L.llz_autocorr.argtypes = [_dp, C.c_int, C.c_int, _dp]
        L.llz_crosscorr.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp]
        L.llz_corr_cof.restype = C.c_double
        L.llz_corr_cof.argtypes = [_dp, _dp, C.c_int]
        L.llz_autocorr_fast_init.restype = ul
        L.llz_autocorr_fast_init.argtypes = [C.c_int]
        L.llz_autocorr_fast_uninit.argtypes = [ul]
        L.llz_autocorr_fast.argtypes = [ul, _dp, C.c_int, C.c_int, _dp]
        for n in ("llz_analysis_fft_init", "llz_synthesis_fft_init"):
            getattr(L, n).restype = ul
            getattr(L, n).argtypes = [C.c_int, C.c_int, C.c_int]
        L.llz_analysis_fft.argtypes = [ul, _dp, _dp, _dp]
        L.llz_synthesis_fft.argtypes = [ul, _dp, _dp, _dp]
        L.llz_analysis_fft_uninit.argtypes = [ul]
        L.llz_synthesis_fft_uninit.argtypes = [ul]
        L.llz_mdct_sine.argtypes = [_dp, C.c_int]
        L.llz_mdct_kbd.argtypes = [_dp, C.c_int, C.c_double]
        L.llz_mdct_init.restype = ul
        L.llz_mdct_init.argtypes = [C.c_int, C.c_int]
        L.llz_mdct.argtypes = [ul, _dp, _dp]
        L.llz_imdct.argtypes = [ul, _dp, _dp]
        L.llz_mdct_uninit.argtypes = [ul]
        L.llz_mdct_fixed_init.restype = ul
        L.llz_mdct_fixed_init.argtypes = [C.c_int, C.c_int]
        L.llz_mdct_fixed.argtypes = [ul, _ip, _ip]
        L.llz_imdct_fixed.argtypes = [ul, _ip, _ip]
        L.llz_mdct_fixed_uninit.argtypes = [ul]
        for n in ("llz_analysis_mdct_init", "llz_synthesis_mdct_init"):
            getattr(L, n).restype = ul
            getattr(L, n).argtypes = [C.c_int, C.c_int]
        L.llz_analysis_mdct.argtypes = [ul, _dp, _dp]
        L.llz_synthesis_mdct.argtypes = [ul, _dp, _dp]
        L.llz_analysis_mdct_uninit.argtypes = [ul]
        L.llz_synthesis_mdct_uninit.argtypes = [ul]
        self._libc = C.CDLL(None)
        self._libc.free.argtypes = [C.c_void_p]

    def window(self, win, n, beta=None):
        w = np.zeros(n)
        if beta is not None:
            self.lib.llz_kaiser_beta(_ptr(w), n, beta)
        else:
            (self.lib.llz_hamming, self.lib.llz_blackman, self.lib.llz_kaiser)[win](_ptr(w), n)
        return w

    def cof_num(self, win, ftrans, atten=90.0):
        if win == KAISER:
            return self.lib.llz_kaiser_cof_num(ftrans, atten)
        return (self.lib.llz_hamming_cof_num, self.lib.llz_blackman_cof_num)[win](ftrans)

    def atten2beta(self, atten):
        return self.lib.llz_kaiser_atten2beta(atten)

    def fir_design(self, kind, n, fc1, fc2=0.0, win=HAMMING):
        hp = _dp()
        if kind == LPF:
            m = self.lib.llz_fir_lpf_cof(C.byref(hp), n, fc1, win)
        elif kind == HPF:
            m = self.lib.llz_fir_hpf_cof(C.byref(hp), n, fc1, win)
        elif kind == BPF:
            m = self.lib.llz_fir_bandpass_cof(C.byref(hp), n, fc1, fc2, win)
        else:
            m = self.lib.llz_fir_bandstop_cof(C.byref(hp), n, fc1, fc2, win)
        h = np.ctypeslib.as_array(hp, shape=(m,)).copy()
        self._libc.free(hp)
        return h

    def conv(self, x_hist, h):
        x = _f64(x_hist)
        h = _f64(h)
        newest = C.cast(C.c_void_p(x.ctypes.data + 8 * (len(h) - 1)), _dp)
        return self.lib.llz_conv(newest, _ptr(h), len(h))

    def fir_stream(self, kind, frame_len, flt_len, fc1, fc2, win, x, flush=True):
        if kind == LPF:
            f = self.lib.llz_fir_filter_lpf_init(frame_len, flt_len, fc1, win)
        elif kind == HPF:
            f = self.lib.llz_fir_filter_hpf_init(frame_len, flt_len, fc1, win)
        elif kind == BPF:
            f = self.lib.llz_fir_filter_bandpass_init(frame_len, flt_len, fc1, fc2, win)
        else:
            f = self.lib.llz_fir_filter_bandstop_init(frame_len, flt_len, fc1, fc2, win)
        taps = self.fir_design(kind, flt_len, fc1, fc2, win)
        T = len(taps)
        x = _f64(x)
        y = np.zeros_like(x)
        for o in range(0, len(x), frame_len):
            xi = np.ascontiguousarray(x[o:o + frame_len])
            yo = np.zeros(frame_len)
            self.lib.llz_fir_filter(f, _ptr(xi), _ptr(yo), frame_len)
            y[o:o + frame_len] = yo
        tail = np.zeros(max(T - 1, 1))
        if flush:
            # the reference's flush reads past its buffer when flt_len-2 >= frame_len
            assert T - 2 < frame_len, "flush would read out of bounds in the reference"
            self.lib.llz_fir_filter_flush(f, _ptr(tail))
        self.lib.llz_fir_filter_uninit(f)
        return y, tail[:T - 1], taps

    def iir_stream(self, a, b, x, frame_len=None, flush=True):
        a = _f64(a)
        b = _f64(b)
        M, N = len(a) - 1, len(b) - 1
        f = self.lib.llz_iir_filter_init(M, _ptr(a), N, _ptr(b))
        x = _f64(x)
        y = np.zeros_like(x)
        frame_len = frame_len or len(x)
        for o in range(0, len(x), frame_len):
            xi = np.ascontiguousarray(x[o:o + frame_len])
            yo = np.zeros(len(xi))
            self.lib.llz_iir_filter(f, _ptr(xi), _ptr(yo), len(xi))
            y[o:o + len(xi)] = yo
        tail = np.zeros(max(N, 1))
        if flush:
            self.lib.llz_iir_filter_flush(f, _ptr(tail))
        self.lib.llz_iir_filter_uninit(f)
        return y, tail[:N]

    def stft_analysis(self, overlap_hint, frame_len, win, x):
        x = _f64(x)
        h = self.lib.llz_analysis_fft_init(overlap_hint, frame_len, win)
        bins = (frame_len << (2 if overlap_hint == 0 else 1)) // 2 + 1
        frames = len(x) // frame_len
        re, im = np.zeros((frames, bins)), np.zeros((frames, bins))
        for f in range(frames):
            xi = np.ascontiguousarray(x[f * frame_len:(f + 1) * frame_len])
            r, i = np.zeros(bins), np.zeros(bins)
            self.lib.llz_analysis_fft(h, _ptr(xi), _ptr(r), _ptr(i))
            re[f], im[f] = r, i
        self.lib.llz_analysis_fft_uninit(h)
        return re, im

    def stft_synthesis(self, overlap_hint, frame_len, win, re, im):
        re, im = np.ascontiguousarray(re, dtype=np.float64), np.ascontiguousarray(im, dtype=np.float64)
        h = self.lib.llz_synthesis_fft_init(overlap_hint, frame_len, win)
        x = np.zeros(re.shape[0] * frame_len)
        for f in range(re.shape[0]):
            r, i, xo = np.ascontiguousarray(re[f]), np.ascontiguousarray(im[f]), np.zeros(frame_len)
            self.lib.llz_synthesis_fft(h, _ptr(r), _ptr(i), _ptr(xo))
            x[f * frame_len:(f + 1) * frame_len] = xo
        self.lib.llz_synthesis_fft_uninit(h)
        return x

    def mdct_window(self, win, n, alpha=6.0):
        w = np.zeros(n)
        if win == 0:
            self.lib.llz_mdct_sine(_ptr(w), n)
        else:
            self.lib.llz_mdct_kbd(_ptr(w), n, alpha)
        return w

    def mdct(self, type_, x):
        x = _f64(x)
        h = self.lib.llz_mdct_init(type_, len(x))
        X = np.zeros(len(x) // 2)
        self.lib.llz_mdct(h, _ptr(x), _ptr(X))
        self.lib.llz_mdct_uninit(h)
        return X

    def imdct(self, type_, X):
        X = _f64(X)
        h = self.lib.llz_mdct_init(type_, 2 * len(X))
        x = np.zeros(2 * len(X))
        self.lib.llz_imdct(h, _ptr(X), _ptr(x))
        self.lib.llz_mdct_uninit(h)
        return x

    def mdct_fixed(self, type_, x, inverse=False):
        x = np.ascontiguousarray(x, dtype=np.int32)
        n = 2 * len(x) if inverse else len(x)
        h = self.lib.llz_mdct_fixed_init(type_, n)
        out = np.zeros(n if inverse else n // 2, dtype=np.int32)
        (self.lib.llz_imdct_fixed if inverse else self.lib.llz_mdct_fixed)(h, x.ctypes.data_as(_ip), out.ctypes.data_as(_ip))
        self.lib.llz_mdct_fixed_uninit(h)
        return out

    def mdct_frames(self, frame_len, win, x):
        x = _f64(x)
        a = self.lib.llz_analysis_mdct_init(frame_len, win)
        s = self.lib.llz_synthesis_mdct_init(frame_len, win)
        frames = len(x) // frame_len
        X, y = np.zeros((frames, frame_len)), np.zeros(frames * frame_len)
        for f in range(frames):
            xi, Xi, yi = np.ascontiguousarray(x[f * frame_len:(f + 1) * frame_len]), np.zeros(frame_len), np.zeros(frame_len)
            self.lib.llz_analysis_mdct(a, _ptr(xi), _ptr(Xi))
            self.lib.llz_synthesis_mdct(s, _ptr(Xi), _ptr(yi))
            X[f], y[f * frame_len:(f + 1) * frame_len] = Xi, yi
        self.lib.llz_analysis_mdct_uninit(a)
        self.lib.llz_synthesis_mdct_uninit(s)
        return X, y

    def _rs_open(self, mode, L, M, gain, win):
        if mode == 0:
            return self.lib.llz_decimate_init(M, gain, win), self.lib.llz_decimate, self.lib.llz_decimate_uninit
        if mode == 1:
            return self.lib.llz_interp_init(L, gain, win), self.lib.llz_interp, self.lib.llz_interp_uninit
        return (self.lib.llz_resample_filter_init(L, M, gain, win), self.lib.llz_resample,
                self.lib.llz_resample_filter_uninit)

    def rs_bytes_in(self, mode, L, M, gain, win):
        h, _, un = self._rs_open(mode, L, M, gain, win)
        if h == C.c_ulong(-1).value:
            return None
        b = self.lib.llz_get_resample_framelen_bytes(h)
        un(h)
        return b

    def rs_stream(self, mode, L, M, gain, win, pcm, pad_tail=0):
        """pad_tail: extra zero int16 samples placed behind each frame in the buffer handed to the
        reference, so llz_interp's read past the frame end hits defined zeros."""
        h, run, un = self._rs_open(mode, L, M, gain, win)
        if h == C.c_ulong(-1).value:
            return None
        nin = self.lib.llz_get_resample_framelen_bytes(h) // 2
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        frames = len(pcm) // nin
        outs = []
        ob = C.c_int(0)
        for f in range(frames):
            xi = np.zeros(nin + pad_tail, dtype=np.int16)
            xi[:nin] = pcm[f * nin:(f + 1) * nin]
            yo = np.zeros(nin * 16 + 16, dtype=np.int16)
            run(h, xi.ctypes.data, 2 * nin, yo.ctypes.data, C.byref(ob))
            outs.append(yo[:ob.value // 2].copy())
        un(h)
        return np.concatenate(outs) if outs else np.zeros(0, dtype=np.int16)

    def autocorr(self, x, p):
        x = _f64(x)
        r = np.zeros(p + 1)
        self.lib.llz_autocorr(_ptr(x), len(x), p, _ptr(r))
        return r

    def crosscorr(self, x, y, p):
        x, y = _f64(x), _f64(y)
        r = np.zeros(p + 1)
        self.lib.llz_crosscorr(_ptr(x), _ptr(y), len(x), p, _ptr(r))
        return r

    def corr_cof(self, a, b):
        a, b = _f64(a), _f64(b)
        return self.lib.llz_corr_cof(_ptr(a), _ptr(b), len(a))

    def autocorr_fast(self, x, p):
        x = _f64(x)
        h = self.lib.llz_autocorr_fast_init(len(x))
        r = np.zeros(p + 1)
        self.lib.llz_autocorr_fast(h, _ptr(x), len(x), p, _ptr(r))
        self.lib.llz_autocorr_fast_uninit(h)
        return r

    def fft(self, data, inverse=False):
        z = np.ascontiguousarray(data, dtype=np.complex128).copy()
        f = self.lib.llz_fft_init(len(z))
        (self.lib.llz_ifft if inverse else self.lib.llz_fft)(f, _ptr(z.view(np.float64)))
        self.lib.llz_fft_uninit(f)
        return z

    def fft_fixed(self, data, inverse=False):
        z = np.ascontiguousarray(data, dtype=np.int32).copy()
        f = self.lib.llz_fft_fixed_init(len(z) // 2)
        (self.lib.llz_ifft_fixed if inverse else self.lib.llz_fft_fixed)(f, _ptr(z, _ip))
        self.lib.llz_fft_fixed_uninit(f)
        return z
