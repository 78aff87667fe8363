// iir.hip -- K2: IIR kernels for gfx950.
//
//  k_iir_df1_f64_exact   the reference's general direct-form-I recurrence for ONE channel in double, operation for
//                        operation (reference libllzfilter/llz_iir.c:103-132): feed-forward sum first (ascending k,
//                        rounded multiply then rounded add), then the feedback terms subtracted one at a time.
//                        Behind the single-channel llz_iir_filter symbol; a recurrence over one channel has no
//                        parallelism, so this is one lane and exists for drop-in parity, not speed.
//  k_iir_cascade_f32     many channels x cascade of second-order sections ("8-biquad cascade" = 8
//                        chained M=N=2 handles). One lane per channel walks time; float32 I/O is staged through
//                        LDS in 64x64 tiles so HBM sees 256-byte rows although lanes own channels. State and
//                        arithmetic are double (the recurrence amplifies rounding by ~1/(1-r)^2 for pole radius r).
#include "common.hpp"

namespace {

constexpr int IIR_MAX_ORDER = 1024;

__global__ void __launch_bounds__(64)
k_iir_df1_f64_exact(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ a,
                    const double *__restrict__ b, double *__restrict__ xs, double *__restrict__ ys, int M, int N,
                    int n)
{
#pragma clang fp contract(off)
    // circular delay lines; slot of x(t-k) is (t-k) mod (N+1), same for y with M+1
    __shared__ double xh[IIR_MAX_ORDER + 1], yh[IIR_MAX_ORDER + 1], bc[IIR_MAX_ORDER + 1], ac[IIR_MAX_ORDER + 1];
    if (threadIdx.x != 0) return;
    const int PX = N + 1, PY = M + 1;
    // reference layout: x[N] newest ... x[0] oldest (llz_iir.c:117-122). Place x(-1-k) = xs[N-k]... the newest
    // stored input xs[N] is x(-1) seen from this call's first sample t = 0.
    for (int k = 0; k <= N; k++) { bc[k] = b[k]; xh[((-1 - k) % PX + PX) % PX] = xs[N - k]; }
    for (int k = 0; k <= M; k++) { ac[k] = a[k]; yh[((-1 - k) % PY + PY) % PY] = ys[M - k]; }
    int px = 0, py = 0;                                   // slot of time t
    for (int t = 0; t < n; t++) {
        xh[px] = in[t];
        double acc = 0.;
        int s = px;
        for (int k = 0; k <= N; k++) {                    // y += b[k] * x(t-k)
            const double prod = bc[k] * xh[s];
            acc = acc + prod;
            s = s == 0 ? PX - 1 : s - 1;
        }
        s = py == 0 ? PY - 1 : py - 1;
        for (int k = 1; k <= M; k++) {                    // y -= a[k] * y(t-k)
            const double prod = ac[k] * yh[s];
            acc = acc - prod;
            s = s == 0 ? PY - 1 : s - 1;
        }
        yh[py] = acc;
        out[t] = acc;
        px = px + 1 == PX ? 0 : px + 1;
        py = py + 1 == PY ? 0 : py + 1;
    }
    // write the delay lines back in the reference's order: xs[N-k] = x(n-1-k)
    for (int k = 0; k <= N; k++) xs[N - k] = xh[(((n - 1 - k) % PX) + PX) % PX];
    for (int k = 0; k <= M; k++) ys[M - k] = yh[(((n - 1 - k) % PY) + PY) % PY];
}

constexpr int CAS_TILE = 64;            // samples per staged tile (and channels per wave)
constexpr int CAS_PITCH = CAS_TILE + 1; // LDS row pitch: lane-per-row column walks are conflict-free

template <int S>
__global__ void __launch_bounds__(64)
k_iir_cascade_f32(const float *__restrict__ in, float *__restrict__ out, const double *__restrict__ coef,
                  double *__restrict__ state, int channels, int n, long in_pitch, long out_pitch, int stages)
{
    __shared__ float tin[CAS_TILE * CAS_PITCH];
    __shared__ float tout[CAS_TILE * CAS_PITCH];
    const int lane = threadIdx.x;
    const int c0 = blockIdx.x * CAS_TILE;
    const int c = c0 + lane;
    const bool live = c < channels;

    double b0[S], b1[S], b2[S], a1[S], a2[S], x1[S], x2[S], y1[S], y2[S];
#pragma unroll
    for (int s = 0; s < S; s++) {
        const bool on = s < stages;
        b0[s] = on ? coef[5 * s + 0] : 1.0;
        b1[s] = on ? coef[5 * s + 1] : 0.0;
        b2[s] = on ? coef[5 * s + 2] : 0.0;
        a1[s] = on ? coef[5 * s + 3] : 0.0;
        a2[s] = on ? coef[5 * s + 4] : 0.0;
        const double *st = state + ((size_t)(live ? c : 0) * stages + (on ? s : 0)) * 4;
        x1[s] = (on && live) ? st[0] : 0.0;
        x2[s] = (on && live) ? st[1] : 0.0;
        y1[s] = (on && live) ? st[2] : 0.0;
        y2[s] = (on && live) ? st[3] : 0.0;
    }

    for (int t0 = 0; t0 < n; t0 += CAS_TILE) {
        const int len = min(CAS_TILE, n - t0);
        // stage: row r = channel c0+r, lanes sweep time -> 256-byte coalesced reads
        for (int r = 0; r < CAS_TILE; r++) {
            float v = 0.f;
            if (c0 + r < channels && lane < len) v = in[(size_t)(c0 + r) * in_pitch + t0 + lane];
            tin[r * CAS_PITCH + lane] = v;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < len; i++) {
            double v = (double)tin[lane * CAS_PITCH + i];
#pragma unroll
            for (int s = 0; s < S; s++) {
                if (s < stages) {
                    // same association as the chained reference handles: ((b0 x + b1 x1) + b2 x2) - a1 y1 - a2 y2
                    double acc = b0[s] * v;
                    acc = __builtin_fma(b1[s], x1[s], acc);
                    acc = __builtin_fma(b2[s], x2[s], acc);
                    acc = __builtin_fma(-a1[s], y1[s], acc);
                    acc = __builtin_fma(-a2[s], y2[s], acc);
                    x2[s] = x1[s]; x1[s] = v;
                    y2[s] = y1[s]; y1[s] = acc;
                    v = acc;
                }
            }
            tout[lane * CAS_PITCH + i] = (float)v;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        for (int r = 0; r < CAS_TILE; r++)
            if (c0 + r < channels && lane < len)
                out[(size_t)(c0 + r) * out_pitch + t0 + lane] = tout[r * CAS_PITCH + lane];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }

    if (live) {
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (s < stages) {
                double *st = state + ((size_t)c * stages + s) * 4;
                st[0] = x1[s]; st[1] = x2[s]; st[2] = y1[s]; st[3] = y2[s];
            }
        }
    }
}

} // namespace

extern "C" int llzs_iir_df1_f64(const double *in, double *out, const double *a, const double *b, double *xs,
                                double *ys, int M, int N, int n, void *stream)
{
    if (!in || !out || !a || !b || !xs || !ys || M < 0 || N < 0 || n <= 0 || M > IIR_MAX_ORDER ||
        N > IIR_MAX_ORDER) {
        llzs_set_error("iir_df1_f64: bad arguments (M=%d N=%d n=%d, order limit %d)", M, N, n, IIR_MAX_ORDER);
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_iir_df1_f64_exact, dim3(1), dim3(64), 0, as_stream(stream), in, out, a, b, xs, ys, M, N,
                       n);
    LLZ_LAUNCH_CHECK("k_iir_df1_f64_exact");
    return LLZ_OK;
}

extern "C" int llzs_iir_cascade_f32(const float *in, float *out, const double *coef, double *state, int channels,
                                    int n, long in_pitch, long out_pitch, int stages, void *stream)
{
    if (!in || !out || !coef || !state || channels <= 0 || n <= 0 || stages < 1 || stages > 16 ||
        in_pitch < n || out_pitch < n) {
        llzs_set_error("iir_cascade_f32: bad arguments (channels=%d n=%d stages=%d, at most 16 stages)", channels,
                       n, stages);
        return LLZ_ERR_ARG;
    }
    dim3 grid((unsigned)((channels + CAS_TILE - 1) / CAS_TILE));
#define LLZ_CAS_LAUNCH(S)                                                                                       \
    hipLaunchKernelGGL(k_iir_cascade_f32<S>, grid, dim3(64), 0, as_stream(stream), in, out, coef, state,       \
                       channels, n, in_pitch, out_pitch, stages)
    if (stages <= 1) LLZ_CAS_LAUNCH(1);
    else if (stages <= 2) LLZ_CAS_LAUNCH(2);
    else if (stages <= 4) LLZ_CAS_LAUNCH(4);
    else if (stages <= 8) LLZ_CAS_LAUNCH(8);
    else LLZ_CAS_LAUNCH(16);
#undef LLZ_CAS_LAUNCH
    LLZ_LAUNCH_CHECK("k_iir_cascade_f32");
    return LLZ_OK;
}
