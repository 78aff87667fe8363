// iir_df1.hip -- the reference's GENERAL direct-form-I filter (any pole order M, zero order N; reference
// libllzfilter/llz_iir.c:103-132) for many channels at once: float32 in and out, double arithmetic in the reference's
// operation order (feed-forward sum first, ascending k, rounded multiply then rounded add; then the feedback terms subtracted
// one at a time).  The only in-tree caller of llz_iir_filter uses order 3 (libllzaudio/llz_musicpitch.c:1277-1285); a
// second-order-section cascade (iir.hip) cannot stand in for odd or unfactored orders.
//
// Parallelism: a lane owns a (channel, time segment).  Segment 0 of a channel starts from the handle's true delay lines; a
// later segment starts `warm` samples early from zero delay lines and drops those outputs -- `warm` is probed on the host
// (the filter's own recurrence from a unit state: the length after which it stays below 1e-13 of its peak), and a channel is
// only split when its segments are several times longer than that.  With one segment per channel the result is the
// reference's own double sequence, rounded once to float32.
#include "common.hpp"

namespace {

constexpr int DF1_MAX = 8;              // largest order of either side the register form holds

// one lane: n samples of one row from position p0 (outputs before `keep_from` are warm-up and not stored)
template <int ORD>
__global__ void __launch_bounds__(256)
k_iir_df1_mc(const float *__restrict__ in, float *__restrict__ out, const double *__restrict__ ab /* a[0..ORD], b[0..ORD] */,
             const double *__restrict__ state_in, double *__restrict__ state_out /* [channels][2][ORD+1]: xs then ys */,
             int channels, long n, long in_pitch, long out_pitch, int M, int N, int segs, long seg_len, int warm)
{
#pragma clang fp contract(off)
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    if (item >= (long)channels * segs) return;
    const int c = (int)(item / segs), seg = (int)(item - (long)c * segs);
    const float *x = in + (size_t)c * in_pitch;
    float *y = out + (size_t)c * out_pitch;
    double a[ORD + 1], b[ORD + 1], xd[ORD + 1], yd[ORD + 1];      // xd[k] = x(t-1-k), yd[k] = y(t-1-k) in front of sample t
#pragma unroll
    for (int k = 0; k <= ORD; k++) {
        a[k] = k <= M ? ab[k] : 0.0;
        b[k] = k <= N ? ab[DF1_MAX + 1 + k] : 0.0;
        xd[k] = yd[k] = 0.0;
    }
    const long start = seg * seg_len;
    const long stop = seg == segs - 1 ? n : start + seg_len;
    long t = start;
    if (seg == 0) {
        // the reference keeps x[N] newest .. x[0] oldest (llz_iir.c:117-122): xs[N-k] = x(-1-k), ys[M-k] = y(-1-k)
        const double *xs = state_in + (size_t)c * 2 * (DF1_MAX + 1), *ys = xs + (DF1_MAX + 1);
#pragma unroll
        for (int k = 0; k <= ORD; k++) {
            if (k <= N) xd[k] = xs[N - k];
            if (k <= M) yd[k] = ys[M - k];
        }
    } else {
        t = start - warm;                                          // (the host guarantees warm <= seg_len)
    }
    // one sample of the reference's loop (llz_iir.c:103-132)
    auto step = [&](float xf) {
        const double xt = (double)xf;
        double acc = 0.0;
        {
            const double prod = b[0] * xt;                         // y = sum_k b[k] x(t-k), ascending k
            acc = acc + prod;
        }
#pragma unroll
        for (int k = 1; k <= ORD; k++) {
            if (k <= N) {
                const double prod = b[k] * xd[k - 1];
                acc = acc + prod;
            }
        }
#pragma unroll
        for (int k = 1; k <= ORD; k++) {                           // y -= a[k] y(t-k), one term at a time
            if (k <= M) {
                const double prod = a[k] * yd[k - 1];
                acc = acc - prod;
            }
        }
#pragma unroll
        for (int k = ORD; k >= 1; k--) { xd[k] = xd[k - 1]; yd[k] = yd[k - 1]; }
        xd[0] = xt;
        yd[0] = acc;
        return (float)acc;
    };
    // A lane's row is its own: adjacent lanes are a whole segment apart, so a 4-byte access per lane and sample touches 64 cache
    // lines per wave instruction and every line 32 times (measured 8 % of the HBM roofline whatever the order).  The row is
    // walked in blocks of 16 samples instead -- four 16-byte loads, the next block's issued before this block's 16 steps, four
    // 16-byte stores -- with single samples in front (warm-up outputs are dropped one by one, and up to the first 4-sample
    // boundary of the row) and behind.
    for (; t < stop && (t < start || ((t & 3) != 0)); t++) {
        const float yv = step(x[t]);
        if (t >= start) y[t] = yv;
    }
    constexpr int BLK = 16;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(4))) u4 { f32x4 v; };
    if (t + BLK <= stop) {
        f32x4 cur[BLK / 4], nxt[BLK / 4];
#pragma unroll
        for (int q = 0; q < BLK / 4; q++) cur[q] = reinterpret_cast<const u4 *>(x + t + 4 * q)->v;
        for (; t + BLK <= stop; t += BLK) {
            const bool more = t + 2 * BLK <= stop;
            if (more) {
#pragma unroll
                for (int q = 0; q < BLK / 4; q++) nxt[q] = reinterpret_cast<const u4 *>(x + t + BLK + 4 * q)->v;
            }
#pragma unroll
            for (int q = 0; q < BLK / 4; q++) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = step(cur[q][e]);
                reinterpret_cast<u4 *>(y + t + 4 * q)->v = o;
            }
            if (more) {
#pragma unroll
                for (int q = 0; q < BLK / 4; q++) cur[q] = nxt[q];
            }
        }
    }
    for (; t < stop; t++) y[t] = step(x[t]);
    if (seg == segs - 1) {
        double *xs = state_out + (size_t)c * 2 * (DF1_MAX + 1), *ys = xs + (DF1_MAX + 1);
#pragma unroll
        for (int k = 0; k <= ORD; k++) {
            if (k <= N) xs[N - k] = xd[k];
            if (k <= M) ys[M - k] = yd[k];
        }
    }
}

} // namespace

extern "C" int llzs_iir_df1_mc_max_order(void) { return DF1_MAX; }

// ab: DF1_MAX + 1 doubles a[] then DF1_MAX + 1 doubles b[] (zero padded); state_in / state_out: [channels][2][DF1_MAX + 1],
// two different buffers (segment 0 reads the start state while the last segment writes the end state); segs >= 1 time
// segments per channel, each warmed up over `warm` samples (ignored when segs == 1)
extern "C" int llzs_iir_df1_mc_f32(const float *in, float *out, const double *ab, const double *state_in, double *state_out,
                                   int channels, long n, long in_pitch, long out_pitch, int M, int N, int segs, int warm,
                                   void *stream)
{
    if (!in || !out || !ab || !state_in || !state_out || state_in == state_out || channels <= 0 || n <= 0 || in_pitch < n ||
        out_pitch < n || M < 0 || N < 0 || M > DF1_MAX || N > DF1_MAX || segs < 1 || warm < 0) {
        llzs_set_error("iir_df1_mc_f32: bad arguments (channels=%d n=%ld M=%d N=%d segs=%d)", channels, n, M, N, segs);
        return LLZ_ERR_ARG;
    }
    long seg_len = (n + segs - 1) / segs;
    while (segs > 1 && (seg_len < warm || (long)(segs - 1) * seg_len >= n)) {       // every segment non-empty and >= warm
        segs--;
        seg_len = (n + segs - 1) / segs;
    }
    const long items = (long)channels * segs;
    // (ORD: the orders the unrolled loops run to; the tables keep the stride of DF1_MAX)
    if (M <= 4 && N <= 4)
        hipLaunchKernelGGL(k_iir_df1_mc<4>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, as_stream(stream), in, out, ab,
                           state_in, state_out, channels, n, in_pitch, out_pitch, M, N, segs, seg_len, warm);
    else
        hipLaunchKernelGGL(k_iir_df1_mc<DF1_MAX>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, as_stream(stream), in, out,
                           ab, state_in, state_out, channels, n, in_pitch, out_pitch, M, N, segs, seg_len, warm);
    LLZ_LAUNCH_CHECK("k_iir_df1_mc");
    return LLZ_OK;
}
