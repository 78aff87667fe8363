// resample.hip -- K3: rational L/M resampler, decimator and interpolator for gfx950.
//
// Replaces the hot loops of reference libllzfilter/llz_resample.c:583-603 (llz_resample), :457-483 (llz_decimate)
// and :515-536 (llz_interp).  Indexing is the reference's:  y[i] = gain * sum_{k<Q} x[(i*M)/L - k] * g[i mod L][k].
//
//  k_resample<float,float>    float32 in/out, float accumulate (FMA), no clamp: the batch extension.
//  k_resample<short,double>   int16 in/out: double accumulate in ascending k with a rounded multiply and a rounded
//                             add per tap (no contraction), * gain, clamp to [-32768,32767], truncate toward zero --
//                             bit-exact with the reference for every sample.
//
// One workgroup = one channel x 256 consecutive outputs. The input span those outputs touch ((256*M)/L + Q samples)
// is staged once in LDS with coalesced reads, so HBM sees each input sample about once (+Q/span halo from L2).
// Tap rows come through the vector L1 (one row for L = 1: every lane reads the same address).
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr int RS_THREADS = 256;

template <typename T>
struct rs_traits;
template <>
struct rs_traits<float> {
    typedef float acc_t;
    typedef float tap_t;
};
template <>
struct rs_traits<short> {
    typedef double acc_t;
    typedef double tap_t;
};

__device__ __forceinline__ float rs_finish(float acc, float gain, float *) { return acc * gain; }

__device__ __forceinline__ short rs_finish(double acc, double gain, short *)
{
#pragma clang fp contract(off)
    double y = acc * gain;                        // llz_resample.c:594
    if (y > 32767) y = 32767;
    if (y < -32768) y = -32768;
    return (short)y;                              // C conversion: toward zero (v_cvt_i32_f64 truncates)
}

template <typename T>
__global__ void __launch_bounds__(RS_THREADS)
k_resample(const T *__restrict__ in, T *__restrict__ out, const T *__restrict__ hist,
           const typename rs_traits<T>::tap_t *__restrict__ g, long n_in, long n_out, long in_pitch,
           long out_pitch, int L, int M, int Q, typename rs_traits<T>::acc_t gain, long long i0, long long in0,
           int span_max)
{
#pragma clang fp contract(off)
    typedef typename rs_traits<T>::acc_t acc_t;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *xs = reinterpret_cast<T *>(smem_raw);
    const int c = blockIdx.y;
    const long o0 = (long)blockIdx.x * RS_THREADS;                 // first output (local) of this tile
    const long olast = min(o0 + RS_THREADS, n_out) - 1;
    // local input index of the newest sample an output needs: ((i0+i)*M)/L - in0
    const long pos_first = (long)(((i0 + o0) * M) / L - in0);
    const long pos_last = (long)(((i0 + olast) * M) / L - in0);
    const long base = pos_first - (Q - 1);                         // oldest local input index staged
    const int span = (int)(pos_last - base + 1);                   // <= span_max by construction
    const T *row = in + (size_t)c * in_pitch;
    const T *hrow = hist ? hist + (size_t)c * (Q - 1) : nullptr;
    for (int p = threadIdx.x; p < span; p += RS_THREADS) {
        const long idx = base + p;
        T v = 0;
        if (idx >= 0) {
            if (idx < n_in) v = row[idx];
        } else if (hrow && idx >= -(long)(Q - 1)) {
            v = hrow[(Q - 1) + idx];
        }
        xs[p] = v;
    }
    (void)span_max;
    __syncthreads();
    const long i = o0 + threadIdx.x;
    if (i >= n_out) return;
    const long long gi = i0 + i;
    const int newest = (int)((gi * M) / L - in0 - base);           // LDS index of x[(gi*M)/L]
    const typename rs_traits<T>::tap_t *grow = g + (size_t)(gi % L) * Q;
    acc_t acc = 0;
    if (sizeof(T) == sizeof(float)) {
        for (int k = 0; k < Q; k++) acc = (acc_t)__builtin_fmaf((float)xs[newest - k], (float)grow[k], (float)acc);
    } else {
        for (int k = 0; k < Q; k++) {                              // llz_resample.c:590-592, exact order
            const acc_t prod = (acc_t)xs[newest - k] * (acc_t)grow[k];
            acc = acc + prod;
        }
    }
    out[(size_t)c * out_pitch + i] = rs_finish(acc, gain, (T *)nullptr);
}

// general L/M, float32, with the tap matrix in LDS ("per-phase sub-filters in LDS"): a workgroup owns RSL_R * 256
// consecutive outputs of one channel, stages the whole L x Q matrix once (rows padded to qpad floats, qpad/4 odd, so that
// the 16-byte row reads of 16 consecutive phases fall in distinct banks) next to the input span, and each lane walks
// RSL_R outputs with one ds_read_b128 of taps per four MACs.  The first version fetched every tap through the vector
// memory path per MAC (6-7 % of the HBM roofline).
constexpr int RSL_R = 8;

__global__ void __launch_bounds__(RS_THREADS)
k_resample_f32_lds(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                   const float *__restrict__ g, long n_in, long n_out, long in_pitch, long out_pitch, int L, int M, int Q,
                   float gain, long long i0, long long in0, int qpad, int tiles_per_block)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *gl = reinterpret_cast<float *>(smem_raw);               // [L][qpad], zero beyond Q
    float *xs = gl + (size_t)L * qpad + 8;                          // 8 zero floats in front: the padded taps read there
    const int c = blockIdx.y;
    // the tap matrix is staged ONCE per workgroup and serves tiles_per_block tiles of RSL_R * 256 outputs (with one tile per
    // workgroup a 147 x 47 matrix, 30 KB, was re-read from L2 for every 2048 outputs: more bytes than the signal itself)
    for (int e = threadIdx.x; e < L * qpad; e += RS_THREADS) {
        const int l = e / qpad, k = e - l * qpad;
        gl[e] = k < Q ? g[(size_t)l * Q + k] : 0.f;
    }
    if (threadIdx.x < 8) xs[-8 + (int)threadIdx.x] = 0.f;
    const float *row = in + (size_t)c * in_pitch;
    const float *hrow = hist ? hist + (size_t)c * (Q - 1) : nullptr;
    const int dq = (RS_THREADS * M) / L, dr = (RS_THREADS * M) % L, dph = RS_THREADS % L;
#pragma unroll 1
    for (int tile = 0; tile < tiles_per_block; tile++) {
        const long o0 = ((long)blockIdx.x * tiles_per_block + tile) * (RS_THREADS * RSL_R);
        if (o0 >= n_out) break;                                     // uniform
        const long olast = min(o0 + RS_THREADS * RSL_R, n_out) - 1;
        const long pos_first = (long)(((i0 + o0) * M) / L - in0);
        const long pos_last = (long)(((i0 + olast) * M) / L - in0);
        const long base = pos_first - (Q - 1);
        const int span = (int)(pos_last - base + 1);
        if (tile) __syncthreads();                                  // the previous tile's readers are done with xs
        for (int p = threadIdx.x; p < span; p += RS_THREADS) {
            const long idx = base + p;
            float v = 0.f;
            if (idx >= 0) {
                if (idx < n_in) v = row[idx];
            } else if (hrow && idx >= -(long)(Q - 1)) {
                v = hrow[(Q - 1) + idx];
            }
            xs[p] = v;
        }
        __syncthreads();
        // index arithmetic once per lane and tile, then by increments: output i+256 sits (256*M) input-steps/L further,
        // i.e. the position advances by dq (+1 on a carry of the remainder) and the phase by 256 mod L -- no division per
        // output
        const long i_first = o0 + threadIdx.x;
        if (i_first >= n_out) continue;                             // (every lane still reaches the barriers above)
        const long long gi0 = i0 + i_first;
        const long long num = gi0 * M;
        int pos = (int)(num / L - in0 - base);                      // LDS index of x[(gi*M)/L]
        int rem = (int)(num % L);
        int ph = (int)(gi0 % L);
#pragma unroll 1
        for (int r = 0; r < RSL_R; r++) {
            const long i = i_first + r * RS_THREADS;
            if (i >= n_out) break;
            const float *xp = xs + pos;                             // taps walk backwards from here
            const float *trow = gl + (size_t)ph * qpad;
            float acc = 0.f;
            for (int k = 0; k < qpad; k += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(trow + k);
                acc = __builtin_fmaf(xp[-k], t.x, acc);
                acc = __builtin_fmaf(xp[-k - 1], t.y, acc);
                acc = __builtin_fmaf(xp[-k - 2], t.z, acc);
                acc = __builtin_fmaf(xp[-k - 3], t.w, acc);
            }
            out[(size_t)c * out_pitch + i] = acc * gain;
            pos += dq; rem += dr;
            if (rem >= L) { rem -= L; pos++; }
            ph += dph;
            if (ph >= L) ph -= L;
        }
    }
}

// Small L and M (2:3, 3:2, 3:4, 4:3, 2:1 ...), float32: the kernel above is bound by LDS reads (a sample and a quarter of
// a tap row per MAC).  An L:M resampler is L decimators by M interleaved at the output: output L m + f uses tap row f at
// input position M m + floor(f M / L).  Here a lane computes R consecutive m of one phase at a time: their sample windows
// overlap (they are M apart), so a chunk of K = M R taps needs M(R-1)+K window samples in registers for K R MACs, and the
// tap row of the phase is the same for every lane (LDS broadcast, four taps per read): ~0.3 LDS reads per MAC instead of
// 1.25.  The staged span is laid out column-major in units of M R samples ([M R][SW] floats): window element j of all
// lanes is one conflict-free run, and because the tap count is padded to a multiple of K with zero taps every window
// address is the lane's pointer, stepped back one float per chunk, plus a compile-time offset.
constexpr int RSW_SW = 289;                                         // floats per row: 256 lanes + (Qp + M) / (M R) <= 32, odd

template <int L, int M, int R>
__global__ void __launch_bounds__(RS_THREADS)
k_resample_f32_win(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                   const float *__restrict__ g, long n_in, long n_out, long in_pitch, long out_pitch, int Q, float gain,
                   long long i0, long long in0)
{
    constexpr int RM = M * R, K = RM, W = M * (R - 1) + K, SW = RSW_SW;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *xs = reinterpret_cast<float *>(smem_raw);                // [RM][SW]
    const int c = blockIdx.y, tid = threadIdx.x;
    const int chunks = (Q + K - 1) / K, Qp = chunks * K;
    float *gs = xs + RM * SW;                                       // [L][Qp], zero beyond Q
    // m counts periods of L outputs / M inputs from the start of the stream; this workgroup owns 256 R of them
    const long long m_first = i0 / L;
    const long long mb = m_first + (long long)blockIdx.x * (RS_THREADS * R);
    const long base = (long)(mb * M - in0) - (Qp - 1);              // row index of staged sample 0
    const int span = RS_THREADS * RM + Qp + M;
    const float *row = in + (size_t)c * in_pitch;
    const float *hrow = hist ? hist + (size_t)c * (Q - 1) : nullptr;
    for (int p = tid; p < span; p += RS_THREADS) {
        const long idx = base + p;
        float v = 0.f;
        if (idx >= 0) {
            if (idx < n_in) v = row[idx];
        } else if (hrow && idx >= -(long)(Q - 1)) {
            v = hrow[(Q - 1) + idx];
        }
        xs[(p % RM) * SW + p / RM] = v;
    }
    for (int e = tid; e < L * Qp; e += RS_THREADS) {
        const int f = e / Qp, k = e - f * Qp;
        gs[e] = k < Q ? g[(size_t)f * Q + k] : 0.f;
    }
    __syncthreads();
    // lane tid, period r of it, phase f, tap k read staged sample RM tid + Qp-1 + d_f + M r - k, d_f = floor(f M / L)
    float *orow = out + (size_t)c * out_pitch;
    float res[L][R];
#pragma unroll
    for (int f = 0; f < L; f++) {
        const int df = (f * M) / L;
        const float *wp = xs + tid + (chunks - 1);
        const float *gp = gs + f * Qp;
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = 0.f;
        for (int ch = 0; ch < chunks; ch++, wp--, gp += K) {
            float xw[W];
#pragma unroll
            for (int j = 0; j < W; j++) xw[j] = wp[((j + df) % RM) * SW + (j + df) / RM];
#pragma unroll
            for (int u4 = 0; u4 < K; u4 += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(gp + u4);      // same address in every lane
                const float tk[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int uu = 0; uu < 4; uu++) {
                    const int u = u4 + uu;
#pragma unroll
                    for (int r = 0; r < R; r++) acc[r] = __builtin_fmaf(xw[M * r - u + (K - 1)], tk[uu], acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) res[f][r] = acc[r] * gain;
    }
    // the workgroup's 256 R L outputs are one contiguous run: through LDS (the sample image is no longer needed) so that
    // HBM sees whole lines instead of R L stride-L dword stores per lane (those cost more than the arithmetic)
    __syncthreads();
#pragma unroll
    for (int f = 0; f < L; f++)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int idx = (R * tid + r) * L + f;
            xs[idx + (idx >> 5)] = res[f][r];
        }
    __syncthreads();
    const long long i_block = mb * L - i0;                          // output index of the run's first element; < 0 in tile 0
#pragma unroll
    for (int q = 0; q < R * L; q++) {
        const int idx = q * RS_THREADS + tid;
        const long long i = i_block + idx;
        if (i >= 0 && i < n_out) orow[i] = xs[idx + (idx >> 5)];
    }
}

// int16 in/out, the reference's arithmetic (llz_resample.c:583-603) with the per-tap overhead taken out: the input span
// is converted to double ONCE while it is staged in LDS (the int16 -> double conversion is exact, so converting before
// or after the LDS round trip gives the same operand), and for L = 1 every lane uses the same tap row, which the
// compiler then fetches through the scalar cache (the multiply takes the tap from an SGPR pair).  Per tap that leaves
// ds_read_b64 + v_mul_f64 + v_add_f64 instead of ds_read_i16 + v_cvt_f64_i32 + a vector tap load + mul + add.
// Products and sums are the reference's, in its order: rounded multiply, rounded add, ascending k.
template <bool UNIFORM_TAPS>
__global__ void __launch_bounds__(RS_THREADS)
k_resample_i16_exact(const short *__restrict__ in, short *__restrict__ out, const short *__restrict__ hist,
                     const double *__restrict__ g, long n_in, long n_out, long in_pitch, long out_pitch, int L, int M,
                     int Q, double gain, long long i0, long long in0)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *xs = reinterpret_cast<double *>(smem_raw);
    const int c = blockIdx.y;
    const long o0 = (long)blockIdx.x * RS_THREADS;
    const long olast = min(o0 + RS_THREADS, n_out) - 1;
    const long pos_first = (long)(((i0 + o0) * M) / L - in0);
    const long pos_last = (long)(((i0 + olast) * M) / L - in0);
    const long base = pos_first - (Q - 1);
    const int span = (int)(pos_last - base + 1);
    const short *row = in + (size_t)c * in_pitch;
    const short *hrow = hist ? hist + (size_t)c * (Q - 1) : nullptr;
    for (int p = threadIdx.x; p < span; p += RS_THREADS) {
        const long idx = base + p;
        short v = 0;
        if (idx >= 0) {
            if (idx < n_in) v = row[idx];
        } else if (hrow && idx >= -(long)(Q - 1)) {
            v = hrow[(Q - 1) + idx];
        }
        xs[p] = (double)v;
    }
    __syncthreads();
    const long i = o0 + threadIdx.x;
    if (i >= n_out) return;
    const long long gi = i0 + i;
    const double *xp = xs + (int)((gi * M) / L - in0 - base);          // x[(gi*M)/L]; taps walk backwards from here
    const double *grow = UNIFORM_TAPS ? g : g + (size_t)(gi % L) * Q;
    double acc = 0;
    int k = 0;
    for (; k + 8 <= Q; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double prod = xp[-(k + u)] * grow[k + u];
            acc = acc + prod;
        }
    }
    for (; k < Q; k++) {
        const double prod = xp[-k] * grow[k];
        acc = acc + prod;
    }
    out[(size_t)c * out_pitch + i] = rs_finish(acc, gain, (short *)nullptr);
}

// history update for int16 (float uses k_fir_tail_f32): hist_new = last `keep` of concat(hist_old, in[0:n])
__global__ void __launch_bounds__(256)
k_tail_i16(const short *__restrict__ in, const short *__restrict__ hist_old, short *__restrict__ hist_new, long n,
           long in_pitch, int keep)
{
    const int c = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= keep) return;
    const long idx = n - keep + j;
    hist_new[(size_t)c * keep + j] = idx >= 0 ? in[(size_t)c * in_pitch + idx]
                                              : hist_old[(size_t)c * keep + (keep + idx)];
}

// llz_decimate inner loops (llz_resample.c:457-483): buf = [n history samples | num_in new samples];
// y_i = sum_{m<M} sum_{k<K} buf[i*M + m + M*k] * p[m][k], m outer, k inner, one rounded multiply and add per term
__global__ void __launch_bounds__(256)
k_decimate_i16(const short *__restrict__ buf, short *__restrict__ out, const double *__restrict__ p, int M, int K,
               int num_out, double gain)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= num_out) return;
    const short *x = buf + (size_t)i * M;
    double y = 0.0;
    for (int m = 0; m < M; m++)
        for (int k = 0; k < K; k++) {
            const double prod = (double)x[m + M * k] * p[(size_t)m * K + k];
            y = y + prod;
        }
    out[i] = rs_finish(y, gain, (short *)nullptr);
}

// llz_interp inner loops (llz_resample.c:515-536): x has K zero samples behind the frame (the reference reads
// past its input there); out[i*L + (L-1-m)] = clamp(gain * sum_k x[i+k] * p[m][k])
__global__ void __launch_bounds__(256)
k_interp_i16(const short *__restrict__ x, short *__restrict__ out, const double *__restrict__ p, int L, int K,
             int num_in, double gain)
{
#pragma clang fp contract(off)
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= num_in * L) return;
    const int i = t / L, m = t - i * L;
    double y = 0.0;
    for (int k = 0; k < K; k++) {
        const double prod = (double)x[i + k] * p[(size_t)m * K + k];
        y = y + prod;
    }
    out[(size_t)i * L + (L - 1 - m)] = rs_finish(y, gain, (short *)nullptr);
}


// ---------------------------------------------------------------------------------------------------------------
// L = 1 fast path (decimation by M, BASELINE config 5: 48 kHz -> 16 kHz): polyphase in LDS.
//   y[i] = sum_k g[k] x[iM-k] = sum_{m<M} sum_j g[jM+m] * x_m[i-j],   x_m[n] = x[nM-m]
// The input tile is de-interleaved into its M phase planes while it is staged (coalesced dword reads, one
// multiply-high per element to split the index), after which every phase is an ordinary stride-1 FIR and reuses the
// register sliding window of fir_td.hip: 64 FMAs per two ds_read_b128.  Taps are wave-uniform (scalar cache).
#ifndef LLZ_DEC_THREADS
#define LLZ_DEC_THREADS 256
#endif
constexpr int DEC_R = 8;
constexpr int DEC_THREADS = LLZ_DEC_THREADS;        // (64- and 128-thread workgroups measured 5 % slower: more halo per tile)

constexpr int DEC_TILE = DEC_R * DEC_THREADS;       // outputs per workgroup

__device__ __forceinline__ int dec_phys(int p) { return p + ((p >> 3) << 2); }   // same padding as fir_td.hip

// One workgroup = one channel x 2048 outputs (a persistent, register-prefetching variant was tried and dropped:
// hipcc spilled the prefetch registers at every launch bound and ran 1.7x slower).
// Staging: lane t takes the M consecutive samples of decimated positions q = t + 256 j, i.e. input indices
// first + M*q + r (r = 0..M-1): one contiguous M-dword load per lane (a wave covers 64*M*4 contiguous bytes), sample r
// belongs to phase M-1-r at plane position q, and since 256 j is a multiple of 8 the padded LDS address is
// phys(t) + 384 j -- no division, constant offsets (the first version spent 12 VALU instructions per element here).
template <int M>
__global__ void __launch_bounds__(DEC_THREADS)
k_resample_dec_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                   const float *__restrict__ gp /* [M][tp] phase taps, zero padded */, long n_in, long n_out,
                   long in_pitch, long out_pitch, int Q, int tp, float gain, int plane_pitch)
{
    extern __shared__ __attribute__((aligned(16))) float lds_dec[];
    const int c = blockIdx.y;
    const int tid = threadIdx.x;
    const long o0 = (long)blockIdx.x * DEC_TILE;
    const long base_n = o0 - tp;                               // decimated index of logical position 0
    const int npos = DEC_TILE + tp;                            // decimated positions staged per phase
    const long first = base_n * M - (M - 1);                   // input index of (q = 0, r = 0)
    const float *row = in + (size_t)c * in_pitch;
    const float *hrow = hist ? hist + (size_t)c * (Q - 1) : nullptr;
    float *dst = lds_dec + dec_phys(tid);
    constexpr int NJ = 12;                                     // positions per lane: covers 2048 + tp up to 3072
    if (first >= 0 && first + (long)npos * M <= n_in) {
        // interior tile: independent, unconditional loads, 4 positions (4*M dwords) in flight per lane
        const float *src = row + first + (long)M * tid;
#pragma unroll
        for (int j0 = 0; j0 < NJ; j0 += 4) {
            if (j0 * DEC_THREADS >= npos) break;
            float v[4 * M];
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int q = min(tid + (j0 + jj) * DEC_THREADS, npos - 1);      // clamp: re-reads the last position
#pragma unroll
                for (int r = 0; r < M; r++) v[jj * M + r] = row[first + (long)M * q + r];
            }
            (void)src;
#pragma unroll
            for (int jj = 0; jj < 4; jj++)
                if (tid + (j0 + jj) * DEC_THREADS < npos) {
#pragma unroll
                    for (int r = 0; r < M; r++) dst[(M - 1 - r) * plane_pitch + 384 * (j0 + jj)] = v[jj * M + r];
                }
        }
    } else {
        for (int q = tid; q < npos; q += DEC_THREADS) {         // first / last tile of a channel
#pragma unroll
            for (int r = 0; r < M; r++) {
                const long idx = first + (long)M * q + r;
                float v = 0.f;
                if (idx >= 0) {
                    if (idx < n_in) v = row[idx];
                } else if (hrow && idx >= -(long)(Q - 1)) {
                    v = hrow[(Q - 1) + idx];
                }
                lds_dec[(M - 1 - r) * plane_pitch + dec_phys(q)] = v;
            }
        }
    }
    __syncthreads();

    float acc[DEC_R];
#pragma unroll
    for (int r = 0; r < DEC_R; r++) acc[r] = 0.f;
    const int p0 = tp + tid * DEC_R;
    for (int m = 0; m < M; m++) {
        const float *plane = lds_dec + m * plane_pitch;
        const float *taps = gp + m * tp;
        float wa[8], wb[8];
        auto load8 = [&](float (&w)[8], int p) {
            const float4 a = *reinterpret_cast<const float4 *>(&plane[dec_phys(p)]);
            const float4 b = *reinterpret_cast<const float4 *>(&plane[dec_phys(p + 4)]);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
            w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        };
        auto mac8 = [&](const float (&lo)[8], const float (&hi)[8], const float *h8) {
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const float h = h8[kk];
#pragma unroll
                for (int r = 0; r < DEC_R; r++) {
                    const int slot = 8 + r - kk;
                    acc[r] = __builtin_fmaf(h, slot >= 8 ? hi[slot - 8] : lo[slot], acc[r]);
                }
            }
        };
        load8(wa, p0);
        for (int kc = 0; kc < tp; kc += 16) {
            load8(wb, p0 - kc - 8);
            mac8(wb, wa, taps + kc);
            load8(wa, p0 - kc - 16);
            mac8(wa, wb, taps + kc + 8);
        }
    }
    float *orow = out + (size_t)c * out_pitch;
    const long oi = o0 + (long)tid * DEC_R;
    if (oi + DEC_R <= n_out && ((out_pitch & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
        *reinterpret_cast<float4 *>(orow + oi) = make_float4(acc[0] * gain, acc[1] * gain, acc[2] * gain, acc[3] * gain);
        *reinterpret_cast<float4 *>(orow + oi + 4) =
            make_float4(acc[4] * gain, acc[5] * gain, acc[6] * gain, acc[7] * gain);
    } else {
#pragma unroll
        for (int r = 0; r < DEC_R; r++)
            if (oi + r < n_out) orow[oi + r] = acc[r] * gain;
    }
}

template <typename T>
int launch_resample(const T *in, T *out, const T *hist, const typename rs_traits<T>::tap_t *g, int channels,
                    long n_in, long n_out, long in_pitch, long out_pitch, int L, int M, int Q,
                    typename rs_traits<T>::acc_t gain, long long i0, long long in0, void *stream, const char *name)
{
    if (!in || !out || !g || channels <= 0 || channels > 65535 || n_in <= 0 || n_out <= 0 || L < 1 || M < 1 ||
        Q < 1 || in_pitch < n_in || out_pitch < n_out) {
        llzs_set_error("%s: bad arguments", name);
        return LLZ_ERR_ARG;
    }
    // newest-input positions of 256 consecutive outputs span at most ceil(255*M/L)+1 samples
    const int span_max = (int)((255L * M + L - 1) / L) + 1 + Q;
    const size_t lds = (size_t)span_max * sizeof(T);
    if (lds > 160 * 1024) {
        llzs_set_error("%s: L=%d M=%d Q=%d needs %zu B of LDS", name, L, M, Q, lds);
        return LLZ_ERR_RANGE;
    }
    if (lds > 64 * 1024)
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resample<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)((n_out + RS_THREADS - 1) / RS_THREADS), (unsigned)channels);
    hipLaunchKernelGGL(k_resample<T>, grid, dim3(RS_THREADS), lds, as_stream(stream), in, out, hist, g, n_in,
                       n_out, in_pitch, out_pitch, L, M, Q, gain, i0, in0, span_max);
    LLZ_LAUNCH_CHECK(name);
    return LLZ_OK;
}

} // namespace

extern "C" int llzs_resample_f32(const float *in, float *out, const float *hist, const float *g, int channels,
                                 long n_in, long n_out, long in_pitch, long out_pitch, int L, int M, int Q,
                                 float gain, long long i0, long long in0, void *stream)
{
    if (in && out && g && channels > 0 && channels <= 65535 && n_in > 0 && n_out > 0 && L >= 1 && M >= 1 && Q >= 1 &&
        in_pitch >= n_in && out_pitch >= n_out && llzs_tune(LLZS_TUNE_RS_GENERIC) < 1) {
        // small L, M: the register-window kernel (R periods per lane; M R = 8 .. 24 samples per lane and tile)
#define LLZ_RSW_TRY(LL, MM, RR)                                                                                      \
    if (L == LL && M == MM) {                                                                                        \
        const int RM = MM * RR, Qp = (Q + RM - 1) / RM * RM;                                                         \
        if ((Qp + MM + RM - 1) / RM <= RSW_SW - 1 - RS_THREADS) {                                                    \
            size_t lds = ((size_t)RM * RSW_SW + (size_t)LL * Qp) * sizeof(float);                                    \
            const size_t lds_out = ((size_t)RS_THREADS * RR * LL * 33 / 32 + 8) * sizeof(float);   /* the output run */ \
            if (lds < lds_out) lds = lds_out;                                                                        \
            const long long m_first = i0 / LL, m_last = (i0 + n_out - 1) / LL;                                       \
            const long tiles = (long)((m_last - m_first) / (RS_THREADS * RR) + 1);                                   \
            if (lds > 64 * 1024)                                                                                     \
                LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resample_f32_win<LL, MM, RR>),    \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
            hipLaunchKernelGGL((k_resample_f32_win<LL, MM, RR>), dim3((unsigned)tiles, (unsigned)channels),          \
                               dim3(RS_THREADS), lds, as_stream(stream), in, out, hist, g, n_in, n_out, in_pitch,    \
                               out_pitch, Q, gain, i0, in0);                                                         \
            LLZ_LAUNCH_CHECK("k_resample_f32_win");                                                                  \
            return LLZ_OK;                                                                                           \
        }                                                                                                            \
    }
        LLZ_RSW_TRY(2, 3, 8) LLZ_RSW_TRY(3, 2, 8) LLZ_RSW_TRY(3, 4, 4) LLZ_RSW_TRY(4, 3, 8)
        LLZ_RSW_TRY(2, 1, 8) LLZ_RSW_TRY(3, 1, 8) LLZ_RSW_TRY(4, 1, 8)
#undef LLZ_RSW_TRY
    }
    if (in && out && g && channels > 0 && channels <= 65535 && n_in > 0 && n_out > 0 && L >= 1 && M >= 1 && Q >= 1 &&
        in_pitch >= n_in && out_pitch >= n_out) {
        int qpad = (Q + 3) & ~3;
        if (((qpad >> 2) & 1) == 0) qpad += 4;                       // qpad/4 odd: conflict-free 16-byte row reads
        const int span_max = (int)(((long)(RS_THREADS * RSL_R - 1) * M + L - 1) / L) + 1 + Q;
        const size_t lds = ((size_t)L * qpad + 8 + span_max) * sizeof(float);
        if (lds <= 64 * 1024 && llzs_tune(LLZS_TUNE_RS_GENERIC) < 2) {
            // tiles per workgroup: enough to amortise the tap matrix (its size in tile-spans, x4), while the grid keeps
            // at least ~8 workgroups per CU
            const long tiles = (n_out + RS_THREADS * RSL_R - 1) / (RS_THREADS * RSL_R);
            long tpb = (4L * L * qpad + span_max - 1) / span_max;
            const long cap = tiles * channels / 2048;
            if (tpb > cap) tpb = cap;
            if (tpb < 1) tpb = 1;
            if (const int v = llzs_tune(LLZS_TUNE_RS_TILES); v >= 1) tpb = v;
            dim3 grid((unsigned)((tiles + tpb - 1) / tpb), (unsigned)channels);
            hipLaunchKernelGGL(k_resample_f32_lds, grid, dim3(RS_THREADS), lds, as_stream(stream), in, out, hist, g, n_in,
                               n_out, in_pitch, out_pitch, L, M, Q, gain, i0, in0, qpad, (int)tpb);
            LLZ_LAUNCH_CHECK("k_resample_f32_lds");
            return LLZ_OK;
        }
    }
    return launch_resample<float>(in, out, hist, g, channels, n_in, n_out, in_pitch, out_pitch, L, M, Q, gain, i0,
                                  in0, stream, "k_resample<float>");
}

extern "C" int llzs_resample_i16(const short *in, short *out, const short *hist, const double *g, int channels,
                                 long n_in, long n_out, long in_pitch, long out_pitch, int L, int M, int Q,
                                 double gain, long long i0, long long in0, void *stream)
{
    if (!in || !out || !g || channels <= 0 || channels > 65535 || n_in <= 0 || n_out <= 0 || L < 1 || M < 1 ||
        Q < 1 || in_pitch < n_in || out_pitch < n_out) {
        llzs_set_error("k_resample_i16_exact: bad arguments");
        return LLZ_ERR_ARG;
    }
    const int span_max = (int)((255L * M + L - 1) / L) + 1 + Q;
    const size_t lds = (size_t)span_max * sizeof(double);
    if (lds > 160 * 1024)                                           // very long spans: the int16-staged kernel
        return launch_resample<short>(in, out, hist, g, channels, n_in, n_out, in_pitch, out_pitch, L, M, Q, gain, i0,
                                      in0, stream, "k_resample<short>");
    dim3 grid((unsigned)((n_out + RS_THREADS - 1) / RS_THREADS), (unsigned)channels);
    if (L == 1) {
        if (lds > 64 * 1024)
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resample_i16_exact<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_resample_i16_exact<true>, grid, dim3(RS_THREADS), lds, as_stream(stream), in, out, hist, g,
                           n_in, n_out, in_pitch, out_pitch, L, M, Q, gain, i0, in0);
    } else {
        if (lds > 64 * 1024)
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resample_i16_exact<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_resample_i16_exact<false>, grid, dim3(RS_THREADS), lds, as_stream(stream), in, out, hist, g,
                           n_in, n_out, in_pitch, out_pitch, L, M, Q, gain, i0, in0);
    }
    LLZ_LAUNCH_CHECK("k_resample_i16_exact");
    return LLZ_OK;
}

extern "C" int llzs_tail_i16(const short *in, const short *hist_old, short *hist_new, int channels, long n,
                             long in_pitch, int keep, void *stream)
{
    if (keep <= 0) return LLZ_OK;
    if (!in || !hist_old || !hist_new || channels <= 0 || channels > 65535 || n <= 0) {
        llzs_set_error("tail_i16: bad arguments");
        return LLZ_ERR_ARG;
    }
    dim3 grid((unsigned)((keep + 255) / 256), (unsigned)channels);
    hipLaunchKernelGGL(k_tail_i16, grid, dim3(256), 0, as_stream(stream), in, hist_old, hist_new, n, in_pitch,
                       keep);
    LLZ_LAUNCH_CHECK("k_tail_i16");
    return LLZ_OK;
}

extern "C" int llzs_decimate_i16(const short *buf, short *out, const double *p, int M, int K, int n, int num_out,
                                 double gain, void *stream)
{
    (void)n;
    if (!buf || !out || !p || M < 1 || K < 1 || num_out < 1) {
        llzs_set_error("decimate_i16: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_decimate_i16, dim3((unsigned)((num_out + 255) / 256)), dim3(256), 0, as_stream(stream),
                       buf, out, p, M, K, num_out, gain);
    LLZ_LAUNCH_CHECK("k_decimate_i16");
    return LLZ_OK;
}

extern "C" int llzs_interp_i16(const short *x, short *out, const double *p, int L, int K, int num_in, double gain,
                               void *stream)
{
    if (!x || !out || !p || L < 1 || K < 1 || num_in < 1) {
        llzs_set_error("interp_i16: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_interp_i16, dim3((unsigned)((num_in * L + 255) / 256)), dim3(256), 0, as_stream(stream),
                       x, out, p, L, K, num_in, gain);
    LLZ_LAUNCH_CHECK("k_interp_i16");
    return LLZ_OK;
}

// phase taps gp[m][j] = g[j*M + m], rows padded with zeros to tp (multiple of 16); see k_resample_dec_f32
extern "C" int llzs_resample_dec_f32(const float *in, float *out, const float *hist, const float *gp, int channels,
                                     long n_in, long n_out, long in_pitch, long out_pitch, int M, int Q, int tp,
                                     float gain, void *stream)
{
    if (!in || !out || !gp || channels <= 0 || channels > 65535 || n_in <= 0 || n_out <= 0 || M < 1 || Q < 1 ||
        tp < 16 || (tp & 15) || in_pitch < n_in || out_pitch < n_out) {
        llzs_set_error("resample_dec_f32: bad arguments");
        return LLZ_ERR_ARG;
    }
    int plane = DEC_TILE + tp;
    plane = plane + (plane >> 3) * 4 + 4;                      // dec_phys image of one plane ...
    while ((plane & 31) != 12) plane += 4;                     // ... phases start 12 banks apart, 16-byte aligned
    const size_t lds = (size_t)plane * M * sizeof(float);
    if (lds > 64 * 1024 || M > 8 || DEC_TILE + tp > 12 * DEC_THREADS) {
        llzs_set_error("resample_dec_f32: M=%d tp=%d needs %zu B of LDS (fast path: 64 KiB, M <= 8)", M, tp, lds);
        return LLZ_ERR_RANGE;
    }
    const unsigned magic = (unsigned)((0x100000000ull + (unsigned)M - 1) / (unsigned)M);
    dim3 grid((unsigned)((n_out + DEC_TILE - 1) / DEC_TILE), (unsigned)channels);
    (void)magic;
#define LLZ_DEC_LAUNCH(MM)                                                                                        \
    hipLaunchKernelGGL(k_resample_dec_f32<MM>, grid, dim3(DEC_THREADS), lds, as_stream(stream), in, out, hist, gp, \
                       n_in, n_out, in_pitch, out_pitch, Q, tp, gain, plane)
    switch (M) {
    case 1: LLZ_DEC_LAUNCH(1); break;
    case 2: LLZ_DEC_LAUNCH(2); break;
    case 3: LLZ_DEC_LAUNCH(3); break;
    case 4: LLZ_DEC_LAUNCH(4); break;
    case 5: LLZ_DEC_LAUNCH(5); break;
    case 6: LLZ_DEC_LAUNCH(6); break;
    case 7: LLZ_DEC_LAUNCH(7); break;
    default: LLZ_DEC_LAUNCH(8); break;
    }
#undef LLZ_DEC_LAUNCH
    LLZ_LAUNCH_CHECK("k_resample_dec_f32");
    return LLZ_OK;
}

extern "C" int llzs_resample_dec_f32_fits(int M, int tp)
{
    int plane = DEC_TILE + tp;
    plane = plane + (plane >> 3) * 4 + 4 + 32;
    return (size_t)plane * M * sizeof(float) <= 64 * 1024 && M <= 8 && DEC_TILE + tp <= 12 * DEC_THREADS;
}
