// fir_ols.hip -- K4: overlap-save long FIR (up to 257 taps) with a 1024-point complex FFT that never leaves the CU.
//
// New functionality relative to the reference (llz_fir.c is time-domain only); its end-to-end
// oracle is the time-domain llz_fir_filter (llz_fir.c:547-584), its FFT stage follows the sign/scale
// convention of llz_fft.c:61-130,142-198 (forward e^{-j}, unscaled; inverse e^{+j}, divided by N -- the 1/N is
// folded into the filter spectrum on the host).
//
// Why: 257 taps in the time domain is 514 flop per 8 B of HBM traffic -> VALU-bound at ~30 % of the HBM roofline.
// Overlap-save costs ~55 flop per sample and is HBM-bound.
//
// Mapping to CDNA4 (wave64):
//   * job = one channel x 1536 new samples. Two consecutive 1024-sample blocks (each: 256 overlap + 768 new)
//     are packed as real and imaginary part of ONE complex 1024-point transform -- the filter is real, so
//     IFFT(FFT(xa + j xb) H) = ya + j yb and nothing has to be untangled.
//   * a half-wave (32 lanes) owns a job: 1024 = 32 x 32, every lane keeps 32 complex values in registers.
//     A transform is two passes of 32-point in-register FFTs (constant twiddles fold into the code) with one
//     32x32 transpose through LDS between them (row pitch 33 floats: conflict-free ds_write_b32/ds_read_b32).
//     Forward leaves bins digit-reversed in the register index, which is free (register renaming), so
//     FFT -> multiply by H -> IFFT needs no permutation pass at all.
//   * per workgroup (4 waves): LDS = 8 KB inter-pass twiddles W_1024^(a*b) + 8 KB filter spectrum + 4 x 8.25 KB
//     transpose buffers = 49 KB -> 3 workgroups (12 waves) per CU.
//   * waves are independent (no workgroup barrier after the table load) and walk the job list with a grid
//     stride; neighbouring jobs of a channel sit in neighbouring waves so the 256-sample halo re-read is an L2 hit.
// HBM traffic: 4 B read + 4 B written per sample (+1/6 halo re-read served by L2).
#include "common.hpp"

namespace {

struct cf {
    float x, y;
};

__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }

// a * w (CONJ = false) or a * conj(w) (CONJ = true)
template <bool CONJ>
__device__ __forceinline__ cf cmul(cf a, cf w)
{
    if (CONJ) return {__builtin_fmaf(a.y, w.y, a.x * w.x), __builtin_fmaf(-a.x, w.y, a.y * w.x)};
    return {__builtin_fmaf(-a.y, w.y, a.x * w.x), __builtin_fmaf(a.x, w.y, a.y * w.x)};
}

// cos(2*pi*q/32), q = 0..8
__device__ constexpr float kCos32[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f,
                                        0.83146961230254523708f, 0.70710678118654752440f,
                                        0.55557023301960222474f, 0.38268343236508977173f,
                                        0.19509032201612826785f, 0.0f};

// d * W32^q with W32 = exp(-2*pi*j/32) (INV: exp(+2*pi*j/32)), q in 0..15 known at compile time after unrolling
template <bool INV>
__device__ __forceinline__ cf tw32(cf d, int q)
{
    if (q == 0) return d;
    if (q == 8) return INV ? cf{-d.y, d.x} : cf{d.y, -d.x};
    constexpr float r = 0.70710678118654752440f;
    if (q == 4) return INV ? cf{(d.x - d.y) * r, (d.x + d.y) * r} : cf{(d.x + d.y) * r, (d.y - d.x) * r};
    if (q == 12) return INV ? cf{-(d.x + d.y) * r, (d.x - d.y) * r} : cf{(d.y - d.x) * r, -(d.x + d.y) * r};
    const float c = q <= 8 ? kCos32[q] : -kCos32[16 - q];
    const float s = q <= 8 ? kCos32[8 - q] : kCos32[q - 8];
    // forward: d*(c - js); inverse: d*(c + js)
    return cmul<!INV>(d, cf{c, s});
}

__device__ constexpr int brev5(int r)
{
    return ((r & 1) << 4) | ((r & 2) << 2) | (r & 4) | ((r & 8) >> 2) | ((r & 16) >> 4);
}

// 32-point radix-2 decimation-in-frequency FFT on registers; natural order in, v[r] = X[brev5(r)] out
template <bool INV>
__device__ __forceinline__ void fft32(cf (&v)[32])
{
#pragma unroll
    for (int span = 32; span >= 2; span >>= 1) {
        const int half = span >> 1;
        const int tstep = 32 / span;
#pragma unroll
        for (int blk = 0; blk < 32; blk += span) {
#pragma unroll
            for (int q = 0; q < half; q++) {
                const cf a = v[blk + q], b = v[blk + q + half];
                v[blk + q] = cadd(a, b);
                v[blk + q + half] = tw32<INV>(csub(a, b), q * tstep);
            }
        }
    }
}

constexpr int OLS_N = 1024;
constexpr int OLS_OVERLAP = 256;
constexpr int OLS_VALID = OLS_N - OLS_OVERLAP;     // 768
constexpr int OLS_JOB = 2 * OLS_VALID;             // 1536 new samples per complex transform
constexpr int OLS_WAVES = 4;
constexpr int OLS_THREADS = 64 * OLS_WAVES;
constexpr int OLS_PITCH = 33;
constexpr int OLS_XBUF = 32 * OLS_PITCH;           // floats per job transpose buffer (one plane)

// 32x32 transpose of one float plane inside a half-wave: lane l5 writes its 32 registers down a column,
// then reads its row. reg index r of the source is stored at row brev5(r) (undoing the FFT's output order).
#define OLS_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <bool INV>
__device__ __forceinline__ void transpose_twiddle(cf (&v)[32], float *buf, const float2 *__restrict__ tw, int l5)
{
    // inter-pass twiddle W_1024^(+-brev5(r)*l5) applied on the way out, then two single-plane transposes
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const float2 w = tw[brev5(r) * 32 + l5];
        v[r] = cmul<INV>(v[r], cf{w.x, w.y});
    }
#pragma unroll
    for (int r = 0; r < 32; r++) buf[brev5(r) * OLS_PITCH + l5] = v[r].x;
    OLS_WAVE_SYNC();
#pragma unroll
    for (int cidx = 0; cidx < 32; cidx++) v[cidx].x = buf[l5 * OLS_PITCH + cidx];
    OLS_WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 32; r++) buf[brev5(r) * OLS_PITCH + l5] = v[r].y;
    OLS_WAVE_SYNC();
#pragma unroll
    for (int cidx = 0; cidx < 32; cidx++) v[cidx].y = buf[l5 * OLS_PITCH + cidx];
    OLS_WAVE_SYNC();
}

__global__ void __launch_bounds__(OLS_THREADS)
k_fir_ols_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
              const float2 *__restrict__ hfreq, const float2 *__restrict__ twid, int channels, int n,
              long in_pitch, long out_pitch, int flt_len, int jobs_per_channel, long total_jobs)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_tw = reinterpret_cast<float2 *>(smem);           // [32][32]  W_1024^(a*b)
    float2 *s_h = s_tw + 1024;                                  // [1024]    FFT(taps)/1024, natural bins
    float *s_x = reinterpret_cast<float *>(s_h + 1024);         // per half-wave transpose buffers

    for (int i = threadIdx.x; i < 1024; i += OLS_THREADS) {
        s_tw[i] = twid[i];
        s_h[i] = hfreq[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const int l5 = lane & 31;
    float *buf = s_x + (wave * 2 + half) * OLS_XBUF;

    const long waves_total = (long)gridDim.x * OLS_WAVES;
    const long pairs = (total_jobs + 1) >> 1;
    const int keep = flt_len - 1;

    for (long pair = (long)blockIdx.x * OLS_WAVES + wave; pair < pairs; pair += waves_total) {
        const long job = pair * 2 + half;
        const bool live = job < total_jobs;                    // odd job count: the last upper half idles
        const int c = live ? (int)(job / jobs_per_channel) : 0;
        const int j = live ? (int)(job - (long)c * jobs_per_channel) : 0;
        const int s = j * OLS_JOB;                             // first new sample of this job
        const float *row = in + (size_t)c * in_pitch;
        float *orow = out + (size_t)c * out_pitch;
        const float *hrow = hist ? hist + (size_t)c * keep : nullptr;

        // block A = samples [s-256, s+768), block B = [s+512, s+1536)
        const int a0 = s - OLS_OVERLAP + l5;
        const int b0 = s + OLS_VALID - OLS_OVERLAP + l5;
        cf v[32];
        // the whole wave takes the unguarded path only when both of its jobs are interior
        const bool safe = live && (s >= OLS_OVERLAP) && (s + OLS_JOB <= n);
        if (__builtin_amdgcn_read_exec() == ~0ull && __all(safe)) {
#pragma unroll
            for (int n1 = 0; n1 < 32; n1++) {
                v[n1].x = row[a0 + 32 * n1];
                v[n1].y = row[b0 + 32 * n1];
            }
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 32; n1++) {
                const int ia = a0 + 32 * n1, ib = b0 + 32 * n1;
                float xa = 0.f, xb = 0.f;
                if (live) {
                    if (ia >= 0) { if (ia < n) xa = row[ia]; }
                    else if (hrow && ia >= -keep) xa = hrow[keep + ia];
                    if (ib >= 0) { if (ib < n) xb = row[ib]; }
                    else if (hrow && ib >= -keep) xb = hrow[keep + ib];
                }
                v[n1].x = xa;
                v[n1].y = xb;
            }
        }

        // ---- forward: pass 1 over n1 (in registers), twiddle + transpose, pass 2 over n2
        fft32<false>(v);
        transpose_twiddle<false>(v, buf, s_tw, l5);
        fft32<false>(v);                                        // v[r] = X[l5 + 32*brev5(r)]

        // ---- filter in the frequency domain (spectrum already carries the 1/1024)
        cf u[32];
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const float2 h = s_h[l5 + 32 * brev5(r)];
            u[brev5(r)] = cmul<false>(v[r], cf{h.x, h.y});     // back to natural k2 order: renaming only
        }

        // ---- inverse: pass over k2, conj twiddle + transpose, pass over k1
        fft32<true>(u);
        transpose_twiddle<true>(u, buf, s_tw, l5);
        fft32<true>(u);                                         // u[r] = y[32*brev5(r) + l5]

        // ---- keep the 768 valid samples of each block: n1 = brev5(r) >= 8
        if (live) {
            const int oa = s + l5 - OLS_OVERLAP;                // + 32*n1
            const int ob = s + OLS_VALID + l5 - OLS_OVERLAP;
            if (s + OLS_JOB <= n) {
#pragma unroll
                for (int r = 0; r < 32; r++) {
                    const int n1 = brev5(r);
                    if (n1 >= 8) {
                        orow[oa + 32 * n1] = u[r].x;
                        orow[ob + 32 * n1] = u[r].y;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 32; r++) {
                    const int n1 = brev5(r);
                    if (n1 >= 8) {
                        if (oa + 32 * n1 < n) orow[oa + 32 * n1] = u[r].x;
                        if (ob + 32 * n1 < n) orow[ob + 32 * n1] = u[r].y;
                    }
                }
            }
        }
    }
}

} // namespace

extern "C" int llzs_fir_ols_f32(const float *in, float *out, const float *hist, const float *hfreq,
                                const float *twid, int channels, int n, long in_pitch, long out_pitch,
                                int flt_len, void *stream)
{
    if (!in || !out || !hfreq || !twid || channels <= 0 || n <= 0 || in_pitch < n || out_pitch < n) {
        llzs_set_error("fir_ols_f32: bad arguments");
        return LLZ_ERR_ARG;
    }
    if (flt_len < 1 || flt_len > LLZS_OLS_MAX_TAPS) {
        llzs_set_error("fir_ols_f32: flt_len %d outside 1..%d", flt_len, LLZS_OLS_MAX_TAPS);
        return LLZ_ERR_RANGE;
    }
    const int jobs_per_channel = (n + OLS_JOB - 1) / OLS_JOB;
    const long total_jobs = (long)jobs_per_channel * channels;
    const long pairs = (total_jobs + 1) / 2;
    const size_t lds_bytes = 2 * 1024 * sizeof(float2) + (size_t)OLS_WAVES * 2 * OLS_XBUF * sizeof(float);
    long blocks = (pairs + OLS_WAVES - 1) / OLS_WAVES;
    const long max_blocks = 256L * 3;            // 3 workgroups per CU by LDS: one resident set, grid stride
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(k_fir_ols_f32, dim3((unsigned)blocks), dim3(OLS_THREADS), lds_bytes, as_stream(stream),
                       in, out, hist, reinterpret_cast<const float2 *>(hfreq),
                       reinterpret_cast<const float2 *>(twid), channels, n, in_pitch, out_pitch, flt_len,
                       jobs_per_channel, total_jobs);
    LLZ_LAUNCH_CHECK("k_fir_ols_f32");
    return LLZ_OK;
}
