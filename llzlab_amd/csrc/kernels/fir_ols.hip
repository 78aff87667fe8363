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
#include <stdlib.h>
#include "common.hpp"
#include "fft32.hpp"

#ifndef LLZ_DIAG
#define LLZ_DIAG 0     /* 1 / 2: timing-only builds for ablation (EXTRA_HIPFLAGS=-DLLZ_DIAG=n), never shipped */
#endif

namespace {

constexpr int OLS_N = 1024;
constexpr int OLS_OVERLAP = 256;
constexpr int OLS_VALID = OLS_N - OLS_OVERLAP;     // 768
constexpr int OLS_JOB = 2 * OLS_VALID;             // 1536 new samples per complex transform
#ifndef LLZ_OLS_WAVES
#define LLZ_OLS_WAVES 4
#endif
constexpr int OLS_WAVES = LLZ_OLS_WAVES;
constexpr int OLS_THREADS = 64 * OLS_WAVES;
// everything a half-wave needs to know about its job
struct ols_job {
    const float *row;      // input row of the job's channel
    float *orow;           // output row
    const float *hrow;     // history row (flt_len-1 samples) or nullptr
    int s;                 // first new sample
    bool live;             // false: the idle upper half of an odd last pair
};

__device__ __forceinline__ ols_job ols_locate(long pair, int half, const float *in, float *out, const float *hist,
                                              long in_pitch, long out_pitch, int keep, int jobs_per_channel,
                                              long total_jobs)
{
    ols_job jb;
    const long job = pair * 2 + half;
    jb.live = job < total_jobs;
    const int c = jb.live ? (int)(job / jobs_per_channel) : 0;
    const int j = jb.live ? (int)(job - (long)c * jobs_per_channel) : 0;
    jb.s = j * OLS_JOB;
    jb.row = in + (size_t)c * in_pitch;
    jb.orow = out + (size_t)c * out_pitch;
    jb.hrow = hist ? hist + (size_t)c * keep : nullptr;
    return jb;
}

// block A = samples [s-256, s+768) -> real parts, block B = [s+512, s+1536) -> imaginary parts;
// register n1 of lane l5 holds sample 32*n1 + l5 of each block
__device__ __forceinline__ void ols_load(cf (&v)[32], const ols_job &jb, int l5, int n, int keep)
{
    const int a0 = jb.s - OLS_OVERLAP + l5;
    const int b0 = jb.s + OLS_VALID - OLS_OVERLAP + l5;
    // the whole wave takes the unguarded path only when both of its jobs are interior
    const bool safe = jb.live && (jb.s >= OLS_OVERLAP) && (jb.s + OLS_JOB <= n);
    if (__all(safe)) {
#pragma unroll
        for (int n1 = 0; n1 < 32; n1++) {
            v[n1].x = jb.row[a0 + 32 * n1];
            v[n1].y = jb.row[b0 + 32 * n1];
        }
    } else {
        // edge jobs (first / last of a channel). Step 1, branch-free: every address clamped into the row, samples
        // outside [0, n) zeroed afterwards.
#pragma unroll
        for (int n1 = 0; n1 < 32; n1++) {
            const int ia = a0 + 32 * n1, ib = b0 + 32 * n1;
            const float xa = jb.row[min(max(ia, 0), n - 1)], xb = jb.row[min(max(ib, 0), n - 1)];
            v[n1].x = (jb.live && ia >= 0 && ia < n) ? xa : 0.f;
            v[n1].y = (jb.live && ib >= 0 && ib < n) ? xb : 0.f;
        }
        // Step 2: only the first job of a channel reaches back before the stream start, and only with block A's
        // first 256 samples (registers 0..7): those come from the history row.
        if (jb.live && jb.hrow != nullptr && jb.s < OLS_OVERLAP) {
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) {
                const int ia = a0 + 32 * n1;
                if (ia < 0 && ia >= -keep) v[n1].x = jb.hrow[keep + ia];
            }
        }
    }
}

// FFT -> multiply by the filter spectrum -> IFFT, all in registers + one LDS transpose each way
__device__ __forceinline__ void ols_filter(cf (&v)[32], cf (&u)[32], float *buf, const float2 *s_tw,
                                           const float2 *s_h, int l5)
{
    // ---- forward: pass 1 over n1 (in registers), twiddle + transpose, pass 2 over n2
    fft32<false>(v);
    transpose_twiddle<false>(v, buf, s_tw, l5);
    fft32<false>(v);                                        // v[r] = X[l5 + 32*brev5(r)]
    // ---- filter in the frequency domain (spectrum already carries the 1/1024)
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const float2 h = s_h[l5 + 32 * brev5(r)];
        u[brev5(r)] = cmul<false>(v[r], cf{h.x, h.y});     // back to natural k2 order: renaming only
    }
    // ---- inverse: pass over k2, conj twiddle + transpose, pass over k1
    fft32<true>(u);
    transpose_twiddle<true>(u, buf, s_tw, l5);
    fft32<true>(u);                                         // u[r] = y[32*brev5(r) + l5]
}

#ifndef LLZ_OLS_NT
#define LLZ_OLS_NT 1      /* streaming (non-temporal) output stores: measured +1 % */
#endif
#if LLZ_OLS_NT
#define OLS_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define OLS_ST(p, v) (*(p) = (v))
#endif

// keep the 768 valid samples of each block: n1 = brev5(r) >= 8
__device__ __forceinline__ void ols_store(const cf (&u)[32], const ols_job &jb, int l5, int n)
{
    if (!jb.live) return;
    const int oa = jb.s + l5 - OLS_OVERLAP;                 // + 32*n1
    const int ob = jb.s + OLS_VALID + l5 - OLS_OVERLAP;
    if (jb.s + OLS_JOB <= n) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int n1 = brev5(r);
            if (n1 >= 8) {
                OLS_ST(&jb.orow[oa + 32 * n1], u[r].x);
                OLS_ST(&jb.orow[ob + 32 * n1], u[r].y);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int n1 = brev5(r);
            if (n1 >= 8) {
                if (oa + 32 * n1 < n) jb.orow[oa + 32 * n1] = u[r].x;
                if (ob + 32 * n1 < n) jb.orow[ob + 32 * n1] = u[r].y;
            }
        }
    }
}

// PREFETCH: the next pair's 64 input dwords per lane are requested before the current pair is transformed, so
// HBM latency hides under ~2200 VALU instructions instead of under other waves only (costs 64 VGPRs: 2 waves/SIMD)
template <bool PREFETCH>
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
    long pair = (long)blockIdx.x * OLS_WAVES + wave;
    if (pair >= pairs) return;

    if (PREFETCH) {
        ols_job cur = ols_locate(pair, half, in, out, hist, in_pitch, out_pitch, keep, jobs_per_channel, total_jobs);
        cf nxt[32];
        ols_load(nxt, cur, l5, n, keep);
        while (true) {
            cf v[32], u[32];
#pragma unroll
            for (int r = 0; r < 32; r++) v[r] = nxt[r];
            const long np = pair + waves_total;
            const bool more = np < pairs;
            ols_job nj = cur;
            if (more) {
                nj = ols_locate(np, half, in, out, hist, in_pitch, out_pitch, keep, jobs_per_channel, total_jobs);
                ols_load(nxt, nj, l5, n, keep);
            }
#if LLZ_DIAG == 1    /* memory only: loads feed the stores directly (wrong results; timing build) */
#pragma unroll
            for (int r = 0; r < 32; r++) u[r] = v[brev5(r)];
#else
            ols_filter(v, u, buf, s_tw, s_h, l5);
#endif
            ols_store(u, cur, l5, n);
            if (!more) break;
            cur = nj;
            pair = np;
        }
    } else {
        cf v[32], u[32];
#if LLZ_DIAG == 2
#pragma unroll
        for (int r = 0; r < 32; r++) u[r] = cf{0.f, 0.f};
#endif
        for (; pair < pairs; pair += waves_total) {
            const ols_job jb = ols_locate(pair, half, in, out, hist, in_pitch, out_pitch, keep, jobs_per_channel,
                                          total_jobs);
#if LLZ_DIAG == 2    /* compute only: one load per wave, stores suppressed unless a value is NaN (timing build) */
            if (pair == (long)blockIdx.x * OLS_WAVES + wave) ols_load(v, jb, l5, n, keep);
            else {
#pragma unroll
                for (int r = 0; r < 32; r++) { v[r].x = u[r].x; v[r].y = u[r].y; }
            }
            ols_filter(v, u, buf, s_tw, s_h, l5);
            if (u[0].x != u[0].x) ols_store(u, jb, l5, n);
#elif LLZ_DIAG == 1
            ols_load(v, jb, l5, n, keep);
#pragma unroll
            for (int r = 0; r < 32; r++) u[r] = v[brev5(r)];
            ols_store(u, jb, l5, n);
#else
            ols_load(v, jb, l5, n, keep);
            ols_filter(v, u, buf, s_tw, s_h, l5);
            ols_store(u, jb, l5, n);
#endif
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// "walk" form: a half-wave owns a SEGMENT of consecutive jobs of one channel and carries the 256-sample overlap in
// registers from job to job, so every input sample is requested from memory exactly once:
//   * block A's first 256 samples (registers 0..7, real part) are the previous job's last 256 (kept in 8 VGPRs),
//   * block B's first 256 samples (registers 0..7, imaginary part) are block A's last 256 (registers 24..31),
// which leaves 48 instead of 64 loads per lane and job and removes the halo re-read from L2/HBM altogether.
// Segments are dealt round-robin to half-waves; both halves of a wave run the same number of jobs (a segment that
// is short at the end of a channel idles its tail).
#ifndef LLZ_OLS_SEG
#define LLZ_OLS_SEG 16
#endif
constexpr int OLS_SEG = LLZ_OLS_SEG;                        // jobs per segment (16 x 1536 samples of one channel)

struct ols_raw {                                            // the 1536 new samples of one job, 48 per lane
    float a[24], b[24];
};

#ifndef LLZ_OLS_NTLOAD
#define LLZ_OLS_NTLOAD 1   /* streaming (non-temporal) input loads: measured +0.8 % (6.69 -> 6.635 ms) */
#endif
#if LLZ_OLS_NTLOAD
#define OLS_LD(p) __builtin_nontemporal_load(p)
#else
#define OLS_LD(p) (*(p))
#endif
__device__ __forceinline__ void walk_load(ols_raw &raw, const float *row, int s, int l5, int n, bool live)
{
    // sample index of a[i]: s + 32*i + l5 ; of b[i]: s + 768 + 32*i + l5
    const bool safe = live && (s + OLS_JOB <= n);
    if (__all(safe)) {
#pragma unroll
        for (int i = 0; i < 24; i++) {
            raw.a[i] = OLS_LD(&row[s + 32 * i + l5]);
            raw.b[i] = OLS_LD(&row[s + OLS_VALID + 32 * i + l5]);
        }
    } else {
        // (also taken by the prefetch past the end of a segment: those loads are issued and discarded on purpose --
        // skipping them with a wave-uniform branch measured 8 % SLOWER, 7.4 vs 6.8 ms; they cost 1/16 extra reads)
#pragma unroll
        for (int i = 0; i < 24; i++) {
            const int ia = s + 32 * i + l5, ib = ia + OLS_VALID;
            const float xa = row[min(ia, n - 1)], xb = row[min(ib, n - 1)];
            raw.a[i] = (live && ia < n) ? xa : 0.f;
            raw.b[i] = (live && ib < n) ? xb : 0.f;
        }
    }
}

// the 256 samples in front of a segment: from the row (s > 0) or from the history / zeros (s == 0)
__device__ __forceinline__ void walk_load_halo(float (&halo)[8], const float *row, const float *hrow, int s, int l5,
                                               int keep, bool live)
{
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int idx = s - OLS_OVERLAP + 32 * i + l5;
        float v = 0.f;
        if (live) {
            if (idx >= 0) v = row[idx];
            else if (hrow && idx >= -keep) v = hrow[keep + idx];
        }
        halo[i] = v;
    }
}

template <bool PREFETCH>
__global__ void __launch_bounds__(OLS_THREADS)
k_fir_ols_walk_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                   const float2 *__restrict__ hfreq, const float2 *__restrict__ twid, int channels, int n,
                   long in_pitch, long out_pitch, int flt_len, int jobs_per_channel, int segs_per_channel,
                   long total_segs, int seg_len)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_tw = reinterpret_cast<float2 *>(smem);
    float2 *s_h = s_tw + 1024;
    float *s_x = reinterpret_cast<float *>(s_h + 1024);
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
    const int keep = flt_len - 1;
    const long halves_total = (long)gridDim.x * OLS_WAVES * 2;
    const long first = ((long)blockIdx.x * OLS_WAVES + wave) * 2;          // this wave's first segment pair

    for (long sp = first; sp < total_segs; sp += halves_total) {
        const long seg = sp + half;
        const bool seg_live = seg < total_segs;
        const int c = seg_live ? (int)(seg / segs_per_channel) : 0;
        const int j0 = seg_live ? (int)(seg - (long)c * segs_per_channel) * seg_len : 0;
        const int jcount = seg_live ? min(seg_len, jobs_per_channel - j0) : 0;
        const float *row = in + (size_t)c * in_pitch;
        float *orow = out + (size_t)c * out_pitch;
        const float *hrow = hist ? hist + (size_t)c * keep : nullptr;

        float halo[8];
        walk_load_halo(halo, row, hrow, j0 * OLS_JOB, l5, keep, seg_live);
        ols_raw raw;
        if (PREFETCH) walk_load(raw, row, j0 * OLS_JOB, l5, n, jcount > 0);

#pragma unroll 1
        for (int jj = 0; jj < seg_len; jj++) {
            const int s = (j0 + jj) * OLS_JOB;
            const bool live = jj < jcount;
            if (!__any(live)) break;
            if (!PREFETCH) walk_load(raw, row, s, l5, n, live);
            cf v[32], u[32];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                v[i].x = halo[i];                         // block A [0,256)   = carried overlap
                v[i].y = raw.a[16 + i];                   // block B [0,256)   = block A [768,1024)
                halo[i] = raw.b[16 + i];                  // next overlap      = block B [768,1024)
            }
#pragma unroll
            for (int i = 0; i < 24; i++) {
                v[8 + i].x = raw.a[i];
                v[8 + i].y = raw.b[i];
            }
            if (PREFETCH) walk_load(raw, row, s + OLS_JOB, l5, n, (jj + 1) < jcount);
#if LLZ_DIAG == 1    /* memory only: the loaded samples go straight to the stores (wrong results; timing build) */
#pragma unroll
            for (int r = 0; r < 32; r++) u[r] = v[brev5(r)];
#elif LLZ_DIAG == 2  /* compute only: transform the first job's data over and over, no further loads */
            ols_filter(v, u, buf, s_tw, s_h, l5);
#else
            ols_filter(v, u, buf, s_tw, s_h, l5);
#endif
            ols_job jb;
            jb.row = row; jb.orow = orow; jb.hrow = hrow; jb.s = s; jb.live = live;
            ols_store(u, jb, l5, n);
        }
    }
}


// Chain form: the walk form with the prefetch carried ACROSS segments.  In the walk form the prefetch issued during a
// segment's last job runs past the segment's end and is discarded (1/16 of the input fetched twice), and every segment
// starts with an exposed load of its halo and first job.  Here the last job of a segment prefetches the first job and
// the halo of the half-wave's NEXT segment instead, so every load is used and no segment start waits on memory.
__global__ void __launch_bounds__(OLS_THREADS)
k_fir_ols_chain_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                    const float2 *__restrict__ hfreq, const float2 *__restrict__ twid, int channels, int n,
                    long in_pitch, long out_pitch, int flt_len, int jobs_per_channel, int segs_per_channel,
                    long total_segs, int seg_len)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_tw = reinterpret_cast<float2 *>(smem);
    float2 *s_h = s_tw + 1024;
    float *s_x = reinterpret_cast<float *>(s_h + 1024);
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
    const int keep = flt_len - 1;
    const long halves_total = (long)gridDim.x * OLS_WAVES * 2;
    const long first = ((long)blockIdx.x * OLS_WAVES + wave) * 2;

    struct seginfo {
        bool live;
        int c, j0, jcount;
    };
    auto locate = [&](long seg) {
        seginfo g;
        g.live = seg < total_segs;
        g.c = g.live ? (int)(seg / segs_per_channel) : 0;
        g.j0 = g.live ? (int)(seg - (long)g.c * segs_per_channel) * seg_len : 0;
        g.jcount = g.live ? min(seg_len, jobs_per_channel - g.j0) : 0;
        return g;
    };

    seginfo cur = locate(first + half);
    float halo[8];
    ols_raw raw;
    {
        const float *row = in + (size_t)cur.c * in_pitch;
        walk_load_halo(halo, row, hist ? hist + (size_t)cur.c * keep : nullptr, cur.j0 * OLS_JOB, l5, keep, cur.live);
        walk_load(raw, row, cur.j0 * OLS_JOB, l5, n, cur.jcount > 0);
    }
    for (long sp = first; sp < total_segs; sp += halves_total) {
        const seginfo nxt = locate(sp + halves_total + half);
        const float *row = in + (size_t)cur.c * in_pitch;
        float *orow = out + (size_t)cur.c * out_pitch;
        const float *hrow = hist ? hist + (size_t)cur.c * keep : nullptr;
        const float *nrow = in + (size_t)nxt.c * in_pitch;
        const float *nhrow = hist ? hist + (size_t)nxt.c * keep : nullptr;
        float halo_n[8];
#pragma unroll
        for (int i = 0; i < 8; i++) halo_n[i] = 0.f;

#pragma unroll 1
        for (int jj = 0; jj < seg_len; jj++) {
            const int s = (cur.j0 + jj) * OLS_JOB;
            const bool live = jj < cur.jcount;
            if (!__any(live)) break;
            cf v[32], u[32];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                v[i].x = halo[i];
                v[i].y = raw.a[16 + i];
                halo[i] = raw.b[16 + i];
            }
#pragma unroll
            for (int i = 0; i < 24; i++) {
                v[8 + i].x = raw.a[i];
                v[8 + i].y = raw.b[i];
            }
            // next job of this segment, or (from the segment's last job on) the first job and halo of the next segment
            const bool in_seg = (jj + 1) < cur.jcount;
            walk_load(raw, in_seg ? row : nrow, in_seg ? s + OLS_JOB : nxt.j0 * OLS_JOB, l5, n,
                      in_seg ? true : nxt.jcount > 0);
            if (__any(!in_seg))
                walk_load_halo(halo_n, nrow, nhrow, nxt.j0 * OLS_JOB, l5, keep, !in_seg && nxt.live);
            ols_filter(v, u, buf, s_tw, s_h, l5);
            ols_job jb;
            jb.row = row; jb.orow = orow; jb.hrow = hrow; jb.s = s; jb.live = live;
            ols_store(u, jb, l5, n);
        }
        cur = nxt;
#pragma unroll
        for (int i = 0; i < 8; i++) halo[i] = halo_n[i];
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
    // tuning knobs (measurement only): LLZ_OLS_VARIANT bit 0 = register prefetch, bit 1 = walk form, bit 2 / bit 3 = force /
    // forbid the chain form;
    // LLZ_OLS_WG_PER_CU = resident workgroups per CU the grid is sized for
    static int variant = -1, wg_per_cu = -1;
    if (variant < 0) {
        const char *e = getenv("LLZ_OLS_VARIANT");
        variant = e ? atoi(e) : 3;              /* default: walk form + register prefetch (fastest measured) */
        e = getenv("LLZ_OLS_WG_PER_CU");
        wg_per_cu = e ? atoi(e) : 0;
    }
    const bool prefetch = (variant & 1) != 0;
    const int per_cu = wg_per_cu > 0 ? wg_per_cu : (prefetch ? 2 : 3);
    const long max_blocks = 256L * per_cu;       // one resident set of workgroups, grid stride over the work list
    const float2 *hf = reinterpret_cast<const float2 *>(hfreq), *tw = reinterpret_cast<const float2 *>(twid);
    if (variant >= 2) {
        // chain form (prefetch carried across segments) when a half-wave walks several segments; on a batch that fits one
        // round the walk form is faster (64 ch x 63 taps: 0.14 vs 0.16 ms).  LLZ_OLS_VARIANT bit 2 forces it, bit 3 forbids
        const bool large = (long)((jobs_per_channel + OLS_SEG - 1) / OLS_SEG) * channels >= 4 * max_blocks * OLS_WAVES * 2;
        const bool chain = prefetch && ((variant & 4) != 0 || (large && (variant & 8) == 0));
        // jobs per segment: a half-wave walks seg_len consecutive jobs of one channel (the overlap stays in registers), at
        // most OLS_SEG.  Small batches (BASELINE config 2: 64 channels) would leave half-wave slots idle or quantise badly
        // into rounds with the full length, so take the length that minimises rounds x (length + halo reload)
        const long slots = max_blocks * OLS_WAVES * 2;
        int seg_len = OLS_SEG;
        if ((long)((jobs_per_channel + OLS_SEG - 1) / OLS_SEG) * channels < 4 * slots) {
            // (large batches keep the full length: on 4096 channels a shorter segment measured 2.6 % slower, the start of
            // a segment costs about one job: halo reload and an empty prefetch pipeline)
            double best = 1e300;
            for (int sl = OLS_SEG; sl >= 1; sl--) {
                const long segs = (long)((jobs_per_channel + sl - 1) / sl) * channels;
                const double cost = (double)((segs + slots - 1) / slots) * (sl + 1.0);
                if (cost < best * 0.999) { best = cost; seg_len = sl; }
            }
        }
        const int segs_per_channel = (jobs_per_channel + seg_len - 1) / seg_len;
        const long total_segs = (long)segs_per_channel * channels;
        blocks = (total_segs + 2 * OLS_WAVES - 1) / (2 * OLS_WAVES);
        if (blocks > max_blocks) blocks = max_blocks;
        if (chain)
            hipLaunchKernelGGL(k_fir_ols_chain_f32, dim3((unsigned)blocks), dim3(OLS_THREADS), lds_bytes,
                               as_stream(stream), in, out, hist, hf, tw, channels, n, in_pitch, out_pitch, flt_len,
                               jobs_per_channel, segs_per_channel, total_segs, seg_len);
        else if (prefetch)
            hipLaunchKernelGGL(k_fir_ols_walk_f32<true>, dim3((unsigned)blocks), dim3(OLS_THREADS), lds_bytes,
                               as_stream(stream), in, out, hist, hf, tw, channels, n, in_pitch, out_pitch, flt_len,
                               jobs_per_channel, segs_per_channel, total_segs, seg_len);
        else
            hipLaunchKernelGGL(k_fir_ols_walk_f32<false>, dim3((unsigned)blocks), dim3(OLS_THREADS), lds_bytes,
                               as_stream(stream), in, out, hist, hf, tw, channels, n, in_pitch, out_pitch, flt_len,
                               jobs_per_channel, segs_per_channel, total_segs, seg_len);
        LLZ_LAUNCH_CHECK("k_fir_ols_walk_f32");
        return LLZ_OK;
    }
    if (blocks > max_blocks) blocks = max_blocks;
    if (prefetch)
        hipLaunchKernelGGL(k_fir_ols_f32<true>, dim3((unsigned)blocks), dim3(OLS_THREADS), lds_bytes,
                           as_stream(stream), in, out, hist, hf, tw, channels, n, in_pitch, out_pitch, flt_len,
                           jobs_per_channel, total_jobs);
    else
        hipLaunchKernelGGL(k_fir_ols_f32<false>, dim3((unsigned)blocks), dim3(OLS_THREADS), lds_bytes,
                           as_stream(stream), in, out, hist, hf, tw, channels, n, in_pitch, out_pitch, flt_len,
                           jobs_per_channel, total_jobs);
    LLZ_LAUNCH_CHECK("k_fir_ols_f32");
    return LLZ_OK;
}
