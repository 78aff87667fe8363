// fir_ols.hip -- K4 / K4b: overlap-save FIR with register FFTs that never leave the CU.
//   k_fir_ols_chain_f32      up to 257 taps   1024-point transforms, a half-wave per job (described first, below)
//   k_fir_ols2k_chain_f32<O> up to 1025 taps  2048-point transforms, a whole wave per job (one radix-2 step over the half-waves)
//   k_fir_ols4k_f32<O>       up to 3073 taps  4096-point transforms, a whole wave per job (two radix-2 steps)
//   k_fir_ols8k_f32<O>       up to 6145 taps  8192-point transforms, a pair of waves per job (one more radix-2 step, across the pair)
//
// New functionality relative to the reference (llz_fir.c is time-domain only); its end-to-end
// oracle is the time-domain llz_fir_filter (llz_fir.c:547-584), its FFT stage follows the sign/scale
// convention of llz_fft.c:61-130,142-198 (forward e^{-j}, unscaled; inverse e^{+j}, divided by N -- the 1/N is
// folded into the filter spectrum on the host).
//
// Why: 257 taps in the time domain is 514 flop per 8 B of HBM traffic -> VALU-bound at ~30 % of the HBM roofline.
// Overlap-save costs ~46 vector lane-operations per sample and is HBM-bound.
//
// Mapping to CDNA4 (wave64):
//   * job = one channel x 1536 new samples.  Two consecutive 1024-sample blocks (each: 256 overlap + 768 new) are
//     the real and imaginary part of ONE complex 1024-point transform -- the filter is real, so
//     IFFT(FFT(xa + j xb) H) = ya + j yb and nothing has to be untangled.
//   * a half-wave (32 lanes) owns a job: 1024 = 32 x 32, every lane keeps 32 complex values in registers.  A
//     transform is two passes of 32-point in-register FFTs (fft32.hpp) with one 32 x 32 transpose through LDS
//     between them (row pitch 33 floats).  Forward leaves bins digit-reversed in the register index, which is
//     free (register renaming), so FFT -> multiply by H -> IFFT needs no permutation pass at all.
//   * a half-wave walks a SEGMENT of up to 16 consecutive jobs of one channel and carries the 256-sample overlap in
//     registers from job to job, so every input sample is requested from memory exactly once; the next job's samples
//     are requested (into registers) before the current job is transformed, across segment boundaries as well
//     (k_fir_ols_chain_f32: the last job of a segment requests the first job and the halo of the half-wave's next segment).
//   * HBM access shape: one dword per lane, a half-wave instruction moves 128 contiguous bytes of its job.  Wider shapes
//     were built and measured and do not pay (profiles/r02/): a copy kernel in exactly this walk structure runs at the same
//     rate with 4, 8 or 16 bytes per lane (6.04 / 6.02 / 6.02 ms for 32 GiB), and an 8-byte form of this kernel (512
//     contiguous bytes per wave instruction, samples handed to their lanes by v_permlane16/32_swap transposes, commit
//     cda1552) was 7 % slower: the 96 extra vector instructions per job pair cost more than the shorter request stream saves.
//   * per workgroup (4 waves): LDS = 8 KB inter-pass twiddles W_1024^(a*b) + 8 KB filter spectrum + 8 x 4.1 KB
//     transpose buffers = 49 KB; two workgroups per CU (the prefetch registers allow two waves per SIMD).
// HBM traffic: 4 B read + 4 B written per sample, nothing re-read (PMC: profiles/pmc_traffic.json).
#include <stdlib.h>
#include "common.hpp"
#include "fft32.hpp"

namespace {

constexpr int OLS_N = 1024;
constexpr int OLS_OVERLAP = 256;
constexpr int OLS_VALID = OLS_N - OLS_OVERLAP;     // 768
constexpr int OLS_JOB = 2 * OLS_VALID;             // 1536 new samples per complex transform
constexpr int OLS_WAVES = 4;
constexpr int OLS_THREADS = 64 * OLS_WAVES;
constexpr int OLS_SEG = 16;                        // jobs per segment at most (16 x 1536 samples of one channel)

// where a lane sits: which of the wave's two jobs it transforms and which column of the 32 x 32 decomposition it owns
struct ols_lane {
    int half, col;
};

__device__ __forceinline__ ols_lane ols_lane_of(int lane)
{
    ols_lane g;
    g.half = lane >> 5;
    g.col = lane & 31;
    return g;
}

// one segment of one half-wave: `count` consecutive jobs of channel c starting at job j0
struct ols_seg {
    bool live;
    int c, j0, count;
};

struct ols_geom {
    int n, keep, jobs_per_channel, segs_per_channel, seg_len;
    long total_segs, in_pitch, out_pitch;
};

__device__ __forceinline__ ols_seg ols_locate(long seg, const ols_geom &G)
{
    ols_seg g;
    g.live = seg < G.total_segs;
    g.c = g.live ? (int)(seg / G.segs_per_channel) : 0;
    g.j0 = g.live ? (int)(seg - (long)g.c * G.segs_per_channel) * G.seg_len : 0;
    g.count = g.live ? min(G.seg_len, G.jobs_per_channel - g.j0) : 0;
    return g;
}

// the 1536 new samples of a job, 48 registers per lane: a[r] / b[r] = row r of block A's / block B's new part, this
// lane's column
struct ols_raw {
    float a[24], b[24];
};

// request the 1536 new samples of the wave's two jobs (job h: row[h] + s[h], h = 0 lower / 1 upper half-wave); every
// lane takes its own job's column
__device__ __forceinline__ void ols_load(ols_raw &raw, const float *const (&row)[2], const int (&s)[2],
                                         const bool (&live)[2], const ols_lane &g, int n)
{
    const bool whole = live[0] && live[1] && s[0] + OLS_JOB <= n && s[1] + OLS_JOB <= n;   // wave-uniform
    const float *r = g.half ? row[1] : row[0];
    const int so = g.half ? s[1] : s[0];
    const bool lv = g.half ? live[1] : live[0];
    if (whole) {
#pragma unroll
        for (int i = 0; i < 24; i++) {
            raw.a[i] = __builtin_nontemporal_load(&r[so + 32 * i + g.col]);
            raw.b[i] = __builtin_nontemporal_load(&r[so + OLS_VALID + 32 * i + g.col]);
        }
    } else {
        // ragged last job of a row, or an idle half: addresses clamped into the row, samples outside [0, n) are zero
#pragma unroll
        for (int i = 0; i < 24; i++) {
            const int ia = so + 32 * i + g.col, ib = ia + OLS_VALID;
            const float xa = r[min(ia, n - 1)], xb = r[min(ib, n - 1)];
            raw.a[i] = (lv && ia < n) ? xa : 0.f;
            raw.b[i] = (lv && ib < n) ? xb : 0.f;
        }
    }
}

// the 256 samples in front of a segment, rows 0..7 of this lane's column: from the row (s > 0) or from the history
// (the previous call's last flt_len-1 samples) / zeros (s == 0)
__device__ __forceinline__ void ols_load_halo(float (&halo)[8], const float *row, const float *hrow, int s, int col,
                                              int keep, bool live)
{
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int idx = s - OLS_OVERLAP + 32 * i + col;
        float v = 0.f;
        if (live) {
            if (idx >= 0) v = row[idx];
            else if (hrow && idx >= -keep) v = hrow[keep + idx];
        }
        halo[i] = v;
    }
}

// FFT -> multiply by the filter spectrum -> IFFT, all in registers + one LDS transpose each way.  `col` is the lane's
// column n2 of the input / output decomposition n = 32 n1 + n2; between the transposes lane l5 owns bin row k1 = l5.
__device__ __forceinline__ void ols_filter(cf (&v)[32], cf (&u)[32], float *buf, const float2 *s_tw,
                                           const float2 *s_h, int l5, int col)
{
    // ---- forward: pass 1 over n1 (in registers), twiddle W^(k1 n2) + transpose, pass 2 over n2
    fft32<false>(v);
    transpose_twiddle<false>(v, buf, s_tw, col, l5);
    fft32<false>(v);                                        // v[r] = X[l5 + 32*brev5(r)]
    // ---- filter in the frequency domain (spectrum already carries the 1/1024)
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const float2 h = s_h[l5 + 32 * brev5(r)];
        u[brev5(r)] = cmul<false>(v[r], cf{h.x, h.y});     // back to natural k2 order: renaming only
    }
    // ---- inverse: pass over k2, conj twiddle + transpose, pass over k1
    fft32<true>(u);
    transpose_twiddle<true>(u, buf, s_tw, l5, col);
    fft32<true>(u);                                         // u[r] = y[32*brev5(r) + col]
}

// keep the 768 valid samples of each block: rows n1 = brev5(r) >= 8
__device__ __forceinline__ void ols_store(const cf (&u)[32], float *const (&orow)[2], const int (&s)[2],
                                          const bool (&live)[2], const ols_lane &g, int n)
{
    const bool whole = live[0] && live[1] && s[0] + OLS_JOB <= n && s[1] + OLS_JOB <= n;   // wave-uniform
    const bool lv = g.half ? live[1] : live[0];
    if (!lv) return;
    float *o = g.half ? orow[1] : orow[0];
    const int so = g.half ? s[1] : s[0];
    const int oa = so + g.col - OLS_OVERLAP;                // + 32*n1
    const int ob = oa + OLS_VALID;
    if (whole) {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int n1 = brev5(r);
            if (n1 >= 8) {
                __builtin_nontemporal_store(u[r].x, &o[oa + 32 * n1]);
                __builtin_nontemporal_store(u[r].y, &o[ob + 32 * n1]);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 32; r++) {
            const int n1 = brev5(r);
            if (n1 >= 8) {
                if (oa + 32 * n1 < n) o[oa + 32 * n1] = u[r].x;
                if (ob + 32 * n1 < n) o[ob + 32 * n1] = u[r].y;
            }
        }
    }
}

// blocks A and B of a job from the carried overlap and the job's new samples; the new overlap is block B's last 256
__device__ __forceinline__ void ols_assemble(cf (&v)[32], float (&halo)[8], const ols_raw &raw)
{
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
}

struct ols_smem {
    float2 *tw, *h;
    float *buf;
};

__device__ __forceinline__ ols_smem ols_tables(char *smem, const float2 *__restrict__ hfreq, const float2 *__restrict__ twid)
{
    ols_smem m;
    m.tw = reinterpret_cast<float2 *>(smem);                 // [32][32]  W_1024^(a*b)
    m.h = m.tw + 1024;                                        // [1024]    FFT(taps)/1024, natural bins
    for (int i = threadIdx.x; i < 1024; i += OLS_THREADS) {
        m.tw[i] = twid[i];
        m.h[i] = hfreq[i];
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, half = (threadIdx.x >> 5) & 1;
    m.buf = reinterpret_cast<float *>(m.h + 1024) + (wave * 2 + half) * OLS_XBUF;     // per half-wave transpose buffer
    return m;
}

// A half-wave runs one segment after the other (grid stride over segment pairs), with the prefetch carried ACROSS segments: the
// last job of a segment requests the first job and the halo of the half-wave's NEXT segment, so every load is used and no
// segment start waits on memory (a first form requested past the segment's end and discarded the data: 1/16 of the input
// fetched twice, 3 % slower on the headline; it was also slower on batches of a single round: 0.21 against 0.13 ms on 64
// channels x 63 taps).  Both halves of a wave step through their segments together (a segment that is short at the end of a
// channel idles its tail).
__global__ void __launch_bounds__(OLS_THREADS, 2)
k_fir_ols_chain_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                    const float2 *__restrict__ hfreq, const float2 *__restrict__ twid, ols_geom G)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ols_smem S = ols_tables(smem, hfreq, twid);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l5 = lane & 31;
    const ols_lane g = ols_lane_of(lane);
    const long halves_total = (long)gridDim.x * OLS_WAVES * 2;
    const long first = ((long)blockIdx.x * OLS_WAVES + wave) * 2;

    ols_seg cur[2] = {ols_locate(first, G), ols_locate(first + 1, G)};
    float halo[8];
    ols_raw raw;
    {
        const float *const row[2] = {in + (size_t)cur[0].c * G.in_pitch, in + (size_t)cur[1].c * G.in_pitch};
        const ols_seg own = g.half ? cur[1] : cur[0];
        ols_load_halo(halo, g.half ? row[1] : row[0], hist ? hist + (size_t)own.c * G.keep : nullptr, own.j0 * OLS_JOB,
                      g.col, G.keep, own.live);
        const int s0[2] = {cur[0].j0 * OLS_JOB, cur[1].j0 * OLS_JOB};
        const bool lv[2] = {cur[0].count > 0, cur[1].count > 0};
        ols_load(raw, row, s0, lv, g, G.n);
    }
    for (long sp = first; sp < G.total_segs; sp += halves_total) {
        const ols_seg nxt[2] = {ols_locate(sp + halves_total, G), ols_locate(sp + halves_total + 1, G)};
        const float *const row[2] = {in + (size_t)cur[0].c * G.in_pitch, in + (size_t)cur[1].c * G.in_pitch};
        float *const orow[2] = {out + (size_t)cur[0].c * G.out_pitch, out + (size_t)cur[1].c * G.out_pitch};
        const float *const nrow[2] = {in + (size_t)nxt[0].c * G.in_pitch, in + (size_t)nxt[1].c * G.in_pitch};
        const ols_seg nown = g.half ? nxt[1] : nxt[0];
        float halo_n[8];
#pragma unroll
        for (int i = 0; i < 8; i++) halo_n[i] = 0.f;

        const int jmax = max(cur[0].count, cur[1].count);
#pragma unroll 1
        for (int jj = 0; jj < jmax; jj++) {
            const int s[2] = {(cur[0].j0 + jj) * OLS_JOB, (cur[1].j0 + jj) * OLS_JOB};
            const bool live[2] = {jj < cur[0].count, jj < cur[1].count};
            cf v[32], u[32];
            ols_assemble(v, halo, raw);
            // next job of this segment pair, or (after the pair's last job) the first jobs and halos of the next pair.  A
            // half whose own segment has ended while its partner's has not requests nothing.
            const bool in_pair = jj + 1 < jmax;                                  // wave-uniform
            {
                const float *const lrow[2] = {in_pair ? row[0] : nrow[0], in_pair ? row[1] : nrow[1]};
                const int sn[2] = {in_pair ? s[0] + OLS_JOB : nxt[0].j0 * OLS_JOB,
                                   in_pair ? s[1] + OLS_JOB : nxt[1].j0 * OLS_JOB};
                const bool ln[2] = {in_pair ? jj + 1 < cur[0].count : nxt[0].count > 0,
                                    in_pair ? jj + 1 < cur[1].count : nxt[1].count > 0};
                ols_load(raw, lrow, sn, ln, g, G.n);
            }
            if (!in_pair)
                ols_load_halo(halo_n, g.half ? nrow[1] : nrow[0], hist ? hist + (size_t)nown.c * G.keep : nullptr,
                              nown.j0 * OLS_JOB, g.col, G.keep, nown.live);
            ols_filter(v, u, S.buf, S.tw, S.h, l5, g.col);
            ols_store(u, orow, s, live, g, G.n);
        }
        cur[0] = nxt[0];
        cur[1] = nxt[1];
#pragma unroll
        for (int i = 0; i < 8; i++) halo[i] = halo_n[i];
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// 258 .. 1025 taps: the same walk on 2048-point transforms, a WHOLE WAVE per job.
//
// A 2048-point transform held by one half-wave needs 64 complex registers per lane and spills (round 1's kernel of that
// shape: 12.4 ms at 513 taps, 17 ms at 1025).  Here one radix-2 step splits it over the wave's two half-waves, each of which then runs the
// 1024-point machinery above unchanged:
//     forward (decimation in frequency):  X[2k]   = FFT_1024( a[n] + a[n+1024] )            -> lower half-wave
//                                         X[2k+1] = FFT_1024( (a[n] - a[n+1024]) W_2048^n ) -> upper half-wave
//     inverse (decimation in time):       y[n], y[n+1024] = S'[n] +- W_2048^-n D'[n],   S' / D' = the halves' inverse transforms
// A block is 64 rows of 32 samples; lane (half h, l5) owns the rows of parity h (2p + h, p < 32) at column l5, so rows p
// and p + 16 of a lane are 1024 samples apart: the butterflies are in-lane, and ONE v_permlane32_swap per register pair
// then hands all sums to the lower and all differences to the upper half-wave (and back after the inverse transforms).
// Because a row's parity survives the walk's shifts (the 512-sample overlap is 16 rows, a block advances by 48 rows), the
// carried overlap stays in the lane that needs it, and ols_assemble / ols_raw are the 1024-point kernel's.  Lower and upper
// lanes load adjacent rows: a wave instruction moves 256 contiguous bytes.  Job = two blocks (real / imaginary part) =
// 3072 new samples of one channel.
// Template parameter O = the overlap: 512 samples (16 rows, up to 513 taps, 1536 new samples per block) or 1024 (32 rows,
// up to 1025 taps, 1024 new samples per block).

template <int O>
__global__ void __launch_bounds__(OLS_THREADS, 2)
k_fir_ols2k_chain_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                     const float2 *__restrict__ hfreq2 /* [2][1024]: even bins, odd bins, / 2048 */,
                     const float2 *__restrict__ twid /* [32][32] W_1024^(ab) */, const float2 *__restrict__ tw2k /* [1024] W_2048^n */,
                     ols_geom G)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_tw = reinterpret_cast<float2 *>(smem);           // [1024]
    float2 *s_h = s_tw + 1024;                                  // [2][1024]
    float2 *s_w = s_h + 2048;                                   // [1024]
    for (int i = threadIdx.x; i < 1024; i += OLS_THREADS) {
        s_tw[i] = twid[i];
        s_h[i] = hfreq2[i];
        s_h[1024 + i] = hfreq2[1024 + i];
        s_w[i] = tw2k[i];
    }
    __syncthreads();
    constexpr int O2K_OVERLAP = O, O2K_VALID = 2048 - O, O2K_JOB = 2 * O2K_VALID;
    constexpr int HP = O / 64, NEW = 32 - HP;                       // a lane's carried and new rows per block
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l5 = lane & 31;
    float *buf = reinterpret_cast<float *>(s_w + 1024) + (wave * 2 + half) * OLS_XBUF;
    const float2 *my_h = s_h + half * 1024;
    const long waves_total = (long)gridDim.x * OLS_WAVES;
    const int rowoff = 32 * half + l5;                              // this lane inside a pair of rows

    // segment -> (channel row, first job, job count); a wave past the last segment gets an empty one
    struct segd { const float *row; float *orow; const float *hrow; int j0, count; };
    auto locate = [&](long seg) {
        segd d;
        const bool live = seg < G.total_segs;
        const int c = live ? (int)(seg / G.segs_per_channel) : 0;
        d.j0 = live ? (int)(seg - (long)c * G.segs_per_channel) * G.seg_len : 0;
        d.count = live ? min(G.seg_len, G.jobs_per_channel - d.j0) : 0;
        d.row = in + (size_t)c * G.in_pitch;
        d.orow = out + (size_t)c * G.out_pitch;
        d.hrow = hist ? hist + (size_t)c * G.keep : nullptr;
        return d;
    };
    const int n = G.n;
    // rows 2i + h (i < HP) in front of a segment: from the row, the history or zeros
    auto load_halo = [&](float (&halo)[HP], const segd &d) {
#pragma unroll
        for (int i = 0; i < HP; i++) {
            const int idx = d.j0 * O2K_JOB - O2K_OVERLAP + 64 * i + rowoff;
            float v = 0.f;
            if (d.count > 0) {
                if (idx >= 0) v = d.row[idx];
                else if (d.hrow && idx >= -G.keep) v = d.hrow[G.keep + idx];
            }
            halo[i] = v;
        }
    };
    ols_raw raw;
    auto load = [&](const float *row, int s, bool live) {          // the new samples of the job at s: 2 NEW per lane
        if (live && s + O2K_JOB <= n) {
#pragma unroll
            for (int i = 0; i < NEW; i++) {
                raw.a[i] = __builtin_nontemporal_load(&row[s + 64 * i + rowoff]);
                raw.b[i] = __builtin_nontemporal_load(&row[s + O2K_VALID + 64 * i + rowoff]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NEW; i++) {
                const int ia = s + 64 * i + rowoff, ib = ia + O2K_VALID;
                const float xa = row[min(ia, n - 1)], xb = row[min(ib, n - 1)];
                raw.a[i] = (live && ia < n) ? xa : 0.f;
                raw.b[i] = (live && ib < n) ? xb : 0.f;
            }
        }
    };

    long seg = (long)blockIdx.x * OLS_WAVES + wave;
    segd cur = locate(seg);
    float halo[HP];
    load_halo(halo, cur);
    load(cur.row, cur.j0 * O2K_JOB, cur.count > 0);
    for (; seg < G.total_segs; seg += waves_total) {
        const segd nxt = locate(seg + waves_total);
        float halo_n[HP];
#pragma unroll
        for (int i = 0; i < HP; i++) halo_n[i] = 0.f;
#pragma unroll 1
        for (int jj = 0; jj < cur.count; jj++) {
            const int s = (cur.j0 + jj) * O2K_JOB;
            const float *row = cur.row;
            float *orow = cur.orow;
            cf v[32], u[32];
            // v[p] = row 2p + h of blocks A (re) and B (im): block A's first HP positions are the carried overlap, block B's
            // are block A's last HP; the new overlap is block B's last HP
#pragma unroll
            for (int i = 0; i < HP; i++) {
                v[i].x = halo[i];
                v[i].y = raw.a[NEW - HP + i];
                halo[i] = raw.b[NEW - HP + i];
            }
#pragma unroll
            for (int i = 0; i < NEW; i++) {
                v[HP + i].x = raw.a[i];
                v[HP + i].y = raw.b[i];
            }
            // the next job of this segment, or -- after its last job -- the first job and the halo of the wave's next segment
            const bool in_seg = jj + 1 < cur.count;                  // wave-uniform
            load(in_seg ? row : nxt.row, in_seg ? s + O2K_JOB : nxt.j0 * O2K_JOB, in_seg || nxt.count > 0);
            if (!in_seg) load_halo(halo_n, nxt);
            // ---- radix-2 step down: rows p and p + 16 are 1024 samples apart.  Every lane twiddles the differences of its
            // own 16 rows BEFORE they move to the upper half-wave (half the multiplies the upper lanes alone would need)
            cf w[32];
#pragma unroll
            for (int p = 0; p < 16; p++) {
                cf sm = cadd(v[p], v[p + 16]);
                const float2 t = s_w[64 * p + rowoff];              // W_2048^n, n = 32 (2p + h) + l5
                cf df = cmul<false>(csub(v[p], v[p + 16]), cf{t.x, t.y});
                swap32(sm.x, df.x);                                 // lower: (s_2p, s_2p+1)   upper: (d_2p, d_2p+1)
                swap32(sm.y, df.y);
                w[2 * p] = sm;
                w[2 * p + 1] = df;
            }
            // ---- 1024 points per half-wave: forward, spectrum product, inverse
            ols_filter(w, u, buf, s_tw, my_h, l5, l5);              // u[r] = S' or D' at row brev5(r)
            // ---- radix-2 step up and the stores of the valid rows (positions >= HP)
            const bool whole = s + O2K_JOB <= n;
            const int oa = s - O2K_OVERLAP + rowoff, ob = oa + O2K_VALID;
#pragma unroll
            for (int p = 0; p < 16; p++) {
                cf P = u[brev5(2 * p)], Q = u[brev5(2 * p + 1)];
                swap32(P.x, Q.x);                                   // lower: (S_2p, D_2p)   upper: (S_2p+1, D_2p+1)
                swap32(P.y, Q.y);
                const float2 t = s_w[64 * p + rowoff];
                Q = cmul<true>(Q, cf{t.x, t.y});                    // W_2048^-n on the lane's own 16 differences
                const cf y0 = cadd(P, Q), y1 = csub(P, Q);          // rows 2p + h and 2p + h + 32 = positions p, p + 16
                if (whole) {
                    if (p >= HP) {
                        __builtin_nontemporal_store(y0.x, &orow[oa + 64 * p]);
                        __builtin_nontemporal_store(y0.y, &orow[ob + 64 * p]);
                    }
                    __builtin_nontemporal_store(y1.x, &orow[oa + 64 * (p + 16)]);
                    __builtin_nontemporal_store(y1.y, &orow[ob + 64 * (p + 16)]);
                } else {
                    if (p >= HP) {
                        if (oa + 64 * p < n) orow[oa + 64 * p] = y0.x;
                        if (ob + 64 * p < n) orow[ob + 64 * p] = y0.y;
                    }
                    if (oa + 64 * (p + 16) < n) orow[oa + 64 * (p + 16)] = y1.x;
                    if (ob + 64 * (p + 16) < n) orow[ob + 64 * (p + 16)] = y1.y;
                }
            }
        }
        cur = nxt;
#pragma unroll
        for (int i = 0; i < HP; i++) halo[i] = halo_n[i];
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// 514 .. 3073 taps: 4096-point transforms, again a whole wave per job, as radix-2 step + TWO 2048-point problems run one
// after the other through the split above (each of them: radix-2 step + a 1024-point transform per half-wave).
//     X[2k] = FFT_2048( a[n] + a[n+2048] ),   X[2k+1] = FFT_2048( (a[n] - a[n+2048]) W_4096^n );   back: E'[n] +- W_4096^-n O'[n]
// A block is 128 rows of 32 samples, a lane owns the 64 rows of its parity (positions p: row 2p + h), rows p and p + 32 are
// 2048 samples apart (in-lane butterflies).  The overlap O is 512 / 1024 / 2048 / 3072 samples by tap count (positions
// below O / 64 of a lane; with 2048 only the "minus" half of the last step is ever stored).  64 complex values per lane leave
// no room for a register prefetch or a carried overlap: a job's new dwords and those of block A's overlap (an L2 hit: the
// same wave read them one job ago) are requested when the job starts, and the second wave of the SIMD covers the wait.  A
// workgroup is 8 waves (one per CU: the four spectra planes, three twiddle tables and sixteen transpose buffers are 132 KB
// of LDS).
//
// The spectrum is split four ways: plane j = 2 half + sub holds H[4m + j] / 4096 (sub 0 = the even-bin problem E, 1 = the
// odd-bin problem O; inside a problem the lower half-wave has its even bins, the upper its odd bins).
// the lane id, formed afresh where it is needed: what the phases of a job derive from it (buffer addresses, the lane's twiddle)
// then lives inside that phase only -- carried across the 4096-point problem, which needs every register there is, each of
// them is a spill
__device__ __forceinline__ int o8k_lane()
{
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}
constexpr int O4K_WAVES = 8, O4K_THREADS = 64 * O4K_WAVES;

// one 2048-point problem of the wave: v[p] = row 2p + h (p < 32) in, y[p] out (same ownership); see k_fir_ols2k_chain_f32
template <int WS = 1>                                                 // WS = 2: s_w is the W_4096^n table (every other entry)
__device__ __forceinline__ void ols2k_core(cf (&v)[32], cf (&y)[32], float *buf, const float2 *s_tw, const float2 *my_h,
                                           const float2 *s_w, int l5, int rowoff)
{
    cf w[32], u[32];
#pragma unroll
    for (int p = 0; p < 16; p++) {
        cf sm = cadd(v[p], v[p + 16]);
        const float2 t = s_w[WS * (64 * p + rowoff)];
        cf df = cmul<false>(csub(v[p], v[p + 16]), cf{t.x, t.y});
        swap32(sm.x, df.x);
        swap32(sm.y, df.y);
        w[2 * p] = sm;
        w[2 * p + 1] = df;
    }
    ols_filter(w, u, buf, s_tw, my_h, l5, l5);
#pragma unroll
    for (int p = 0; p < 16; p++) {
        cf P = u[brev5(2 * p)], Q = u[brev5(2 * p + 1)];
        swap32(P.x, Q.x);
        swap32(P.y, Q.y);
        const float2 t = s_w[WS * (64 * p + rowoff)];
        Q = cmul<true>(Q, cf{t.x, t.y});
        y[p] = cadd(P, Q);
        y[p + 16] = csub(P, Q);
    }
}

// O = overlap, a multiple of 64 that holds flt_len - 1 samples (512, 768, 1024, 1536, 2048, 2560, 3072); a job is two blocks =
// 2 (4096 - O) new samples.
// Lane position i (0..63) of a block is sample 64 i + rowoff; positions below O / 64 are overlap.
template <int O>
__global__ void __launch_bounds__(O4K_THREADS, 2)
k_fir_ols4k_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                const float2 *__restrict__ hfreq4 /* [4][1024] */, const float2 *__restrict__ twid /* [32][32] W_1024^(ab) */,
                const float2 *__restrict__ tw2k /* [1024] W_2048^n */, const float2 *__restrict__ tw4k /* [2048] W_4096^n */,
                ols_geom G)
{
    constexpr int V = 4096 - O, JOB = 2 * V, PO = O / 64, PV = V / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_tw = reinterpret_cast<float2 *>(smem);           // [1024]
    float2 *s_w2 = s_tw + 1024;                                 // [1024]
    float2 *s_w4 = s_w2 + 1024;                                 // [2048]
    float2 *s_h = s_w4 + 2048;                                  // [4][1024]
    for (int i = threadIdx.x; i < 1024; i += O4K_THREADS) {
        s_tw[i] = twid[i];
        s_w2[i] = tw2k[i];
        s_w4[i] = tw4k[i];
        s_w4[1024 + i] = tw4k[1024 + i];
#pragma unroll
        for (int j = 0; j < 4; j++) s_h[1024 * j + i] = hfreq4[1024 * j + i];
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *s_buf = reinterpret_cast<float *>(s_h + 4096);
    const long waves_total = (long)gridDim.x * O4K_WAVES;
    const int n = G.n;

    for (long seg = (long)blockIdx.x * O4K_WAVES + wave; seg < G.total_segs; seg += waves_total) {
        const int c = (int)(seg / G.segs_per_channel);
        const int j0 = (int)(seg - (long)c * G.segs_per_channel) * G.seg_len;
        const int count = min(G.seg_len, G.jobs_per_channel - j0);
        const float *row = in + (size_t)c * G.in_pitch;
        float *orow = out + (size_t)c * G.out_pitch;
        const float *hrow = hist ? hist + (size_t)c * G.keep : nullptr;
#pragma unroll 1
        for (int jj = 0; jj < count; jj++) {
            const int s = (j0 + jj) * JOB;
            // block A (real parts) = [s - O, s + V), block B (imaginary parts) = [s + V - O, s + 2V).  Block A's overlap was
            // read by this wave one job ago: it is read again (from L2) rather than carried -- 32 more registers would spill.
            // Block B's first O samples are block A's last O.
            cf v[64];
            const bool whole = s >= O && s + JOB <= n;
            {
            const int rowoff = o8k_lane();
            if (whole) {
#pragma unroll
                for (int i = 0; i < PO; i++) v[i].x = row[s - O + 64 * i + rowoff];
#pragma unroll
                for (int i = PO; i < 64; i++) v[i].x = __builtin_nontemporal_load(&row[s - O + 64 * i + rowoff]);
#pragma unroll
                for (int i = PO; i < 64; i++) v[i].y = __builtin_nontemporal_load(&row[s + V - O + 64 * i + rowoff]);
            } else {
#pragma unroll
                for (int i = 0; i < 64; i++) {
                    const int ia = s - O + 64 * i + rowoff;
                    float xa = 0.f;
                    if (ia >= 0) xa = row[min(ia, n - 1)];
                    else if (hrow && ia >= -G.keep) xa = hrow[G.keep + ia];
                    v[i].x = ia < n ? xa : 0.f;
                }
#pragma unroll
                for (int i = PO; i < 64; i++) {
                    const int ib = s + V - O + 64 * i + rowoff;
                    const float xb = row[min(ib, n - 1)];
                    v[i].y = ib < n ? xb : 0.f;
                }
            }
            }
#pragma unroll
            for (int i = 0; i < PO; i++) v[i].y = v[i + PV].x;
            const int rowoff = o8k_lane(), half = rowoff >> 5, l5 = rowoff & 31;
            float *buf = s_buf + (wave * 2 + half) * OLS_XBUF;
            // ---- radix-2 step down: positions p and p + 32 are 2048 samples apart
            cf e[32], o[32];
#pragma unroll
            for (int p = 0; p < 32; p++) {
                e[p] = cadd(v[p], v[p + 32]);
                const float2 t = s_w4[64 * p + rowoff];             // W_4096^n, n = 32 (2p + h) + l5
                o[p] = cmul<false>(csub(v[p], v[p + 32]), cf{t.x, t.y});
            }
            // ---- the even-bin and the odd-bin 2048-point problems, one after the other
            cf ye[32], yo[32];
            ols2k_core(e, ye, buf, s_tw, s_h + (2 * half) * 1024, s_w2, l5, rowoff);
            ols2k_core(o, yo, buf, s_tw, s_h + (2 * half + 1) * 1024, s_w2, l5, rowoff);
            // ---- radix-2 step up: position p = E' + W^-n O', position p + 32 = E' - W^-n O'; positions >= O / 64 are outputs
            const bool all_out = s + JOB <= n;
#pragma unroll
            for (int p = 0; p < 32; p++) {
                if (p + 32 < PO) continue;
                const float2 t = s_w4[64 * p + rowoff];
                const cf q = cmul<true>(yo[p], cf{t.x, t.y});
                if (p >= PO) {
                    const cf y0 = cadd(ye[p], q);
                    const int oa = s + 64 * (p - PO) + rowoff, ob = oa + V;
                    if (all_out) {
                        __builtin_nontemporal_store(y0.x, &orow[oa]);
                        __builtin_nontemporal_store(y0.y, &orow[ob]);
                    } else {
                        if (oa < n) orow[oa] = y0.x;
                        if (ob < n) orow[ob] = y0.y;
                    }
                }
                const cf y1 = csub(ye[p], q);
                const int oa = s + 64 * (p + 32 - PO) + rowoff, ob = oa + V;
                if (all_out) {
                    __builtin_nontemporal_store(y1.x, &orow[oa]);
                    __builtin_nontemporal_store(y1.y, &orow[ob]);
                } else {
                    if (oa < n) orow[oa] = y1.x;
                    if (ob < n) orow[ob] = y1.y;
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// 2 .. 6145 taps (the library's choice from 1026 on): 8192-point transforms on a PAIR of waves.  With 3072 samples of overlap a 4096-point block yields 1024
// outputs (25 % of the transform), an 8192-point block 5120 (62.5 %) at 13/12 of the work per point.
//     X[2k] = FFT_4096( a[n] + a[n+4096] )  -> wave 0 of the pair,     X[2k+1] = FFT_4096( (a[n] - a[n+4096]) W_8192^n )  -> wave 1,
// each wave then runs the 4096-point problem of k_fir_ols4k_f32 (radix-2 step + two 2048-point problems); back:
// y[n] = E'[n] + W_8192^-n O'[n], y[n+4096] = E'[n] - W_8192^-n O'[n].
// A block is 128 rows of 64 samples (position i = samples 64 i + lane).  Wave w requests positions 32w .. 32w+31 and the
// same + 64 of both blocks (64 complex values per lane, as the 4096-point kernel), forms its 32 sums and 32 twiddled
// differences, and the two waves SWAP halves through LDS: wave 0 gives its differences for wave 1's sums.  On the way back
// wave 1 twiddles its results, the waves swap halves again and each forms and stores the outputs among "its" 64 positions.
// W_8192^(64 i + lane) = W_128^i (a literal after unrolling) x W_8192^lane (a 512-byte LDS table, read where it is used): no
// 32 KB table, and nothing lane-derived is carried across the 4096-point problem (o8k_lane()).
// Wave 1 keeps the two halves of its register array exchanged (slot j = position j + 32 mod 64), so that both waves keep slots
// 0..31 and swap slots 32..63 through one code path; the 4096-point problem is indifferent to it.
//
// The swap goes through the transpose buffers the two waves own anyway (8.25 KB per wave: two rounds of 16 values per lane),
// and the pair synchronises on four LDS words of its own instead of a workgroup barrier, so that the four pairs of a workgroup
// drift apart and one pair's memory wait is another's arithmetic -- a barrier would hold all eight waves in the same phase
// (64 complex values per lane leave no registers for a prefetch).  The pair is the two waves of ONE SIMD (waves p, p + 4): a
// wave that waits for its partner leaves the SIMD to exactly that partner.  Both waves of a pair run the same loop bounds, hence the
// same number of rounds; the waits are bounded (a pair out of step would produce wrong samples, which the tests see, never
// a hang).  LDS: W_1024^(ab) 8 KB, W_4096^n 16 KB (W_2048^n = every other entry), spectrum 8 x 8 KB, buffers 66 KB = 154 KB.
// O8K_TRACE (a measurement build only): shader-clock time of each phase of a job, summed per wave (tools/trace_ols8k.py)
#ifdef O8K_TRACE
__device__ unsigned long long o8k_trace_buf[2048 * 8];
#define O8K_T0() unsigned long long o8_t = __builtin_readcyclecounter(), o8_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define O8K_MARK(i) do { const unsigned long long now = __builtin_readcyclecounter(); o8_acc[i] += now - o8_t; o8_t = now; } while (0)
#define O8K_PIN(x) asm volatile("s_nop 0" ::"v"(x))
#define O8K_DUMP() do { if ((threadIdx.x & 63) == 0) { const unsigned wv_ = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 2047u; \
    for (int i = 0; i < 8; i++) o8k_trace_buf[wv_ * 8 + i] = o8_acc[i]; } } while (0)
#else
#define O8K_T0() do { } while (0)
#define O8K_MARK(i) do { } while (0)
#define O8K_PIN(x) do { } while (0)
#define O8K_DUMP() do { } while (0)
#endif
struct w128_tab {
    float c[64], s[64];
};
constexpr double o8k_sin(double x)
{
    double term = x, sum = x;
    for (int k = 1; k < 16; k++) {
        term *= -x * x / (double)((2 * k) * (2 * k + 1));
        sum += term;
    }
    return sum;
}
constexpr double o8k_cos(double x)
{
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 16; k++) {
        term *= -x * x / (double)((2 * k - 1) * (2 * k));
        sum += term;
    }
    return sum;
}
constexpr w128_tab o8k_make_w128()
{
    w128_tab t{};
    for (int j = 0; j < 64; j++) {
        const double a = 3.14159265358979323846 * (double)j / 64.0;
        t.c[j] = j == 32 ? 0.f : (float)o8k_cos(a);
        t.s[j] = j == 0 ? 0.f : (float)o8k_sin(a);
    }
    return t;
}
__device__ constexpr w128_tab kW128 = o8k_make_w128();          // W_128^j = c[j] - i s[j]

constexpr int O8K_WAVES = 8, O8K_THREADS = 64 * O8K_WAVES, O8K_PAIRS = O8K_WAVES / 2;
#ifndef O8K_SPIN
#define O8K_SPIN (1 << 18)                 // (A/B builds: -DO8K_SPIN=0 takes the waiting out, with wrong samples)
#endif
constexpr int O8K_SPIN_LIMIT = O8K_SPIN;

typedef __attribute__((address_space(3))) volatile int *o8k_flag_p;          // 32-bit LDS addresses: generic pointers in the
typedef float o8k_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) o8k_f2 *o8k_box_p;                 // struct would become 64-bit flat accesses
struct o8k_pair {
    o8k_flag_p my_free, my_sent, pt_free, pt_sent;             // round counters, one writer each
    o8k_box_p my_box, pt_box;                                   // [16][64]: the wave's own transpose buffers (wave-uniform bases)
    int k;
};
__device__ __forceinline__ void o8k_signal(o8k_flag_p flag, int k)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // what this wave wrote / read before is done
    *flag = k;
}
__device__ __forceinline__ void o8k_wait(o8k_flag_p flag, int k)
{
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(*flag) < k && ++spins < O8K_SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
// the wave hands a[OFF .. OFF+31] to its partner and takes the partner's 32 values in their place
template <int OFF>
__device__ __forceinline__ void o8k_swap(cf (&a)[64], o8k_pair &ps)
{
    const int lane = o8k_lane();
    const o8k_box_p mine = ps.my_box + lane, theirs = ps.pt_box + lane;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        ps.k++;
        o8k_signal(ps.my_free, ps.k);                           // my buffers may take round k
        o8k_wait(ps.pt_free, ps.k);
#pragma unroll
        for (int i = 0; i < 16; i++) theirs[64 * i] = (o8k_f2){a[OFF + 16 * r + i].x, a[OFF + 16 * r + i].y};
        o8k_signal(ps.my_sent, ps.k);
        o8k_wait(ps.pt_sent, ps.k);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const o8k_f2 t = mine[64 * i];
            a[OFF + 16 * r + i] = cf{t[0], t[1]};
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the buffers go back to the transposes
}

// the 4096-point problem of one wave, in place: a[p] = position p (sample 64 p + lane of the 4096) in and out; h_e / h_o = the
// spectrum planes of its even-bin and odd-bin 2048-point problems for this half-wave
__device__ __forceinline__ void ols4k_core(cf (&a)[64], float *buf, const float2 *s_tw, const float2 *s_w4, const float2 *h_e,
                                           const float2 *h_o, int l5, int rowoff)
{
    cf e[32], o[32];
#pragma unroll
    for (int p = 0; p < 32; p++) {
        e[p] = cadd(a[p], a[p + 32]);
        const float2 t = s_w4[64 * p + rowoff];
        o[p] = cmul<false>(csub(a[p], a[p + 32]), cf{t.x, t.y});
    }
    cf ye[32], yo[32];
    ols2k_core<2>(e, ye, buf, s_tw, h_e, s_w4, l5, rowoff);
    ols2k_core<2>(o, yo, buf, s_tw, h_o, s_w4, l5, rowoff);
#pragma unroll
    for (int p = 0; p < 32; p++) {
        const float2 t = s_w4[64 * p + rowoff];
        const cf q = cmul<true>(yo[p], cf{t.x, t.y});
        a[p] = cadd(ye[p], q);
        a[p + 32] = csub(ye[p], q);
    }
}

// O = overlap, a multiple of 64 with flt_len - 1 <= O <= 6144; a job is two blocks = 2 (8192 - O) new samples of one channel
// (beyond 4096 block B starts in front of the row for the first job: the history serves it too, and rows 64 .. O/64 - 1 of a
// block are overlap as well)
template <int O>
__global__ void __launch_bounds__(O8K_THREADS, 2)
k_fir_ols8k_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                const float2 *__restrict__ hfreq8 /* [8][1024] */, const float2 *__restrict__ twid /* [32][32] W_1024^(ab) */,
                const float2 *__restrict__ tw4k /* [2048] W_4096^n */, ols_geom G)
{
    constexpr int V = 8192 - O, JOB = 2 * V, PO = O / 64;
    static_assert(O % 64 == 0 && O <= 6144, "at least 2048 new samples per block");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_tw = reinterpret_cast<float2 *>(smem);           // [1024]
    float2 *s_w4 = s_tw + 1024;                                 // [2048]
    float2 *s_h = s_w4 + 2048;                                  // [8][1024]
    float *s_buf = reinterpret_cast<float *>(s_h + 8192);       // [16][OLS_XBUF]
    int *s_flag = reinterpret_cast<int *>(s_buf + 2 * O8K_WAVES * OLS_XBUF);                      // [8][2]
    float2 *s_wl = reinterpret_cast<float2 *>(s_flag + 2 * O8K_WAVES);                            // [64] W_8192^lane
    for (int i = threadIdx.x; i < 1024; i += O8K_THREADS) {
        s_tw[i] = twid[i];
        s_w4[i] = tw4k[i];
        s_w4[1024 + i] = tw4k[1024 + i];
#pragma unroll
        for (int j = 0; j < 8; j++) s_h[1024 * j + i] = hfreq8[1024 * j + i];
    }
    if (threadIdx.x < 2 * O8K_WAVES) s_flag[threadIdx.x] = 0;
    if (threadIdx.x < 64) {
        float sn, cs;
        sincospif((float)threadIdx.x * (1.0f / 4096.0f), &sn, &cs);
        s_wl[threadIdx.x] = make_float2(cs, -sn);
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the pair = waves p and p + 4 of the workgroup: the two waves of ONE SIMD.  Whenever one of them waits for the other, the
    // other is the wave its SIMD runs, so the waits cost no issue slots anybody wanted; with neighbouring waves (two SIMDs)
    // a waiting wave leaves its SIMD to a wave of another pair, which may be waiting as well (12.23 against 11.42 ms at 3073
    // taps, same box: -DO8K_CROSS_SIMD)
#ifdef O8K_CROSS_SIMD
    const int w = wave & 1, pair = wave >> 1, partner = wave ^ 1;
#else
    const int w = wave >> 2, pair = wave & 3, partner = wave ^ 4;
#endif
    o8k_pair ps;
    ps.my_free = (o8k_flag_p)(s_flag + 2 * wave);
    ps.my_sent = (o8k_flag_p)(s_flag + 2 * wave + 1);
    ps.pt_free = (o8k_flag_p)(s_flag + 2 * partner);
    ps.pt_sent = (o8k_flag_p)(s_flag + 2 * partner + 1);
    ps.my_box = (o8k_box_p)(reinterpret_cast<o8k_f2 *>(s_buf + (wave * 2) * OLS_XBUF));
    ps.pt_box = (o8k_box_p)(reinterpret_cast<o8k_f2 *>(s_buf + (partner * 2) * OLS_XBUF));
    ps.k = 0;
    O8K_T0();
    const long pairs_total = (long)gridDim.x * O8K_PAIRS;
    const int n = G.n;
    const int pos0 = 32 * w;

    for (long seg = (long)blockIdx.x * O8K_PAIRS + pair; seg < G.total_segs; seg += pairs_total) {
        const int c = (int)(seg / G.segs_per_channel);
        const int j0 = (int)(seg - (long)c * G.segs_per_channel) * G.seg_len;
        const int count = min(G.seg_len, G.jobs_per_channel - j0);
        const float *row = in + (size_t)c * G.in_pitch;
        float *orow = out + (size_t)c * G.out_pitch;
        const float *hrow = hist ? hist + (size_t)c * G.keep : nullptr;
#pragma unroll 1
        for (int jj = 0; jj < count; jj++) {
            const int s = (j0 + jj) * JOB;
            // block A (real parts) = [s - O, s + V), block B (imaginary parts) = [s + V - O, s + 2V): B's first O samples are
            // A's last O, the same addresses (requested again: cache hits, the other wave may hold them)
            O8K_MARK(6);                                                 // loop overhead
            cf a[64];
            const bool whole = s >= O && s + JOB <= n;
            if (whole) {
                const float *ra = row + s - O + 64 * pos0 + o8k_lane(), *rb = ra + V;
#pragma unroll
                for (int j = 0; j < 32; j++) {
                    a[j].x = __builtin_nontemporal_load(&ra[64 * j]);
                    a[32 + j].x = __builtin_nontemporal_load(&ra[64 * (j + 64)]);
                }
#pragma unroll
                for (int j = 0; j < 32; j++) {
                    a[j].y = __builtin_nontemporal_load(&rb[64 * j]);
                    a[32 + j].y = __builtin_nontemporal_load(&rb[64 * (j + 64)]);
                }
            } else {
                const int lane = o8k_lane();
#pragma unroll
                for (int j = 0; j < 64; j++) {
                    const int pos = pos0 + (j < 32 ? j : j + 32);
                    const int ia = s - O + 64 * pos + lane, ib = ia + V;
                    float xa = 0.f;
                    if (ia >= 0) xa = row[min(ia, n - 1)];
                    else if (hrow && ia >= -G.keep) xa = hrow[G.keep + ia];
                    a[j].x = ia < n ? xa : 0.f;
                    float xb;
                    if (O > 4096 && ib < 0) xb = (hrow && ib >= -G.keep) ? hrow[G.keep + ib] : 0.f;
                    else xb = row[min(ib, n - 1)];
                    a[j].y = ib < n ? xb : 0.f;
                }
            }
            O8K_PIN(a[0].x); O8K_PIN(a[63].y); O8K_PIN(a[31].y); O8K_PIN(a[32].x);
            O8K_MARK(0);                                                 // requests, awaited
            // ---- radix-2 step down over the wave's 32 positions.  Wave 1 works with the halves of its arrays exchanged (slot j =
            // position j + 32 mod 64), so that BOTH waves keep slots 0..31 and swap slots 32..63 (one code path, no register
            // shuffles where two would join): the 4096-point problem is indifferent to it -- exchanged input halves flip the
            // sign of its odd-bin branch, which exchanges the output halves the same way.
            // (the twiddles are formed where they are used: hoisted out of the job loop they would be 192 more live registers)
            const float2 wd = s_wl[o8k_lane()];                          // W_8192^lane, times W_128^(32 w) = (-i)^w
            const cf wv = w ? cf{wd.y, -wd.x} : cf{wd.x, wd.y};
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const cf sm = cadd(a[j], a[32 + j]), df = csub(a[j], a[32 + j]);
                const cf t = cmul<false>(cf{kW128.c[j], -kW128.s[j]}, wv);
                const cf dt = cmul<false>(df, t);
                a[j] = w ? dt : sm;                                      // kept: wave 0 its sums (positions 0..31), wave 1 its
                a[32 + j] = w ? sm : dt;                                 // differences (positions 32..63); the other 32 go over
            }
            O8K_PIN(a[0].x); O8K_PIN(a[63].y);
            O8K_MARK(1);                                                 // step down
            o8k_swap<32>(a, ps);
            O8K_PIN(a[32].x); O8K_PIN(a[63].y);
            O8K_MARK(2);                                                 // swap
            {
                const int ln = o8k_lane(), hf = ln >> 5;
                ols4k_core(a, s_buf + (wave * 2 + hf) * OLS_XBUF, s_tw, s_w4, s_h + (4 * hf + w) * 1024, s_h + (4 * hf + 2 + w) * 1024,
                           ln & 31, ln);
            }
            O8K_PIN(a[0].x); O8K_PIN(a[63].y); O8K_PIN(a[32].x);
            O8K_MARK(3);                                                 // the 4096-point problem
            // ---- back: wave 1 turns its results by W_8192^-n (slot j = position j + 32 mod 64), gives the 32 of positions 0..31
            // for wave 0's E' of positions 32..63, and position 32 w + j gets E' + q, position 32 w + j + 64 gets E' - q
            if (w) {
                const float2 wr = s_wl[o8k_lane()];
                const cf wu = cf{wr.x, wr.y};
#pragma unroll
                for (int p = 0; p < 64; p++) {
                    const cf t = cmul<false>(cf{kW128.c[p ^ 32], -kW128.s[p ^ 32]}, wu);
                    a[p] = cmul<true>(a[p], t);
                }
            }
            o8k_swap<32>(a, ps);
            O8K_PIN(a[32].x); O8K_PIN(a[63].y); O8K_PIN(a[0].x);
            O8K_MARK(4);                                                 // turn + swap
            const float sg = w ? -1.f : 1.f;                             // wave 0: a[j] = E', a[32 + j] = q; wave 1: the reverse
            const int lane = o8k_lane();
            const bool all_out = s + JOB <= n;
            const int jmin = PO - pos0;                                  // positions below PO are overlap
            float *oa0 = orow + s + 64 * (pos0 - PO) + lane;
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const cf y0 = cadd(a[j], a[32 + j]), d1 = csub(a[j], a[32 + j]);
                const cf y1 = cf{d1.x * sg, d1.y * sg};
                const int oa = s + 64 * (pos0 + j - PO) + lane, ob = oa + V;
                if (all_out) {
                    if (j >= jmin) {
                        __builtin_nontemporal_store(y0.x, &oa0[64 * j]);
                        __builtin_nontemporal_store(y0.y, &oa0[64 * j + V]);
                    }
                    if (PO <= 64 || j >= jmin - 64) {
                        __builtin_nontemporal_store(y1.x, &oa0[64 * (j + 64)]);
                        __builtin_nontemporal_store(y1.y, &oa0[64 * (j + 64) + V]);
                    }
                } else {
                    if (j >= jmin) {
                        if (oa < n) orow[oa] = y0.x;
                        if (ob < n) orow[ob] = y0.y;
                    }
                    if (PO <= 64 || j >= jmin - 64) {
                        if (oa + 4096 < n) orow[oa + 4096] = y1.x;
                        if (ob + 4096 < n) orow[ob + 4096] = y1.y;
                    }
                }
            }
            O8K_MARK(5);                                                 // outputs formed, stores issued
#ifdef O8K_TRACE
            o8_acc[7]++;
#endif
        }
    }
    O8K_DUMP();
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
    ols_geom G;
    G.n = n;
    G.keep = flt_len - 1;
    G.in_pitch = in_pitch;
    G.out_pitch = out_pitch;
    G.jobs_per_channel = (n + OLS_JOB - 1) / OLS_JOB;
    const size_t lds_bytes = 2 * 1024 * sizeof(float2) + (size_t)OLS_WAVES * 2 * OLS_XBUF * sizeof(float);
    // one resident set of workgroups (two per CU: 49 KB of LDS and ~240 registers per lane), grid stride over the segments
    const int tuned_per_cu = llzs_tune(LLZS_TUNE_OLS_WG_PER_CU);
    const long max_blocks = 256L * (tuned_per_cu > 0 ? tuned_per_cu : 2);
    const long slots = max_blocks * OLS_WAVES * 2;
    // jobs per segment: a half-wave walks seg_len consecutive jobs of one channel (the overlap stays in registers), at most
    // OLS_SEG.  Small batches (BASELINE config 2: 64 channels) would leave half-wave slots idle or quantise badly into
    // rounds with the full length, so they take the length that minimises rounds x (length + one job of start-up).  Large
    // batches keep the full length: on 4096 channels a shorter segment measured 2.6 % slower.
    const bool large = (long)((G.jobs_per_channel + OLS_SEG - 1) / OLS_SEG) * channels >= 4 * slots;
    int seg_len = OLS_SEG;
    if (!large) {
        double best = 1e300;
        for (int sl = OLS_SEG; sl >= 1; sl--) {
            const long segs = (long)((G.jobs_per_channel + sl - 1) / sl) * channels;
            const double cost = (double)((segs + slots - 1) / slots) * (sl + 1.0);
            if (cost < best * 0.999) { best = cost; seg_len = sl; }
        }
    }
    if (const int v = llzs_tune(LLZS_TUNE_OLS_SEG_LEN); v >= 1 && v <= 1024) seg_len = v;
    G.seg_len = seg_len;
    G.segs_per_channel = (G.jobs_per_channel + seg_len - 1) / seg_len;
    G.total_segs = (long)G.segs_per_channel * channels;
    long blocks = (G.total_segs + 2 * OLS_WAVES - 1) / (2 * OLS_WAVES);
    if (blocks > max_blocks) blocks = max_blocks;
    const float2 *hf = reinterpret_cast<const float2 *>(hfreq), *tw = reinterpret_cast<const float2 *>(twid);
    const dim3 grid((unsigned)blocks), block(OLS_THREADS);
    hipLaunchKernelGGL(k_fir_ols_chain_f32, grid, block, lds_bytes, as_stream(stream), in, out, hist, hf, tw, G);
    LLZ_LAUNCH_CHECK("k_fir_ols_f32");
    return LLZ_OK;
}

// 258 .. 1025 taps on 2048-point transforms split over the two half-waves (k_fir_ols2k_chain_f32): hfreq2 = [2][1024] complex,
// even then odd bins of DFT_2048(taps) / 2048 in natural order; twid as above; tw2k = [1024] complex W_2048^n
extern "C" int llzs_fir_ols2k_f32(const float *in, float *out, const float *hist, const float *hfreq2, const float *twid,
                                  const float *tw2k, int channels, int n, long in_pitch, long out_pitch, int flt_len,
                                  void *stream)
{
    if (!in || !out || !hfreq2 || !twid || !tw2k || channels <= 0 || n <= 0 || in_pitch < n || out_pitch < n ||
        flt_len < 2 || flt_len > 1025) {
        llzs_set_error("fir_ols2k_f32: bad arguments (flt_len=%d, 2..1025)", flt_len);
        return LLZ_ERR_ARG;
    }
    const int overlap = flt_len <= 513 ? 512 : 1024;
    const int O2K_JOB = 2 * (2048 - overlap);
    ols_geom G;
    G.n = n;
    G.keep = flt_len - 1;
    G.in_pitch = in_pitch;
    G.out_pitch = out_pitch;
    G.jobs_per_channel = (n + O2K_JOB - 1) / O2K_JOB;
    const size_t lds_bytes = 4 * 1024 * sizeof(float2) + (size_t)OLS_WAVES * 2 * OLS_XBUF * sizeof(float);
    const long max_blocks = 256L * 2;
    const long slots = max_blocks * OLS_WAVES;                  // a wave per segment
    int seg_len = OLS_SEG;
    if ((long)((G.jobs_per_channel + OLS_SEG - 1) / OLS_SEG) * channels < 4 * slots) {
        double best = 1e300;
        for (int sl = OLS_SEG; sl >= 1; sl--) {
            const long segs = (long)((G.jobs_per_channel + sl - 1) / sl) * channels;
            const double cost = (double)((segs + slots - 1) / slots) * (sl + 1.0);
            if (cost < best * 0.999) { best = cost; seg_len = sl; }
        }
    }
    G.seg_len = seg_len;
    G.segs_per_channel = (G.jobs_per_channel + seg_len - 1) / seg_len;
    G.total_segs = (long)G.segs_per_channel * channels;
    long blocks = (G.total_segs + OLS_WAVES - 1) / OLS_WAVES;
    if (blocks > max_blocks) blocks = max_blocks;
    const float2 *hf = reinterpret_cast<const float2 *>(hfreq2), *tw = reinterpret_cast<const float2 *>(twid),
                 *w2 = reinterpret_cast<const float2 *>(tw2k);
    if (overlap == 512) {
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_ols2k_chain_f32<512>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(k_fir_ols2k_chain_f32<512>, dim3((unsigned)blocks), dim3(OLS_THREADS), lds_bytes, as_stream(stream),
                           in, out, hist, hf, tw, w2, G);
    } else {
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_ols2k_chain_f32<1024>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(k_fir_ols2k_chain_f32<1024>, dim3((unsigned)blocks), dim3(OLS_THREADS), lds_bytes, as_stream(stream),
                           in, out, hist, hf, tw, w2, G);
    }
    LLZ_LAUNCH_CHECK("k_fir_ols2k_chain_f32");
    return LLZ_OK;
}

// 514 .. 3073 taps on 4096-point transforms (k_fir_ols4k_f32; overlap 1024 / 2048 / 3072 by tap count): hfreq4 = [4][1024]
// complex, plane j = bins 4m + j of DFT_4096(taps) / 4096; twid [32][32] W_1024^(ab); tw2k [1024] W_2048^n; tw4k [2048] W_4096^n
template <int O>
static int ols4k_launch(const float *in, float *out, const float *hist, const float *hfreq4, const float *twid,
                        const float *tw2k, const float *tw4k, int channels, int n, long in_pitch, long out_pitch, int flt_len,
                        void *stream)
{
    constexpr int JOB = 2 * (4096 - O);
    ols_geom G;
    G.n = n;
    G.keep = flt_len - 1;
    G.in_pitch = in_pitch;
    G.out_pitch = out_pitch;
    G.jobs_per_channel = (n + JOB - 1) / JOB;
    const size_t lds_bytes = 8 * 1024 * sizeof(float2) + (size_t)O4K_WAVES * 2 * OLS_XBUF * sizeof(float);
    const long max_blocks = 256L;                               // one 8-wave workgroup per CU
    const long slots = max_blocks * O4K_WAVES;
    int seg_len = OLS_SEG;
    if ((long)((G.jobs_per_channel + OLS_SEG - 1) / OLS_SEG) * channels < 4 * slots) {
        double best = 1e300;
        for (int sl = OLS_SEG; sl >= 1; sl--) {
            const long segs = (long)((G.jobs_per_channel + sl - 1) / sl) * channels;
            const double cost = (double)((segs + slots - 1) / slots) * sl;
            if (cost < best * 0.999) { best = cost; seg_len = sl; }
        }
    }
    G.seg_len = seg_len;
    G.segs_per_channel = (G.jobs_per_channel + seg_len - 1) / seg_len;
    G.total_segs = (long)G.segs_per_channel * channels;
    long blocks = (G.total_segs + O4K_WAVES - 1) / O4K_WAVES;
    if (blocks > max_blocks) blocks = max_blocks;
    LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_ols4k_f32<O>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(k_fir_ols4k_f32<O>, dim3((unsigned)blocks), dim3(O4K_THREADS), lds_bytes, as_stream(stream), in, out, hist,
                       reinterpret_cast<const float2 *>(hfreq4), reinterpret_cast<const float2 *>(twid),
                       reinterpret_cast<const float2 *>(tw2k), reinterpret_cast<const float2 *>(tw4k), G);
    LLZ_LAUNCH_CHECK("k_fir_ols4k_f32");
    return LLZ_OK;
}

extern "C" int llzs_fir_ols4k_f32(const float *in, float *out, const float *hist, const float *hfreq4, const float *twid,
                                  const float *tw2k, const float *tw4k, int channels, int n, long in_pitch, long out_pitch,
                                  int flt_len, void *stream)
{
    if (!in || !out || !hfreq4 || !twid || !tw2k || !tw4k || channels <= 0 || n <= 0 || in_pitch < n || out_pitch < n ||
        flt_len < 2 || flt_len > LLZS_OLS4K_MAX_TAPS) {
        llzs_set_error("fir_ols4k_f32: bad arguments (flt_len=%d, 2..%d)", flt_len, LLZS_OLS4K_MAX_TAPS);
        return LLZ_ERR_ARG;
    }
    // the smallest overlap of the ladder that holds flt_len - 1 samples: a block yields 4096 - O outputs
#define LLZ_OLS4K_GO(O) return ols4k_launch<O>(in, out, hist, hfreq4, twid, tw2k, tw4k, channels, n, in_pitch, out_pitch, flt_len, stream)
    if (flt_len <= 513) LLZ_OLS4K_GO(512);
    if (flt_len <= 769) LLZ_OLS4K_GO(768);
    if (flt_len <= 1025) LLZ_OLS4K_GO(1024);
    if (flt_len <= 1537) LLZ_OLS4K_GO(1536);
    if (flt_len <= 2049) LLZ_OLS4K_GO(2048);
    if (flt_len <= 2561) LLZ_OLS4K_GO(2560);
    LLZ_OLS4K_GO(3072);
#undef LLZ_OLS4K_GO
}

// 2 .. 6145 taps on 8192-point transforms (k_fir_ols8k_f32: a pair of waves per job; overlap 1536 / 2304 / ... / 6144 by tap
// count): hfreq8 = [8][1024] complex, plane j = bins 8m + j of DFT_8192(taps) / 8192; twid [32][32] W_1024^(ab); tw4k [2048] W_4096^n
template <int O>
static int ols8k_launch(const float *in, float *out, const float *hist, const float *hfreq8, const float *twid,
                        const float *tw4k, int channels, int n, long in_pitch, long out_pitch, int flt_len, void *stream)
{
    constexpr int JOB = 2 * (8192 - O);
    ols_geom G;
    G.n = n;
    G.keep = flt_len - 1;
    G.in_pitch = in_pitch;
    G.out_pitch = out_pitch;
    G.jobs_per_channel = (n + JOB - 1) / JOB;
    const size_t lds_bytes = (1024 + 2048 + 8192) * sizeof(float2) + (size_t)O8K_WAVES * 2 * OLS_XBUF * sizeof(float) +
                             2 * O8K_WAVES * sizeof(int) + 64 * sizeof(float2);
    const long max_blocks = 256L;                               // one 8-wave workgroup per CU
    const long slots = max_blocks * O8K_PAIRS;
    int seg_len = OLS_SEG;
    if ((long)((G.jobs_per_channel + OLS_SEG - 1) / OLS_SEG) * channels < 4 * slots) {
        double best = 1e300;
        for (int sl = OLS_SEG; sl >= 1; sl--) {
            const long segs = (long)((G.jobs_per_channel + sl - 1) / sl) * channels;
            const double cost = (double)((segs + slots - 1) / slots) * sl;
            if (cost < best * 0.999) { best = cost; seg_len = sl; }
        }
    }
    G.seg_len = seg_len;
    G.segs_per_channel = (G.jobs_per_channel + seg_len - 1) / seg_len;
    G.total_segs = (long)G.segs_per_channel * channels;
    long blocks = (G.total_segs + O8K_PAIRS - 1) / O8K_PAIRS;
    if (blocks > max_blocks) blocks = max_blocks;
    LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_ols8k_f32<O>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(k_fir_ols8k_f32<O>, dim3((unsigned)blocks), dim3(O8K_THREADS), lds_bytes, as_stream(stream), in, out, hist,
                       reinterpret_cast<const float2 *>(hfreq8), reinterpret_cast<const float2 *>(twid),
                       reinterpret_cast<const float2 *>(tw4k), G);
    LLZ_LAUNCH_CHECK("k_fir_ols8k_f32");
    return LLZ_OK;
}

extern "C" int llzs_fir_ols8k_f32(const float *in, float *out, const float *hist, const float *hfreq8, const float *twid,
                                  const float *tw4k, int channels, int n, long in_pitch, long out_pitch, int flt_len, void *stream)
{
    if (!in || !out || !hfreq8 || !twid || !tw4k || channels <= 0 || n <= 0 || in_pitch < n || out_pitch < n ||
        flt_len < 2 || flt_len > LLZS_OLS8K_MAX_TAPS) {
        llzs_set_error("fir_ols8k_f32: bad arguments (flt_len=%d, 2..%d)", flt_len, LLZS_OLS8K_MAX_TAPS);
        return LLZ_ERR_ARG;
    }
#define LLZ_OLS8K_GO(O) return ols8k_launch<O>(in, out, hist, hfreq8, twid, tw4k, channels, n, in_pitch, out_pitch, flt_len, stream)
    if (flt_len <= 1537) LLZ_OLS8K_GO(1536);
    // (a job costs 27 .. 31 us whatever the overlap between 2048 and 2560 -- 10.96 / 10.92 / 10.90 ms for 2048 / 2304 / 2560 on
    // the headline batch -- so the ladder is coarse there)
    if (flt_len <= 2305) LLZ_OLS8K_GO(2304);
    if (flt_len <= 2561) LLZ_OLS8K_GO(2560);
    if (flt_len <= 3073) LLZ_OLS8K_GO(3072);
    if (flt_len <= 3585) LLZ_OLS8K_GO(3584);
    if (flt_len <= 4097) LLZ_OLS8K_GO(4096);
    if (flt_len <= 5121) LLZ_OLS8K_GO(5120);
    LLZ_OLS8K_GO(6144);
#undef LLZ_OLS8K_GO
}

#ifdef O8K_TRACE
extern "C" int llzs_o8k_trace_read(unsigned long long *dst, int count)
{
    LLZ_HIP_CHECK(hipDeviceSynchronize());
    LLZ_HIP_CHECK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(o8k_trace_buf), sizeof(unsigned long long) * (size_t)count));
    return LLZ_OK;
}
#endif
