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
#include <mutex>
#include "common.hpp"
#include <type_traits>
#include <stdlib.h>

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


// ---------------------------------------------------------------------------------------------------------------
// k_iir_cascade_pipe_f32: the fast path.  A recurrence has no parallelism along time inside one section, and 1024
// channels x one lane each would leave the chip at 1/64 occupancy.  Two independent sources of parallelism are used:
//
//  (1) stage pipeline: a workgroup owns ONE channel, wave s runs biquad section s.  Waves march over 1024-sample
//      chunks in a software pipeline (wave s works on chunk t-s at step t) and hand chunks to the next section through
//      LDS ([k][lane] layout: the consumer lane reads back exactly what the same lane of the producer wrote, so the
//      8 KB slots need neither padding nor a transpose).  Section 0 reads float32 from HBM one chunk ahead, the
//      last section writes float32 back; each lane owns 16 consecutive samples = one 64-byte segment.
//  (2) lanes along time: inside a chunk lane l owns samples [16 l, 16 l + 16).  The feed-forward part needs no scan
//      (the two samples in front of a lane come from lane l-1).  The feedback part is an affine map of the 2-vector
//      (y[n-1], y[n-2]):  after 16 samples  s' = P s + z  with P = A^16, A = [[-a1,-a2],[1,0]], and z the lane's
//      zero-state response.  A 6-step Hillis-Steele scan over the z vectors with the host-built powers P^(2^d), plus
//      P^l applied to the chunk's incoming state, gives every lane its exact start state; the lane then runs the true
//      recurrence from there.  All of it in double: 7 DFMA per sample and section + ~30 per 16 samples for the scan.
//
// Bound: FP64 VALU (56 DFMA per sample for 8 sections) is within a factor ~1 of the HBM time for 8 B/sample; the
// kernel is compute/latency bound, not a streaming kernel.

// cross-lane moves of a double as two DPP dword moves (VALU, no LDS round trip). Lanes whose source is out of
// range, or whose row is masked off, receive 0.0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_ROW_SHR = 0x110;     // + n: shift right by n inside each row of 16 lanes
constexpr int DPP_WAVE_SHR1 = 0x138;   // whole-wave shift right by one lane
constexpr int DPP_BCAST15 = 0x142;     // lane 15 of each row -> the next row
constexpr int DPP_BCAST31 = 0x143;     // lane 31 -> rows 2 and 3

__device__ __forceinline__ double lane63_f64(double v)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// scalar helpers of the pipelined kernel for both working precisions
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_(double v) { return dpp_f64<CTRL, ROW_MASK>(v); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ double lane63_(double v) { return lane63_f64(v); }
__device__ __forceinline__ float lane63_(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }

// R = double: the general form.  R = float: taken only when the host has checked every section of the cascade for a low
// rounding-noise gain (see llz_iir_cascade_mc_init): same algorithm, float32 arithmetic and 4 KB hand-over slots.
// RR samples per lane and chunk (16 in both shipped instantiations)
template <typename R, int RR>
__global__ void __launch_bounds__(1024)
k_iir_cascade_pipe(const float *__restrict__ in, float *__restrict__ out, const double *__restrict__ coef,
                       const double *__restrict__ pd /* [S][6][4] */, const double *__restrict__ pl /* [S][64][12] */,
                       const double *__restrict__ state_in, double *__restrict__ state, int nchunks_total, long in_pitch,
                       long out_pitch, int stages, int segs, int seg_chunks, int warm)
{
    extern __shared__ __attribute__((aligned(16))) char slots_raw[];
    R *slots = reinterpret_cast<R *>(slots_raw);        // [stages-1][(64 * RR)]: one per section boundary
    const int lane = threadIdx.x & 63;
    const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // this wave's section (wave-uniform)
    // (3) segments along time when there are too few channels to fill the chip: workgroup (c, seg) owns chunks
    //     [seg*seg_chunks, (seg+1)*seg_chunks) of channel c.  Segment 0 starts from the channel's true state; a later one
    //     starts `warm` chunks early from the zero state and discards those outputs: the host has checked on the
    //     cascade's own homogeneous responses that a state error decays below 1e-13 within `warm` chunks.
    const int c = blockIdx.x / segs, seg = blockIdx.x - c * segs;
    const int skip = seg > 0 ? warm : 0;                                   // leading chunks computed but not written
    const int chunk0 = seg * seg_chunks - skip;
    const int nchunks = min(nchunks_total, (seg + 1) * seg_chunks) - chunk0;
    const R b0 = (R)coef[5 * s + 0], b1 = (R)coef[5 * s + 1], b2 = (R)coef[5 * s + 2];
    const R a1 = (R)coef[5 * s + 3], a2 = (R)coef[5 * s + 4];
    R P[4][4];                                                       // P^1, P^2, P^4, P^8 (wave-uniform)
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
        for (int j = 0; j < 4; j++) P[d][j] = (R)pd[(s * 6 + d) * 4 + j];
    // per-lane powers: L = P^lane (chunk state), M1 = P^(lane%16 + 1) (row hand-over), M2 = P^(lane%32 + 1) (half)
    const double *plane_tab = pl + (size_t)(s * 64 + lane) * 12;
    const R L00 = (R)plane_tab[0], L01 = (R)plane_tab[1], L10 = (R)plane_tab[2], L11 = (R)plane_tab[3];
    const R M1a = (R)plane_tab[4], M1b = (R)plane_tab[5], M1c = (R)plane_tab[6], M1d = (R)plane_tab[7];
    const R M2a = (R)plane_tab[8], M2b = (R)plane_tab[9], M2c = (R)plane_tab[10], M2d = (R)plane_tab[11];
    // the frame's start state is read from state_in, its end state written to another buffer (state): segment 0 and the
    // last segment of a channel are different workgroups of one launch, and nothing orders them
    const double *st_in = state_in + ((size_t)c * stages + s) * 4;
    double *st = state + ((size_t)c * stages + s) * 4;
    R su1 = 0, su2 = 0, sy1 = 0, sy2 = 0;                   // x(n-1), x(n-2), y(n-1), y(n-2)
    if (seg == 0) { su1 = (R)st_in[0]; su2 = (R)st_in[1]; sy1 = (R)st_in[2]; sy2 = (R)st_in[3]; }

    const float *row = in + (size_t)c * in_pitch + (size_t)chunk0 * (64 * RR) + lane * RR;
    float *orow = out + (size_t)c * out_pitch + (size_t)chunk0 * (64 * RR) + lane * RR;
    R *my_in = slots + (size_t)(s > 0 ? s - 1 : 0) * (64 * RR) + lane;    // boundary s-1 | s
    R *my_out = slots + (size_t)s * (64 * RR) + lane;                      // boundary s | s+1 (unused by the last)
    const bool first = (s == 0), last = (s == stages - 1);

    float4 pre[RR / 4];
    if (first && nchunks > 0) {
#pragma unroll
        for (int q = 0; q < RR / 4; q++) pre[q] = *reinterpret_cast<const float4 *>(row + 4 * q);
    }
    const int steps = nchunks + stages - 1;
    for (int t = 0; t < steps; t++) {
        const int chunk = t - s;
        const bool active = chunk >= 0 && chunk < nchunks;
        R u[RR];
        if (active) {
            if (first) {
#pragma unroll
                for (int q = 0; q < RR / 4; q++) {
                    u[4 * q + 0] = (R)pre[q].x; u[4 * q + 1] = (R)pre[q].y;
                    u[4 * q + 2] = (R)pre[q].z; u[4 * q + 3] = (R)pre[q].w;
                }
                if (chunk + 1 < nchunks) {
#pragma unroll
                    for (int q = 0; q < RR / 4; q++)
                        pre[q] = *reinterpret_cast<const float4 *>(row + (size_t)(chunk + 1) * (64 * RR) + 4 * q);
                }
            } else {
#pragma unroll
                for (int k = 0; k < RR; k++) u[k] = my_in[k * 64];
            }
        }
        __syncthreads();                           // every section has taken its input: slots may be rewritten
        if (active) {
            // the two samples in front of this lane: from lane-1, or from the previous chunk for lane 0
            R um1 = dpp_<DPP_WAVE_SHR1, 0xF>(u[RR - 1]), um2 = dpp_<DPP_WAVE_SHR1, 0xF>(u[RR - 2]);
            if (lane == 0) { um1 = su1; um2 = su2; }
            const R nu1 = lane63_(u[RR - 1]), nu2 = lane63_(u[RR - 2]);
            // feed-forward part in place: u[k] <- b0 u[k] + b1 u[k-1] + b2 u[k-2]   (same association as the oracle)
            {
                R p1 = um1, p2 = um2;
#pragma unroll
                for (int k = 0; k < RR; k++) {
                    const R x = u[k];
                    R acc = b0 * x;
                    acc = fma_(b1, p1, acc);
                    acc = fma_(b2, p2, acc);
                    u[k] = acc;
                    p2 = p1; p1 = x;
                }
            }
            // zero-state response of this lane's 16 samples -> z = (y[15], y[14])
            R z1 = 0.0, z2 = 0.0;
#pragma unroll
            for (int k = 0; k < RR; k++) {
                // the term with the OLDER output first: one DFMA latency per sample on the critical path, not two
                const R y = fma_(-a1, z1, fma_(-a2, z2, u[k]));
                z2 = z1; z1 = y;
            }
            // inclusive scan of the affine maps over the lanes, all with DPP moves:
            //   inside each row of 16 lanes: z_l <- z_l + P^d z_(l-d), d = 1,2,4,8 (out-of-row sources read as 0)
#define LLZ_ROW_STEP(D, SH)                                                                                   \
            {                                                                                                 \
                const R q1 = dpp_<DPP_ROW_SHR + SH, 0xF>(z1), q2 = dpp_<DPP_ROW_SHR + SH, 0xF>(z2);   \
                z1 = fma_(P[D][0], q1, fma_(P[D][1], q2, z1));                              \
                z2 = fma_(P[D][2], q1, fma_(P[D][3], q2, z2));                              \
            }
            LLZ_ROW_STEP(0, 1) LLZ_ROW_STEP(1, 2) LLZ_ROW_STEP(2, 4) LLZ_ROW_STEP(3, 8)
#undef LLZ_ROW_STEP
            //   rows 1 and 3 take the total of the row before them, advanced by (lane%16 + 1) lanes
            {
                const R q1 = dpp_<DPP_BCAST15, 0xA>(z1), q2 = dpp_<DPP_BCAST15, 0xA>(z2);
                z1 = fma_(M1a, q1, fma_(M1b, q2, z1));
                z2 = fma_(M1c, q1, fma_(M1d, q2, z2));
            }
            //   the upper half takes the total of the lower half, advanced by (lane%32 + 1) lanes
            {
                const R q1 = dpp_<DPP_BCAST31, 0xC>(z1), q2 = dpp_<DPP_BCAST31, 0xC>(z2);
                z1 = fma_(M2a, q1, fma_(M2b, q2, z1));
                z2 = fma_(M2c, q1, fma_(M2d, q2, z2));
            }
            // exact state in front of this lane: exclusive prefix + P^lane applied to the chunk's incoming state
            const R e1 = dpp_<DPP_WAVE_SHR1, 0xF>(z1), e2 = dpp_<DPP_WAVE_SHR1, 0xF>(z2);
            R y1 = fma_(L00, sy1, fma_(L01, sy2, e1));
            R y2 = fma_(L10, sy1, fma_(L11, sy2, e2));
            // the true recurrence from that state
#pragma unroll
            for (int k = 0; k < RR; k++) {
                const R y = fma_(-a1, y1, fma_(-a2, y2, u[k]));
                u[k] = y;
                y2 = y1; y1 = y;
            }
            su1 = nu1; su2 = nu2;
            sy1 = lane63_(y1); sy2 = lane63_(y2);
            if (last) {
                float *dst = orow + (size_t)chunk * (64 * RR);
                if (chunk >= skip) {                              // warm-up chunks of a later segment are not written
#pragma unroll
                    for (int q = 0; q < RR / 4; q++)
                        *reinterpret_cast<float4 *>(dst + 4 * q) = make_float4((float)u[4 * q], (float)u[4 * q + 1],
                                                                                (float)u[4 * q + 2], (float)u[4 * q + 3]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < RR; k++) my_out[k * 64] = u[k];
            }
        }
        __syncthreads();                           // outputs visible before the next step's reads
    }
    if (lane == 0 && seg == segs - 1) { st[0] = (double)su1; st[1] = (double)su2; st[2] = (double)sy1; st[3] = (double)sy2; }
}


// ---------------------------------------------------------------------------------------------------------------
// k_iir_cascade_wave_pk: the float32 wave-autonomous cascade on PACKED float32 arithmetic (v_pk_fma_f32: two FMAs per
// lane and issue slot).  The kernel above is bound by VALU issue (about 160 instructions per section and 1024-sample
// chunk), so the lane's 16 samples are held as 8 pairs U[j] = (u[j], u[j+8]) and every part of the section step is
// rewritten so that both halves of a pair do the same thing:
//   * feed-forward part: acc[j] = b0*U[j] + b1*U[j-1] + b2*U[j-2] is pair-aligned by construction (U[-1] = (u[-1], u[7]));
//   * feedback part: the two half runs [0,8) and [8,16) run as ONE packed 8-step recurrence from the zero state; a half
//     run's true start state then enters through the section's homogeneous responses h1[k], h2[k] (k < 8, wave-uniform,
//     host-computed): y[k] = w[k] + h1[k]*y[-1] + h2[k]*y[-2] -- again two independent packed FMAs per pair;
//   * the lane scan carries (z1, z2) as one pair: each step is two packed FMAs on broadcast halves.
// About 100 instructions per section and chunk instead of 160.  Tables: pl32 [S][64][12] as above (read into LDS with the
// 2x2 matrices transposed so that matrix columns are register pairs), ph32 [S][24] = (h1[k], h2[k]) k < 8, then the
// section's b0 b1 b2 a1 a2 and 3 pad.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ f2 dpp2_(f2 v) { return f2{dpp_<CTRL, ROW_MASK>(v.x), dpp_<CTRL, ROW_MASK>(v.y)}; }

typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const volatile f4v lds_cv_f4v;
struct pk_tabs { f4v l0, l1, l2, p0, p1, p2, p3; };   // per lane: P^lane, P^(lane%16+1), P^(lane%32+1); uniform: P^(2^d), d < 4

// one section over the lane's 16 samples U[j] = (u[j], u[j + 8]); H = (h1[k], h2[k]) k < 8, cc = b0 b1 b2 a1 a2
__device__ __forceinline__ void pk_section(f2 (&U)[8], const f16v H, const f8v cc, const pk_tabs &T, const int lane,
                                           float &su1, float &su2, float &sy1, float &sy2)
{
    const float b0 = cc[0], b1 = cc[1], b2 = cc[2], a1 = cc[3], a2 = cc[4];
    float um1 = dpp_<DPP_WAVE_SHR1, 0xF>(U[7].y), um2 = dpp_<DPP_WAVE_SHR1, 0xF>(U[6].y);
    if (lane == 0) { um1 = su1; um2 = su2; }
    su1 = lane63_(U[7].y); su2 = lane63_(U[6].y);
    {   // feed-forward part in place, from the top so that U[j-1], U[j-2] are still the inputs
        const f2 Um1 = f2{um1, U[7].x}, Um2 = f2{um2, U[6].x};
        const f2 B0 = splat(b0), B1 = splat(b1), B2 = splat(b2);
#pragma unroll
        for (int j = 7; j >= 2; j--) U[j] = pk_fma(B2, U[j - 2], pk_fma(B1, U[j - 1], B0 * U[j]));
        U[1] = pk_fma(B2, Um1, pk_fma(B1, U[0], B0 * U[1]));
        U[0] = pk_fma(B2, Um2, pk_fma(B1, Um1, B0 * U[0]));
    }
    {   // both half runs from the zero state: U[j] <- (w[j], w[j + 8])
        const f2 A1 = splat(-a1), A2 = splat(-a2);
        f2 Z1 = U[0], Z2 = splat(0.f);
        U[1] = pk_fma(A1, U[0], U[1]);
        Z2 = Z1; Z1 = U[1];
#pragma unroll
        for (int j = 2; j < 8; j++) {
            const f2 Y = pk_fma(A1, Z1, pk_fma(A2, Z2, U[j]));
            U[j] = Y;
            Z2 = Z1; Z1 = Y;
        }
    }
    // the lane's zero-state end state (yz[15], yz[14]): the second half run started from (w[7], w[6])
    f2 ZZ;
    ZZ.x = fma_(H[14], U[7].x, fma_(H[15], U[6].x, U[7].y));
    ZZ.y = fma_(H[12], U[7].x, fma_(H[13], U[6].x, U[6].y));
#define LLZ_SCAN_STEP(CTRL, MASK, M)                                                                                 \
    {                                                                                                                \
        const f2 Q = dpp2_<CTRL, MASK>(ZZ);                                                                          \
        ZZ = pk_fma(f2{M.x, M.y}, splat(Q.x), pk_fma(f2{M.z, M.w}, splat(Q.y), ZZ));                                 \
    }
    LLZ_SCAN_STEP(DPP_ROW_SHR + 1, 0xF, T.p0) LLZ_SCAN_STEP(DPP_ROW_SHR + 2, 0xF, T.p1)
    LLZ_SCAN_STEP(DPP_ROW_SHR + 4, 0xF, T.p2) LLZ_SCAN_STEP(DPP_ROW_SHR + 8, 0xF, T.p3)
    LLZ_SCAN_STEP(DPP_BCAST15, 0xA, T.l1) LLZ_SCAN_STEP(DPP_BCAST31, 0xC, T.l2)
#undef LLZ_SCAN_STEP
    // the lane's start state (y[-1], y[-2])
    const f2 Y0 = pk_fma(f2{T.l0.x, T.l0.y}, splat(sy1), pk_fma(f2{T.l0.z, T.l0.w}, splat(sy2), dpp2_<DPP_WAVE_SHR1, 0xF>(ZZ)));
    // the true (y[7], y[6]): the start state of the second half run
    const float y7 = fma_(H[14], Y0.x, fma_(H[15], Y0.y, U[7].x));
    const float y6 = fma_(H[12], Y0.x, fma_(H[13], Y0.y, U[6].x));
    const f2 SA = f2{Y0.x, y7}, SB = f2{Y0.y, y6};
#pragma unroll
    for (int j = 0; j < 8; j++) U[j] = pk_fma(splat(H[2 * j]), SA, pk_fma(splat(H[2 * j + 1]), SB, U[j]));
    sy1 = lane63_(U[7].y); sy2 = lane63_(U[6].y);
}

// S = the number of sections exactly
// (forcing four waves per SIMD with amdgpu_waves_per_eu spills 6 registers inside the loop: 2.54 against 2.34 ms)
template <int S>
__global__ void __launch_bounds__(256)
k_iir_cascade_wave_pk(const float *__restrict__ in, float *__restrict__ out,
                      const float *__restrict__ pd32 /* [S][16], P^(2^d) d < 4, row major 2x2 each */,
                      const float *__restrict__ pl32 /* [S][64][12] */, const float *__restrict__ ph32 /* [S][24] */,
                      const double *__restrict__ state_in, double *__restrict__ state, int nchunks_total, long in_pitch,
                      long out_pitch, int segs, int seg_chunks, int warm, long items)
{
    __shared__ __attribute__((aligned(16))) float s_pl[S * 64 * 12];
    __shared__ __attribute__((aligned(16))) float s_pd[S * 16];
    // (m00, m01, m10, m11) -> (m00, m10, m01, m11): columns become aligned pairs
    for (int e = threadIdx.x; e < S * 768; e += 256) s_pl[e] = pl32[(e & ~3) | ((e & 1) << 1) | ((e >> 1) & 1)];
    if (threadIdx.x < S * 16) {
        const int e = threadIdx.x;
        s_pd[e] = pd32[(e & ~3) | ((e & 1) << 1) | ((e >> 1) & 1)];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (item >= items) return;
    const int c = (int)(item / segs), seg = (int)(item - (long)c * segs);
    const int skip = seg > 0 ? warm : 0;
    const int chunk0 = seg * seg_chunks - skip;
    const int nchunks = min(nchunks_total, (seg + 1) * seg_chunks) - chunk0;

    float su1[S], su2[S], sy1[S], sy2[S];
#pragma unroll
    for (int s = 0; s < S; s++) {
        su1[s] = su2[s] = sy1[s] = sy2[s] = 0;
        if (seg == 0) {
            const double *st = state_in + ((size_t)c * S + s) * 4;
            su1[s] = (float)st[0]; su2[s] = (float)st[1]; sy1[s] = (float)st[2]; sy2[s] = (float)st[3];
        }
    }
    const float *row = in + (size_t)c * in_pitch + (size_t)chunk0 * 1024 + lane * 16;
    float *orow = out + (size_t)c * out_pitch + (size_t)chunk0 * 1024 + lane * 16;
    float4 pre[4];
    if (nchunks > 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) pre[q] = *reinterpret_cast<const float4 *>(row + 4 * q);
    }
    // A section's constants are fetched one section AHEAD into the other of two register sets: 24 wave-uniform dwords
    // through the scalar cache, 7 x 16 bytes of powers from LDS.  With two waves per SIMD a load that is waited for where
    // it is issued costs its whole latency in every section, and the compiler sinks invariant loads to their first use:
    // hence the volatile LDS reads and the written-out scalar loads, whose wait names the registers (which orders
    // their users behind it).  Both kinds count on lgkmcnt; one wait per section covers all of them.
    f16v hA, hB; f8v cA, cB;
    pk_tabs TA, TB;
#define LLZ_PK_TIE asm volatile("" : "+v"(U[0]), "+v"(U[7]))
#define LLZ_PK_FETCH(SEC, HX, CX, TX)                                                                                \
    {                                                                                                                \
        const lds_cv_f4v *tl = (const lds_cv_f4v *)(s_pl + ((SEC) * 64 + lane) * 12);                                \
        const lds_cv_f4v *tp = (const lds_cv_f4v *)(s_pd + (SEC) * 16);                                              \
        TX.l0 = tl[0]; TX.l1 = tl[1]; TX.l2 = tl[2];                                                                 \
        TX.p0 = tp[0]; TX.p1 = tp[1]; TX.p2 = tp[2]; TX.p3 = tp[3];                                                  \
        asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx8 %1, %2, %4"                                        \
                     : "=&s"(HX), "=&s"(CX) : "s"(ph32), "n"((SEC) * 96), "n"((SEC) * 96 + 64) : "memory");           \
        LLZ_PK_TIE;                                                                                                  \
    }
    // (volatile asm statements keep their order; the empty ones name U[0] and U[7] so that the scheduler cannot move a
    //  section's arithmetic across its wait and the next section's fetch)
#define LLZ_PK_WAIT(HX, CX)                                                                                          \
    {                                                                                                                \
        LLZ_PK_TIE;                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(HX), "+s"(CX));                                                   \
    }
    f2 U[8];                                                           // U[j] = (u[j], u[j + 8])
    U[0] = U[7] = splat(0.f);
    LLZ_PK_FETCH(0, hA, cA, TA)
    for (int chunk = 0; chunk < nchunks; chunk++) {
        U[0] = f2{pre[0].x, pre[2].x}; U[1] = f2{pre[0].y, pre[2].y}; U[2] = f2{pre[0].z, pre[2].z}; U[3] = f2{pre[0].w, pre[2].w};
        U[4] = f2{pre[1].x, pre[3].x}; U[5] = f2{pre[1].y, pre[3].y}; U[6] = f2{pre[1].z, pre[3].z}; U[7] = f2{pre[1].w, pre[3].w};
        if (chunk + 1 < nchunks) {
#pragma unroll
            for (int q = 0; q < 4; q++) pre[q] = *reinterpret_cast<const float4 *>(row + (size_t)(chunk + 1) * 1024 + 4 * q);
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            if ((s & 1) == 0) {
                LLZ_PK_WAIT(hA, cA);
                if (s + 1 < S) LLZ_PK_FETCH(s + 1, hB, cB, TB)
                pk_section(U, hA, cA, TA, lane, su1[s], su2[s], sy1[s], sy2[s]);
                if (s + 1 == S) LLZ_PK_FETCH(0, hA, cA, TA)            // odd S: set A is free only now (once per chunk)
            } else {
                LLZ_PK_WAIT(hB, cB);
                LLZ_PK_FETCH((s + 1 < S ? s + 1 : 0), hA, cA, TA)
                pk_section(U, hB, cB, TB, lane, su1[s], su2[s], sy1[s], sy2[s]);
            }
        }
        if (chunk >= skip) {
            float *dst = orow + (size_t)chunk * 1024;
            *reinterpret_cast<float4 *>(dst) = make_float4(U[0].x, U[1].x, U[2].x, U[3].x);
            *reinterpret_cast<float4 *>(dst + 4) = make_float4(U[4].x, U[5].x, U[6].x, U[7].x);
            *reinterpret_cast<float4 *>(dst + 8) = make_float4(U[0].y, U[1].y, U[2].y, U[3].y);
            *reinterpret_cast<float4 *>(dst + 12) = make_float4(U[4].y, U[5].y, U[6].y, U[7].y);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // nothing in flight when the wave ends
#undef LLZ_PK_FETCH
#undef LLZ_PK_TIE
#undef LLZ_PK_WAIT
    if (lane == 0 && seg == segs - 1) {
#pragma unroll
        for (int s = 0; s < S; s++) {
            double *st = state + ((size_t)c * S + s) * 4;
            st[0] = (double)su1[s]; st[1] = (double)su2[s]; st[2] = (double)sy1[s]; st[3] = (double)sy2[s];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// k_iir_cascade_wave_pk32: the packed float32 cascade with 32 samples per lane and the b0 gains folded out.
//
// In k_iir_cascade_wave_pk a section costs ~114 vector instructions per 1024-sample chunk, of which the part that does not
// grow with the lane's run -- the lane scan (6 steps), the state hand-over and the half-run end values -- is ~45.  With
// 32 samples per lane (16 pairs U[j] = (u[j], u[j+16]), 2048-sample chunks) that part is paid once per 32 samples instead
// of once per 16.  And because a section's b0 only scales its output, the cascade's b0's are folded into ONE gain applied
// to the input: a section computes  w = u + b1' u[-1] + b2' u[-2]  (b' = b / b0, two packed FMAs per pair instead of a
// multiply and two FMAs), and its delay-line states are kept scaled by the product of the b0's still to come (xfac for the
// inputs, yfac for the outputs; the handle's state stays in the true convention: scaled on load, unscaled on store).
// Per sample and section: 2 (feed-forward) + 2 (zero-state recurrence) + 2 (start-state correction) packed-pair FMAs and
// ~1.4 instructions of scan and bookkeeping, against 7 + 2.8 before.
// Tables: pd32 / pl32 as above but for P = A^32; ph32 [S][40] = (h1[k], h2[k]) k < 16, then 1 b1' b2' a1 a2 xfac yfac pad.
// The 32 h values of a section are fetched when the section starts (they are first used after its ~64 recurrence
// instructions); the 8 coefficients and the scan powers one section ahead, as before.
template <int S>
__global__ void __launch_bounds__(256)
k_iir_cascade_wave_pk32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ pd32,
                        const float *__restrict__ pl32, const float *__restrict__ ph32 /* [S][40] */,
                        const double *__restrict__ state_in, double *__restrict__ state, int nchunks_total, long in_pitch,
                        long out_pitch, int segs, int seg_chunks, int warm, long items, float in_gain)
{
    constexpr int HP = 16, RUN = 32, CHUNK = 64 * RUN;
    // per-lane scan powers: P^lane for 64 lanes, P^(lane%16+1) for 16, P^(lane%32+1) for 32 (the global table repeats the
    // last two for every lane; keeping one copy leaves room for three workgroups per CU next to the turn buffers)
    __shared__ __attribute__((aligned(16))) float s_pl[S * 448];
    __shared__ __attribute__((aligned(16))) float s_pd[S * 16];
    // A lane owns 32 CONSECUTIVE samples (the recurrence runs along them), i.e. 128 bytes.  Loaded straight into the lane
    // that consumes them, a wave instruction touches 64 different 128-byte lines for 16 bytes each, and the eight
    // instructions of a chunk fetch every line eight times over from L2 (measured: the kernel sat on L2 -> L1 bandwidth,
    // 2.3 ms whatever the arithmetic cost).  So a chunk travels between HBM and registers in LINEAR order (a wave
    // instruction = 1 KB contiguous) and is turned lane-major through a wave-private LDS buffer: 32 floats of data per 36 of
    // pitch, which makes both the linear 16-byte accesses and the per-lane 16-byte accesses bank-conflict free.
    __shared__ __attribute__((aligned(16))) float s_turn[4][CHUNK + CHUNK / 8];
    for (int e = threadIdx.x; e < S * 448; e += 256) {
        const int sec = e / 448, r = e - sec * 448;
        // r < 256: P^lane (lane = r / 4); r < 320: P^(i+1), i = (r - 256) / 4 < 16; else P^(i+1), i = (r - 320) / 4 < 32
        const int lane_src = r < 256 ? r >> 2 : (r < 320 ? (r - 256) >> 2 : (r - 320) >> 2);
        const int part = r < 256 ? 0 : (r < 320 ? 4 : 8);
        const int el = r & 3, el_t = ((el & 1) << 1) | ((el >> 1) & 1);       // (m00, m01, m10, m11) -> (m00, m10, m01, m11)
        s_pl[e] = pl32[(sec * 64 + lane_src) * 12 + part + el_t];
    }
    if (threadIdx.x < S * 16) {
        const int e = threadIdx.x;
        s_pd[e] = pd32[(e & ~3) | ((e & 1) << 1) | ((e >> 1) & 1)];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (item >= items) return;
    const int c = (int)(item / segs), seg = (int)(item - (long)c * segs);
    const int skip = seg > 0 ? warm : 0;
    const int chunk0 = seg * seg_chunks - skip;
    const int nchunks = min(nchunks_total, (seg + 1) * seg_chunks) - chunk0;

    float su1[S], su2[S], sy1[S], sy2[S];
#pragma unroll
    for (int s = 0; s < S; s++) {
        su1[s] = su2[s] = sy1[s] = sy2[s] = 0;
        if (seg == 0) {
            const double *st = state_in + ((size_t)c * S + s) * 4;
            const float xf = ph32[s * 40 + 37], yf = ph32[s * 40 + 38];
            su1[s] = (float)st[0] * xf; su2[s] = (float)st[1] * xf; sy1[s] = (float)st[2] * yf; sy2[s] = (float)st[3] * yf;
        }
    }
    // linear order: instruction q of a chunk moves floats [256 q, 256 q + 256), 16 bytes per lane
    const float *row = in + (size_t)c * in_pitch + (size_t)chunk0 * CHUNK + 4 * lane;
    float *orow = out + (size_t)c * out_pitch + (size_t)chunk0 * CHUNK + 4 * lane;
    float *turn = s_turn[threadIdx.x >> 6];
    float *t_lin = turn + 4 * lane + 4 * (lane >> 3);       // + 288 q: float 256 q + 4 lane at pitch 36 per 32
    float *t_own = turn + 36 * lane;                        // + 4 j: the lane's own run
    float4 pre[8];
    if (nchunks > 0) {
#pragma unroll
        for (int q = 0; q < 8; q++) { const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(row + 256 * q)); pre[q] = make_float4(t.x, t.y, t.z, t.w); }
    }
    f16v h0, h1;                    // the current section's (h1[k], h2[k]): k < 8 and 8 <= k < 16
    f8v cA, cB;
    pk_tabs TA, TB;
#define LLZ_PK_TIE asm volatile("" : "+v"(U[0]), "+v"(U[HP - 1]))
#define LLZ_PK_FETCH(SEC, CX, TX)                                                                                    \
    {                                                                                                                \
        const float *ts = s_pl + (SEC) * 448;                                                                        \
        const lds_cv_f4v *tp = (const lds_cv_f4v *)(s_pd + (SEC) * 16);                                              \
        TX.l0 = *(const lds_cv_f4v *)(ts + 4 * lane);                                                                \
        TX.l1 = *(const lds_cv_f4v *)(ts + 256 + 4 * (lane & 15));                                                   \
        TX.l2 = *(const lds_cv_f4v *)(ts + 320 + 4 * (lane & 31));                                                   \
        TX.p0 = tp[0]; TX.p1 = tp[1]; TX.p2 = tp[2]; TX.p3 = tp[3];                                                  \
        asm volatile("s_load_dwordx8 %0, %1, %2" : "=&s"(CX) : "s"(ph32), "n"((SEC) * 160 + 128) : "memory");         \
        LLZ_PK_TIE;                                                                                                  \
    }
#define LLZ_PK_FETCH_H(SEC)                                                                                          \
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4"                                          \
                 : "=&s"(h0), "=&s"(h1) : "s"(ph32), "n"((SEC) * 160), "n"((SEC) * 160 + 64) : "memory")
#define LLZ_PK_WAIT(CX)                                                                                              \
    {                                                                                                                \
        LLZ_PK_TIE;                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(CX));                                                             \
    }
    f2 U[HP];                                                          // U[j] = (u[j], u[j + 16])
    U[0] = U[HP - 1] = splat(0.f);
    LLZ_PK_FETCH(0, cA, TA)
    const f2 G = splat(in_gain);
    for (int chunk = 0; chunk < nchunks; chunk++) {
#pragma unroll
        for (int q = 0; q < 8; q++) *reinterpret_cast<float4 *>(t_lin + 288 * q) = pre[q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 8; q++) pre[q] = *reinterpret_cast<const float4 *>(t_own + 4 * q);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; q++) {
            U[4 * q + 0] = G * f2{pre[q].x, pre[q + 4].x}; U[4 * q + 1] = G * f2{pre[q].y, pre[q + 4].y};
            U[4 * q + 2] = G * f2{pre[q].z, pre[q + 4].z}; U[4 * q + 3] = G * f2{pre[q].w, pre[q + 4].w};
        }
        if (chunk + 1 < nchunks) {
#pragma unroll
            for (int q = 0; q < 8; q++)
                { const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(row + (size_t)(chunk + 1) * CHUNK + 256 * q)); pre[q] = make_float4(t.x, t.y, t.z, t.w); }
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            // this section's constants: waited for here, the next section's requested, then the section's own h values
            f8v &cc = (s & 1) ? cB : cA;
            pk_tabs &T = (s & 1) ? TB : TA;
            if ((s & 1) == 0) {
                LLZ_PK_WAIT(cA);
                if (s + 1 < S) LLZ_PK_FETCH(s + 1, cB, TB)
            } else {
                LLZ_PK_WAIT(cB);
                LLZ_PK_FETCH((s + 1 < S ? s + 1 : 0), cA, TA)
            }
            LLZ_PK_FETCH_H(s);
            const float b1 = cc[1], b2 = cc[2], a1 = cc[3], a2 = cc[4];
            float um1 = dpp_<DPP_WAVE_SHR1, 0xF>(U[HP - 1].y), um2 = dpp_<DPP_WAVE_SHR1, 0xF>(U[HP - 2].y);
            if (lane == 0) { um1 = su1[s]; um2 = su2[s]; }
            su1[s] = lane63_(U[HP - 1].y); su2[s] = lane63_(U[HP - 2].y);
            {   // feed-forward part in place, from the top so that U[j-1], U[j-2] are still the inputs (b0 folded out)
                const f2 Um1 = f2{um1, U[HP - 1].x}, Um2 = f2{um2, U[HP - 2].x};
                const f2 B1 = splat(b1), B2 = splat(b2);
#pragma unroll
                for (int j = HP - 1; j >= 2; j--) U[j] = pk_fma(B2, U[j - 2], pk_fma(B1, U[j - 1], U[j]));
                U[1] = pk_fma(B2, Um1, pk_fma(B1, U[0], U[1]));
                U[0] = pk_fma(B2, Um2, pk_fma(B1, Um1, U[0]));
            }
            {   // both half runs from the zero state: U[j] <- (w[j], w[j + 16])
                const f2 A1 = splat(-a1), A2 = splat(-a2);
                f2 Z1 = U[0], Z2;
                U[1] = pk_fma(A1, U[0], U[1]);
                Z2 = Z1; Z1 = U[1];
#pragma unroll
                for (int j = 2; j < HP; j++) {
                    const f2 Y = pk_fma(A1, Z1, pk_fma(A2, Z2, U[j]));
                    U[j] = Y;
                    Z2 = Z1; Z1 = Y;
                }
            }
            LLZ_PK_TIE;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(h0), "+s"(h1));     // the section's h values (requested ~64 instructions ago)
            // the lane's zero-state end state (yz[31], yz[30]): the second half run started from (w[15], w[14])
            f2 ZZ;
            ZZ.x = fma_(h1[14], U[HP - 1].x, fma_(h1[15], U[HP - 2].x, U[HP - 1].y));
            ZZ.y = fma_(h1[12], U[HP - 1].x, fma_(h1[13], U[HP - 2].x, U[HP - 2].y));
#define LLZ_SCAN_STEP(CTRL, MASK, M)                                                                                 \
            {                                                                                                        \
                const f2 Q = dpp2_<CTRL, MASK>(ZZ);                                                                  \
                ZZ = pk_fma(f2{M.x, M.y}, splat(Q.x), pk_fma(f2{M.z, M.w}, splat(Q.y), ZZ));                         \
            }
            LLZ_SCAN_STEP(DPP_ROW_SHR + 1, 0xF, T.p0) LLZ_SCAN_STEP(DPP_ROW_SHR + 2, 0xF, T.p1)
            LLZ_SCAN_STEP(DPP_ROW_SHR + 4, 0xF, T.p2) LLZ_SCAN_STEP(DPP_ROW_SHR + 8, 0xF, T.p3)
            LLZ_SCAN_STEP(DPP_BCAST15, 0xA, T.l1) LLZ_SCAN_STEP(DPP_BCAST31, 0xC, T.l2)
#undef LLZ_SCAN_STEP
            // the lane's start state (y[-1], y[-2]), then the true (y[15], y[14]): the start state of the second half run
            const f2 Y0 = pk_fma(f2{T.l0.x, T.l0.y}, splat(sy1[s]), pk_fma(f2{T.l0.z, T.l0.w}, splat(sy2[s]), dpp2_<DPP_WAVE_SHR1, 0xF>(ZZ)));
            const float ym1 = fma_(h1[14], Y0.x, fma_(h1[15], Y0.y, U[HP - 1].x));
            const float ym2 = fma_(h1[12], Y0.x, fma_(h1[13], Y0.y, U[HP - 2].x));
            const f2 SA = f2{Y0.x, ym1}, SB = f2{Y0.y, ym2};
#pragma unroll
            for (int j = 0; j < 8; j++) U[j] = pk_fma(splat(h0[2 * j]), SA, pk_fma(splat(h0[2 * j + 1]), SB, U[j]));
#pragma unroll
            for (int j = 0; j < 8; j++) U[8 + j] = pk_fma(splat(h1[2 * j]), SA, pk_fma(splat(h1[2 * j + 1]), SB, U[8 + j]));
            sy1[s] = lane63_(U[HP - 1].y); sy2[s] = lane63_(U[HP - 2].y);
            if ((s & 1) == 0 && s + 1 == S) LLZ_PK_FETCH(0, cA, TA)   // odd S: set A is free only now (once per chunk)
        }
        if (chunk >= skip) {
            float *dst = orow + (size_t)chunk * CHUNK;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                *reinterpret_cast<float4 *>(t_own + 4 * q) = make_float4(U[4 * q].x, U[4 * q + 1].x, U[4 * q + 2].x, U[4 * q + 3].x);
                *reinterpret_cast<float4 *>(t_own + 16 + 4 * q) = make_float4(U[4 * q].y, U[4 * q + 1].y, U[4 * q + 2].y, U[4 * q + 3].y);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f4v v = *reinterpret_cast<const f4v *>(t_lin + 288 * q);
                __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(dst + 256 * q));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // read out before the next chunk is written in
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // nothing in flight when the wave ends
#undef LLZ_PK_FETCH
#undef LLZ_PK_FETCH_H
#undef LLZ_PK_TIE
#undef LLZ_PK_WAIT
    if (lane == 0 && seg == segs - 1) {
#pragma unroll
        for (int s = 0; s < S; s++) {
            double *st = state + ((size_t)c * S + s) * 4;
            const float xf = ph32[s * 40 + 37], yf = ph32[s * 40 + 38];
            st[0] = (double)su1[s] / (double)xf; st[1] = (double)su2[s] / (double)xf;
            st[2] = (double)sy1[s] / (double)yf; st[3] = (double)sy2[s] / (double)yf;
        }
    }
}
// ---------------------------------------------------------------------------------------------------------------
// k_iir_cascade_wave_pf64: the double wave-autonomous cascade with the same one-section-ahead fetch.  k_iir_cascade_wave
// <double, 8> needs 348 VGPRs (the compiler hoists every section's powers out of the chunk loop) and so runs one wave per
// SIMD with its scalar loads exposed; here a section's powers live in one of two register sets, filled from LDS a section
// ahead, and the five coefficients come by written-out scalar loads: two waves per SIMD.
typedef double d2v __attribute__((ext_vector_type(2)));
typedef double d4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const volatile d2v lds_cv_d2v;
struct pf_tabs { d2v l[6], p[8]; };     // per lane: P^lane, P^(lane%16+1), P^(lane%32+1) (row major); uniform: P^(2^d), d < 4

__device__ __forceinline__ void pf_section(double (&u)[16], const d4v c4, const double a2, const pf_tabs &T, const int lane,
                                           double &su1, double &su2, double &sy1, double &sy2)
{
    const double b0 = c4[0], b1 = c4[1], b2 = c4[2], a1 = c4[3];
    double um1 = dpp_<DPP_WAVE_SHR1, 0xF>(u[15]), um2 = dpp_<DPP_WAVE_SHR1, 0xF>(u[14]);
    if (lane == 0) { um1 = su1; um2 = su2; }
    su1 = lane63_(u[15]); su2 = lane63_(u[14]);
    {
        double p1 = um1, p2 = um2;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const double x = u[k];
            u[k] = fma_(b2, p2, fma_(b1, p1, b0 * x));
            p2 = p1; p1 = x;
        }
    }
    double z1 = 0.0, z2 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const double y = fma_(-a1, z1, fma_(-a2, z2, u[k]));
        z2 = z1; z1 = y;
    }
#define LLZ_SCAN_STEP(CTRL, MASK, M0, M1)                                                                            \
    {                                                                                                                \
        const double q1 = dpp_<CTRL, MASK>(z1), q2 = dpp_<CTRL, MASK>(z2);                                           \
        z1 = fma_(M0.x, q1, fma_(M0.y, q2, z1));                                                                     \
        z2 = fma_(M1.x, q1, fma_(M1.y, q2, z2));                                                                     \
    }
    LLZ_SCAN_STEP(DPP_ROW_SHR + 1, 0xF, T.p[0], T.p[1]) LLZ_SCAN_STEP(DPP_ROW_SHR + 2, 0xF, T.p[2], T.p[3])
    LLZ_SCAN_STEP(DPP_ROW_SHR + 4, 0xF, T.p[4], T.p[5]) LLZ_SCAN_STEP(DPP_ROW_SHR + 8, 0xF, T.p[6], T.p[7])
    LLZ_SCAN_STEP(DPP_BCAST15, 0xA, T.l[2], T.l[3]) LLZ_SCAN_STEP(DPP_BCAST31, 0xC, T.l[4], T.l[5])
#undef LLZ_SCAN_STEP
    const double e1 = dpp_<DPP_WAVE_SHR1, 0xF>(z1), e2 = dpp_<DPP_WAVE_SHR1, 0xF>(z2);
    double y1 = fma_(T.l[0].x, sy1, fma_(T.l[0].y, sy2, e1));
    double y2 = fma_(T.l[1].x, sy1, fma_(T.l[1].y, sy2, e2));
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const double y = fma_(-a1, y1, fma_(-a2, y2, u[k]));
        u[k] = y;
        y2 = y1; y1 = y;
    }
    sy1 = lane63_(y1); sy2 = lane63_(y2);
}

template <int S>
__global__ void __launch_bounds__(256)
k_iir_cascade_wave_pf64(const float *__restrict__ in, float *__restrict__ out,
                        const double *__restrict__ coef /* [S][5] */, const double *__restrict__ pd /* [S][24] */,
                        const double *__restrict__ pl /* [S][64][12] */, const double *__restrict__ state_in,
                        double *__restrict__ state, int nchunks_total, long in_pitch, long out_pitch, int segs,
                        int seg_chunks, int warm, long items)
{
    __shared__ __attribute__((aligned(16))) double s_pl[S * 64 * 12];
    __shared__ __attribute__((aligned(16))) double s_pd[S * 16];
    // chunks move between HBM and registers in linear order and are turned lane-major through a wave-private LDS buffer
    // (16 floats of data per 20 of pitch: conflict-free both ways), as in k_iir_cascade_wave_pk32
    extern __shared__ __attribute__((aligned(16))) float s_turn_dyn[];
    for (int e = threadIdx.x; e < S * 768; e += 256) s_pl[e] = pl[e];
    if (threadIdx.x < S * 16) s_pd[threadIdx.x] = pd[(threadIdx.x >> 4) * 24 + (threadIdx.x & 15)];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (item >= items) return;
    const int c = (int)(item / segs), seg = (int)(item - (long)c * segs);
    const int skip = seg > 0 ? warm : 0;
    const int chunk0 = seg * seg_chunks - skip;
    const int nchunks = min(nchunks_total, (seg + 1) * seg_chunks) - chunk0;

    double su1[S], su2[S], sy1[S], sy2[S];
#pragma unroll
    for (int s = 0; s < S; s++) {
        su1[s] = su2[s] = sy1[s] = sy2[s] = 0;
        if (seg == 0) {
            const double *st = state_in + ((size_t)c * S + s) * 4;
            su1[s] = st[0]; su2[s] = st[1]; sy1[s] = st[2]; sy2[s] = st[3];
        }
    }
    const float *row = in + (size_t)c * in_pitch + (size_t)chunk0 * 1024 + 4 * lane;
    float *orow = out + (size_t)c * out_pitch + (size_t)chunk0 * 1024 + 4 * lane;
    float *turn = s_turn_dyn + (threadIdx.x >> 6) * 1280;
    float *t_lin = turn + 4 * lane + 4 * (lane >> 2);       // + 320 q: float 256 q + 4 lane at pitch 20 per 16
    float *t_own = turn + 20 * lane;                        // + 4 j: the lane's own run
    float4 pre[4];
    if (nchunks > 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) pre[q] = *reinterpret_cast<const float4 *>(row + 256 * q);
    }
    d4v cA, cB; double aA, aB;
    pf_tabs TA, TB;
    double u[16];
    u[0] = u[15] = 0.0;
#define LLZ_PF_TIE asm volatile("" : "+v"(u[0]), "+v"(u[15]))
#define LLZ_PF_FETCH(SEC, CX, AX, TX)                                                                                \
    {                                                                                                                \
        const lds_cv_d2v *tl = (const lds_cv_d2v *)(s_pl + ((SEC) * 64 + lane) * 12);                                \
        const lds_cv_d2v *tp = (const lds_cv_d2v *)(s_pd + (SEC) * 16);                                              \
        _Pragma("unroll") for (int i = 0; i < 6; i++) TX.l[i] = tl[i];                                               \
        _Pragma("unroll") for (int i = 0; i < 8; i++) TX.p[i] = tp[i];                                               \
        asm volatile("s_load_dwordx8 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4"                                         \
                     : "=&s"(CX), "=&s"(AX) : "s"(coef), "n"((SEC) * 40), "n"((SEC) * 40 + 32) : "memory");           \
        LLZ_PF_TIE;                                                                                                  \
    }
#define LLZ_PF_WAIT(CX, AX)                                                                                          \
    {                                                                                                                \
        LLZ_PF_TIE;                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(CX), "+s"(AX));                                                   \
    }
    LLZ_PF_FETCH(0, cA, aA, TA)
    for (int chunk = 0; chunk < nchunks; chunk++) {
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<float4 *>(t_lin + 320 * q) = pre[q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; q++) pre[q] = *reinterpret_cast<const float4 *>(t_own + 4 * q);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; q++) { u[4 * q] = pre[q].x; u[4 * q + 1] = pre[q].y; u[4 * q + 2] = pre[q].z; u[4 * q + 3] = pre[q].w; }
        if (chunk + 1 < nchunks) {
#pragma unroll
            for (int q = 0; q < 4; q++) pre[q] = *reinterpret_cast<const float4 *>(row + (size_t)(chunk + 1) * 1024 + 256 * q);
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            if ((s & 1) == 0) {
                LLZ_PF_WAIT(cA, aA);
                if (s + 1 < S) LLZ_PF_FETCH(s + 1, cB, aB, TB)
                pf_section(u, cA, aA, TA, lane, su1[s], su2[s], sy1[s], sy2[s]);
                if (s + 1 == S) LLZ_PF_FETCH(0, cA, aA, TA)            // odd S: set A is free only now (once per chunk)
            } else {
                LLZ_PF_WAIT(cB, aB);
                LLZ_PF_FETCH((s + 1 < S ? s + 1 : 0), cA, aA, TA)
                pf_section(u, cB, aB, TB, lane, su1[s], su2[s], sy1[s], sy2[s]);
            }
        }
        if (chunk >= skip) {
            float *dst = orow + (size_t)chunk * 1024;
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<float4 *>(t_own + 4 * q) = make_float4((float)u[4 * q], (float)u[4 * q + 1], (float)u[4 * q + 2], (float)u[4 * q + 3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f4v v = *reinterpret_cast<const f4v *>(t_lin + 320 * q);
                __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(dst + 256 * q));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // read out before the next chunk is written in
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // nothing in flight when the wave ends
#undef LLZ_PF_FETCH
#undef LLZ_PF_TIE
#undef LLZ_PF_WAIT
    if (lane == 0 && seg == segs - 1) {
#pragma unroll
        for (int s = 0; s < S; s++) {
            double *st = state + ((size_t)c * S + s) * 4;
            st[0] = su1[s]; st[1] = su2[s]; st[2] = sy1[s]; st[3] = sy2[s];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// k_iir_cascade_wave_pf64w: the double cascade with 32 samples per lane and the b0 gains folded out -- what
// k_iir_cascade_wave_pk32 does for float32, without the packing (v_fma_f64 already is the two-slot instruction).  Per sample
// and section: 2 (feed-forward, b' = b / b0) + 2 (zero-state recurrence, end state only) + 2 (recurrence from the true start
// state) FMAs and 28 / 32 of the lane scan, against 3 + 2 + 2 and 28 / 16 in k_iir_cascade_wave_pf64: 6.9 instead of 8.75.
// Register budget for two waves per SIMD: the run is 64 registers, so only the four coefficients are fetched a section
// ahead (two SGPR sets); the scan's uniform powers (SGPRs) and the lane's powers (LDS, one compact copy per section as in
// pk32) are requested when the section starts and first used ~130 FMAs later.
// Tables: cw [S][8] = b1', b2', a1, a2, xfac, yfac, 0, 0;  pd [S][16] = P^(2^d), d < 4, P = A^32, row major;
// plc [S][448] = P^lane (64 x 4), P^(i+1) i < 16, P^(i+1) i < 32.
typedef double d8v __attribute__((ext_vector_type(8)));
template <int S>
__global__ void __launch_bounds__(256, 2)
k_iir_cascade_wave_pf64w(const float *__restrict__ in, float *__restrict__ out, const double *__restrict__ cw,
                         const double *__restrict__ pd, const double *__restrict__ plc,
                         const double *__restrict__ state_in, double *__restrict__ state, int nchunks_total, long in_pitch,
                         long out_pitch, int segs, int seg_chunks, int warm, long items, double in_gain)
{
    constexpr int RUN = 32, CHUNK = 64 * RUN;
    __shared__ __attribute__((aligned(16))) double s_pl[S * 448];
    __shared__ __attribute__((aligned(16))) float s_turn[4][CHUNK + CHUNK / 8];
    for (int e = threadIdx.x; e < S * 448; e += 256) s_pl[e] = plc[e];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    if (item >= items) return;
    const int c = (int)(item / segs), seg = (int)(item - (long)c * segs);
    const int skip = seg > 0 ? warm : 0;
    const int chunk0 = seg * seg_chunks - skip;
    const int nchunks = min(nchunks_total, (seg + 1) * seg_chunks) - chunk0;

    double su1[S], su2[S], sy1[S], sy2[S];
#pragma unroll
    for (int s = 0; s < S; s++) {
        su1[s] = su2[s] = sy1[s] = sy2[s] = 0;
        if (seg == 0) {
            const double *st = state_in + ((size_t)c * S + s) * 4;
            const double xf = cw[s * 8 + 4], yf = cw[s * 8 + 5];
            su1[s] = st[0] * xf; su2[s] = st[1] * xf; sy1[s] = st[2] * yf; sy2[s] = st[3] * yf;
        }
    }
    const float *row = in + (size_t)c * in_pitch + (size_t)chunk0 * CHUNK + 4 * lane;
    float *orow = out + (size_t)c * out_pitch + (size_t)chunk0 * CHUNK + 4 * lane;
    float *turn = s_turn[threadIdx.x >> 6];
    float *t_lin = turn + 4 * lane + 4 * (lane >> 3);       // + 288 q: float 256 q + 4 lane at pitch 36 per 32
    float *t_own = turn + 36 * lane;                        // + 4 j: the lane's own run
    float4 pre[8];
    if (nchunks > 0) {
#pragma unroll
        for (int q = 0; q < 8; q++) { const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(row + 256 * q)); pre[q] = make_float4(t.x, t.y, t.z, t.w); }
    }
    d4v cA, cB;                     // b1', b2', a1, a2 of the current / next section
    d8v pw0, pw1;                   // the current section's P^1, P^2 and P^4, P^8
    d2v tl[6];                      // rows of P^lane, P^(lane%16+1), P^(lane%32+1)
    double u[RUN];
    u[0] = u[RUN - 1] = 0.0;
#define LLZ_PFW_TIE asm volatile("" : "+v"(u[0]), "+v"(u[RUN - 1]))
#define LLZ_PFW_FETCH_C(SEC, CX)                                                                                     \
    {                                                                                                                \
        asm volatile("s_load_dwordx8 %0, %1, %2" : "=&s"(CX) : "s"(cw), "n"((SEC) * 64) : "memory");                  \
        LLZ_PFW_TIE;                                                                                                 \
    }
#define LLZ_PFW_FETCH_P(SEC)                                                                                         \
    {                                                                                                                \
        const double *ts = s_pl + (SEC) * 448;                                                                       \
        tl[0] = *(lds_cv_d2v *)(ts + 4 * lane); tl[1] = *(lds_cv_d2v *)(ts + 4 * lane + 2);                          \
        tl[2] = *(lds_cv_d2v *)(ts + 256 + 4 * (lane & 15)); tl[3] = *(lds_cv_d2v *)(ts + 256 + 4 * (lane & 15) + 2); \
        tl[4] = *(lds_cv_d2v *)(ts + 320 + 4 * (lane & 31)); tl[5] = *(lds_cv_d2v *)(ts + 320 + 4 * (lane & 31) + 2); \
        asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4"                                       \
                     : "=&s"(pw0), "=&s"(pw1) : "s"(pd), "n"((SEC) * 128), "n"((SEC) * 128 + 64) : "memory");         \
    }
#define LLZ_PFW_WAIT(CX)                                                                                             \
    {                                                                                                                \
        LLZ_PFW_TIE;                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(CX));                                                             \
    }
    LLZ_PFW_FETCH_C(0, cA)
    for (int chunk = 0; chunk < nchunks; chunk++) {
#pragma unroll
        for (int q = 0; q < 8; q++) *reinterpret_cast<float4 *>(t_lin + 288 * q) = pre[q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 8; q++) pre[q] = *reinterpret_cast<const float4 *>(t_own + 4 * q);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 8; q++) {
            u[4 * q] = in_gain * (double)pre[q].x; u[4 * q + 1] = in_gain * (double)pre[q].y;
            u[4 * q + 2] = in_gain * (double)pre[q].z; u[4 * q + 3] = in_gain * (double)pre[q].w;
        }
        if (chunk + 1 < nchunks) {
#pragma unroll
            for (int q = 0; q < 8; q++)
                { const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(row + (size_t)(chunk + 1) * CHUNK + 256 * q)); pre[q] = make_float4(t.x, t.y, t.z, t.w); }
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            d4v &cc = (s & 1) ? cB : cA;
            if ((s & 1) == 0) {
                LLZ_PFW_WAIT(cA);
                if (s + 1 < S) LLZ_PFW_FETCH_C(s + 1, cB)
            } else {
                LLZ_PFW_WAIT(cB);
                LLZ_PFW_FETCH_C((s + 1 < S ? s + 1 : 0), cA)
            }
            LLZ_PFW_FETCH_P(s)
            const double b1 = cc[0], b2 = cc[1], a1 = cc[2], a2 = cc[3];
            double um1 = dpp_<DPP_WAVE_SHR1, 0xF>(u[RUN - 1]), um2 = dpp_<DPP_WAVE_SHR1, 0xF>(u[RUN - 2]);
            if (lane == 0) { um1 = su1[s]; um2 = su2[s]; }
            su1[s] = lane63_(u[RUN - 1]); su2[s] = lane63_(u[RUN - 2]);
            double z1 = 0.0, z2 = 0.0;
            {   // feed-forward part in place and, behind it, the run from the zero state (only its end state is kept)
                double p1 = um1, p2 = um2;
#pragma unroll
                for (int k = 0; k < RUN; k++) {
                    const double x = u[k];
                    const double w = fma_(b2, p2, fma_(b1, p1, x));
                    u[k] = w;
                    p2 = p1; p1 = x;
                    const double y = fma_(-a1, z1, fma_(-a2, z2, w));
                    z2 = z1; z1 = y;
                }
            }
            LLZ_PFW_TIE;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pw0), "+s"(pw1));     // the section's powers (requested ~130 FMAs ago)
#define LLZ_SCAN_STEP(CTRL, MASK, M00, M01, M10, M11)                                                                \
            {                                                                                                        \
                const double q1 = dpp_<CTRL, MASK>(z1), q2 = dpp_<CTRL, MASK>(z2);                                   \
                z1 = fma_(M00, q1, fma_(M01, q2, z1));                                                               \
                z2 = fma_(M10, q1, fma_(M11, q2, z2));                                                               \
            }
            LLZ_SCAN_STEP(DPP_ROW_SHR + 1, 0xF, pw0[0], pw0[1], pw0[2], pw0[3])
            LLZ_SCAN_STEP(DPP_ROW_SHR + 2, 0xF, pw0[4], pw0[5], pw0[6], pw0[7])
            LLZ_SCAN_STEP(DPP_ROW_SHR + 4, 0xF, pw1[0], pw1[1], pw1[2], pw1[3])
            LLZ_SCAN_STEP(DPP_ROW_SHR + 8, 0xF, pw1[4], pw1[5], pw1[6], pw1[7])
            LLZ_SCAN_STEP(DPP_BCAST15, 0xA, tl[2].x, tl[2].y, tl[3].x, tl[3].y)
            LLZ_SCAN_STEP(DPP_BCAST31, 0xC, tl[4].x, tl[4].y, tl[5].x, tl[5].y)
#undef LLZ_SCAN_STEP
            const double e1 = dpp_<DPP_WAVE_SHR1, 0xF>(z1), e2 = dpp_<DPP_WAVE_SHR1, 0xF>(z2);
            double y1 = fma_(tl[0].x, sy1[s], fma_(tl[0].y, sy2[s], e1));
            double y2 = fma_(tl[1].x, sy1[s], fma_(tl[1].y, sy2[s], e2));
#pragma unroll
            for (int k = 0; k < RUN; k++) {
                const double y = fma_(-a1, y1, fma_(-a2, y2, u[k]));
                u[k] = y;
                y2 = y1; y1 = y;
            }
            sy1[s] = lane63_(y1); sy2[s] = lane63_(y2);
            if ((s & 1) == 0 && s + 1 == S) LLZ_PFW_FETCH_C(0, cA)   // odd S: set A is free only now (once per chunk)
        }
        if (chunk >= skip) {
            float *dst = orow + (size_t)chunk * CHUNK;
#pragma unroll
            for (int q = 0; q < 8; q++)
                *reinterpret_cast<float4 *>(t_own + 4 * q) = make_float4((float)u[4 * q], (float)u[4 * q + 1], (float)u[4 * q + 2], (float)u[4 * q + 3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const f4v v = *reinterpret_cast<const f4v *>(t_lin + 288 * q);
                __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(dst + 256 * q));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // read out before the next chunk is written in
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // nothing in flight when the wave ends
#undef LLZ_PFW_FETCH_C
#undef LLZ_PFW_FETCH_P
#undef LLZ_PFW_TIE
#undef LLZ_PFW_WAIT
    if (lane == 0 && seg == segs - 1) {
#pragma unroll
        for (int s = 0; s < S; s++) {
            double *st = state + ((size_t)c * S + s) * 4;
            const double xf = cw[s * 8 + 4], yf = cw[s * 8 + 5];
            st[0] = su1[s] / xf; st[1] = su2[s] / xf; st[2] = sy1[s] / yf; st[3] = sy2[s] / yf;
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

// pd: [stages][6][4] = P^(2^d) row major, P = A^16; pl: [stages][64][12] = P^lane, P^(lane%16+1), P^(lane%32+1).  n must be a multiple of 1024 and
// the rows 16-byte aligned (pitches % 4 == 0); the caller runs the remainder through llzs_iir_cascade_f32.
extern "C" int llzs_iir_cascade_pipe_f32(const float *in, float *out, const double *coef, const double *pd,
                                         const double *pl, const double *state_in, double *state, int channels, int n,
                                         long in_pitch, long out_pitch, int stages, int warm_chunks, int float32,
                                         void *stream)
{
    // 16 samples per lane in both precisions: 32 in float32 measured slower (4.27 vs 3.62 ms: 128 VGPRs with spills under
    // the 1024-thread bound, and twice as long dependent recurrences per lane)
    const int RR = 16, chunk = 64 * RR;
    if (!in || !out || !coef || !pd || !pl || !state || !state_in || state == state_in || channels <= 0 || n <= 0 || (n % chunk) ||
        stages < 1 || stages > 16 || in_pitch < n || out_pitch < n || (in_pitch & 3) || (out_pitch & 3) ||
        (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) {
        llzs_set_error("iir_cascade_pipe_f32: bad arguments (n=%d must be a multiple of %d, rows 16-byte aligned)", n,
                       chunk);
        return LLZ_ERR_ARG;
    }
    const size_t lds = (size_t)(stages > 1 ? stages - 1 : 1) * chunk * (float32 ? sizeof(float) : sizeof(double));
    if (lds > 64 * 1024) {
        if (float32)
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_iir_cascade_pipe<float, 16>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        else
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_iir_cascade_pipe<double, 16>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // time segments: only when the channels alone leave most of the chip idle (fewer than two workgroups per CU) and the
    // warm-up stays a small part of a segment.  warm_chunks counts 1024-sample chunks.
    const int nchunks = n / chunk;
    const int warm = (warm_chunks * 1024 + chunk - 1) / chunk;
    int segs = 1;
    if (warm_chunks > 0 && channels < 512) {
        segs = (512 + channels - 1) / channels;
        if (segs > 16) segs = 16;
        while (segs > 1 && nchunks / segs < 8 * warm) segs--;
    }
    if (const int v = llzs_tune(LLZS_TUNE_IIR_SEGS); v >= 1 && v <= 64 && (v == 1 || warm_chunks > 0)) segs = v;
    while (segs > 1 && nchunks / segs < warm_chunks) segs--;        // (a forced count too must leave room for the warm-up)
    const int seg_chunks = (nchunks + segs - 1) / segs;
    segs = (nchunks + seg_chunks - 1) / seg_chunks;
    if (float32)
        hipLaunchKernelGGL((k_iir_cascade_pipe<float, 16>), dim3((unsigned)((long)channels * segs)), dim3(64 * stages), lds,
                           as_stream(stream), in, out, coef, pd, pl, state_in, state, nchunks, in_pitch, out_pitch, stages, segs,
                           seg_chunks, warm);
    else
        hipLaunchKernelGGL((k_iir_cascade_pipe<double, 16>), dim3((unsigned)((long)channels * segs)), dim3(64 * stages), lds,
                           as_stream(stream), in, out, coef, pd, pl, state_in, state, nchunks, in_pitch, out_pitch, stages, segs,
                           seg_chunks, warm);
    LLZ_LAUNCH_CHECK("k_iir_cascade_pipe");
    return LLZ_OK;
}

// wave-autonomous form (see k_iir_cascade_wave).  n a multiple of 1024, rows 16-byte aligned, warm_chunks > 0.
template <typename R>
static int launch_iir_wave(const float *in, float *out, const R *coef, const R *pd, const R *pl, const float *ph32,
                           const double *state_in, double *state, int channels, int n, long in_pitch, long out_pitch,
                           int stages, int warm_chunks, int pd_stride, void *stream)
{
    if (!in || !out || !coef || !pd || !pl || !state || !state_in || state == state_in || channels <= 0 || n <= 0 || (n % 1024) || stages < 1 ||
        stages > 8 /* 16 sections in registers spill */ || warm_chunks < 1 || in_pitch < n || out_pitch < n || (in_pitch & 3) || (out_pitch & 3) ||
        (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) {
        llzs_set_error("iir_cascade_wave: bad arguments");
        return LLZ_ERR_ARG;
    }
    const int nchunks = n / 1024;
    // Time segments per channel.  `slots` = waves the chip holds of this kernel.  Measured on config 4 (1024 channels) and
    // on 128 channels: about three rounds of items are best while a segment stays long (>= 64 chunks: the per-item cost
    // of warm-up and table load, ~1.4 chunks, stays small and the hardware balances the rounds), otherwise exactly one
    // round; segments at least 8 x the warm-up.
    // the one-section-ahead kernels: packed float32 (needs the h table) or double
    if (std::is_same<R, float>::value && ph32 == nullptr) {
        llzs_set_error("iir_cascade_wave: the float32 form needs its h table");
        return LLZ_ERR_ARG;
    }
    const void *kfn = nullptr;
    if constexpr (std::is_same<R, float>::value) {
        static const void *const tab[8] = {
            (const void *)k_iir_cascade_wave_pk<1>, (const void *)k_iir_cascade_wave_pk<2>, (const void *)k_iir_cascade_wave_pk<3>,
            (const void *)k_iir_cascade_wave_pk<4>, (const void *)k_iir_cascade_wave_pk<5>, (const void *)k_iir_cascade_wave_pk<6>,
            (const void *)k_iir_cascade_wave_pk<7>, (const void *)k_iir_cascade_wave_pk<8>};
        kfn = tab[stages - 1];
    } else {
        static const void *const tab[8] = {
            (const void *)k_iir_cascade_wave_pf64<1>, (const void *)k_iir_cascade_wave_pf64<2>, (const void *)k_iir_cascade_wave_pf64<3>,
            (const void *)k_iir_cascade_wave_pf64<4>, (const void *)k_iir_cascade_wave_pf64<5>, (const void *)k_iir_cascade_wave_pf64<6>,
            (const void *)k_iir_cascade_wave_pf64<7>, (const void *)k_iir_cascade_wave_pf64<8>};
        kfn = tab[stages - 1];
    }
    // (queried once per kernel and process: same answer on every device of a node)
    static struct { const void *fn; long slots; } seen[24];
    static int nseen = 0;
    static std::mutex seen_lock;
    long slots = 0;
    std::lock_guard<std::mutex> guard(seen_lock);
    for (int i = 0; i < nseen; i++)
        if (seen[i].fn == kfn) slots = seen[i].slots;
    if (!slots) {
        int blocks_per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kfn, 256, 0) != hipSuccess || blocks_per_cu < 1) {
            (void)hipGetLastError();
            blocks_per_cu = 2;
        }
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
            (void)hipGetLastError();
            cus = 256;
        }
        slots = 4L * blocks_per_cu * cus;
        if (nseen < 24) { seen[nseen].slots = slots; seen[nseen].fn = kfn; nseen++; }
    }
    int segs = 1;
    for (int rounds = 3; rounds >= 1; rounds--) {
        segs = (int)((rounds * slots + channels / 2) / channels);
        if (segs < 1) segs = 1;
        if (segs > 64) segs = 64;
        if (rounds == 1 || nchunks / segs >= 64) break;
    }
    while (segs > 1 && nchunks / segs < 8 * warm_chunks) segs--;
    if (const int v = llzs_tune(LLZS_TUNE_IIR_SEGS); v >= 1 && v <= 64) segs = v;
    while (segs > 1 && nchunks / segs < warm_chunks) segs--;        // (a forced count too: a segment must hold its own warm-up,
                                                                    //  or its first chunk would lie in front of the row)
    const int seg_chunks = (nchunks + segs - 1) / segs;
    segs = (nchunks + seg_chunks - 1) / seg_chunks;
    const long items = (long)channels * segs;
    const dim3 grid((unsigned)((items + 3) / 4));
    {
        if constexpr (std::is_same<R, float>::value) {
#define LLZ_AHEAD_LAUNCH(S)                                                                                          \
    hipLaunchKernelGGL((k_iir_cascade_wave_pk<S>), grid, dim3(256), 0, as_stream(stream), in, out, pd, pl, ph32, state_in, state, \
                       nchunks, in_pitch, out_pitch, segs, seg_chunks, warm_chunks, items)
            switch (stages) {
            case 1: LLZ_AHEAD_LAUNCH(1); break; case 2: LLZ_AHEAD_LAUNCH(2); break; case 3: LLZ_AHEAD_LAUNCH(3); break;
            case 4: LLZ_AHEAD_LAUNCH(4); break; case 5: LLZ_AHEAD_LAUNCH(5); break; case 6: LLZ_AHEAD_LAUNCH(6); break;
            case 7: LLZ_AHEAD_LAUNCH(7); break; default: LLZ_AHEAD_LAUNCH(8); break;
            }
#undef LLZ_AHEAD_LAUNCH
        } else {
#define LLZ_AHEAD_LAUNCH(S)                                                                                          \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_iir_cascade_wave_pf64<S>),                            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 1280 * (int)sizeof(float));            \
    hipLaunchKernelGGL((k_iir_cascade_wave_pf64<S>), grid, dim3(256), 4 * 1280 * sizeof(float), as_stream(stream), in, \
                       out, coef, pd, pl, state_in, state, nchunks, in_pitch, out_pitch, segs, seg_chunks, warm_chunks, items)
            switch (stages) {
            case 1: LLZ_AHEAD_LAUNCH(1); break; case 2: LLZ_AHEAD_LAUNCH(2); break; case 3: LLZ_AHEAD_LAUNCH(3); break;
            case 4: LLZ_AHEAD_LAUNCH(4); break; case 5: LLZ_AHEAD_LAUNCH(5); break; case 6: LLZ_AHEAD_LAUNCH(6); break;
            case 7: LLZ_AHEAD_LAUNCH(7); break; default: LLZ_AHEAD_LAUNCH(8); break;
            }
#undef LLZ_AHEAD_LAUNCH
        }
        LLZ_LAUNCH_CHECK("k_iir_cascade_wave (one section ahead)");
        return LLZ_OK;
    }
}

extern "C" int llzs_iir_cascade_wave_f32(const float *in, float *out, const float *coef32, const float *pd32,
                                         const float *pl32, const float *ph32 /* NULL: unpacked kernel */,
                                         const double *state_in, double *state, int channels, int n, long in_pitch,
                                         long out_pitch, int stages, int warm_chunks, void *stream)
{
    return launch_iir_wave<float>(in, out, coef32, pd32, pl32, ph32, state_in, state, channels, n, in_pitch, out_pitch, stages,
                                  warm_chunks, 16, stream);
}

// the same in double, from the pipelined kernel's own tables (pd: [S][6][4])
extern "C" int llzs_iir_cascade_wave_f64(const float *in, float *out, const double *coef, const double *pd,
                                         const double *pl, const double *state_in, double *state, int channels, int n,
                                         long in_pitch, long out_pitch, int stages, int warm_chunks, void *stream)
{
    return launch_iir_wave<double>(in, out, coef, pd, pl, nullptr, state_in, state, channels, n, in_pitch, out_pitch, stages, warm_chunks,
                                   24, stream);
}

// 32 samples per lane, b0 folded out (k_iir_cascade_wave_pk32): n a multiple of 2048, rows 16-byte aligned, 1..8 sections,
// warm_chunks in units of 1024 samples as everywhere; tables for P = A^32: pd32 [S][16], pl32 [S][64][12], ph32 [S][40];
// in_gain = the product of the b0's
extern "C" int llzs_iir_cascade_wave32_f32(const float *in, float *out, const float *pd32, const float *pl32,
                                           const float *ph32, const double *state_in, double *state, int channels, int n,
                                           long in_pitch, long out_pitch, int stages, int warm_chunks, float in_gain,
                                           void *stream)
{
    if (!in || !out || !pd32 || !pl32 || !ph32 || !state || !state_in || state == state_in || channels <= 0 || n <= 0 ||
        (n % 2048) || stages < 1 || stages > 8 || warm_chunks < 1 || in_pitch < n || out_pitch < n || (in_pitch & 3) ||
        (out_pitch & 3) || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) {
        llzs_set_error("iir_cascade_wave32_f32: bad arguments");
        return LLZ_ERR_ARG;
    }
    static const void *const tab[8] = {
        (const void *)k_iir_cascade_wave_pk32<1>, (const void *)k_iir_cascade_wave_pk32<2>, (const void *)k_iir_cascade_wave_pk32<3>,
        (const void *)k_iir_cascade_wave_pk32<4>, (const void *)k_iir_cascade_wave_pk32<5>, (const void *)k_iir_cascade_wave_pk32<6>,
        (const void *)k_iir_cascade_wave_pk32<7>, (const void *)k_iir_cascade_wave_pk32<8>};
    const void *kfn = tab[stages - 1];
    int blocks_per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kfn, 256, 0) != hipSuccess || blocks_per_cu < 1) {
        (void)hipGetLastError();
        blocks_per_cu = 2;
    }
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
        (void)hipGetLastError();
        cus = 256;
    }
    const long slots = 4L * blocks_per_cu * cus;
    const int nchunks = n / 2048, warm = (warm_chunks + 1) / 2;
    // time segments per channel as for the 16-sample kernels: about three rounds of (channel, segment) items while a
    // segment stays long (>= 32 chunks of 2048), else one round; segments at least 8 x the warm-up
    int segs = 1;
    for (int rounds = 3; rounds >= 1; rounds--) {
        segs = (int)((rounds * slots + channels / 2) / channels);
        if (segs < 1) segs = 1;
        if (segs > 64) segs = 64;
        if (rounds == 1 || nchunks / segs >= 32) break;
    }
    while (segs > 1 && nchunks / segs < 8 * warm) segs--;
    if (const int v = llzs_tune(LLZS_TUNE_IIR_SEGS); v >= 1 && v <= 64) segs = v;
    while (segs > 1 && nchunks / segs < warm_chunks) segs--;        // (a forced count too: a segment must hold its own warm-up,
                                                                    //  or its first chunk would lie in front of the row)
    const int seg_chunks = (nchunks + segs - 1) / segs;
    segs = (nchunks + seg_chunks - 1) / seg_chunks;
    const long items = (long)channels * segs;
    const dim3 grid((unsigned)((items + 3) / 4));
#define LLZ_PK32_LAUNCH(S)                                                                                           \
    hipLaunchKernelGGL((k_iir_cascade_wave_pk32<S>), grid, dim3(256), 0, as_stream(stream), in, out, pd32, pl32, ph32,  \
                       state_in, state, nchunks, in_pitch, out_pitch, segs, seg_chunks, warm, items, in_gain)
    switch (stages) {
    case 1: LLZ_PK32_LAUNCH(1); break; case 2: LLZ_PK32_LAUNCH(2); break; case 3: LLZ_PK32_LAUNCH(3); break;
    case 4: LLZ_PK32_LAUNCH(4); break; case 5: LLZ_PK32_LAUNCH(5); break; case 6: LLZ_PK32_LAUNCH(6); break;
    case 7: LLZ_PK32_LAUNCH(7); break; default: LLZ_PK32_LAUNCH(8); break;
    }
#undef LLZ_PK32_LAUNCH
    LLZ_LAUNCH_CHECK("k_iir_cascade_wave_pk32");
    return LLZ_OK;
}

// the double cascade with 32 samples per lane and b0 folded out (k_iir_cascade_wave_pf64w): n a multiple of 2048, rows
// 16-byte aligned, 1..8 sections; cw [S][8], pd [S][16], plc [S][448] as described at the kernel; in_gain = the product of
// the b0's
extern "C" int llzs_iir_cascade_wave32_f64(const float *in, float *out, const double *cw, const double *pd,
                                           const double *plc, const double *state_in, double *state, int channels, int n,
                                           long in_pitch, long out_pitch, int stages, int warm_chunks, double in_gain,
                                           void *stream)
{
    if (!in || !out || !cw || !pd || !plc || !state || !state_in || state == state_in || channels <= 0 || n <= 0 ||
        (n % 2048) || stages < 1 || stages > 8 || warm_chunks < 1 || in_pitch < n || out_pitch < n || (in_pitch & 3) ||
        (out_pitch & 3) || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) {
        llzs_set_error("iir_cascade_wave32_f64: bad arguments");
        return LLZ_ERR_ARG;
    }
    static const void *const tab[8] = {
        (const void *)k_iir_cascade_wave_pf64w<1>, (const void *)k_iir_cascade_wave_pf64w<2>, (const void *)k_iir_cascade_wave_pf64w<3>,
        (const void *)k_iir_cascade_wave_pf64w<4>, (const void *)k_iir_cascade_wave_pf64w<5>, (const void *)k_iir_cascade_wave_pf64w<6>,
        (const void *)k_iir_cascade_wave_pf64w<7>, (const void *)k_iir_cascade_wave_pf64w<8>};
    const void *kfn = tab[stages - 1];
    int blocks_per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kfn, 256, 0) != hipSuccess || blocks_per_cu < 1) {
        (void)hipGetLastError();
        blocks_per_cu = 2;
    }
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
        (void)hipGetLastError();
        cus = 256;
    }
    const long slots = 4L * blocks_per_cu * cus;
    const int nchunks = n / 2048, warm = (warm_chunks + 1) / 2;
    int segs = 1;
    for (int rounds = 3; rounds >= 1; rounds--) {
        segs = (int)((rounds * slots + channels / 2) / channels);
        if (segs < 1) segs = 1;
        if (segs > 64) segs = 64;
        if (rounds == 1 || nchunks / segs >= 32) break;
    }
    while (segs > 1 && nchunks / segs < 8 * warm) segs--;
    if (const int v = llzs_tune(LLZS_TUNE_IIR_SEGS); v >= 1 && v <= 64) segs = v;
    while (segs > 1 && nchunks / segs < warm_chunks) segs--;        // (a forced count too: a segment must hold its own warm-up,
                                                                    //  or its first chunk would lie in front of the row)
    const int seg_chunks = (nchunks + segs - 1) / segs;
    segs = (nchunks + seg_chunks - 1) / seg_chunks;
    const long items = (long)channels * segs;
    const dim3 grid((unsigned)((items + 3) / 4));
#define LLZ_PFW_LAUNCH(S)                                                                                            \
    hipLaunchKernelGGL((k_iir_cascade_wave_pf64w<S>), grid, dim3(256), 0, as_stream(stream), in, out, cw, pd, plc,      \
                       state_in, state, nchunks, in_pitch, out_pitch, segs, seg_chunks, warm, items, in_gain)
    switch (stages) {
    case 1: LLZ_PFW_LAUNCH(1); break; case 2: LLZ_PFW_LAUNCH(2); break; case 3: LLZ_PFW_LAUNCH(3); break;
    case 4: LLZ_PFW_LAUNCH(4); break; case 5: LLZ_PFW_LAUNCH(5); break; case 6: LLZ_PFW_LAUNCH(6); break;
    case 7: LLZ_PFW_LAUNCH(7); break; default: LLZ_PFW_LAUNCH(8); break;
    }
#undef LLZ_PFW_LAUNCH
    LLZ_LAUNCH_CHECK("k_iir_cascade_wave_pf64w");
    return LLZ_OK;
}
