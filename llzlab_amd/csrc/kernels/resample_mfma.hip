// resample_mfma.hip -- K3g: rational L/M resampling with large L and M (147:160, 160:147 ...) on the fp32 matrix cores.
//
//   reference (llz_resample.c:583-603):   y[i] = gain * sum_{k<Q} x[(i M) / L - k] * g[i mod L][k]
//
// An L:M resampler is L decimators by M interleaved at the output: with i = L m + f (period m, phase f < L)
//       y[L m + f] = sum_k x[M m + c_f - k] * g_f[k],        c_f = floor(f M / L).
// For ONE period m every phase reads inside the same window of M + Q input samples, and the taps do not depend on m:
//       Y[f][m] = sum_u W[f][u] * X[u][m],    W[f][u] = g_f[c_f - u]  (zero outside 0 <= c_f - u < Q),   X[u][m] = x[M m + u]
// -- a product of a FIXED banded matrix W (L x (M + Q)) with the signal laid out as overlapping columns.  The VALU form of
// this (k_resample_f32_lds, resample.hip) needs one LDS read per multiply-add and stops at 12-13 % of the HBM roofline;
// v_mfma_f32_16x16x4_f32 takes one LDS read per 16 multiply-adds and keeps fp32 products and sums (no split needed).
//
// Mapping.  A wave owns a tile of 16 phases x 16 periods: D[r][n] = output (phase 16 T + r, period m_n).  The band of 16
// consecutive phases is c_{16T+15} - c_{16T} + Q wide (64 samples at 147:160, Q = 47), walked 4 samples per MFMA starting
// at u0 = c_{16T} - (Q-1):   A[r][t] = gain * g_{16T+r}[c_{16T+r} - u0 - t],   B[t][n] = x[M m_n + u0 + t].
// A is built once on the host ([phase tile][step][64 lanes]) and lives in the wave's registers while it sweeps a phase
// tile; the signal span of a workgroup (64 periods + Q-1 samples of history) is staged once in LDS, one float per
// sample, with the period stride made odd (one pad float per period when M is even) so that the 16 columns of a B read
// fall in 16 different banks.  A workgroup = 4 waves = 4 period tiles; every wave sweeps all phase tiles, so the work
// is balanced for any L, and a wave's stores of one tile are 16 runs of 16 consecutive floats that the next phase tile
// extends (whole lines form in L2 before they leave for HBM).
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RM_MAX_WAVES = 4;                   // a workgroup is 1..4 waves = 16..64 periods (fewer for long periods)
constexpr int RM_CHUNK = 2;                       // phase tiles per output run: 32 consecutive outputs of a period
constexpr int RM_SROW = 16 * RM_CHUNK + 4;        // floats per row of the per-wave output buffer (+4: 16-byte writes of the
                                                  // 16 columns land in different banks)

struct rm_shape {
    int L, M, Q;
    int ntiles;          // phase tiles: ceil(L / 16)
    int pad;             // pad floats per period: column stride M + pad = 2 mod 32, so that the 32 lanes of a B read
                         // (16 columns x 2 consecutive samples) fall in 32 different banks
    int H;               // history samples in front of the span: Q - 1
    int img;             // floats of the signal image (with padding)
    unsigned m_magic;    // ceil(2^32 / M): e / M for e < 2^32 / M by multiply-high
};

template <int KS>
__global__ void __launch_bounds__(RM_MAX_WAVES * 64)
k_resample_mfma_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                    const float *__restrict__ atab /* [ntiles][KS][64] */, const int *__restrict__ c0tab /* [ntiles] */,
                    long n_in, long n_out, long in_pitch, long out_pitch, rm_shape sh)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = blockDim.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int c = blockIdx.y;
    const int periods = 16 * waves;
    const long m0 = (long)blockIdx.x * periods;                     // first period of this workgroup
    const long first = m0 * sh.M - sh.H;                            // input index of image element 0
    // (4 KS more than the span: the zero-padded tail of a band reads there; the values only have to be finite)
    const int count = periods * sh.M + sh.H + 4 * KS;
    const float *row = in + (size_t)c * in_pitch;
    auto pos_of = [&](int e) { return e + sh.pad * (int)__umulhi((unsigned)e, sh.m_magic); };
    // 16 loads in flight per lane (a one-load-per-iteration loop waits out the full memory latency 40 times per span)
    const bool interior = first >= 0 && first + count <= n_in;
    const float *hrow = hist ? hist + (size_t)c * (sh.Q - 1) : nullptr;
    for (int base = 0; base < count; base += 16 * (int)blockDim.x) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int p = base + j * (int)blockDim.x + tid;
            const long idx = first + p;
            v[j] = 0.f;
            if (interior) {
                if (p < count) v[j] = __builtin_nontemporal_load(&row[idx]);
            } else if (p < count) {
                if (idx >= 0) {
                    if (idx < n_in) v[j] = row[idx];
                } else if (hrow && idx >= -(long)(sh.Q - 1)) {
                    v[j] = hrow[(sh.Q - 1) + idx];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int p = base + j * (int)blockDim.x + tid;
            if (p < count) xs[pos_of(p)] = v[j];
        }
    }
    __syncthreads();

    // this wave's 16 periods: column n is period m0 + 16 wave + n; its window for a phase tile starts at image element
    // (16 wave + n) M + c0(tile)   [history H and band start c0 - (Q-1) cancel]
    const long mw = m0 + 16 * wave;                                  // the wave's first period
    float *orow = out + (size_t)c * out_pitch;
    // results leave through a per-wave LDS buffer [16 periods][64 phases]: a wave instruction then stores 64 CONSECUTIVE
    // outputs of one period (stored straight from the accumulators a wave instruction would touch 64 separate dwords)
    float *stage = xs + sh.img + wave * (16 * RM_SROW);
    // Window sample d = c0 + kq + 4 s of column n sits at image position colpos + d + pad * (d / M); d < M + 4 KS, so with
    // M >= 4 KS the quotient is 0 or 1: two base addresses and one select per step instead of a division
    const int colpos = (16 * wave + n) * (sh.M + sh.pad);
    auto load_tile = [&](int t, float (&av)[KS], float (&bv)[KS]) {
        const float *ap = atab + ((size_t)t * KS) * 64 + lane;
        const int d0 = c0tab[t] + kq;
        const float *lo = xs + colpos + d0, *hi = lo + sh.pad;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            av[s] = ap[s * 64];
            bv[s] = (d0 + 4 * s >= sh.M ? hi : lo)[4 * s];
        }
    };
    auto finish_tile = [&](int t, const f32x4 &acc) {
        // D[4 kq + j][n] = output (phase 16 t + 4 kq + j, period mw + n) -> stage[n][16 (t % CHUNK) + 4 kq + j]
        *reinterpret_cast<f32x4 *>(&stage[n * RM_SROW + 16 * (t % RM_CHUNK) + 4 * kq]) = acc;
        if ((t % RM_CHUNK) == RM_CHUNK - 1 || t == sh.ntiles - 1) {
            const int f0 = 16 * (t - t % RM_CHUNK);                  // first phase of the run
            const int run = min(16 * RM_CHUNK, sh.L - f0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own LDS writes have landed
            // two periods per instruction: lanes 0..31 one run of 32 outputs, lanes 32..63 the next period's
            const int half = lane >> 5, l5 = lane & 31;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const long o = (mw + r + half) * sh.L + f0 + l5;
                if (l5 < run && o < n_out) __builtin_nontemporal_store(stage[(r + half) * RM_SROW + l5], &orow[o]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next chunk overwrites the buffer
        }
    };
    auto products = [&](const float (&av)[KS], const float (&bv)[KS]) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
        return acc;
    };
    // two register sets in turn: a tile's taps and samples arrive during the previous tile's products
    float a0[KS], b0[KS], a1[KS], b1[KS];
    load_tile(0, a0, b0);
#pragma unroll 1
    for (int t = 0; t < sh.ntiles; t += 2) {
        if (t + 1 < sh.ntiles) load_tile(t + 1, a1, b1);
        finish_tile(t, products(a0, b0));
        if (t + 1 < sh.ntiles) {
            if (t + 2 < sh.ntiles) load_tile(t + 2, a0, b0);
            finish_tile(t + 1, products(a1, b1));
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// k_resample_mfma_pt_f32: the same product with the roles turned round -- a wave owns a PHASE tile and keeps its A operands
// (the tile's taps, KS registers) for the whole launch, and sweeps the period tiles of the spans its workgroup walks.  In
// the form above every wave re-reads the whole tap table (40 KB at 147:160) through the vector memory path for each 16
// periods, and load -> wait -> compute -> store run in sequence.  Here a workgroup = ceil(L / 16) waves (4..16; more phase
// tiles than waves: the RELOAD instantiation, a wave takes several tiles in turn and re-reads A once per tile and span)
// walking consecutive spans of 16 PT periods of one channel:
//   1. a span's input is REQUESTED one span ahead into 16 registers per thread and written to the LDS image when its turn
//      comes.  Image = one column per period holding the period's whole window (M + 4 KS samples: the first 4 KS samples of a
//      period are stored a second time behind the previous period's M), column stride = 2 mod 32: a B operand is then
//      column base + immediate offset -- one address register for the KS reads of a tile, all issued back to back;
//   2. wave (tile t) x period tile p: KS LDS reads + KS MFMAs, D written to an LDS image of the OUTPUT in memory order
//      (period-major rows of L floats, odd row stride: the 16 columns of a D write spread over the banks);
//   3. the whole workgroup streams the output image out: 16 PT x L consecutive floats, one dword per lane (a wave
//      instruction = 256 contiguous, 256-byte aligned bytes).
// Measured by knocking phases out (147:160, 256 ch): moving the data alone 0.35 ms, the products alone 0.85 ms against 0.27 ms
// of matrix-pipe busy time -- the pipe idles while a workgroup moves data, and two workgroups per CU do not interleave well
// enough.  A form with specialised waves (10 multiplying + 6 data waves, one workgroup per CU, both images double-buffered,
// one barrier per span) was built and measured at 1.57 ms against 0.90 ms for this one; it was dropped.
struct rp_shape {
    int L, M, Q, ntiles;
    int pt;              // period tiles per span
    int cstride;         // floats per column of the input image: M + 4 KS + pad, = 2 mod 32
    int ostride;         // floats per row of the output image (L | 1)
    int img;             // floats of the input image: 16 pt cstride
    unsigned m_magic;    // ceil(2^32 / M)
    unsigned l_magic;    // ceil(2^32 / L)
    long spans;          // spans of a channel: ceil(periods / (16 pt))
    int spans_per_wg;    // consecutive spans a workgroup walks
};

// Round 3: the structure that took the bit-exact int16 sibling (resample_i8.hip) from 0.99 to 0.57 ms, here:
//   * a lane's four results of a period tile are four consecutive phases of one period = 16 contiguous bytes of the output row:
//     stored straight from the accumulator (global_store_dwordx4 from an SGPR base at a 4-byte aligned address) -- no output
//     image in LDS, no copy-out phase, no second barrier; the waves of a period fill its 4 L bytes within a span;
//   * the input image is DOUBLE-BUFFERED: span i + 1 is written while slower waves still read span i, ONE barrier per span;
//   * the next span is requested by hand (global_load_dword from an SGPR base) and awaited with the count of younger stores,
//     so staging does not wait for the previous span's stores to be acknowledged; spans at a frame's edges (history in front,
//     zeros behind) and the first span of a walk are staged sample by sample.
__device__ __forceinline__ float rp_load_nt(const float *base, int off)
{
    float r;
    asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(r) : "v"(off), "s"(base) : "memory");
    return r;
}
// at most n vector-memory operations (the youngest) still in flight; n = 0 .. 8
__device__ __forceinline__ void rp_wait_vm(int n)
{
    switch (n) {
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void rp_pin(float (&v)[16])
{
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
    asm volatile("" : "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) : : "memory");
}

template <int KS, bool RELOAD>
__global__ void __launch_bounds__(1024, KS <= 16 ? 5 : 4)
k_resample_mfma_pt_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                       const float *__restrict__ atab /* [ntiles][KS][64] */, const int *__restrict__ c0tab /* [ntiles] */,
                       long n_in, long n_out, long in_pitch, long out_pitch, rp_shape rp)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];      // two input images of rp.img floats
    const int tid = threadIdx.x, lane = tid & 63, waves = blockDim.x >> 6, threads = (int)blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int c = blockIdx.y;
    const int periods = 16 * rp.pt;
    const int H = rp.Q - 1;
    const int count = periods * rp.M + 4 * KS;                      // input samples a span touches (<= 16 per thread)
    const float *row = in + (size_t)c * in_pitch;
    const float *hrow = hist ? hist + (size_t)c * H : nullptr;
    float *obase = out + (size_t)c * out_pitch;
    const long total_periods = (n_out + rp.L - 1) / rp.L;

    // the wave's tile (RELOAD: tiles wave, wave + waves, ... per span, set up again for each)
    float av[KS];
    int c0 = 0, f0 = 16 * wave + 4 * kq;
    auto tile_setup = [&](int t) {
        const float *ap = atab + ((size_t)t * KS) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; s++) av[s] = ap[s * 64];
        c0 = c0tab[t];
        f0 = 16 * t + 4 * kq;
    };
    if (!RELOAD) {
        if (wave < rp.ntiles) {
            tile_setup(wave);
        } else {
#pragma unroll
            for (int s = 0; s < KS; s++) av[s] = 0.f;
        }
        // (the tile's registers pass through empty asm here: the compiler waits for their loads once, in front of it, instead of
        //  keeping them "possibly in flight" at the head of the period-tile loop -- see resample_i8.hip)
#pragma unroll
        for (int s = 0; s < KS; s++) asm volatile("" : "+v"(av[s]) : : "memory");
    }

    // a span streams when all 16 x threads samples of its request lie inside the frame (lanes past `count` read on in the row)
    auto first_of = [&](long sp) { return sp * periods * rp.M - H; };
    auto streams = [&](long first) { return first >= 0 && first + 16L * threads <= n_in; };

    // ---- a span's samples into image b ----
    auto stage = [&](long sp, int b, float (&v)[16], bool streamed, int young) {
        float *img = xs + (size_t)b * rp.img;
        if (streamed) {
            rp_wait_vm(young);
            rp_pin(v);
        } else {
            // a frame edge (history in front, zeros behind) or the first span of the walk: sample by sample (behind whatever
            // request is still on its way into the same registers)
            rp_wait_vm(0);
            rp_pin(v);
            const long first = first_of(sp);
            const int lo = first < 0 ? (int)-first : 0;
            const long room = n_in - first;
            const int hi = room < (long)count ? (int)(room < 0 ? 0 : room) : count;
            const float *bp = row + first;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int p = j * threads + tid;
                float x = 0.f;
                if (p >= lo && p < hi) x = bp[p];
                else if (p < lo && hrow && p >= lo - H) x = hrow[H - lo + p];
                v[j] = x;
            }
        }
        // The image positions do not depend on the span; left to itself the compiler keeps all 32 of them (and their
        // conditions) live across the whole walk and spills them -- recomputed per span from a value it cannot see through.
        int tid_now = tid;
        asm volatile("" : "+v"(tid_now));
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int p = j * threads + tid_now;
            const int q = (int)__umulhi((unsigned)p, rp.m_magic), r = p - q * rp.M;
            if (p < count) {
                if (q < periods) img[q * rp.cstride + r] = v[j];
                if (r < 4 * KS && q > 0) img[(q - 1) * rp.cstride + rp.M + r] = v[j];
            }
        }
    };

    // ---- products and stores of one phase tile over the span in image b; returns the number of store instructions issued
    //      (0 when it is not the same for every lane) ----
    auto products = [&](long sp, int b) {
        const long m0 = sp * periods;
        const long left = total_periods - m0;                       // periods of this span that exist
        const bool whole = f0 - 4 * kq + 15 < rp.L && left >= periods;
        float *ospan = obase + m0 * rp.L;                            // (wave-uniform)
        const float *bp = xs + (size_t)b * rp.img + n * rp.cstride + c0 + kq;       // + 16 p cstride + 4 s
        unsigned o_at = 4u * (unsigned)(n * rp.L + f0);              // byte offset of the lane's 16 output bytes in the span
        const unsigned o_step = 64u * (unsigned)rp.L;
        const int tail = f0 + 3 < rp.L ? 4 : (f0 < rp.L ? rp.L - f0 : 0);           // values of the lane that are phases
#pragma unroll 1
        for (int p = 0; p < rp.pt; p++, bp += 16 * rp.cstride, o_at += o_step) {
            float bv[KS];
#pragma unroll
            for (int s = 0; s < KS; s++) bv[s] = bp[4 * s];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
            if (whole) {
                asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(o_at), "v"(acc), "s"(ospan) : "memory");
            } else if (16 * p + n < left) {
                float *op = reinterpret_cast<float *>(reinterpret_cast<char *>(ospan) + o_at);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (j < tail) op[j] = acc[j];
            }
        }
        return whole ? rp.pt : 0;
    };

    // One loop, one request site.  Iteration i runs the products of span i (none in the first iteration), writes span i + 1
    // into the other image and requests span i + 2; the barrier at its end publishes image i + 1 and retires the readers of i.
    // A request that does not stream (or has no span) reads the row's head instead: in bounds -- the launcher requires
    // n_in >= 16 x threads -- and never looked at.
    const long span0 = (long)blockIdx.x * rp.spans_per_wg;
    const long span1 = min(span0 + rp.spans_per_wg, rp.spans);
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = 0.f;
    bool requested = false;                             // the registers hold span i + 1
    int b = 0;
    const int lane_off = 4 * tid;
    for (long i = span0 - 1; i < span1; i++) {
        int young = 0;
        if (i >= span0) {
            if (!RELOAD) {
                if (wave < rp.ntiles) young = products(i, b);
            } else {
                bool counted = true;
                for (int t = wave; t < rp.ntiles; t += waves) {
                    tile_setup(t);
                    const int stores = products(i, b);
                    counted = counted && stores > 0;
                    young += stores;
                }
                // (fewer than the true number of younger stores only waits for more; the tile loads above are younger too)
                young = counted ? (young < 8 ? young : 8) : 0;
            }
        }
        if (i + 1 < span1) stage(i + 1, b ^ 1, v, requested, young);
        const long fnext = first_of(i + 2);
        requested = i + 2 < span1 && streams(fnext);
        const float *src = requested ? row + fnext : row;
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = rp_load_nt(src + (size_t)j * threads, lane_off);
        __syncthreads();
        b ^= 1;
    }
}

} // namespace

extern "C" int llzs_resample_mfma_f32_ksteps(int L, int M, int Q)
{
    // widest band over the phase tiles, in steps of 4 samples
    int widest = 0;
    for (int f0 = 0; f0 < L; f0 += 16) {
        const int f1 = f0 + 15 < L ? f0 + 15 : L - 1;
        const int w = (int)(((long)f1 * M) / L - ((long)f0 * M) / L) + Q;
        if (w > widest) widest = w;
    }
    return (widest + 3) / 4;
}

static int rm_pad(int M) { return ((2 - M) % 32 + 32) % 32; }

// waves per workgroup: as many period tiles (16 periods each) as fit ~48 KB of signal image, at most 4
static int rm_waves(int M)
{
    int w = (48 * 1024) / (16 * (M + rm_pad(M)) * (int)sizeof(float));
    return w < 1 ? 1 : (w > RM_MAX_WAVES ? RM_MAX_WAVES : w);
}

static size_t rm_image_floats_n(int M, int Q, int periods, int kst)
{
    const size_t span = (size_t)periods * M + (Q - 1) + 4 * kst;
    return span + (size_t)rm_pad(M) * (span / M) + 8;
}
static size_t rm_image_floats(int M, int Q, int waves, int kst) { return rm_image_floats_n(M, Q, 16 * waves, kst); }

extern "C" int llzs_resample_mfma_f32_table_steps(int L, int M, int Q)
{
    const int ks = llzs_resample_mfma_f32_ksteps(L, M, Q);
    return ks <= 8 ? 8 : ks <= 16 ? 16 : ks <= 24 ? 24 : 32;
}

extern "C" int llzs_resample_mfma_f32_fits(int L, int M, int Q)
{
    // small L: the register-window kernel; short periods (M < a band's 4 KS samples) give a workgroup too little to do per
    // staged span: the LDS form is faster there (8:7: 1.8 against 3.8 ms)
    if (L < 5 || M < 1 || Q < 1) return 0;
    if (llzs_resample_mfma_f32_ksteps(L, M, Q) > 32) return 0;
    const int waves = rm_waves(M), kst = llzs_resample_mfma_f32_table_steps(L, M, Q);
    const size_t lds = (rm_image_floats(M, Q, waves, kst) + 4 + (size_t)waves * 16 * RM_SROW) * sizeof(float);
    return M >= 4 * kst && lds <= 160 * 1024 && (long)16 * waves * M + Q + 128 < (long)(0x100000000ull / (unsigned)M);
}

// atab: [ceil(L/16)][steps][64] floats (steps = llzs_resample_mfma_f32_table_steps), gain folded in; c0tab: [ceil(L/16)]
// ints = floor(16 t M / L).  The call must start on a period boundary.
extern "C" int llzs_resample_mfma_f32(const float *in, float *out, const float *hist, const float *atab, const int *c0tab,
                                      int channels, long n_in, long n_out, long in_pitch, long out_pitch, int L, int M, int Q,
                                      void *stream)
{
    if (!in || !out || !atab || !c0tab || channels <= 0 || channels > 65535 || n_in <= 0 || n_out <= 0 ||
        in_pitch < n_in || out_pitch < n_out || !llzs_resample_mfma_f32_fits(L, M, Q)) {
        llzs_set_error("resample_mfma_f32: bad arguments (channels=%d L=%d M=%d Q=%d)", channels, L, M, Q);
        return LLZ_ERR_ARG;
    }
    const int kst = llzs_resample_mfma_f32_table_steps(L, M, Q);
    if (llzs_tune(LLZS_TUNE_RS_MFMA_FORM) != 1) {
        // a wave per phase tile (k_resample_mfma_pt_f32): the span is as many period tiles (at most 4) as keep the two input
        // images of a workgroup within 128 KB and its samples within 16 per thread; the frame must be at least one request long
        // (requests without a span read the row's head): shorter frames take the period-tile form below
        rp_shape rp;
        rp.L = L; rp.M = M; rp.Q = Q;
        rp.ntiles = (L + 15) / 16;
        rp.ostride = L | 1;
        rp.cstride = M + 4 * kst;
        rp.cstride += ((2 - rp.cstride) % 32 + 32) % 32;
        const int waves_min = rp.ntiles < 4 ? 4 : (rp.ntiles > 16 ? 16 : rp.ntiles);
        int waves = waves_min, pt = 4;
        if (const int forced = llzs_tune(LLZS_TUNE_RS_TILES); forced >= 1 && forced <= 4) pt = forced;
        size_t lds = 0;
        bool ok = false;
        for (; pt >= 1; pt--) {
            rp.img = 16 * pt * rp.cstride;
            lds = 2 * (size_t)rp.img * sizeof(float);
            const long count = (long)16 * pt * M + 4 * kst;
            waves = waves_min;
            while (waves < 16 && count > 16L * 64 * waves) waves++;
            ok = lds <= 128 * 1024 && count <= 16L * 64 * waves && n_in >= 16L * 64 * waves;
            if (ok) break;
        }
        if (ok) {
            rp.pt = pt;
            rp.m_magic = (unsigned)((0x100000000ull + (unsigned)M - 1) / (unsigned)M);
            rp.l_magic = (unsigned)((0x100000000ull + (unsigned)L - 1) / (unsigned)L);
            const long periods = (n_out + L - 1) / L;
            rp.spans = (periods + 16 * pt - 1) / (16 * pt);
            // consecutive spans per workgroup: the walk length that minimises rounds x (length + 1) over the resident
            // workgroups -- a walk's first span waits for memory with nothing to do, and a last round that is half empty costs
            // as much as a full one
            const long resident = lds > 78 * 1024 ? 256 : 512;
            long spw = rp.spans < 4 ? rp.spans : 4;
            double best = 1e300;
            for (long c = spw; c <= rp.spans; c++) {
                const long wgs = ((rp.spans + c - 1) / c) * (long)channels;
                const double cost = (double)((wgs + resident - 1) / resident) * (double)(c + 1);
                if (cost < best * 0.999) { best = cost; spw = c; }
            }
            rp.spans_per_wg = (int)spw;
            const dim3 grid((unsigned)((rp.spans + spw - 1) / spw), (unsigned)channels), block(64 * waves);
            const bool reload = rp.ntiles > waves;
#define RP_GO2(K, R)                                                                                                 \
    do {                                                                                                             \
        if (lds > 64 * 1024)                                                                                         \
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resample_mfma_pt_f32<K, R>),          \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                \
        hipLaunchKernelGGL((k_resample_mfma_pt_f32<K, R>), grid, block, lds, as_stream(stream), in, out, hist, atab, \
                           c0tab, n_in, n_out, in_pitch, out_pitch, rp);                                             \
    } while (0)
#define RP_GO(K) do { if (reload) RP_GO2(K, true); else RP_GO2(K, false); } while (0)
            if (kst == 8) RP_GO(8);
            else if (kst == 16) RP_GO(16);
            else if (kst == 24) RP_GO(24);
            else RP_GO(32);
#undef RP_GO
#undef RP_GO2
            LLZ_LAUNCH_CHECK("k_resample_mfma_pt_f32");
            return LLZ_OK;
        }
    }
    const int waves = rm_waves(M);
    rm_shape sh;
    sh.L = L; sh.M = M; sh.Q = Q;
    sh.ntiles = (L + 15) / 16;
    sh.pad = rm_pad(M);
    sh.H = Q - 1;
    sh.img = (int)rm_image_floats(M, Q, waves, kst);
    sh.img = (sh.img + 3) & ~3;                                              // 16-byte aligned staging rows behind it
    sh.m_magic = (unsigned)((0x100000000ull + (unsigned)M - 1) / (unsigned)M);
    const long periods = (n_out + L - 1) / L;
    const dim3 grid((unsigned)((periods + 16 * waves - 1) / (16 * waves)), (unsigned)channels), block(64 * waves);
    const size_t lds = ((size_t)sh.img + (size_t)waves * 16 * RM_SROW) * sizeof(float);
#define RM_GO(K)                                                                                                     \
    do {                                                                                                             \
        if (lds > 64 * 1024)                                                                                         \
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resample_mfma_f32<K>),                \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                \
        hipLaunchKernelGGL(k_resample_mfma_f32<K>, grid, block, lds, as_stream(stream), in, out, hist, atab, c0tab,  \
                           n_in, n_out, in_pitch, out_pitch, sh);                                                    \
    } while (0)
    if (kst == 8) RM_GO(8);
    else if (kst == 16) RM_GO(16);
    else if (kst == 24) RM_GO(24);
    else RM_GO(32);
#undef RM_GO
    LLZ_LAUNCH_CHECK("k_resample_mfma_f32");
    return LLZ_OK;
}
