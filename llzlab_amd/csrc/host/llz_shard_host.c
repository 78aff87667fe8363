/* llz_shard_host.c -- single-process channel sharding over several GPUs (include/llz_shard.h).
 *
 * One reference handle is one channel (libllzfilter/llz_fir.c:23-33, llz_iir.c:17-26, llz_resample.c:54-78), so a batch
 * handle splits into contiguous channel ranges with nothing shared but the coefficient tables.  A sharded handle is a
 * list of ordinary multi-channel handles, one per entry of devices[], each bound to its device and to a stream of its
 * own; the tables are uploaded once (shard 0) and broadcast (llzs_tables_broadcast: RCCL between devices). */
#include <stdlib.h>
#include <string.h>
#include "llz_host.h"
#include "../../../include/llz_shard.h"

enum { SHARD_FIR = 1, SHARD_IIR = 2, SHARD_RS = 3 };
#define LLZ_TAG_SHARD 0x4c5a5348
#define SHARD_MAX 64

typedef struct {
    int tag, kind, n, channels, rccl_ranks;
    int device[SHARD_MAX], chan0[SHARD_MAX], count[SHARD_MAX];
    unsigned long sub[SHARD_MAX];
    void *stream[SHARD_MAX];
    void *ev_start[SHARD_MAX], *ev_stop[SHARD_MAX];
} shard_t;

int llz_shard_range(int channels, int n_shards, int shard, int *chan0, int *count)
{
    if (channels < 0 || n_shards < 1 || shard < 0 || shard >= n_shards) {
        llzs_set_error("llz_shard_range: shard %d of %d, %d channels", shard, n_shards, channels);
        return LLZ_ERR_ARG;
    }
    const int base = channels / n_shards, rem = channels % n_shards;
    if (chan0) *chan0 = shard * base + (shard < rem ? shard : rem);
    if (count) *count = base + (shard < rem ? 1 : 0);
    return LLZ_OK;
}

static void shard_sub_uninit(int kind, unsigned long h)
{
    if (kind == SHARD_FIR) llz_fir_filter_mc_uninit(h);
    else if (kind == SHARD_IIR) llz_iir_cascade_mc_uninit(h);
    else if (kind == SHARD_RS) llz_resample_mc_uninit(h);
}

static int shard_sub_set_stream(int kind, unsigned long h, void *stream)
{
    if (kind == SHARD_FIR) return llz_fir_filter_mc_set_stream(h, stream);
    if (kind == SHARD_IIR) return llz_iir_cascade_mc_set_stream(h, stream);
    return llz_resample_mc_set_stream(h, stream);
}

static void shard_destroy(shard_t *g)
{
    if (!g) return;
    const int prev = llzs_device_get();
    for (int s = 0; s < g->n; s++) {
        if (llzs_device_set(g->device[s]) != LLZ_OK) continue;
        if (g->sub[s] && g->sub[s] != LLZ_BAD_HANDLE) shard_sub_uninit(g->kind, g->sub[s]);
        llzs_event_destroy(g->ev_start[s]);
        llzs_event_destroy(g->ev_stop[s]);
        llzs_stream_destroy(g->stream[s]);
    }
    if (prev >= 0) llzs_device_set(prev);
    g->tag = 0;
    free(g);
}

typedef unsigned long (*shard_make_fn)(const void *ctx, int channels);

/* build one sub-handle per shard: shard 0 uploads its tables, the others only allocate theirs; then one broadcast */
static unsigned long shard_create(int kind, int channels, const int *devices, int n_devices, shard_make_fn make,
                                  const void *ctx, const char *what)
{
    const int ndev = llz_hip_device_count();
    if (channels < 1 || !devices || n_devices < 1 || n_devices > SHARD_MAX || n_devices > channels) {
        llzs_set_error("%s: %d channels over %d shards (1..%d shards, at most one per channel)", what, channels, n_devices,
                       SHARD_MAX);
        return LLZ_BAD_HANDLE;
    }
    for (int s = 0; s < n_devices; s++)
        if (devices[s] < 0 || devices[s] >= ndev) {
            llzs_set_error("%s: devices[%d] = %d, the node has %d device(s)", what, s, devices[s], ndev);
            return LLZ_BAD_HANDLE;
        }
    shard_t *g = (shard_t *)calloc(1, sizeof(*g));
    if (!g) return LLZ_BAD_HANDLE;
    g->tag = LLZ_TAG_SHARD; g->kind = kind; g->n = n_devices; g->channels = channels;
    const int prev = llzs_device_get();
    llzs_table_ref tabs[SHARD_MAX][LLZS_MAX_TABLES];
    llzs_table_ref *tab_ptr[SHARD_MAX];
    int ntab = -1, rc = LLZ_OK;
    for (int s = 0; s < n_devices && rc == LLZ_OK; s++) {
        g->device[s] = devices[s];
        llz_shard_range(channels, n_devices, s, &g->chan0[s], &g->count[s]);
        tab_ptr[s] = tabs[s];
        rc = llzs_device_set(devices[s]);
        if (rc != LLZ_OK) break;
        g->stream[s] = llzs_stream_create();
        g->ev_start[s] = llzs_event_create();
        g->ev_stop[s] = llzs_event_create();
        if (!g->stream[s] || !g->ev_start[s] || !g->ev_stop[s]) { rc = LLZ_ERR_DEVICE; break; }
        llzs_table_capture(s == 0 ? 1 : 2);
        g->sub[s] = make(ctx, g->count[s]);
        const int got = llzs_table_captured(tabs[s], LLZS_MAX_TABLES);
        llzs_table_capture(0);
        if (g->sub[s] == LLZ_BAD_HANDLE || g->sub[s] == 0) { rc = LLZ_ERR_DEVICE; break; }     /* message set by the init */
        if (ntab < 0) ntab = got;
        if (got != ntab || got > LLZS_MAX_TABLES) {
            llzs_set_error("%s: shard %d built %d coefficient tables, shard 0 %d", what, s, got, ntab);
            rc = LLZ_ERR_DEVICE;
            break;
        }
        rc = shard_sub_set_stream(kind, g->sub[s], g->stream[s]);
    }
    if (rc == LLZ_OK) rc = llzs_tables_broadcast(tab_ptr, ntab, n_devices, g->device, g->stream);
    if (rc == LLZ_OK) g->rccl_ranks = llzs_tables_broadcast_ranks();
    if (prev >= 0) llzs_device_set(prev);
    if (rc != LLZ_OK) {
        shard_destroy(g);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)g;
}

static shard_t *shard_of(unsigned long handle, int kind, const char *what)
{
    if (handle == 0 || handle == LLZ_BAD_HANDLE || ((shard_t *)handle)->tag != LLZ_TAG_SHARD ||
        (kind && ((shard_t *)handle)->kind != kind)) {
        llzs_set_error("%s: not a sharded handle of this kind", what);
        return NULL;
    }
    return (shard_t *)handle;
}

/* ---- FIR ---- */
typedef struct { int frame_len, flt_len, algo; const float *h; } fir_ctx;
static unsigned long fir_make(const void *c, int channels)
{
    const fir_ctx *x = (const fir_ctx *)c;
    return llz_fir_filter_mc_init(channels, x->frame_len, x->h, x->flt_len, x->algo);
}

unsigned long llz_fir_filter_mc_sharded_init(int channels, int frame_len, const float *h, int flt_len, int algo,
                                             const int *devices, int n_devices)
{
    const fir_ctx c = {frame_len, flt_len, algo, h};
    return shard_create(SHARD_FIR, channels, devices, n_devices, fir_make, &c, "llz_fir_filter_mc_sharded_init");
}

int llz_fir_filter_mc_sharded(unsigned long handle, const float *const *in, float *const *out, int frame_len)
{
    shard_t *g = shard_of(handle, SHARD_FIR, "llz_fir_filter_mc_sharded");
    if (!g || !in || !out) return LLZ_ERR_ARG;
    int rc = frame_len;
    for (int s = 0; s < g->n && rc >= 0; s++) {
        const int r = llz_fir_filter_mc(g->sub[s], in[s], out[s], frame_len);       /* binds the shard's device itself */
        if (r < 0) rc = r;
    }
    return rc;
}

int llz_fir_filter_mc_sharded_flush(unsigned long handle, float *const *out)
{
    shard_t *g = shard_of(handle, SHARD_FIR, "llz_fir_filter_mc_sharded_flush");
    if (!g || !out) return LLZ_ERR_ARG;
    int rc = 0;
    for (int s = 0; s < g->n && rc >= 0; s++) rc = llz_fir_filter_mc_flush(g->sub[s], out[s]);
    return rc;
}

/* ---- IIR ---- */
typedef struct { int stages; const double *coef; } iir_ctx;
static unsigned long iir_make(const void *c, int channels)
{
    const iir_ctx *x = (const iir_ctx *)c;
    return llz_iir_cascade_mc_init(channels, x->stages, x->coef);
}

unsigned long llz_iir_cascade_mc_sharded_init(int channels, int stages, const double *coef, const int *devices,
                                              int n_devices)
{
    const iir_ctx c = {stages, coef};
    return shard_create(SHARD_IIR, channels, devices, n_devices, iir_make, &c, "llz_iir_cascade_mc_sharded_init");
}

int llz_iir_cascade_mc_sharded(unsigned long handle, const float *const *x, float *const *y, int frame_len)
{
    shard_t *g = shard_of(handle, SHARD_IIR, "llz_iir_cascade_mc_sharded");
    if (!g || !x || !y) return LLZ_ERR_ARG;
    int rc = frame_len;
    for (int s = 0; s < g->n && rc >= 0; s++) {
        const int r = llz_iir_cascade_mc(g->sub[s], x[s], y[s], frame_len);
        if (r < 0) rc = r;
    }
    return rc;
}

/* ---- resample ---- */
typedef struct { int L, M, fmt; double gain; win_t win; } rs_ctx;
static unsigned long rs_make(const void *c, int channels)
{
    const rs_ctx *x = (const rs_ctx *)c;
    return llz_resample_mc_init(channels, x->L, x->M, x->gain, x->win, x->fmt);
}

unsigned long llz_resample_mc_sharded_init(int channels, int L, int M, double gain, win_t win_type, int pcm_format,
                                           const int *devices, int n_devices)
{
    const rs_ctx c = {L, M, pcm_format, gain, win_type};
    return shard_create(SHARD_RS, channels, devices, n_devices, rs_make, &c, "llz_resample_mc_sharded_init");
}

long llz_resample_mc_sharded(unsigned long handle, const void *const *in, long n_in, void *const *out)
{
    shard_t *g = shard_of(handle, SHARD_RS, "llz_resample_mc_sharded");
    if (!g || !in || !out) return LLZ_ERR_ARG;
    long rc = 0;
    for (int s = 0; s < g->n && rc >= 0; s++) rc = llz_resample_mc(g->sub[s], in[s], n_in, out[s]);
    return rc;
}

/* ---- common ---- */
void llz_sharded_uninit(unsigned long handle)
{
    shard_t *g = shard_of(handle, 0, "llz_sharded_uninit");
    if (g) shard_destroy(g);
}

int llz_sharded_count(unsigned long handle)
{
    shard_t *g = shard_of(handle, 0, "llz_sharded_count");
    return g ? g->n : LLZ_ERR_ARG;
}

int llz_sharded_rccl_ranks(unsigned long handle)
{
    shard_t *g = shard_of(handle, 0, "llz_sharded_rccl_ranks");
    return g ? g->rccl_ranks : LLZ_ERR_ARG;
}

int llz_sharded_shard(unsigned long handle, int shard, int *device, int *chan0, int *count)
{
    shard_t *g = shard_of(handle, 0, "llz_sharded_shard");
    if (!g || shard < 0 || shard >= g->n) return LLZ_ERR_ARG;
    if (device) *device = g->device[shard];
    if (chan0) *chan0 = g->chan0[shard];
    if (count) *count = g->count[shard];
    return LLZ_OK;
}

void *llz_sharded_stream(unsigned long handle, int shard)
{
    shard_t *g = shard_of(handle, 0, "llz_sharded_stream");
    return (g && shard >= 0 && shard < g->n) ? g->stream[shard] : NULL;
}

unsigned long llz_sharded_sub(unsigned long handle, int shard)
{
    shard_t *g = shard_of(handle, 0, "llz_sharded_sub");
    return (g && shard >= 0 && shard < g->n) ? g->sub[shard] : LLZ_BAD_HANDLE;
}

int llz_sharded_synchronize(unsigned long handle)
{
    shard_t *g = shard_of(handle, 0, "llz_sharded_synchronize");
    if (!g) return LLZ_ERR_ARG;
    int rc = LLZ_OK;
    for (int s = 0; s < g->n; s++) {
        const int prev = llzs_device_enter(g->device[s]);
        const int r = llzs_sync(g->stream[s]);
        llzs_device_leave(prev);
        if (r != LLZ_OK) rc = r;
    }
    return rc;
}

static int shard_record(shard_t *g, void *const *ev)
{
    int rc = LLZ_OK;
    for (int s = 0; s < g->n; s++) {
        const int prev = llzs_device_enter(g->device[s]);
        const int r = llzs_event_record(ev[s], g->stream[s]);
        llzs_device_leave(prev);
        if (r != LLZ_OK) rc = r;
    }
    return rc;
}

int llz_sharded_timer_start(unsigned long handle)
{
    shard_t *g = shard_of(handle, 0, "llz_sharded_timer_start");
    return g ? shard_record(g, g->ev_start) : LLZ_ERR_ARG;
}

int llz_sharded_timer_stop(unsigned long handle)
{
    shard_t *g = shard_of(handle, 0, "llz_sharded_timer_stop");
    return g ? shard_record(g, g->ev_stop) : LLZ_ERR_ARG;
}

double llz_sharded_timer_ms(unsigned long handle, double *per_shard_ms)
{
    shard_t *g = shard_of(handle, 0, "llz_sharded_timer_ms");
    if (!g) return -1.0;
    double worst = 0.0;
    for (int s = 0; s < g->n; s++) {
        const int prev = llzs_device_enter(g->device[s]);
        const double ms = llzs_event_elapsed_ms(g->ev_start[s], g->ev_stop[s]);
        llzs_device_leave(prev);
        if (ms < 0) return -1.0;
        if (per_shard_ms) per_shard_ms[s] = ms;
        if (ms > worst) worst = ms;
    }
    return worst;
}
