/* llz_pcm_host.c -- host entry points of the PCM ingest / egress kernels (include/llz_pcm.h) */
#include "../../../include/llz_pcm.h"
#include "llz_host.h"

static int pcm_run(int deint, const void *in, void *out, int channels, long n, float scale, void *stream)
{
    if (!in || !out || channels < 1 || n < 1) {
        llzs_set_error("llz_pcm_*: bad arguments");
        return LLZ_ERR_ARG;
    }
    const size_t count = (size_t)channels * (size_t)n;
    const size_t in_bytes = count * (deint ? sizeof(short) : sizeof(float));
    const size_t out_bytes = count * (deint ? sizeof(float) : sizeof(short));
    const int in_dev = llzs_is_device_ptr(in), out_dev = llzs_is_device_ptr(out);
    if (in_dev < 0 || out_dev < 0) return LLZ_ERR_ARG;            /* device memory of a GPU that is not current */
    void *d_in = (void *)in, *d_out = out;
    int rc = LLZ_OK;
    if (!in_dev) {
        d_in = llzs_malloc(in_bytes);
        rc = d_in ? llzs_h2d(d_in, in, in_bytes, stream) : LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK && !out_dev) {
        d_out = llzs_malloc(out_bytes);
        if (!d_out) rc = LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK)
        rc = deint ? llzs_pcm_deinterleave_i16_f32((const short *)d_in, (float *)d_out, channels, n, scale, stream)
                   : llzs_pcm_interleave_f32_i16((const float *)d_in, (short *)d_out, channels, n, scale, stream);
    if (rc == LLZ_OK && !out_dev) rc = llzs_d2h(out, d_out, out_bytes, stream);
    if (!in_dev) { llzs_sync(stream); llzs_free(d_in); }
    if (!out_dev) llzs_free(d_out);
    return rc;
}

int llz_pcm_deinterleave_i16_f32(const short *in, float *out, int channels, long n, float scale, void *stream)
{
    return pcm_run(1, in, out, channels, n, scale, stream);
}

int llz_pcm_interleave_f32_i16(const float *in, short *out, int channels, long n, float scale, void *stream)
{
    return pcm_run(0, in, out, channels, n, scale, stream);
}
