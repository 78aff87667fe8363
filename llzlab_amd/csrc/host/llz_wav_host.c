/* llz_wav_host.c -- the WAV container around the PCM ingest kernels (include/llz_pcm.h): header walk and writer of the
 * reference's libllzaudio/llz_wavfmt.c:82-213 on a memory image, for any number of channels, and the ingest call that hands
 * the data chunk to the device de-interleaver.  Where the reference prints and exit(0)s (not RIFF/WAVE, not PCM) or would
 * run past the end of a damaged file, these return LLZ_ERR_ARG with a message. */
#include <stdlib.h>
#include <string.h>
#include "../../../include/llz_pcm.h"
#include "llz_host.h"

static unsigned long le_u(const unsigned char *p, int bytes)
{
    unsigned long v = 0;
    while (bytes-- > 0) v = (v << 8) | p[bytes];
    return v;
}

static void le_put(unsigned char *p, unsigned long v, int bytes)
{
    for (int i = 0; i < bytes; i++) p[i] = (unsigned char)(v >> (8 * i));
}

/* position of the chunk `id` at or after `pos` (chunks in between are skipped by their size), or -1 */
static long find_chunk(const unsigned char *b, long len, long pos, const char *id)
{
    while (pos + 8 <= len) {
        if (memcmp(b + pos, id, 4) == 0) return pos;
        pos += 8 + (long)le_u(b + pos + 4, 4);
    }
    return -1;
}

int llz_wav_parse(const unsigned char *image, long len, llz_wav_info *info)
{
    if (!image || !info || len < 12) {
        llzs_set_error("llz_wav_parse: no image");
        return LLZ_ERR_ARG;
    }
    if (memcmp(image, "RIFF", 4) != 0 || memcmp(image + 8, "WAVE", 4) != 0) {
        llzs_set_error("llz_wav_parse: not a RIFF/WAVE image");
        return LLZ_ERR_ARG;
    }
    const long fmt = find_chunk(image, len, 12, "fmt ");
    if (fmt < 0) {
        llzs_set_error("llz_wav_parse: no fmt chunk");
        return LLZ_ERR_ARG;
    }
    const long fmt_size = (long)le_u(image + fmt + 4, 4), body = fmt + 8;
    if (fmt_size < 16 || body + fmt_size > len) {
        llzs_set_error("llz_wav_parse: fmt chunk of %ld bytes", fmt_size);
        return LLZ_ERR_ARG;
    }
    info->format = (int)le_u(image + body, 2);
    info->channels = (int)le_u(image + body + 2, 2);
    info->samplerate = (long)le_u(image + body + 4, 4);
    info->bytes_per_sample = (int)((le_u(image + body + 14, 2) + 7) / 8);
    info->block_align = info->bytes_per_sample * info->channels;
    if (info->format != 1 || info->channels < 1 || info->bytes_per_sample < 1) {
        llzs_set_error("llz_wav_parse: format %d with %d channel(s) of %d byte(s): only PCM is supported", info->format,
                       info->channels, info->bytes_per_sample);
        return LLZ_ERR_ARG;
    }
    const long data = find_chunk(image, len, body + fmt_size, "data");
    if (data < 0) {
        llzs_set_error("llz_wav_parse: no data chunk");
        return LLZ_ERR_ARG;
    }
    info->frames = (long)(le_u(image + data + 4, 4) / (unsigned long)info->block_align);
    info->data_offset = data + 8;
    return LLZ_OK;
}

int llz_wav_write_header(unsigned char header[44], const llz_wav_info *info)
{
    if (!header || !info || info->channels < 1 || info->bytes_per_sample < 1 || info->frames < 0) {
        llzs_set_error("llz_wav_write_header: bad arguments");
        return LLZ_ERR_ARG;
    }
    const unsigned long block = (unsigned long)info->channels * (unsigned long)info->bytes_per_sample;
    const unsigned long bytes = (unsigned long)info->frames * block;
    memcpy(header, "RIFF", 4);
    le_put(header + 4, bytes + 36, 4);
    memcpy(header + 8, "WAVEfmt ", 8);
    le_put(header + 16, 16, 4);
    le_put(header + 20, 1, 2);
    le_put(header + 22, (unsigned long)info->channels, 2);
    le_put(header + 24, (unsigned long)info->samplerate, 4);
    le_put(header + 28, block * (unsigned long)info->samplerate, 4);
    le_put(header + 32, block, 2);
    le_put(header + 34, (unsigned long)info->bytes_per_sample * 8, 2);
    memcpy(header + 36, "data", 4);
    le_put(header + 40, bytes, 4);
    return LLZ_OK;
}

long llz_wav_ingest_f32(const unsigned char *image, long len, float *planar_out, long out_capacity_frames,
                        llz_wav_info *info, void *stream)
{
    llz_wav_info local;
    if (!info) info = &local;
    int rc = llz_wav_parse(image, len, info);
    if (rc != LLZ_OK) return rc;
    if (info->bytes_per_sample != 2) {
        llzs_set_error("llz_wav_ingest_f32: %d-byte samples (16-bit PCM only)", info->bytes_per_sample);
        return LLZ_ERR_ARG;
    }
    long frames = info->frames;
    const long present = (len - info->data_offset) / info->block_align;      /* a truncated file holds fewer */
    if (frames > present) frames = present;
    if (!planar_out || frames < 1 || frames > out_capacity_frames) {
        llzs_set_error("llz_wav_ingest_f32: %ld frame(s), room for %ld", frames, out_capacity_frames);
        return LLZ_ERR_ARG;
    }
    /* the data chunk starts at an even offset in any well-formed file; a misaligned one is copied */
    const short *pcm = (const short *)(const void *)(image + info->data_offset);
    short *aligned = NULL;
    if (((size_t)(image + info->data_offset) & 1) != 0) {
        aligned = (short *)malloc((size_t)frames * info->block_align);
        if (!aligned) return LLZ_ERR_NOMEM;
        memcpy(aligned, image + info->data_offset, (size_t)frames * info->block_align);
        pcm = aligned;
    }
    /* planar rows of `frames` samples (row pitch = frames) */
    rc = llz_pcm_deinterleave_i16_f32(pcm, planar_out, info->channels, frames, 1.0f / 32768.0f, stream);
    free(aligned);
    return rc == LLZ_OK ? frames : rc;
}
