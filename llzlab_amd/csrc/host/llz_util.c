/* llz_util.c -- small host helpers of the C layer */
#include "llz_host.h"

void *llz_stage_reserve(llz_stage_t *s, size_t bytes)
{
    if (s->bytes >= bytes && s->dev) return s->dev;
    if (s->dev) llzs_free(s->dev);
    s->dev = llzs_malloc(bytes);
    s->bytes = s->dev ? bytes : 0;
    return s->dev;
}

void llz_stage_release(llz_stage_t *s)
{
    if (s->dev) llzs_free(s->dev);
    s->dev = NULL;
    s->bytes = 0;
}
