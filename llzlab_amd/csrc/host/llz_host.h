/* llz_host.h -- INTERNAL helpers shared by the host C layer */
#ifndef LLZ_HOST_H
#define LLZ_HOST_H

#include <stddef.h>
#include "../../../include/llz_hip.h"
#include "../../../include/llz_fir.h"
#include "../llz_shim.h"

enum { LLZ_KIND_LPF = 0, LLZ_KIND_HPF = 1, LLZ_KIND_BPF = 2, LLZ_KIND_BSF = 3 };

/* windowed-sinc design shared by the four llz_fir_*_cof symbols; *out is malloc'ed, returns the tap count
 * (odd-forced for HPF/BPF/BSF) or -1 */
int llz_host_design(int kind, double **out, int n, double fc1, double fc2, win_t win);

/* handle tags: a wrong or stale handle fails the tag check instead of dereferencing garbage */
enum {
    LLZ_TAG_FIR1 = 0x4c5a4631, LLZ_TAG_FIRM = 0x4c5a464d, LLZ_TAG_IIR1 = 0x4c5a4931, LLZ_TAG_IIRM = 0x4c5a494d,
    LLZ_TAG_RS1 = 0x4c5a5231, LLZ_TAG_RSM = 0x4c5a524d, LLZ_TAG_FFT1 = 0x4c5a5431, LLZ_TAG_FFTB = 0x4c5a5442,
    LLZ_TAG_FFTX = 0x4c5a5458
};

#define LLZ_HANDLE_OK(h, type, tagv) ((h) != 0 && (h) != LLZ_BAD_HANDLE && ((type *)(h))->tag == (tagv))

/* one frame of a llz_mdct_init() handle between two device buffers of doubles (llz_mdct_host.c; used by the frame handles
 * of llz_asmodel_host.c): forward N samples -> N/2 coefficients, inverse the other way; d_src != d_dst */
int llz_host_mdct_on_device(unsigned long handle, const double *d_src, double *d_dst, int inverse);
/* cos then sin of 2 pi i / size, i < size, as llz_fft_init builds them (llz_fft.c:222-229), uploaded; NULL on failure */
double *llz_host_fft_table_f64(int size);

/* Caller buffers: llzs_is_device_ptr(p) is 1 for device memory of the CURRENT device (used in place), 0 for host memory
 * (staged through the GPU) and LLZ_ERR_ARG, with a message, for device memory that lives on another device -- a handle
 * binds its device before it looks at the caller's pointers, so a buffer of the wrong GPU is refused instead of faulting. */

/* staging buffers for callers that hand over host memory */
typedef struct {
    void *dev;
    size_t bytes;
} llz_stage_t;

/* grow-only device scratch; returns NULL on failure */
void *llz_stage_reserve(llz_stage_t *s, size_t bytes);
void  llz_stage_release(llz_stage_t *s);

#endif
