/*
 * loudscan_ebur128.h -- the slice of libebur128's C API that loudgain's scan
 * module calls, served by the HIP scanner (SURVEY.md section 8b "inner boundary").
 *
 * libebur128 itself is NOT part of the reference tree (third party, pinned
 * >= 1.2.4 by /root/reference/debian/control:10 and README.md:329; version check at
 * /root/reference/src/loudgain.c:179-184).  This header restates the public
 * interface of 1.2.4 for the symbols below so that an UNMODIFIED scan.c links
 * against libloudscan_hip.so instead of -lebur128 (include/compat/ebur128.h makes
 * `#include <ebur128.h>` resolve to this file).
 *
 *   call site in /root/reference/src/scan.c            symbol
 *   :203  ebur128_init(channels, rate, S|I|LRA|SAMPLE_PEAK|TRUE_PEAK)
 *   :448  ebur128_add_frames_short(state, s16, nb_samples)
 *   :294  ebur128_loudness_global          :297  ebur128_loudness_range
 *   :303,:371  ebur128_true_peak(state, ch, &out)   (state->channels read at :300,:368)
 *   :383  ebur128_loudness_global_multiple :388  ebur128_loudness_range_multiple
 *   :102  ebur128_destroy
 *   /root/reference/src/loudgain.c:179  ebur128_get_version
 * plus ebur128_add_frames_float and ebur128_sample_peak (same machinery).
 *
 * How it differs from the CPU library: frames are only collected by
 * ebur128_add_frames_*; the arithmetic runs on the GPU at the first query -- ONE batched scan
 * of every live state that holds frames, as the tracks of one album -- and every later query
 * of those states, single or _multiple, is served from its results until more frames arrive.
 * Momentary / short-term queries, ebur128_set_channel and the histogram mode are not
 * provided (loudgain does not use them).  There is no CPU fallback: without a HIP
 * device ebur128_init returns NULL.
 */
#ifndef LOUDSCAN_EBUR128_H
#define LOUDSCAN_EBUR128_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EBUR128_VERSION_MAJOR 1
#define EBUR128_VERSION_MINOR 2
#define EBUR128_VERSION_PATCH 4

enum mode {
  EBUR128_MODE_M = (1 << 0),
  EBUR128_MODE_S = (1 << 1) | EBUR128_MODE_M,
  EBUR128_MODE_I = (1 << 2) | EBUR128_MODE_M,
  EBUR128_MODE_LRA = (1 << 3) | EBUR128_MODE_S,
  EBUR128_MODE_SAMPLE_PEAK = (1 << 4) | EBUR128_MODE_M,
  EBUR128_MODE_TRUE_PEAK = (1 << 5) | EBUR128_MODE_M | EBUR128_MODE_SAMPLE_PEAK,
  EBUR128_MODE_HISTOGRAM = (1 << 6)
};

enum error {
  EBUR128_SUCCESS = 0,
  EBUR128_ERROR_NOMEM,
  EBUR128_ERROR_INVALID_MODE,
  EBUR128_ERROR_INVALID_CHANNEL_INDEX,
  EBUR128_ERROR_NO_CHANGE
};

struct ebur128_state_internal;

/* same leading layout as libebur128's: scan.c reads ->channels directly */
typedef struct {
  int mode;
  unsigned int channels;
  unsigned long samplerate;
  struct ebur128_state_internal *d;
} ebur128_state;

void ebur128_get_version(int *major, int *minor, int *patch);
ebur128_state *ebur128_init(unsigned int channels, unsigned long samplerate, int mode);
void ebur128_destroy(ebur128_state **st);

int ebur128_add_frames_short(ebur128_state *st, const short *src, size_t frames);
int ebur128_add_frames_float(ebur128_state *st, const float *src, size_t frames);

int ebur128_loudness_global(ebur128_state *st, double *out);
int ebur128_loudness_global_multiple(ebur128_state **sts, size_t size, double *out);
int ebur128_loudness_range(ebur128_state *st, double *out);
int ebur128_loudness_range_multiple(ebur128_state **sts, size_t size, double *out);
int ebur128_sample_peak(ebur128_state *st, unsigned int channel_number, double *out);
int ebur128_true_peak(ebur128_state *st, unsigned int channel_number, double *out);

/* extension: GPU used by states created afterwards (default 0) */
int loudscan_ebur128_set_device(int device);
/* extension: scans (plans) run so far in this process.  A session driven the way loudgain's main drives
 * scan.c (loudgain.c:299-340: every file scanned, then every result queried) costs exactly one. */
unsigned long long loudscan_ebur128_plan_count(void);

#ifdef __cplusplus
}
#endif
#endif
