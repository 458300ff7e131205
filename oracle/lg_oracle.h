/*
 * lg_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Sequential, scalar restatement of the arithmetic behind loudgain's scan path:
 * /root/reference/src/scan.c calls nine libebur128 functions
 * (scan.c:102,203,294,297,303,371,383,388,448); libebur128 itself (pinned
 * >= 1.2.4 by /root/reference/debian/control:10 and README.md:329) is a
 * third-party dependency that is ABSENT from /root/reference, so its published
 * v1.2.4 algorithm is restated here following SURVEY.md Appendix A.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 * for this path and can be neither built nor run in this environment.  This
 * oracle is pinned instead by the ITU-R BS.1770 48 kHz coefficient table,
 * the synthesizable EBU Tech 3341 / 3342 cases and independent scipy/numpy
 * cross-checks (tests/test_oracle_kat.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libloudscan_hip.so) never links it.
 */
#ifndef LG_ORACLE_H
#define LG_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lgo_state lgo_state;

/* ebur128_init(channels, rate, S|I|LRA|SAMPLE_PEAK|TRUE_PEAK) as requested at
 * scan.c:203-207.  NULL on channels==0 || channels>64 || rate<16 || rate>2822400. */
lgo_state *lgo_create(unsigned channels, unsigned long rate);
void lgo_destroy(lgo_state *st);

/* ebur128_add_frames_short (scan.c:448): interleaved S16, scale 1/32768. */
int lgo_add_frames_short(lgo_state *st, const short *src, size_t frames);
/* ebur128_add_frames_float: interleaved f32, scale 1.0 (the f32 entry the GPU
 * bench drives; identical numbers when the floats sit on the S16 grid). */
int lgo_add_frames_float(lgo_state *st, const float *src, size_t frames);

/* ebur128_loudness_global / _global_multiple (scan.c:294,383). 0 = SUCCESS. */
int lgo_loudness_global(lgo_state *st, double *out);
int lgo_loudness_global_multiple(lgo_state **sts, size_t n, double *out);
/* ebur128_loudness_range / _range_multiple (scan.c:297,388). */
int lgo_loudness_range(lgo_state *st, double *out);
int lgo_loudness_range_multiple(lgo_state **sts, size_t n, double *out);
/* ebur128_true_peak (scan.c:303,371): max(true_peak, sample_peak) of channel. */
int lgo_true_peak(lgo_state *st, unsigned ch, double *out);
int lgo_sample_peak(lgo_state *st, unsigned ch, double *out);
unsigned lgo_channels(const lgo_state *st);

/* Introspection for parity tests (integer counts must match bit-exactly). */
size_t lgo_gating_block_count(const lgo_state *st);     /* blocks >= abs gate */
size_t lgo_shortterm_block_count(const lgo_state *st);  /* ST blocks >= abs gate */
const double *lgo_gating_blocks(const lgo_state *st);
const double *lgo_shortterm_blocks(const lgo_state *st);
/* two-pass gating detail: n/sum above abs gate, n/sum above relative gate */
int lgo_gating_detail(lgo_state **sts, size_t n, size_t *n_abs, double *sum_abs,
                      double *rel_threshold, size_t *n_rel, double *sum_rel);
/* filter design (App. A.1): merged b[5], a[5] for a rate */
void lgo_design_filter(unsigned long rate, double b[5], double a[5]);
/* interpolator design (App. A.5): returns factor (4, 2 or 0); coeff/index are
 * [factor][delay] row-major, count[factor]; delay returned through *delay. */
int lgo_design_interp(unsigned long rate, unsigned *delay, unsigned count[4],
                      unsigned index[4 * 25], double coeff[4 * 25]);

/* ---- scan.c-level restatement (scan.c:66-405) ------------------------- */
typedef struct {
  char *file;
  char *container;
  int codec_id;
  double track_gain;
  double track_peak;
  double track_loudness;
  double track_loudness_range;
  double album_gain;
  double album_peak;
  double album_loudness;
  double album_loudness_range;
  double loudness_reference;
} lgo_scan_result;

int lgo_scan_init(unsigned nb_files);
void lgo_scan_deinit(void);
/* replaces the FFmpeg half of scan_file (scan.c:139-272): RIFF/WAVE reader,
 * converts to interleaved S16 exactly as scan_frame does (scan.c:414). */
int lgo_scan_file(const char *file, unsigned index);
int lgo_scan_pcm_s16(const short *pcm, size_t frames, unsigned channels,
                     unsigned long rate, unsigned index);
int lgo_scan_pcm_f32(const float *pcm, size_t frames, unsigned channels,
                     unsigned long rate, unsigned index);
lgo_scan_result *lgo_scan_get_track_result(unsigned index, double pre_gain);
double lgo_scan_get_album_peak(void);
void lgo_scan_set_album_result(lgo_scan_result *result, double pre_gain);
int lgo_scan_album_has_different_codecs(void);
int lgo_scan_album_has_different_containers(void);
int lgo_scan_album_has_opus(void);
lgo_state *lgo_scan_state(unsigned index);

#ifdef __cplusplus
}
#endif
#endif
