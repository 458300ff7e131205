/* A C caller of the drop-in boundary, written the way loudgain.c's main drives scan.h
 * (/root/reference/src/loudgain.c:299-340,651-654) and the way scan.c drives libebur128
 * (/root/reference/src/scan.c:203,448,294-307,383-391,102).  Built with plain gcc against
 * libloudscan_hip.so: proves the C linkage of both headers.  Prints one JSON line.
 *   link_scan <frames> <seed>    (synthetic stereo 48 kHz S16 noise with a level step) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "loudscan.h"
#include <ebur128.h> /* include/compat/ebur128.h */

int main(int argc, char **argv) {
  size_t frames = argc > 1 ? (size_t)atol(argv[1]) : 48000 * 5;
  unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1;
  short *pcm = malloc(frames * 2 * sizeof(short));
  unsigned s = seed * 2654435761u + 1u;
  for (size_t i = 0; i < frames * 2; ++i) { /* LCG noise, -12 dBFS then -30 dBFS halfway */
    s = s * 1664525u + 1013904223u;
    double v = ((int)(s >> 16) - 32768) / 32768.0;
    pcm[i] = (short)lrint(v * (i < frames ? 8000.0 : 1000.0));
  }

  /* scan.h boundary, two files = one album */
  scan_init(2);
  scan_pcm_s16(pcm, frames, 2, 48000, 0);
  scan_pcm_s16(pcm, frames / 2, 2, 48000, 1);
  scan_result *r0 = scan_get_track_result(0, 0.0);
  scan_result *r1 = scan_get_track_result(1, 0.0);
  scan_set_album_result(r0, 0.0);
  scan_set_album_result(r1, 0.0);

  /* libebur128 boundary, the same two tracks */
  int maj, min, pat;
  ebur128_get_version(&maj, &min, &pat);
  ebur128_state *st[2];
  const int mode = EBUR128_MODE_S | EBUR128_MODE_I | EBUR128_MODE_LRA | EBUR128_MODE_SAMPLE_PEAK | EBUR128_MODE_TRUE_PEAK;
  st[0] = ebur128_init(2, 48000, mode);
  st[1] = ebur128_init(2, 48000, mode);
  if (!st[0] || !st[1]) return 2;
  for (size_t off = 0; off < frames; off += 1152) /* decoder-frame sized pieces */
    ebur128_add_frames_short(st[0], pcm + off * 2, frames - off < 1152 ? frames - off : 1152);
  ebur128_add_frames_short(st[1], pcm, frames / 2);
  double l0, l1, lra0, tp0 = 0.0, tmp, la, ra;
  if (ebur128_loudness_global(st[0], &l0) || ebur128_loudness_global(st[1], &l1) ||
      ebur128_loudness_range(st[0], &lra0))
    return 3;
  for (unsigned ch = 0; ch < st[0]->channels; ++ch) {
    if (ebur128_true_peak(st[0], ch, &tmp)) return 4;
    if (tmp > tp0) tp0 = tmp;
  }
  if (ebur128_loudness_global_multiple(st, 2, &la) || ebur128_loudness_range_multiple(st, 2, &ra)) return 5;

  printf("{\"version\": [%d, %d, %d], \"scan\": {\"l0\": %.12f, \"l1\": %.12f, \"lra0\": %.12f, \"peak0\": %.9f, "
         "\"album_l\": %.12f, \"album_lra\": %.12f, \"gain0\": %.12f, \"ref\": %.1f}, "
         "\"ebur128\": {\"l0\": %.12f, \"l1\": %.12f, \"lra0\": %.12f, \"peak0\": %.9f, \"album_l\": %.12f, "
         "\"album_lra\": %.12f}}\n",
         maj, min, pat, r0->track_loudness, r1->track_loudness, r0->track_loudness_range, r0->track_peak,
         r0->album_loudness, r0->album_loudness_range, r0->track_gain, r0->loudness_reference, l0, l1, lra0, tp0, la, ra);
  ebur128_destroy(&st[0]);
  ebur128_destroy(&st[1]);
  free(r0);
  free(r1);
  scan_deinit();
  free(pcm);
  return st[0] == NULL && st[1] == NULL ? 0 : 6;
}
