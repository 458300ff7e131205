/*
 * lg_scan_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * PARITY UNPINNED by reference fixtures; see lg_oracle.h.
 *
 * Restates the scan module of the reference, /root/reference/src/scan.c:
 *   scan_init :66-96, scan_deinit :98-108, scan_file :110-273 (the FFmpeg
 *   demux/decode/swr half is replaced by a RIFF/WAVE reader that produces the
 *   same interleaved S16 the reference feeds at :414,:448),
 *   scan_get_track_result :275-330, album checks :332-357,
 *   scan_get_album_peak :359-378, scan_set_album_result :380-405.
 */
#include "lg_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* FFmpeg AVCodecID values (recalled; FFmpeg headers are absent here).  The
 * reference only tests equality and == OPUS on them (scan.c:310,344,353). */
#define LGO_CODEC_PCM_S16LE 0x10000
#define LGO_CODEC_PCM_U8 0x10005
#define LGO_CODEC_PCM_S32LE 0x10008
#define LGO_CODEC_PCM_S24LE 0x1000C
#define LGO_CODEC_PCM_F32LE 0x10015
#define LGO_CODEC_PCM_F64LE 0x10017
#define LGO_CODEC_OPUS 0x1503C

static lgo_state **g_states = NULL;
static int *g_codecs = NULL;
static char **g_files = NULL;
static char **g_containers = NULL;
static int g_nb = 0;

#define LUFS_TO_RG(L) (-18 - (L))

int lgo_scan_init(unsigned nb_files) {
  g_nb = (int)nb_files;
  g_states = (lgo_state **)calloc(nb_files ? nb_files : 1, sizeof(*g_states));
  g_files = (char **)calloc(nb_files ? nb_files : 1, sizeof(char *));
  g_containers = (char **)calloc(nb_files ? nb_files : 1, sizeof(char *));
  g_codecs = (int *)calloc(nb_files ? nb_files : 1, sizeof(int));
  if (!g_states || !g_files || !g_containers || !g_codecs) {
    fprintf(stderr, "OOM\n");
    exit(EXIT_FAILURE);
  }
  return 0;
}

void lgo_scan_deinit(void) {
  int i;
  for (i = 0; i < g_nb; i++) {
    lgo_destroy(g_states[i]);
    free(g_files[i]);
    free(g_containers[i]);
  }
  free(g_states); free(g_files); free(g_containers); free(g_codecs);
  g_states = NULL; g_files = NULL; g_containers = NULL; g_codecs = NULL;
  g_nb = 0;
}

static char *dupstr(const char *s) {
  size_t n = strlen(s) + 1;
  char *d = (char *)malloc(n);
  if (d) memcpy(d, s, n);
  return d;
}

static int begin_track(unsigned index, const char *name, const char *container, int codec,
                       unsigned channels, unsigned long rate) {
  if ((int)index >= g_nb) return -1;
  lgo_destroy(g_states[index]);
  free(g_files[index]);
  free(g_containers[index]);
  g_files[index] = dupstr(name);
  g_containers[index] = dupstr(container);
  g_codecs[index] = codec;
  g_states[index] = lgo_create(channels, rate);
  if (!g_states[index]) {
    fprintf(stderr, "Could not initialize EBU R128 scanner\n");
    exit(EXIT_FAILURE);
  }
  return 0;
}

int lgo_scan_pcm_s16(const short *pcm, size_t frames, unsigned channels, unsigned long rate,
                     unsigned index) {
  size_t off = 0;
  if (begin_track(index, "<pcm_s16>", "wav", LGO_CODEC_PCM_S16LE, channels, rate)) return -1;
  /* decoder-sized frames, as scan_frame sees them (scan.c:245); results are
   * invariant to this chunking */
  while (off < frames) {
    size_t n = frames - off < 4096 ? frames - off : 4096;
    lgo_add_frames_short(g_states[index], pcm + off * channels, n);
    off += n;
  }
  return 0;
}

int lgo_scan_pcm_f32(const float *pcm, size_t frames, unsigned channels, unsigned long rate,
                     unsigned index) {
  size_t off = 0;
  if (begin_track(index, "<pcm_f32>", "wav", LGO_CODEC_PCM_F32LE, channels, rate)) return -1;
  while (off < frames) {
    size_t n = frames - off < 4096 ? frames - off : 4096;
    lgo_add_frames_float(g_states[index], pcm + off * channels, n);
    off += n;
  }
  return 0;
}

/* ---- RIFF/WAVE -> interleaved S16 (what swr_convert yields at scan.c:442) */
static uint32_t rd32(const unsigned char *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

static short clip16(long v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

int lgo_scan_file(const char *file, unsigned index) {
  FILE *fp;
  unsigned char hdr[12], ck[8], fmt[40];
  unsigned fmt_tag = 0, channels = 0, bits = 0, block_align = 0;
  unsigned long rate = 0;
  int have_fmt = 0, codec = 0;
  if ((int)index >= g_nb) return -1;
  fp = fopen(file, "rb");
  if (!fp) {
    fprintf(stderr, "Could not open input: %s\n", file);
    exit(EXIT_FAILURE);
  }
  if (fread(hdr, 1, 12, fp) != 12 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4)) {
    fprintf(stderr, "Could not find stream info: %s\n", file);
    exit(EXIT_FAILURE);
  }
  while (fread(ck, 1, 8, fp) == 8) {
    uint32_t sz = rd32(ck + 4);
    if (!memcmp(ck, "fmt ", 4)) {
      uint32_t n = sz < sizeof(fmt) ? sz : (uint32_t)sizeof(fmt);
      memset(fmt, 0, sizeof(fmt));
      if (fread(fmt, 1, n, fp) != n) break;
      if (sz > n) fseek(fp, (long)(sz - n), SEEK_CUR);
      if (sz & 1) fseek(fp, 1, SEEK_CUR);
      fmt_tag = rd16(fmt);
      channels = rd16(fmt + 2);
      rate = rd32(fmt + 4);
      block_align = rd16(fmt + 12);
      bits = rd16(fmt + 14);
      if (fmt_tag == 0xFFFE && sz >= 26) fmt_tag = rd16(fmt + 24); /* EXTENSIBLE sub-format */
      have_fmt = 1;
    } else if (!memcmp(ck, "data", 4)) {
      size_t bps, total_frames, done = 0;
      unsigned char *buf;
      short *s16;
      const size_t CH = 4096;
      if (!have_fmt || !channels || !block_align) break;
      bps = bits / 8;
      if (fmt_tag == 1 && bits == 16) codec = LGO_CODEC_PCM_S16LE;
      else if (fmt_tag == 1 && bits == 8) codec = LGO_CODEC_PCM_U8;
      else if (fmt_tag == 1 && bits == 24) codec = LGO_CODEC_PCM_S24LE;
      else if (fmt_tag == 1 && bits == 32) codec = LGO_CODEC_PCM_S32LE;
      else if (fmt_tag == 3 && bits == 32) codec = LGO_CODEC_PCM_F32LE;
      else if (fmt_tag == 3 && bits == 64) codec = LGO_CODEC_PCM_F64LE;
      else {
        fprintf(stderr, "Could not find the codec: %s\n", file);
        exit(EXIT_FAILURE);
      }
      begin_track(index, file, "wav", codec, channels, rate);
      total_frames = sz / block_align;
      buf = (unsigned char *)malloc(CH * block_align);
      s16 = (short *)malloc(CH * channels * sizeof(short));
      while (done < total_frames) {
        size_t n = total_frames - done < CH ? total_frames - done : CH, i, got;
        got = fread(buf, block_align, n, fp);
        if (got == 0) break; /* truncated file: silently shortened, as scan.c:229-240 */
        n = got;
        for (i = 0; i < n * channels; i++) {
          const unsigned char *p = buf + i * bps;
          switch (codec) {
            case LGO_CODEC_PCM_S16LE: s16[i] = (short)rd16(p); break;
            case LGO_CODEC_PCM_U8: s16[i] = (short)(((int)p[0] - 0x80) * 256); break;
            case LGO_CODEC_PCM_S24LE: { /* decoder: <<8 into S32; swr: >>16 */
              int32_t v = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24);
              s16[i] = (short)(v >> 16);
            } break;
            case LGO_CODEC_PCM_S32LE: s16[i] = (short)((int32_t)rd32(p) >> 16); break;
            case LGO_CODEC_PCM_F32LE: {
              float f; uint32_t u = rd32(p);
              memcpy(&f, &u, 4);
              s16[i] = clip16(lrintf(f * 32768.0f));
            } break;
            default: {
              double d; uint64_t u = (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32);
              memcpy(&d, &u, 8);
              s16[i] = clip16(lrint(d * 32768.0));
            } break;
          }
        }
        lgo_add_frames_short(g_states[index], s16, n);
        done += n;
      }
      free(buf);
      free(s16);
      fclose(fp);
      return 0;
    } else {
      fseek(fp, (long)(sz + (sz & 1)), SEEK_CUR);
    }
  }
  fprintf(stderr, "Could not find audio stream: %s\n", file);
  exit(EXIT_FAILURE);
}

lgo_state *lgo_scan_state(unsigned index) { return (int)index < g_nb ? g_states[index] : NULL; }

lgo_scan_result *lgo_scan_get_track_result(unsigned index, double pre_gain) {
  unsigned ch;
  double global, range, peak = 0.0;
  lgo_scan_result *r;
  lgo_state *st;
  if ((int)index >= g_nb) {
    fprintf(stderr, "Index too high\n");
    return NULL;
  }
  r = (lgo_scan_result *)malloc(sizeof(*r));
  st = g_states[index];
  if (lgo_loudness_global(st, &global) != 0) global = 0.0;
  if (lgo_loudness_range(st, &range) != 0) range = 0.0;
  for (ch = 0; ch < lgo_channels(st); ch++) {
    double tmp;
    if (lgo_true_peak(st, ch, &tmp) != 0) continue;
    peak = peak > tmp ? peak : tmp;
  }
  if (g_codecs[index] == LGO_CODEC_OPUS) pre_gain = pre_gain - 5.0f;
  r->file = g_files[index];
  r->container = g_containers[index];
  r->codec_id = g_codecs[index];
  r->track_gain = LUFS_TO_RG(global) + pre_gain;
  r->track_peak = peak;
  r->track_loudness = global;
  r->track_loudness_range = range;
  r->album_gain = 0.f;
  r->album_peak = 0.f;
  r->album_loudness = 0.f;
  r->album_loudness_range = 0.f;
  r->loudness_reference = LUFS_TO_RG(-pre_gain);
  return r;
}

int lgo_scan_album_has_different_containers(void) {
  int i;
  for (i = 0; i < g_nb; i++)
    if (strcmp(g_containers[0], g_containers[i])) return 1;
  return 0;
}

int lgo_scan_album_has_different_codecs(void) {
  int i;
  for (i = 0; i < g_nb; i++)
    if (g_codecs[0] != g_codecs[i]) return 1;
  return 0;
}

int lgo_scan_album_has_opus(void) {
  int i;
  for (i = 0; i < g_nb; i++)
    if (g_codecs[i] == LGO_CODEC_OPUS) return 1;
  return 0;
}

double lgo_scan_get_album_peak(void) {
  double peak = 0.0;
  int i;
  unsigned ch;
  for (i = 0; i < g_nb; i++) {
    lgo_state *st = g_states[i];
    for (ch = 0; ch < lgo_channels(st); ch++) {
      double tmp;
      if (lgo_true_peak(st, ch, &tmp) != 0) continue;
      peak = peak > tmp ? peak : tmp;
    }
  }
  return peak;
}

void lgo_scan_set_album_result(lgo_scan_result *r, double pre_gain) {
  double global, range;
  if (lgo_loudness_global_multiple(g_states, (size_t)g_nb, &global) != 0) global = 0.0;
  if (lgo_loudness_range_multiple(g_states, (size_t)g_nb, &range) != 0) range = 0.0;
  if (lgo_scan_album_has_opus()) pre_gain = pre_gain - 5.0f;
  r->album_gain = LUFS_TO_RG(global) + pre_gain;
  r->album_peak = lgo_scan_get_album_peak();
  r->album_loudness = global;
  r->album_loudness_range = range;
}
