/*
 * lg_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * PARITY UNPINNED by reference fixtures (the reference has none); see
 * lg_oracle.h for what pins this file instead.
 *
 * Restates libebur128 v1.2.4 (third-party, absent from /root/reference; pinned
 * by /root/reference/debian/control:10) as it is driven from
 * /root/reference/src/scan.c:203-207 (all five modes on, no histogram mode)
 * and scan.c:448 (interleaved S16).  Section tags A.1 .. A.7 refer to SURVEY.md
 * Appendix A.  Scalar, single-threaded, double precision -- on purpose the
 * same shape as the library so that it also serves as the timed CPU baseline.
 */
#include "lg_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846264338327950288
#endif

enum { CH_UNUSED = 0, CH_LEFT, CH_RIGHT, CH_CENTER, CH_LS, CH_RS, CH_DUAL_MONO };

typedef struct {
  double *v;
  size_t n, cap;
} dlist;

static int dlist_push(dlist *l, double x) {
  if (l->n == l->cap) {
    size_t nc = l->cap ? l->cap * 2 : 1024;
    double *nv = (double *)realloc(l->v, nc * sizeof(double));
    if (!nv) return 1;
    l->v = nv;
    l->cap = nc;
  }
  l->v[l->n++] = x;
  return 0;
}

typedef struct {
  unsigned factor, delay, taps;
  unsigned count[4];
  unsigned index[4][25];
  double coeff[4][25];
  float *z; /* [channels][delay] */
  unsigned zi;
} interp_t;

struct lgo_state {
  unsigned channels;
  unsigned long rate;
  int *chmap;
  double b[5], a[5];
  double v[5][5];
  size_t s100;
  double *ring;       /* interleaved filtered samples */
  size_t ring_frames; /* A.3: 3 s rounded up to a multiple of s100 */
  size_t ring_index;  /* in samples (frames*channels), like audio_data_index */
  size_t needed;
  size_t st_counter;
  dlist gate, st;
  double *sample_peak, *true_peak, *prev_sample_peak, *prev_true_peak;
  int has_interp;
  interp_t ip;
  float *rs_in, *rs_out;
  size_t rs_frames;
  double abs_gate;
};

/* ---- A.1 filter design ------------------------------------------------ */
void lgo_design_filter(unsigned long rate, double b[5], double a[5]) {
  double f0 = 1681.974450955533, G = 3.999843853973347, Q = 0.7071752369554196;
  double K = tan(M_PI * f0 / (double)rate);
  double Vh = pow(10.0, G / 20.0);
  double Vb = pow(Vh, 0.4996667741545416);
  double pb[3], pa[3] = {1.0, 0.0, 0.0}, rb[3] = {1.0, -2.0, 1.0}, ra[3] = {1.0, 0.0, 0.0};
  double a0 = 1.0 + K / Q + K * K;
  pb[0] = (Vh + Vb * K / Q + K * K) / a0;
  pb[1] = 2.0 * (K * K - Vh) / a0;
  pb[2] = (Vh - Vb * K / Q + K * K) / a0;
  pa[1] = 2.0 * (K * K - 1.0) / a0;
  pa[2] = (1.0 - K / Q + K * K) / a0;

  f0 = 38.13547087602444;
  Q = 0.5003270373238773;
  K = tan(M_PI * f0 / (double)rate);
  ra[1] = 2.0 * (K * K - 1.0) / (1.0 + K / Q + K * K);
  ra[2] = (1.0 - K / Q + K * K) / (1.0 + K / Q + K * K);

  b[0] = pb[0] * rb[0];
  b[1] = pb[0] * rb[1] + pb[1] * rb[0];
  b[2] = pb[0] * rb[2] + pb[1] * rb[1] + pb[2] * rb[0];
  b[3] = pb[1] * rb[2] + pb[2] * rb[1];
  b[4] = pb[2] * rb[2];
  a[0] = pa[0] * ra[0];
  a[1] = pa[0] * ra[1] + pa[1] * ra[0];
  a[2] = pa[0] * ra[2] + pa[1] * ra[1] + pa[2] * ra[0];
  a[3] = pa[1] * ra[2] + pa[2] * ra[1];
  a[4] = pa[2] * ra[2];
}

/* ---- A.5 interpolator design ----------------------------------------- */
static unsigned interp_factor_for(unsigned long rate) {
  if (rate < 96000) return 4;
  if (rate < 192000) return 2;
  return 0;
}

static void interp_design(interp_t *ip, unsigned factor) {
  const unsigned taps = 49;
  unsigned j;
  memset(ip->count, 0, sizeof(ip->count));
  ip->factor = factor;
  ip->taps = taps;
  ip->delay = (taps + factor - 1) / factor;
  for (j = 0; j < taps; j++) {
    double m = (double)j - (double)(taps - 1) / 2.0;
    double c = 1.0;
    if (fabs(m) > 0.000001) c = sin(m * M_PI / (double)factor) / (m * M_PI / (double)factor);
    c *= 0.5 * (1.0 - cos(2.0 * M_PI * (double)j / (double)(taps - 1)));
    if (fabs(c) > 0.000001) {
      unsigned f = j % factor;
      unsigned t = ip->count[f]++;
      ip->coeff[f][t] = c;
      ip->index[f][t] = j / factor;
    }
  }
}

int lgo_design_interp(unsigned long rate, unsigned *delay, unsigned count[4],
                      unsigned index[4 * 25], double coeff[4 * 25]) {
  interp_t ip;
  unsigned f, t, factor = interp_factor_for(rate);
  memset(&ip, 0, sizeof(ip));
  if (!factor) {
    *delay = 0;
    return 0;
  }
  interp_design(&ip, factor);
  *delay = ip.delay;
  for (f = 0; f < 4; f++) {
    count[f] = f < factor ? ip.count[f] : 0;
    for (t = 0; t < 25; t++) {
      index[f * 25 + t] = ip.index[f][t];
      coeff[f * 25 + t] = ip.coeff[f][t];
    }
  }
  return (int)factor;
}

/* per input frame and channel: push into the circular delay line, emit
 * `factor` outputs (double accumulate, cast to float) -- A.5 */
static void interp_run(interp_t *ip, unsigned channels, size_t frames, const float *in,
                       float *out) {
  size_t frame;
  unsigned chan, f, t;
  for (frame = 0; frame < frames; frame++) {
    for (chan = 0; chan < channels; chan++) {
      float *zc = ip->z + (size_t)chan * ip->delay;
      float *outp = out + frame * ip->factor * channels + chan;
      zc[ip->zi] = *in++;
      for (f = 0; f < ip->factor; f++) {
        double acc = 0.0;
        for (t = 0; t < ip->count[f]; t++) {
          int i = (int)ip->zi - (int)ip->index[f][t];
          if (i < 0) i += (int)ip->delay;
          acc += (double)zc[i] * ip->coeff[f][t];
        }
        *outp = (float)acc;
        outp += channels;
      }
    }
    ip->zi++;
    if (ip->zi == ip->delay) ip->zi = 0;
  }
}

/* ---- init / destroy ---------------------------------------------------- */
static void default_channel_map(int *m, unsigned channels) {
  unsigned i;
  if (channels == 4) {
    m[0] = CH_LEFT; m[1] = CH_RIGHT; m[2] = CH_LS; m[3] = CH_RS;
  } else if (channels == 5) {
    m[0] = CH_LEFT; m[1] = CH_RIGHT; m[2] = CH_CENTER; m[3] = CH_LS; m[4] = CH_RS;
  } else {
    for (i = 0; i < channels; i++) {
      switch (i) {
        case 0: m[i] = CH_LEFT; break;
        case 1: m[i] = CH_RIGHT; break;
        case 2: m[i] = CH_CENTER; break;
        case 3: m[i] = CH_UNUSED; break;
        case 4: m[i] = CH_LS; break;
        case 5: m[i] = CH_RS; break;
        default: m[i] = CH_UNUSED; break;
      }
    }
  }
}

lgo_state *lgo_create(unsigned channels, unsigned long rate) {
  lgo_state *st;
  if (channels == 0 || channels > 64) return NULL;
  if (rate < 16 || rate > 2822400) return NULL;
  st = (lgo_state *)calloc(1, sizeof(*st));
  if (!st) return NULL;
  st->channels = channels;
  st->rate = rate;
  st->chmap = (int *)calloc(channels, sizeof(int));
  st->sample_peak = (double *)calloc(channels, sizeof(double));
  st->true_peak = (double *)calloc(channels, sizeof(double));
  st->prev_sample_peak = (double *)calloc(channels, sizeof(double));
  st->prev_true_peak = (double *)calloc(channels, sizeof(double));
  default_channel_map(st->chmap, channels);
  st->s100 = (rate + 5) / 10;
  st->ring_frames = rate * 3;
  if (st->ring_frames % st->s100)
    st->ring_frames = st->ring_frames + st->s100 - (st->ring_frames % st->s100);
  st->ring = (double *)calloc(st->ring_frames * channels, sizeof(double));
  lgo_design_filter(rate, st->b, st->a);
  st->needed = st->s100 * 4;
  st->abs_gate = pow(10.0, (-70.0 + 0.691) / 10.0);
  st->rs_frames = st->s100 * 4;
  {
    unsigned factor = interp_factor_for(rate);
    if (factor) {
      st->has_interp = 1;
      interp_design(&st->ip, factor);
      st->ip.z = (float *)calloc((size_t)channels * st->ip.delay, sizeof(float));
      st->rs_in = (float *)calloc(st->rs_frames * channels, sizeof(float));
      st->rs_out = (float *)calloc(st->rs_frames * channels * factor, sizeof(float));
    }
  }
  return st;
}

void lgo_destroy(lgo_state *st) {
  if (!st) return;
  free(st->chmap); free(st->sample_peak); free(st->true_peak);
  free(st->prev_sample_peak); free(st->prev_true_peak);
  free(st->ring); free(st->gate.v); free(st->st.v);
  free(st->ip.z); free(st->rs_in); free(st->rs_out);
  free(st);
}

unsigned lgo_channels(const lgo_state *st) { return st->channels; }

/* ---- A.2 / E3 / E4: one chunk (<= needed frames) through peak scan,
 * true-peak interpolator and the merged 4th-order DF-II K-weighting ------- */
#define FILTER_BODY(TYPE, SCALE)                                                      \
  const double scale = (SCALE);                                                       \
  size_t i;                                                                           \
  unsigned c;                                                                         \
  double *ring = st->ring + st->ring_index;                                           \
  for (c = 0; c < st->channels; c++) {                                                \
    double mx = 0.0;                                                                  \
    for (i = 0; i < frames; i++) {                                                    \
      double s = (double)src[i * st->channels + c];                                   \
      if (s > mx) mx = s;                                                             \
      else if (-s > mx) mx = -s;                                                      \
    }                                                                                 \
    mx /= scale;                                                                      \
    if (mx > st->prev_sample_peak[c]) st->prev_sample_peak[c] = mx;                   \
  }                                                                                   \
  if (st->has_interp) {                                                               \
    for (c = 0; c < st->channels; c++)                                                \
      for (i = 0; i < frames; i++)                                                    \
        st->rs_in[i * st->channels + c] =                                             \
            (float)((double)src[i * st->channels + c] / scale);                       \
    interp_run(&st->ip, st->channels, frames, st->rs_in, st->rs_out);                 \
    for (c = 0; c < st->channels; c++) {                                              \
      size_t nout = frames * st->ip.factor;                                           \
      for (i = 0; i < nout; i++) {                                                    \
        double o = (double)st->rs_out[i * st->channels + c];                          \
        if (o > st->prev_true_peak[c]) st->prev_true_peak[c] = o;                     \
        else if (-o > st->prev_true_peak[c]) st->prev_true_peak[c] = -o;              \
      }                                                                               \
    }                                                                                 \
  }                                                                                   \
  for (c = 0; c < st->channels; c++) {                                                \
    int ci = st->chmap[c] - 1;                                                        \
    double *v;                                                                        \
    if (ci < 0) continue;                                                             \
    if (ci == CH_DUAL_MONO - 1) ci = 0;                                               \
    v = st->v[ci];                                                                    \
    for (i = 0; i < frames; i++) {                                                    \
      v[0] = (double)src[i * st->channels + c] / scale - st->a[1] * v[1] -            \
             st->a[2] * v[2] - st->a[3] * v[3] - st->a[4] * v[4];                     \
      ring[i * st->channels + c] = st->b[0] * v[0] + st->b[1] * v[1] +                \
                                   st->b[2] * v[2] + st->b[3] * v[3] + st->b[4] * v[4]; \
      v[4] = v[3]; v[3] = v[2]; v[2] = v[1]; v[1] = v[0];                             \
    }                                                                                 \
    v[4] = fabs(v[4]) < DBL_MIN ? 0.0 : v[4];                                         \
    v[3] = fabs(v[3]) < DBL_MIN ? 0.0 : v[3];                                         \
    v[2] = fabs(v[2]) < DBL_MIN ? 0.0 : v[2];                                         \
    v[1] = fabs(v[1]) < DBL_MIN ? 0.0 : v[1];                                         \
  }

static void filter_short(lgo_state *st, const short *src, size_t frames) {
  FILTER_BODY(short, 32768.0)
}
static void filter_float(lgo_state *st, const float *src, size_t frames) {
  FILTER_BODY(float, 1.0)
}

/* ---- A.4 / E5 / E6: block energy over the last `fpb` frames of the ring - */
static double block_energy(const lgo_state *st, size_t fpb) {
  double sum = 0.0;
  unsigned c;
  size_t i;
  const unsigned nch = st->channels;
  for (c = 0; c < nch; c++) {
    double cs = 0.0;
    if (st->chmap[c] == CH_UNUSED) continue;
    if (st->ring_index < fpb * nch) {
      for (i = 0; i < st->ring_index / nch; i++)
        cs += st->ring[i * nch + c] * st->ring[i * nch + c];
      for (i = st->ring_frames - (fpb - st->ring_index / nch); i < st->ring_frames; i++)
        cs += st->ring[i * nch + c] * st->ring[i * nch + c];
    } else {
      for (i = st->ring_index / nch - fpb; i < st->ring_index / nch; i++)
        cs += st->ring[i * nch + c] * st->ring[i * nch + c];
    }
    if (st->chmap[c] == CH_LS || st->chmap[c] == CH_RS) cs *= 1.41;
    else if (st->chmap[c] == CH_DUAL_MONO) cs *= 2.0;
    sum += cs;
  }
  return sum / (double)fpb;
}

/* ---- A.3 / E2: chunk scheduler ----------------------------------------- */
#define ADD_FRAMES_BODY(FILTER)                                                   \
  size_t off = 0;                                                                 \
  unsigned c;                                                                     \
  for (c = 0; c < st->channels; c++) {                                            \
    st->prev_sample_peak[c] = 0.0;                                                \
    st->prev_true_peak[c] = 0.0;                                                  \
  }                                                                               \
  while (frames > 0) {                                                            \
    if (frames >= st->needed) {                                                   \
      double e;                                                                   \
      FILTER(st, src + off, st->needed);                                          \
      off += st->needed * st->channels;                                           \
      frames -= st->needed;                                                       \
      st->ring_index += st->needed * st->channels;                                \
      e = block_energy(st, st->s100 * 4);                                         \
      if (e >= st->abs_gate && dlist_push(&st->gate, e)) return 1;                \
      st->st_counter += st->needed;                                               \
      if (st->st_counter == st->s100 * 30) {                                      \
        e = block_energy(st, st->s100 * 30);                                      \
        if (e >= st->abs_gate && dlist_push(&st->st, e)) return 1;                \
        st->st_counter = st->s100 * 20;                                           \
      }                                                                           \
      if (st->ring_index == st->ring_frames * st->channels) st->ring_index = 0;   \
      st->needed = st->s100;                                                      \
    } else {                                                                      \
      FILTER(st, src + off, frames);                                              \
      st->ring_index += frames * st->channels;                                    \
      st->st_counter += frames;                                                   \
      st->needed -= frames;                                                       \
      frames = 0;                                                                 \
    }                                                                             \
  }                                                                               \
  for (c = 0; c < st->channels; c++) {                                            \
    if (st->prev_sample_peak[c] > st->sample_peak[c])                             \
      st->sample_peak[c] = st->prev_sample_peak[c];                               \
    if (st->prev_true_peak[c] > st->true_peak[c])                                 \
      st->true_peak[c] = st->prev_true_peak[c];                                   \
  }                                                                               \
  return 0;

int lgo_add_frames_short(lgo_state *st, const short *src, size_t frames) {
  ADD_FRAMES_BODY(filter_short)
}
int lgo_add_frames_float(lgo_state *st, const float *src, size_t frames) {
  ADD_FRAMES_BODY(filter_float)
}

/* ---- A.6 / E7: integrated loudness over one or many states ------------- */
static double energy_to_loudness(double e) { return 10.0 * (log(e) / log(10.0)) - 0.691; }

int lgo_gating_detail(lgo_state **sts, size_t n, size_t *n_abs, double *sum_abs,
                      double *rel_threshold, size_t *n_rel, double *sum_rel) {
  size_t i, j, cnt = 0;
  double thr = 0.0, acc = 0.0;
  *n_abs = 0; *sum_abs = 0.0; *rel_threshold = 0.0; *n_rel = 0; *sum_rel = 0.0;
  for (i = 0; i < n; i++)
    for (j = 0; j < sts[i]->gate.n; j++) {
      ++cnt;
      thr += sts[i]->gate.v[j];
    }
  *n_abs = cnt;
  *sum_abs = thr;
  if (!cnt) return 0;
  thr /= (double)cnt;
  thr *= pow(10.0, -10.0 / 10.0);
  *rel_threshold = thr;
  cnt = 0;
  for (i = 0; i < n; i++)
    for (j = 0; j < sts[i]->gate.n; j++)
      if (sts[i]->gate.v[j] >= thr) {
        ++cnt;
        acc += sts[i]->gate.v[j];
      }
  *n_rel = cnt;
  *sum_rel = acc;
  return 0;
}

int lgo_loudness_global_multiple(lgo_state **sts, size_t n, double *out) {
  size_t n_abs, n_rel;
  double sum_abs, thr, sum_rel;
  lgo_gating_detail(sts, n, &n_abs, &sum_abs, &thr, &n_rel, &sum_rel);
  if (!n_abs || !n_rel) {
    *out = -HUGE_VAL;
    return 0;
  }
  *out = energy_to_loudness(sum_rel / (double)n_rel);
  return 0;
}

int lgo_loudness_global(lgo_state *st, double *out) {
  return lgo_loudness_global_multiple(&st, 1, out);
}

/* ---- A.7 / E8: loudness range ----------------------------------------- */
static int dcmp(const void *p, const void *q) {
  double a = *(const double *)p, b = *(const double *)q;
  return (a > b) - (a < b);
}

int lgo_loudness_range_multiple(lgo_state **sts, size_t n, double *out) {
  size_t i, total = 0, k = 0, m;
  double *vec, *rel, power = 0.0, integrated;
  for (i = 0; i < n; i++) total += sts[i]->st.n;
  if (!total) {
    *out = 0.0;
    return 0;
  }
  vec = (double *)malloc(total * sizeof(double));
  if (!vec) return 1;
  for (i = 0; i < n; i++) {
    memcpy(vec + k, sts[i]->st.v, sts[i]->st.n * sizeof(double));
    k += sts[i]->st.n;
  }
  qsort(vec, total, sizeof(double), dcmp);
  for (i = 0; i < total; i++) power += vec[i];
  power /= (double)total;
  integrated = pow(10.0, -20.0 / 10.0) * power;
  rel = vec;
  m = total;
  while (m > 0 && *rel < integrated) {
    ++rel;
    --m;
  }
  if (m) {
    double h = rel[(size_t)((double)(m - 1) * 0.95 + 0.5)];
    double l = rel[(size_t)((double)(m - 1) * 0.1 + 0.5)];
    *out = energy_to_loudness(h) - energy_to_loudness(l);
  } else {
    *out = 0.0;
  }
  free(vec);
  return 0;
}

int lgo_loudness_range(lgo_state *st, double *out) {
  return lgo_loudness_range_multiple(&st, 1, out);
}

/* ---- E9: peaks ---------------------------------------------------------- */
int lgo_true_peak(lgo_state *st, unsigned ch, double *out) {
  if (ch >= st->channels) return 1;
  *out = st->true_peak[ch] > st->sample_peak[ch] ? st->true_peak[ch] : st->sample_peak[ch];
  return 0;
}

int lgo_sample_peak(lgo_state *st, unsigned ch, double *out) {
  if (ch >= st->channels) return 1;
  *out = st->sample_peak[ch];
  return 0;
}

size_t lgo_gating_block_count(const lgo_state *st) { return st->gate.n; }
size_t lgo_shortterm_block_count(const lgo_state *st) { return st->st.n; }
const double *lgo_gating_blocks(const lgo_state *st) { return st->gate.v; }
const double *lgo_shortterm_blocks(const lgo_state *st) { return st->st.v; }
