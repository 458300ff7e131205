// lgd_engine.cpp -- host side of the device-level C ABI (include/loudscan_device.h).
//
// Plans a batch of tracks into wavefront-sized segments, owns the HBM
// workspace, enqueues the kernels of lgd_kernels.hip and copies results out.
// Replaces what /root/reference/src/scan.c gets from libebur128:
// ebur128_init (scan.c:203) -> lgd_plan, the add_frames loop (scan.c:225-250,
// :448) -> lgd_execute, the loudness/range/peak queries (scan.c:294-307,
// :383-391, :359-378) -> lgd_fetch.  There is no CPU fallback anywhere here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <tuple>
#include <string>
#include <vector>

#include "../../include/loudscan_device.h"
#include "lgd_internal.h"

extern "C" const int lgd_chunk_table[];
extern "C" size_t lgd_scan_lds_bytes(int chunk, int nch, int tp, int generic);
extern "C" hipError_t lgd_launch_scan(int chunk, int nch, int tp, int generic, int s16, const LgdSeg *segs,
                                      int n_seg, hipStream_t s);
extern "C" hipError_t lgd_launch_peak_reduce(const LgdTrackMeta *meta, int n_tracks, const float *peaks,
                                             float *hint, hipStream_t s);
extern "C" int lgd_tp_instance(int chunk, int *u_out, int *ns_out);
extern "C" int lgd_tp_magics(int chunk, int nch_wg, int tp, unsigned *magic_nch, unsigned *magic_ns, unsigned *magic_c);
extern "C" hipError_t lgd_launch_tp(int u, int tp, int ns, int s16, const LgdSeg *segs, int n_seg, int rows_max,
                                    hipStream_t s);
extern "C" hipError_t lgd_launch_track_epilogue(const LgdSlice *slices, int n_slices,
                                                const LgdTrackMeta *meta, int n_tracks,
                                                const double *E, double *Z, double *st,
                                                const float *peaks, double *p1, double *p2,
                                                double *pmax_s, double *res, unsigned *done_count,
                                                double abs_gate, double rel_factor, double minus20, int do_tp,
                                                int skip_big, hipStream_t s);
extern "C" size_t lgd_lra_pick_bytes(void);
extern "C" hipError_t lgd_launch_lra(const void *ranges, int n_ranges, const double *st_base,
                                     double minus20, const int *big_idx, int n_big, unsigned *hist,
                                     double *part, void *picks, double *cand, const long long *cand_off,
                                     int small_too, hipStream_t s);
extern "C" hipError_t lgd_launch_album_part1(const double *res, const LgdAlbumMeta *albums,
                                             int n_albums, double *heads, hipStream_t s);
extern "C" hipError_t lgd_launch_album_stage2(const LgdSlice *slices, int n_slices,
                                              const LgdTrackMeta *meta, const double *Z,
                                              const double *p1, double *p2a,
                                              const LgdAlbumMeta *albums, int n_albums,
                                              double *heads_all, int world, long long rec_stride,
                                              double *part1, double *rec2, double abs_gate,
                                              double rel_factor, hipStream_t s);
extern "C" hipError_t lgd_launch_album_final(const double *part1, const double *rec2_all, int world,
                                             int n_albums, double rel_factor, double *album,
                                             hipStream_t s);

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                    \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(e_ == hipErrorOutOfMemory ? LGD_ENOMEM : LGD_EHIP, "%s: %s", #expr,   \
                  hipGetErrorString(e_));                                               \
  } while (0)

// ------------------------------------------------------------ filter design --
// BS.1770 K-weighting as libebur128 designs it for an arbitrary rate
// (SURVEY.md A.1): high-shelf biquad pb/pa and RLB high-pass (1,-2,1)/ra.  The
// library merges them into one 4th-order filter; the kernels keep the chain.
static void design_kfilter(double rate, double pb[3], double pa[2], double ra[2]) {
  const double pi = 3.14159265358979323846264338327950288;
  double K = std::tan(pi * 1681.974450955533 / rate);
  const double Q1 = 0.7071752369554196;
  const double Vh = std::pow(10.0, 3.999843853973347 / 20.0);
  const double Vb = std::pow(Vh, 0.4996667741545416);
  const double d1 = 1.0 + K / Q1 + K * K;
  pb[0] = (Vh + Vb * K / Q1 + K * K) / d1;
  pb[1] = 2.0 * (K * K - Vh) / d1;
  pb[2] = (Vh - Vb * K / Q1 + K * K) / d1;
  pa[0] = 2.0 * (K * K - 1.0) / d1;
  pa[1] = (1.0 - K / Q1 + K * K) / d1;
  K = std::tan(pi * 38.13547087602444 / rate);
  const double Q2 = 0.5003270373238773;
  const double d2 = 1.0 + K / Q2 + K * K;
  ra[0] = 2.0 * (K * K - 1.0) / d2;
  ra[1] = (1.0 - K / Q2 + K * K) / d2;
}

typedef long double ld;

static void mat4_mul(const ld *X, const ld *Y, ld *Z) {
  ld T[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      ld s = 0;
      for (int k = 0; k < 4; ++k) s += X[4 * r + k] * Y[4 * k + c];
      T[4 * r + c] = s;
    }
  memcpy(Z, T, sizeof(T));
}

// Scan-basis constants.  Chain state (q1, q2, p1, p2) steps as
//   q0 = w - ra1 q1 - ra2 q2,  p0 = q0 - pa1 p1 - pa2 p2  ->  (q0, q1, p0, p1)
// i.e. s <- Ac s + Bc w.  With T = [[1,0,0,0],[al,-al*be,0,0],[0,0,ga,0],[0,0,0,ga]]
// the kernels carry T s; M = T Ac T^-1 is well conditioned (all powers <= ~10),
// so its powers are formed by plain repeated multiplication in long double.
static void design_scan_basis(LgdFilt &F, int chunk) {
  const ld ra1 = F.ra[0], ra2 = F.ra[1], pa1 = F.pa[0], pa2 = F.pa[1];
  const ld r = sqrtl(ra2);                 // radius of the (nearly double) RLB pole
  const ld al = 1.0L / (1.0L - r), be = r;
  const ld dc = 1.0L / (1.0L + pa1 + pa2), ga = 1.0L / dc;
  F.alpha = (double)al; F.beta = (double)be;
  F.inv_alpha = (double)(1.0L / al); F.inv_beta = (double)(1.0L / be);
  F.gamma = (double)ga; F.dc = (double)dc;
  // use the rounded doubles the kernel will really apply, so T and T^-1 match it
  const ld alr = F.alpha, ber = F.beta, gar = F.gamma;
  ld Ac[16] = {-ra1, -ra2, 0, 0, 1, 0, 0, 0, -ra1, -ra2, -pa1, -pa2, 0, 0, 1, 0};
  ld T[16] = {1, 0, 0, 0, alr, -alr * ber, 0, 0, 0, 0, gar, 0, 0, 0, 0, gar};
  ld Ti[16] = {1, 0, 0, 0, 1 / ber, -1 / (alr * ber), 0, 0, 0, 0, 1 / gar, 0, 0, 0, 0, 1 / gar};
  ld M[16], Mk[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  mat4_mul(T, Ac, M);
  mat4_mul(M, Ti, M);
  const ld TB[4] = {1, alr, gar, 0};  // T * (1, 0, 1, 0)
  // powers of M up to the chunk length; pick off what the kernel needs:
  // gC[0] = M^(C-1) T Bc, gC[1] = M^(C-2) T Bc (Mk == M^i in the loop)
  for (int i = 0; i <= chunk; ++i) {
    if (i == chunk - 1 || i == chunk - 2) {
      for (int rr = 0; rr < 4; ++rr) {
        ld acc = 0;
        for (int c = 0; c < 4; ++c) acc += Mk[4 * rr + c] * TB[c];
        F.gC[i == chunk - 1 ? 0 : 1][rr] = (double)acc;
      }
    }
    if (i < chunk) mat4_mul(M, Mk, Mk);
  }
  F.pskip = 0;
  for (int j = 0; j < 6; ++j) {
    for (int i = 0; i < 16; ++i) F.P[j][i] = (double)Mk[i];
    // shelf-pole block negligible against double precision (the other entries are O(1))
    const ld blk = std::max(std::max(fabsl(Mk[10]), fabsl(Mk[11])), std::max(fabsl(Mk[14]), fabsl(Mk[15])));
    if (blk < 1e-19L) F.pskip |= 1 << j;
    mat4_mul(Mk, Mk, Mk);
  }
}

static int interp_factor(unsigned rate) { return rate < 96000 ? 4 : (rate < 192000 ? 2 : 0); }

// 49-tap Hann-windowed sinc split into polyphase branches.
// Branch 0 is a pure delay (its maximum is the sample peak), so only the
// non-trivial branches matter: 3 x 12 taps (4x) or 1 x 24 taps (2x); tap t of a
// branch multiplies x[n - t].  The prototype is symmetric about tap 24, so 4x
// branch 3 is branch 1 reversed and branch 2 (like the 2x branch) is its own
// mirror; the kernel gets the unique values only:
//   tp[0..11]  = 4x branch 1 (branch 3 = reversed) | 2x branch, first half
//   tp[12..17] = 4x branch 2, first half
// (a mirrored pair can differ by one double ulp in libebur128's own table,
// i.e. < 1e-16 of an output that is compared at 1e-4).
static void design_interp(int factor, float tp[36]) {
  const double pi = 3.14159265358979323846264338327950288;
  memset(tp, 0, 36 * sizeof(float));
  if (!factor) return;
  double br[4][25];
  int count[4] = {0, 0, 0, 0};
  for (int j = 0; j < 49; ++j) {
    const double m = (double)j - 24.0;
    double c = 1.0;
    if (std::fabs(m) > 0.000001) c = std::sin(m * pi / factor) / (m * pi / factor);
    c *= 0.5 * (1.0 - std::cos(2.0 * pi * j / 48.0));
    if (std::fabs(c) > 0.000001) br[j % factor][count[j % factor]++] = c;  // index j / factor == slot
  }
  if (factor == 4) {
    for (int t = 0; t < 12; ++t) tp[t] = (float)br[1][t];
    for (int t = 0; t < 6; ++t) tp[12 + t] = (float)br[2][t];
    // phase 3 is phase 1 mirrored: with a_k = x[n-k] + x[n-11+k], b_k = x[n-k] - x[n-11+k]
    // the pair is y1 + y3 = sum (c1[k] + c1[11-k]) a_k and y1 - y3 = sum (c1[k] - c1[11-k]) b_k,
    // and max(|y1|, |y3|) = (|y1 + y3| + |y1 - y3|) / 2: the kernel gets the halved
    // sum / difference coefficients as pairs (tp[18 + 2k], tp[19 + 2k])
    for (int k = 0; k < 6; ++k) {
      tp[18 + 2 * k] = (float)(0.5 * (br[1][k] + br[1][11 - k]));
      tp[19 + 2 * k] = (float)(0.5 * (br[1][k] - br[1][11 - k]));
    }
  } else {
    for (int t = 0; t < 12; ++t) tp[t] = (float)br[1][t];
  }
}

// Pruning bound of the interpolator (lgd_tp_kernel): every output of a window is at
// most L1 * max|x| of it, L1 = the largest sum of |taps| of a non-trivial phase (4x: 1.8642
// for phase 2, 1.6312 for phases 1 / 3; 2x: 2.3068).  The kernel evaluates the taps in fp32
// through sums and differences of mirrored samples: ~14 roundings of 2^-24 each on
// quantities bounded by the same L1 * max|x| (and 2.34 * max|x| for the (sum, difference) form of
// phases 1 / 3), i.e. < 4e-6 relative.  The factor below leaves 10x that margin, so a window with
// max|x| <= tp_thr * P can never produce a computed output above P.
static float interp_prune_factor(int factor) {
  if (!factor) return 0.f;
  const double pi = 3.14159265358979323846264338327950288;
  double l1[4] = {0, 0, 0, 0};
  for (int j = 0; j < 49; ++j) {
    const double m = (double)j - 24.0;
    double c = 1.0;
    if (std::fabs(m) > 0.000001) c = std::sin(m * pi / factor) / (m * pi / factor);
    c *= 0.5 * (1.0 - std::cos(2.0 * pi * j / 48.0));
    if (std::fabs(c) > 0.000001) l1[j % factor] += std::fabs((double)(float)c) + 1e-9;
  }
  double worst = 0.0;
  for (int f = 1; f < factor; ++f) worst = std::max(worst, l1[f]);
  float r = (float)(1.0 / (worst * (1.0 + 4e-5)));
  return std::nextafterf(r, 0.f);
}

// Adjacent-pair bound of the interpolator (lgd_tp_kernel, sparse rows).  Tap k of a phase multiplies x[n - k];
// the two centre taps kA = HX / 2, kB = kA + 1 (4x: 5, 6 of 12; 2x: 11, 12 of 24) carry most of the phase's
// weight.  With u = |x[n - kA]|, v = |x[n - kB]|, u + v <= S2, max(u, v) <= M and every other sample <= M:
//   |y| <= a u + b v + R M <= min(a, b) S2 + (|a - b| + R) M,      R = sum of the other |taps|.
// out = (alpha2, beta2, alpha1, beta1): the half-sample phase and the quarter-sample phases of the 4x
// interpolator (2x: its one phase twice), rounded up; out[4] = 1 / (1 + margin) rounded down, the margin
// (4e-5, as interp_prune_factor) covering the fp32 roundings of the interpolator (< 2e-6 M, and beta >= 0.59)
// and of the bound's own two FMAs.
static void interp_pair_bound(int factor, float out[5]) {
  for (int i = 0; i < 5; ++i) out[i] = 0.f;
  if (!factor) return;
  const double pi = 3.14159265358979323846264338327950288;
  const int ntap = factor == 4 ? 12 : 24, kA = ntap / 2 - 1, kB = ntap / 2;
  double al[4] = {0, 0, 0, 0}, be[4] = {0, 0, 0, 0};
  for (int ph = 1; ph < factor; ++ph) {
    double a = 0, b = 0, rest = 0;
    for (int k = 0; k < ntap; ++k) {
      const int j = factor * k + ph;
      const double m = (double)j - 24.0;
      double c = 1.0;
      if (std::fabs(m) > 0.000001) c = std::sin(m * pi / factor) / (m * pi / factor);
      c *= 0.5 * (1.0 - std::cos(2.0 * pi * j / 48.0));
      const double ac = std::fabs((double)(float)c) + 1e-9;
      if (k == kA) a = ac;
      else if (k == kB) b = ac;
      else rest += ac;
    }
    al[ph] = std::min(a, b);
    be[ph] = std::fabs(a - b) + rest;
  }
  auto up = [](double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, 2.f * f + 1.f); return std::nextafterf(f, 2.f * f + 1.f); };
  if (factor == 4) {
    out[0] = up(al[2]); out[1] = up(be[2]);
    out[2] = up(std::max(al[1], al[3])); out[3] = up(std::max(be[1], be[3]));
  } else {
    out[0] = out[2] = up(al[1]); out[1] = out[3] = up(be[1]);
  }
  out[4] = std::nextafterf((float)(1.0 / (1.0 + 4e-5)), 0.f);
}

// ------------------------------------------------------------------ context --
struct Group {  // tracks sharing (rate, channels, channels per workgroup): one set of kernel constants
  unsigned rate, nch, nch_total;  // nch: channels (waves) per workgroup
  int chunk, tp;
  bool generic;
  bool strided;  // channel pairs of a wider interleaved stream (lgd_scan_kernel<.., STR>)
  bool s16;      // its tracks' PCM is interleaved int16 (lgd_plan_formats), read by the S16 kernel variants
  LgdFilt F;
  size_t seg_begin, seg_count;
  int rows_max;  // most true-peak candidate rows (tiles x channels) of any of its segments
  size_t launch;  // the scan launch its segments go out with
  int tp_u, tp_ns;  // the lgd_tp_kernel instance of its chunk (lgd_tp_instance)
};
// One lgd_tp_kernel launch: every interpolating segment whose (window step, factor, slab steps) instance is
// the same, whatever its rate, chunk and channel count -- C5's 48 kHz mono / stereo / 5.1 tracks are one
// launch, its 96 kHz tracks another (nine launches per group became four).  The segments are a second
// descriptor array in this order (WorkSet::d_segs_tp).
struct TpLaunch {
  int u, tp, ns, s16, rows_max;
  size_t seg_begin, seg_count;
};
// One lgd_scan_kernel launch: the groups that run the same kernel instance (chunk, waves per workgroup,
// staging mode) -- e.g. the 48, 96 and 192 kHz stereo tracks of a plan (C = 75 for all three); the
// constants differ per segment (LgdSeg::filt).
struct Launch {
  int chunk, nch, tp, mode;  // tp: some group of it has an interpolator (the kernel records chunk maxima)
  int s16;                   // PCM element format of its groups' tracks (LGD_PCM_*)
  size_t seg_begin, seg_count;
  std::vector<size_t> groups;
};
static const size_t MAX_GROUPS = 64;  // distinct (rate, channels) pairs per plan
static const unsigned LGD_GROUP_CH = 16;  // channels (waves) per workgroup at most

struct lgd_ctx {
  int device = 0;
  long p_chunk = 0, p_seg_sb = 0, p_warm_sb = 2, p_waves_per_cu = 8, p_debug = 0, p_timing = 1, p_overlap = 0,
       p_album_slots = 0, p_tp_prune = 1, p_album_world = 8, p_group_streams = 0, p_strided = 1, p_merge = 1,
       p_tp_dense_min = 32;
  int n_cu = 256;
  // plan
  bool planned = false, executed = false;
  uint32_t flags = 0;
  std::vector<lgd_track> tracks;
  std::vector<uint8_t> fmt;       // per track: LGD_PCM_* of this plan
  std::vector<uint8_t> next_fmt;  // announced by lgd_plan_formats for the next plan
  std::vector<LgdTrackMeta> meta;
  std::vector<LgdSeg> segs;
  std::vector<LgdSeg> segs_tp;       // the interpolating segments once more, in true-peak launch order
  std::vector<TpLaunch> tp_launches;
  std::vector<LgdRange> ranges;
  std::vector<LgdSlice> slices;
  std::vector<LgdAlbumMeta> albums;
  std::vector<Group> groups;
  uint64_t total_sb = 0, total_e = 0, total_st = 0, total_peak_floats = 0, pcm_bytes = 0,
           warm_bytes = 0, rec1_len = 4, total_tp_rows = 0, pcm_resident_bytes = 0;
  // device workspace.  Everything a scan writes exists twice (WorkSet): scans
  // alternate between the two sets and between two streams (see lgd_execute).
  struct WorkSet {
    double *d_E = nullptr, *d_Z = nullptr, *d_st = nullptr, *d_res = nullptr, *d_album = nullptr;
    double *d_p1 = nullptr, *d_p2 = nullptr, *d_p2a = nullptr, *d_pmax = nullptr;  // per-slice partials
    unsigned *d_done = nullptr;  // per track: epilogue workgroups finished (lgd_track_epilogue_kernel)
    size_t cap_done = 0;
    // album: record 1 = {sum_abs, n_abs, peak, n_st | st energies | 0-padding}, record 2 =
    // {sum_rel, n_rel} are what ranks exchange; d_st points into record 1; part1 = folded heads
    // (single-GPU plans may hold many albums: heads / part1 / rec2 / album are per album)
    double *d_rec1 = nullptr, *d_rec2 = nullptr, *d_part1 = nullptr, *d_heads = nullptr;
    LgdRange *d_album_ranges = nullptr;  // per album, into this set's d_st (single-GPU form)
    const double *lra_base = nullptr;  // short-term list of the album (set by stage 2)
    uint64_t lra_n = 0;
    float *d_peaks = nullptr;
    float *d_hint = nullptr;        // per track and channel: sample peak of the whole track (LgdSeg::hint)
    unsigned char *d_tp_rows = nullptr;  // chunk maxima, 1 KB per (group of 8 tiles, channel) (LgdSeg::tp_rows)
    LgdSeg *d_segs = nullptr;       // descriptors carry pointers into this set's E / peaks
    LgdSeg *d_segs_tp = nullptr;    // the same for lgd_tp_kernel's launches (ctx::segs_tp)
    // long loudness-range lists (> LGD_LRA_BIG entries): which ranges, and the scratch of the
    // multi-workgroup kernels (shared by the track launch and the album launch of an execute)
    int *d_big_tr = nullptr, *d_big_al = nullptr, *d_big_one = nullptr;
    long long *d_off_tr = nullptr, *d_off_al = nullptr, *d_off_one = nullptr;
    int n_big_tr = 0, n_big_al = 0;
    unsigned *d_lra_hist = nullptr;
    double *d_lra_part = nullptr, *d_lra_cand = nullptr;
    unsigned char *d_lra_picks = nullptr;
    size_t cap_big_tr = 0, cap_big_al = 0, cap_big_one = 0, cap_off_tr = 0, cap_off_al = 0, cap_off_one = 0,
           cap_lra_hist = 0, cap_lra_part = 0, cap_lra_cand = 0, cap_lra_picks = 0;
    uint64_t lra_dist_cap = 0;  // longest gathered album list the scratch holds (multi-GPU form)
    LgdRange *d_ranges = nullptr, *d_album_range = nullptr;
    LgdRange *h_album_range = nullptr;  // pinned
    size_t cap_segs_tp = 0;
    size_t cap_E = 0, cap_Z = 0, cap_st = 0, cap_res = 0, cap_peaks = 0, cap_segs = 0,
           cap_ranges = 0, cap_p1 = 0, cap_p2 = 0, cap_p2a = 0, cap_rec1 = 0, cap_album = 0,
           cap_part1 = 0, cap_rec2 = 0, cap_heads = 0, cap_album_ranges = 0, cap_pmax = 0, cap_hint = 0, cap_tp_rows = 0;
    hipEvent_t ev_scan = nullptr;   // caller-stream marker the internal stream waits for
    hipEvent_t ev_done = nullptr;   // end of this set's scan + epilogue (lgd_album_join)
    hipEvent_t ev_album = nullptr;  // end of a caller-driven album stage 3 on this set
    bool album_pending = false;
  } ws[4];
  int n_sets = 1;     // 2 pipelined; 4 when the caller drives the album stages (their exchange lags the scans)
  int cur_set = 0;    // set of the last lgd_execute
  hipStream_t side = nullptr;
  hipEvent_t ev_join = nullptr;  // lgd_join: end of the internal stream's work so far
  static const int GSTREAMS = 3;   // extra streams the groups of a mixed plan are launched on
  hipStream_t gstream[GSTREAMS] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_gjoin[GSTREAMS] = {nullptr, nullptr, nullptr};
  std::vector<Launch> launches;     // scan launches, largest first
  LgdSlice *d_slices = nullptr;
  LgdFilt *d_filt = nullptr;  // [MAX_GROUPS] per-group kernel constants
  LgdTrackMeta *d_meta = nullptr;
  LgdAlbumMeta *d_albums = nullptr;
  size_t cap_meta = 0, cap_slices = 0, cap_albums = 0;
  hipStream_t last_stream = nullptr;
  // ring of (start, scan kernel done, all done) event triples, one per execute
  static const int EV_RING = 64;
  hipEvent_t ev[EV_RING][4];  // start, scan + true-peak kernels done, all done, scan kernels done
  bool ev3_valid[EV_RING] = {};
  uint64_t n_exec = 0;
  double abs_gate, rel_factor, minus20;
};

template <typename T>
static int ensure(T **p, size_t *cap, size_t need) {
  if (need <= *cap && *p) return LGD_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  size_t n = need ? need : 1;
  HIPCHK(hipMalloc((void **)p, n * sizeof(T)));
  *cap = n;
  return LGD_OK;
}

extern "C" const char *lgd_last_error(void) { return g_err.c_str(); }

extern "C" lgd_ctx *lgd_create(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
    fail(LGD_EHIP, "no HIP device %d (count %d): the scanner has no CPU fallback", device, n);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    fail(LGD_EHIP, "hipSetDevice(%d) failed", device);
    return nullptr;
  }
  lgd_ctx *c = new lgd_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  c->abs_gate = std::pow(10.0, (-70.0 + 0.691) / 10.0);
  c->rel_factor = std::pow(10.0, -10.0 / 10.0);
  c->minus20 = std::pow(10.0, -20.0 / 10.0);
  bool ok = true;
  memset(c->ev, 0, sizeof(c->ev));
  for (int i = 0; i < lgd_ctx::EV_RING; ++i)
    for (int j = 0; j < 4; ++j) ok = ok && hipEventCreate(&c->ev[i][j]) == hipSuccess;
  for (auto &w : c->ws) {
    ok = ok && hipMalloc((void **)&w.d_album_range, sizeof(LgdRange)) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&w.h_album_range, sizeof(LgdRange)) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&w.ev_scan, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&w.ev_done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&w.ev_album, hipEventDisableTiming) == hipSuccess;
  }
  // HIP deals a process's streams over a few hardware queues (four by default) and streams that share a queue run
  // one after the other: with three more streams per context the caller's stream and `side` once ended up on one
  // queue and "overlap" 1 no longer overlapped (0.285 -> 0.317 ms per step).  Queues are kept per priority level, so
  // the internal stream asks for the highest one -- a queue the caller's (normal) stream cannot be on -- and the group
  // streams are created only when a plan asks for them.
  {
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    ok = ok && hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_high) == hipSuccess;
  }
  ok = ok && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < lgd_ctx::GSTREAMS; ++i)
    ok = ok && hipEventCreateWithFlags(&c->ev_gjoin[i], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipMalloc((void **)&c->d_filt, MAX_GROUPS * sizeof(LgdFilt)) == hipSuccess;
  if (!ok) {
    fail(LGD_ENOMEM, "lgd_create: allocation failed");
    lgd_destroy(c);
    return nullptr;
  }
  return c;
}

extern "C" void lgd_destroy(lgd_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto &w : c->ws) {
    void *ptrs[] = {w.d_E, w.d_Z, w.d_rec1, w.d_res, w.d_album, w.d_part1, w.d_rec2, w.d_peaks, w.d_done,
                    w.d_segs, w.d_segs_tp, w.d_ranges, w.d_album_range, w.d_p1, w.d_p2, w.d_p2a, w.d_heads,
                    w.d_album_ranges, w.d_pmax, w.d_hint, w.d_tp_rows, w.d_big_tr, w.d_big_al, w.d_big_one,
                    w.d_off_tr, w.d_off_al, w.d_off_one, w.d_lra_hist, w.d_lra_part, w.d_lra_cand, w.d_lra_picks};
    for (void *p : ptrs)
      if (p) (void)hipFree(p);
    if (w.h_album_range) (void)hipHostFree(w.h_album_range);
    if (w.ev_scan) (void)hipEventDestroy(w.ev_scan);
    if (w.ev_done) (void)hipEventDestroy(w.ev_done);
    if (w.ev_album) (void)hipEventDestroy(w.ev_album);
  }
  void *ptrs2[] = {c->d_meta, c->d_slices, c->d_filt, c->d_albums};
  for (void *p : ptrs2)
    if (p) (void)hipFree(p);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  for (int i = 0; i < lgd_ctx::GSTREAMS; ++i) {
    if (c->gstream[i]) (void)hipStreamDestroy(c->gstream[i]);
    if (c->ev_gjoin[i]) (void)hipEventDestroy(c->ev_gjoin[i]);
  }
  for (int i = 0; i < lgd_ctx::EV_RING; ++i)
    for (int j = 0; j < 4; ++j)
      if (c->ev[i][j]) (void)hipEventDestroy(c->ev[i][j]);
  delete c;
}

extern "C" int lgd_set_param(lgd_ctx *c, const char *name, long value) {
  if (!c || !name) return fail(LGD_EINVAL, "lgd_set_param: null argument");
  if (value < 0) return fail(LGD_EINVAL, "lgd_set_param(%s): negative value", name);
  if (!strcmp(name, "chunk")) c->p_chunk = value;
  else if (!strcmp(name, "seg_subblocks")) c->p_seg_sb = value;
  else if (!strcmp(name, "warm_subblocks")) c->p_warm_sb = value;
  else if (!strcmp(name, "waves_per_cu")) c->p_waves_per_cu = value ? value : 8;
  else if (!strcmp(name, "debug")) c->p_debug = value;  // kernel floor measurements only
  else if (!strcmp(name, "timing")) { c->p_timing = value; return LGD_OK; }  // hipEvent brackets on/off
  else if (!strcmp(name, "overlap")) c->p_overlap = value;  // 0: every scan on the caller's stream
  else if (!strcmp(name, "album_slots")) c->p_album_slots = value;  // short-term slots of album record 1
  else if (!strcmp(name, "tp_prune")) c->p_tp_prune = value;  // 0: evaluate every interpolator window
  else if (!strcmp(name, "tp_dense_min")) c->p_tp_dense_min = value;  // rows with >= this many flagged chunks (of 64) are walked whole; 65 = never
  else if (!strcmp(name, "group_streams")) c->p_group_streams = value;  // 0: groups one after the other
  else if (!strcmp(name, "merge_launches")) c->p_merge = value;  // 1: groups that share a kernel instance go out in one launch
  else if (!strcmp(name, "strided")) c->p_strided = value;  // channel pair / triple workgroups: 0 never, 1 where measured faster, 2 pairs for every 3+ channel layout, 3 triples wherever the count divides
  else if (!strcmp(name, "album_world")) c->p_album_world = value ? value : 8;  // ranks the multi-GPU album scratch is sized for
  else return fail(LGD_EINVAL, "lgd_set_param: unknown parameter '%s'", name);
  c->planned = false;
  return LGD_OK;
}

static int pick_chunk(long forced, int s100, unsigned nch, int tp) {
  const size_t lds_cap = 160 * 1024;  // per CU == per workgroup limit on gfx950
  if (forced) {
    for (const int *p = lgd_chunk_table; *p; ++p)
      if (*p == forced && s100 % *p == 0 && (nch <= 2 || *p <= 50) &&
          lgd_scan_lds_bytes(*p, (int)nch, tp, nch > 2 && !((nch >= 3 && nch <= 6) || nch == 8)) <= lds_cap)
        return *p;
    return 0;
  }
  // preference: long chunks amortise the wave scan; the tile of all channels has
  // to leave room for >= 2 workgroups per CU where possible
  // (70: the 44.1 kHz family -- 4410 = 63 x 70 -- 0.279 against 0.288 ms at C = 63 for 345.6 M stereo samples)
  static const int pref_fast[] = {75, 70, 63, 50, 49, 45, 35, 25, 0};
  // the run-time-channel-count kernel is compiled for 16 waves (<= 128 VGPRs):
  // short chunks keep its prefetch registers within that
  static const int pref_many[] = {25, 35, 45, 49, 50, 63, 75, 0};
  // 3, 4, 6 (5.1), 8 (7.1) planes per workgroup: the two short chunks compiled for them
  // (measured, tools/rate_sweep.py: with six or eight waves per workgroup the short chunk wins -- four
  // workgroups per CU instead of two -- 43 % vs 38 % of the HBM peak for 5.1 at 48 kHz; three or four
  // planes fit twice at C = 50: 52 / 56 % vs 41 / 50 %)
  static const int pref_34[] = {50, 45, 49, 35, 25, 0};
  static const int pref_51[] = {25, 45, 35, 49, 50, 0};
  static const int pref_5[] = {25, 35, 0};  // 5 planes: longer chunks spill 25+ registers at 3 waves per SIMD
  const bool multi = (nch >= 3 && nch <= 6) || nch == 8;  // planar specialisations exist
  // (eight planes: one workgroup per CU either way, so the long chunk wins -- 0.362 ms against 0.425 ms for
  // 345.6 M samples at 48 kHz with true peak; six planes at C = 50 need more than the 168 VGPRs of
  // three waves per SIMD and lose, 0.47 against 0.39 ms)
  const int *pref = nch <= 2 ? pref_fast
                    : (nch == 5 ? pref_5 : (nch == 4 || nch == 3 || nch == 8 ? pref_34 : (multi ? pref_51 : pref_many)));
  const bool generic = nch > 2 && !multi;
  for (int pass = 0; pass < 2; ++pass)
    for (const int *p = pref; *p; ++p)
      if (s100 % *p == 0 &&
          lgd_scan_lds_bytes(*p, (int)nch, tp, generic) <= (pass || nch == 8 ? lds_cap : (multi ? lds_cap / 2 : lds_cap / 4)))
        return *p;
  return 0;
}

extern "C" int lgd_plan_albums(lgd_ctx *c, const lgd_track *tracks, uint32_t n,
                               const uint32_t *album_of_track, uint32_t n_albums, uint32_t flags);

extern "C" int lgd_plan_formats(lgd_ctx *c, const uint8_t *formats, uint32_t n) {
  if (!c) return fail(LGD_EINVAL, "lgd_plan_formats: null context");
  c->next_fmt.clear();
  if (!formats || !n) return LGD_OK;  // back to "all f32"
  for (uint32_t t = 0; t < n; ++t)
    if (formats[t] != LGD_PCM_F32 && formats[t] != LGD_PCM_S16)
      return fail(LGD_EINVAL, "lgd_plan_formats: track %u: unknown PCM format %u", t, (unsigned)formats[t]);
  c->next_fmt.assign(formats, formats + n);
  return LGD_OK;
}

extern "C" int lgd_plan(lgd_ctx *c, const lgd_track *tracks, uint32_t n, uint32_t flags) {
  return lgd_plan_albums(c, tracks, n, nullptr, 1, flags);
}

extern "C" int lgd_plan_albums(lgd_ctx *c, const lgd_track *tracks, uint32_t n,
                               const uint32_t *album_of_track, uint32_t n_albums, uint32_t flags) {
  if (!c) return fail(LGD_EINVAL, "lgd_plan: null context");
  if (n && !tracks) return fail(LGD_EINVAL, "lgd_plan: null track array");
  if (n_albums == 0) return fail(LGD_EINVAL, "lgd_plan: zero albums");
  if (n_albums > 1 && !(flags & LGD_FLAG_ALBUM))
    return fail(LGD_EINVAL, "lgd_plan: several albums need LGD_FLAG_ALBUM (one GPU per plan)");
  if (n_albums > 1 && !album_of_track) return fail(LGD_EINVAL, "lgd_plan: null album index array");
  for (uint32_t t = 0; t < n && album_of_track; ++t)
    if (album_of_track[t] >= n_albums || (t && album_of_track[t] < album_of_track[t - 1]))
      return fail(LGD_EINVAL, "lgd_plan: album indices must be < n_albums and non-decreasing (track %u)", t);
  HIPCHK(hipSetDevice(c->device));
  c->planned = c->executed = false;
  c->flags = flags;
  c->tracks.assign(tracks, tracks + n);
  c->meta.assign(n, LgdTrackMeta());
  c->albums.assign(n_albums, LgdAlbumMeta{0, 0, 0, 0});
  c->segs.clear();
  c->segs_tp.clear();
  c->tp_launches.clear();
  c->ranges.clear();
  c->slices.clear();
  c->groups.clear();
  c->total_sb = c->total_e = c->total_st = c->total_peak_floats = c->pcm_bytes = c->warm_bytes = c->pcm_resident_bytes = 0;
  c->total_tp_rows = 0;
  // PCM element formats of this plan's tracks (lgd_plan_formats, consumed here; none: all f32)
  if (!c->next_fmt.empty() && c->next_fmt.size() != n) {
    const size_t announced = c->next_fmt.size();
    c->next_fmt.clear();  // (an announcement is consumed by the plan call it was made for, whatever that call returns)
    return fail(LGD_EINVAL, "lgd_plan: lgd_plan_formats announced %zu tracks, the plan has %u", announced, n);
  }
  c->fmt.assign(n, (uint8_t)LGD_PCM_F32);
  if (!c->next_fmt.empty()) c->fmt.swap(c->next_fmt);
  c->next_fmt.clear();

  uint64_t total_ch = 0;
  if ((uint64_t)n * LGD_MAX_CHANNELS > 0x7fffffffull) return fail(LGD_EUNSUP, "too many tracks in one plan");
  for (uint32_t t = 0; t < n; ++t) {
    const lgd_track &tr = tracks[t];
    // ebur128_init's own argument checks
    if (tr.channels == 0 || tr.channels > LGD_MAX_CHANNELS)
      return fail(LGD_EINVAL, "track %u: %u channels (1..64 supported)", t, tr.channels);
    if (tr.rate < 16 || tr.rate > 2822400)
      return fail(LGD_EINVAL, "track %u: sample rate %u out of range", t, tr.rate);
    if (tr.frames && !tr.pcm) return fail(LGD_EINVAL, "track %u: null PCM pointer", t);
    if (((uintptr_t)tr.pcm) & 15)
      return fail(LGD_EINVAL, "track %u: PCM pointer must be 16-byte aligned", t);
    LgdTrackMeta &m = c->meta[t];
    m.s100 = (int)((tr.rate + 5) / 10);
    m.nch = (int)tr.channels;
    m.album = album_of_track ? (int)album_of_track[t] : 0;
    m.hint_off = (int)total_ch;
    m.pad = 0;
    total_ch += tr.channels;
    const uint64_t nsb = tr.frames / (uint64_t)m.s100;
    if (nsb > 0x7fffffffull) return fail(LGD_EUNSUP, "track %u too long", t);
    m.n_sb = (int)nsb;
    m.n_st_slots = m.n_sb >= 30 ? (m.n_sb - 30) / 10 + 1 : 0;
    m.sb_off = (long long)c->total_sb;
    m.e_off = (long long)c->total_e;
    m.st_off = (long long)c->total_st;
    m.slice_off = (int)c->slices.size();
    // (at least one: the track's last epilogue workgroup writes its result record -- a track shorter than
    // 400 ms has one slice without blocks)
    m.n_slices = m.n_sb >= 4 ? (m.n_sb - 3 + LGD_SLICE - 1) / LGD_SLICE : 1;
    for (int sl = 0; sl < m.n_slices; ++sl) c->slices.push_back(LgdSlice{(int)t, sl * LGD_SLICE});
    c->total_sb += nsb;
    c->total_e += nsb * tr.channels;
    c->total_st += (uint64_t)m.n_st_slots;
    c->pcm_bytes += tr.frames * tr.channels * 4ull;  // (algorithmic: 4 B per sample whatever the element format)
    c->pcm_resident_bytes += tr.frames * tr.channels * (c->fmt[t] == LGD_PCM_S16 ? 2ull : 4ull);
  }
  {  // albums = runs of consecutive tracks; an album without tracks is an empty run
    uint32_t t = 0;
    for (uint32_t a = 0; a < n_albums; ++a) {
      LgdAlbumMeta &am = c->albums[a];
      am.t0 = (int)t;
      am.slice0 = t < n ? c->meta[t].slice_off : (int)c->slices.size();
      while (t < n && c->meta[t].album == (int)a) ++t;
      am.t1 = (int)t;
      am.slice1 = t < n ? c->meta[t].slice_off : (int)c->slices.size();
    }
  }

  // launch groups: tracks (or 16-channel groups of wide tracks) that share
  // (rate, channels of the stream, channels per workgroup) run in one launch
  struct Key {
    unsigned rate, nch_total, g_nch, fmt;
    bool operator<(const Key &o) const {
      return std::tie(rate, nch_total, g_nch, fmt) < std::tie(o.rate, o.nch_total, o.g_nch, o.fmt);
    }
  };
  std::map<Key, size_t> group_of;
  std::vector<std::vector<LgdSeg>> group_segs;

  // Segment length, per (rate, channels) -- every such group is one launch and has to fill the
  // GPU by itself.  A CU holds `slots` workgroups of the group's kernel (2 waves per SIMD for
  // mono / stereo: waves_per_cu / channels workgroups; floor(12 / channels) at 3 waves per SIMD
  // for 3 .. 8 channels, 16 waves per CU for the 9 .. 16-channel kernel).  The group's sub-blocks
  // are cut into `rounds` x slots segments of at most ~48 sub-blocks: ONE full round for a
  // C2-sized plan (1000 segments of 36: measured best; 1.5 rounds cost 25 %), many rounds of
  // short segments for album-sized plans (a 1000-track album as a few hundred long segments
  // left half the GPU idle in its last round), never below min_seg (the 200 ms warm-up in front
  // of a segment has to stay small against it).
  const uint64_t min_seg = (uint64_t)std::max<long>(4, 3 * c->p_warm_sb);
  const uint64_t max_seg = 48;
  struct RC { unsigned rate, ch, fmt; bool operator<(const RC &o) const { return std::tie(rate, ch, fmt) < std::tie(o.rate, o.ch, o.fmt); } };
  std::map<RC, uint64_t> key_sb, key_seg;
  // Streams as stereo-shaped workgroups, one per channel pair (the last pair of an odd count
  // overlaps its neighbour): possible whenever a stereo chunk length divides the sub-block.  Every
  // pair's workgroup pulls all of the stream's cache lines through L1 (L2 / Infinity Cache serve the
  // siblings), so it only pays where the many-plane kernels are weak -- measured (345.6 M samples,
  // tools/rate_sweep.py, "strided" 2 = every layout): 5 channels 40 % of the HBM peak instead of
  // 27 %, 7 channels 34 vs 30 %, 24 channels 12 vs 10 %; but 3 / 4 / 5.1 / 7.1 / 12 channels 38 / 52 /
  // 44 / 29 / 19 % against 54 / 56 / 45 / 41 / 42 % on the planar kernels.
  // 5.1 (six channels) goes out as two channel TRIPLES: three-wave workgroups spread evenly over a
  // CU's SIMDs (four per CU at C = 50), the planar kernel's six-wave workgroups do not (two per CU
  // land 4 / 4 / 2 / 2 waves on a quarter of the CUs, tools/hwid_probe.py).  5 and 7 channels as
  // OVERLAPPING triples (0-2 | 2-4; 0-2 | 3-5 | 4-6), one kernel instance and one launch per track set:
  // 48 % / 40 % of the HBM peak against 40 % / 35 % as pairs (round 3, tools/odd_probe.py).
  // Returns the channels per workgroup: 0 = not strided, 2 = pairs, 3 = triples, 4 = quads.
  auto strided_for = [&](unsigned rate, unsigned ch) -> unsigned {
    if (!c->p_strided || ch < 3) return 0;
    const int tp_ = (flags & LGD_FLAG_TRUE_PEAK) ? interp_factor(rate) : 0;
    const int s100_ = (int)((rate + 5) / 10);
    // (three channels: the one-triple form only with an interpolator -- 0.316 against 0.329 ms, where its chunk-maxima
    // records can wait in registers; without, the planar kernel is 1 % ahead)
    // channel QUADS (four-wave workgroups, one 16-byte load per lane and frame): 7 / 8 / 9 channels (0-3 | 3-6; 0-3 | 4-7;
    // 0-3 | 4-7 | 5-8) and everything from 16 channels up.  Measured (round 3, tools/odd_probe.py, % of the HBM peak without /
    // with true peak): 7 ch 52 / 45 against 40 / 37 as triples, 7.1 58 / 50 against 50 / 43 on the eight-plane kernel, 9 ch
    // 46 / 41 against 35 / 26, 16 ch 34 / 30 against 30 / 23, 24 ch 25 / 23 against 12 / 10 as pairs, 64 ch 20 / 17 against
    // 13 / 10; but 5 ch 40 % (triples 48), 5.1 50 % (triples 52), 10 - 15 ch 30 - 34 % against 41 - 52 % on the
    // run-time-channel kernel (sets of nothing but peaks-only channels run ahead of their siblings).  "strided" 4: quads for
    // every layout from 5 channels up.
    if ((c->p_strided == 4 ? ch >= 5 : (c->p_strided == 1 && ((ch >= 7 && ch <= 9) || ch >= 16))) &&
        pick_chunk(c->p_chunk, s100_, 4, tp_) != 0)
      return 4;
    if ((c->p_strided == 3 ? ch % 3 == 0 : ((c->p_strided == 1 || c->p_strided == 4) && (ch == 5 || ch == 6 || ch == 7 || (ch == 3 && tp_)))) &&
        pick_chunk(c->p_chunk, s100_, 3, tp_) != 0)
      return 3;
    if (c->p_strided != 2 && !(ch == 5 || ch == 7 || ch > LGD_GROUP_CH)) return 0;  // (5 / 7 channels: pairs only if no triple chunk divides the rate's sub-block)
    return pick_chunk(c->p_chunk, s100_, 2, tp_) != 0 ? 2 : 0;
  };
  for (uint32_t t = 0; t < n; ++t) {
    const unsigned ch = tracks[t].channels;
    const unsigned sw = strided_for(tracks[t].rate, ch);
    const uint64_t mult = sw ? (ch + sw - 1) / sw : 1;  // one segment set per pair / triple
    key_sb[RC{tracks[t].rate, ch, c->fmt[t]}] += (uint64_t)c->meta[t].n_sb * mult;
  }
  // The kernel instance a (rate, channels) key runs on: keys that share it share ONE launch
  // ("merge_launches" 1, default), and it is the launch that has to fill the GPU -- a plan of a few
  // tracks at each of 48 / 96 / 192 kHz (C5) would otherwise be three short launches, each with its
  // own partial last round.  The segments of a launch are sized to equal numbers of TILES (a
  // sub-block is 1 tile at 48 kHz and C = 75, 4 at 192 kHz).
  struct LK {
    int chunk; unsigned k; int mode; unsigned rate, ch, fmt;  // (rate, ch: 0 when launches merge)
    bool operator<(const LK &o) const {
      return std::tie(chunk, k, mode, rate, ch, fmt) < std::tie(o.chunk, o.k, o.mode, o.rate, o.ch, o.fmt);
    }
  };
  auto launch_key = [&](unsigned rate, unsigned ch, unsigned g_nch, unsigned fmt) -> LK {
    const unsigned sw = strided_for(rate, ch);
    const int tp_ = (flags & LGD_FLAG_TRUE_PEAK) ? interp_factor(rate) : 0;
    const int s100_ = (int)((rate + 5) / 10);
    int chunk = sw ? pick_chunk(c->p_chunk, s100_, sw, tp_)
                   : (((g_nch <= 6 || g_nch == 8) && g_nch == ch) ? pick_chunk(c->p_chunk, s100_, g_nch, tp_) : 0);
    const int mode = sw ? 2 : (chunk ? 0 : 1);
    if (!chunk) chunk = 25;
    return LK{chunk, g_nch, mode, c->p_merge ? 0u : rate, c->p_merge ? 0u : ch, fmt};
  };
  struct ClassAcc { double tiles = 0.0, tps_min = 1e30; std::vector<RC> keys; };
  std::map<LK, ClassAcc> classes;
  for (const auto &kv : key_sb) {
    const unsigned str = strided_for(kv.first.rate, kv.first.ch);
    const unsigned k = str ? str : std::min<unsigned>(kv.first.ch, LGD_GROUP_CH);
    const LK lk = launch_key(kv.first.rate, kv.first.ch, k, kv.first.fmt);
    const double tps = (double)((kv.first.rate + 5) / 10) / (64.0 * lk.chunk);  // tiles per sub-block
    ClassAcc &a = classes[lk];
    a.tiles += (double)kv.second * tps;
    a.tps_min = std::min(a.tps_min, tps);
    a.keys.push_back(kv.first);
  }
  for (const auto &cl : classes) {
    const unsigned k = cl.first.k;
    const unsigned per_cu = k <= 2 ? std::max(1u, (unsigned)c->p_waves_per_cu / k)
                                   : (k > 8 ? std::max(1u, 16u / k) : std::max(1u, 12u / k));
    const uint64_t slots = (uint64_t)c->n_cu * per_cu;
    for (const RC &key : cl.second.keys) {
      uint64_t seg;
      if (cl.second.keys.size() == 1) {
        const uint64_t sb = key_sb[key];
        const uint64_t rounds = std::max<uint64_t>(1, (sb + slots * max_seg - 1) / (slots * max_seg));
        seg = (sb + rounds * slots - 1) / (rounds * slots);
      } else {
        // rounds: the key with the fewest tiles per sub-block gets segments of <= max_seg sub-blocks
        const double per_round = (double)slots * (double)max_seg * cl.second.tps_min;
        const double rounds = std::max(1.0, std::ceil(cl.second.tiles / per_round - 1e-9));
        const double target = cl.second.tiles / (rounds * (double)slots);  // tiles per segment
        const double tps = (double)((key.rate + 5) / 10) / (64.0 * cl.first.chunk);
        seg = (uint64_t)std::max(1.0, std::ceil(target / tps - 1e-9));
      }
      if (c->p_seg_sb) seg = (uint64_t)c->p_seg_sb;
      else seg = std::max(seg, min_seg);
      key_seg[key] = std::max<uint64_t>(1, seg);
    }
  }
  std::map<LK, size_t> launch_of;
  c->launches.clear();

  for (uint32_t t = 0; t < n; ++t) {
    const lgd_track &tr = tracks[t];
    LgdTrackMeta &m = c->meta[t];
    const int s100 = m.s100;
    const uint64_t nsb = (uint64_t)m.n_sb;
    const uint64_t seg_t = key_seg[RC{tr.rate, tr.channels, c->fmt[t]}];
    const uint64_t nseg = nsb ? (nsb + seg_t - 1) / seg_t : 1;
    m.n_seg = (int)nseg;
    m.peak_off = (long long)c->total_peak_floats;
    c->total_peak_floats += nseg * 2ull * tr.channels;
    const unsigned strided = strided_for(tr.rate, tr.channels);
    std::vector<unsigned> ch0s;  // first channel of every workgroup set of this track
    if (strided) {
      // (a last pair / triple that does not fit overlaps its neighbour: the shared channels are left to
      // the neighbour, LgdSeg::skip_mask)
      for (unsigned a = 0; a + strided <= tr.channels; a += strided) ch0s.push_back(a);
      if (tr.channels % strided) ch0s.push_back(tr.channels - strided);
    } else {
      for (unsigned a = 0; a < tr.channels; a += LGD_GROUP_CH) ch0s.push_back(a);
    }
    std::vector<std::vector<LgdSeg>> pair_segs(ch0s.size());
    size_t pair_group = 0;
    for (size_t pi = 0; pi < ch0s.size(); ++pi) {
      const unsigned ch0 = ch0s[pi];
      const unsigned g_nch = strided ? strided : std::min<unsigned>(LGD_GROUP_CH, tr.channels - ch0);
      const Key key{tr.rate, tr.channels, g_nch, c->fmt[t]};
      auto it = group_of.find(key);
      if (it == group_of.end()) {
        Group g;
        g.rate = tr.rate;
        g.nch = g_nch;
        g.nch_total = tr.channels;
        g.tp = (flags & LGD_FLAG_TRUE_PEAK) ? interp_factor(g.rate) : 0;
        g.strided = strided != 0;
        g.s16 = c->fmt[t] == LGD_PCM_S16;
        g.chunk = strided ? pick_chunk(c->p_chunk, s100, strided, g.tp)
                          : (((g_nch <= 6 || g_nch == 8) && g_nch == tr.channels)
                                 ? pick_chunk(c->p_chunk, s100, g_nch, g.tp) : 0);
        // fast kernels: mono / stereo with a chunk that divides the sub-block; anything else
        // (more channels, channel groups, rates such as 11 025 Hz) goes to the generic kernel,
        // where sub-block boundaries may fall inside a chunk
        g.generic = !g.chunk;
        if (g.generic) g.chunk = 25;
        // below ~3.4 kHz the shelf's 1682 Hz corner lies beyond Nyquist and the
        // reference's own filter design is meaningless (it yields inf/NaN loudness)
        if (tr.rate < 4000)
          return fail(LGD_EUNSUP, "track %u: sample rate %u Hz is below the 4 kHz floor", t, tr.rate);
        memset(&g.F, 0, sizeof(g.F));
        design_kfilter((double)g.rate, g.F.pb, g.F.pa, g.F.ra);
        design_scan_basis(g.F, g.chunk);
        design_interp(g.tp, g.F.tp);
        interp_pair_bound(g.tp, g.F.tp + 30);
        g.F.tp_thr = interp_prune_factor(g.tp);
        g.F.tp_prune = c->p_tp_prune ? 1 : 0;
        g.F.tp_dense_min = (int)c->p_tp_dense_min;
        g.F.tp_hx = g.tp == 4 ? 11 : (g.tp == 2 ? 23 : 0);
        g.tp_u = g.tp_ns = 0;
        if (g.tp && lgd_tp_instance(g.chunk, &g.tp_u, &g.tp_ns))
          return fail(LGD_EUNSUP, "no true-peak kernel instance for chunk %d", g.chunk);
        g.F.pbn[0] = g.F.pb[1] / g.F.pb[0];
        g.F.pbn[1] = g.F.pb[2] / g.F.pb[0];
        g.F.pb0sq = g.F.pb[0] * g.F.pb[0];
        g.F.lps = s100 / g.chunk;
        g.F.s100 = s100;
        g.F.pad = (int)c->p_debug;
        g.seg_begin = g.seg_count = 0;
        g.rows_max = 0;
        {
          LK lk = launch_key(tr.rate, tr.channels, g_nch, c->fmt[t]);
          // (the kernel instance is what the group itself settled on)
          lk.chunk = g.chunk;
          lk.mode = g.strided ? 2 : (g.generic ? 1 : 0);
          auto li = launch_of.find(lk);
          if (li == launch_of.end()) {
            li = launch_of.emplace(lk, c->launches.size()).first;
            c->launches.push_back(Launch{g.chunk, (int)g.nch, 0, lk.mode, g.s16 ? 1 : 0, 0, 0, {}});
          }
          g.launch = li->second;
          Launch &L = c->launches[g.launch];
          L.tp = L.tp || g.tp;
          L.groups.push_back(c->groups.size());
        }
        if (c->groups.size() >= MAX_GROUPS) return fail(LGD_EUNSUP, "more than %zu (rate, channels) groups in one plan", MAX_GROUPS);
        it = group_of.emplace(key, c->groups.size()).first;
        c->groups.push_back(g);
        group_segs.emplace_back();
      }
      Group &g = c->groups[it->second];
      const long long tile_f = 64LL * g.chunk;
      const int warm_tiles =
          c->p_warm_sb ? (int)(((long long)c->p_warm_sb * s100 + tile_f - 1) / tile_f) : 0;
      uint64_t sb0 = 0;
      for (uint64_t sgi = 0; sgi < nseg; ++sgi) {
        const uint64_t cnt = nsb / nseg + (sgi < nsb % nseg ? 1 : 0);
        LgdSeg sg;
        sg.pcm = tr.pcm;
        sg.n_floats = (long long)(tr.frames * tr.channels);
        sg.f0 = (long long)(sb0 * (uint64_t)s100);
        sg.n_sb = (int)cnt;
        sg.f_peak_end = (sgi + 1 == nseg) ? (long long)tr.frames : (long long)((sb0 + cnt) * s100);
        sg.n_warm_tiles = sb0 ? warm_tiles : 0;
        // offsets are patched to pointers once the workspace exists
        sg.e_out = (double *)(uintptr_t)(m.e_off + (long long)sb0);
        sg.e_ch_stride = m.n_sb;
        sg.ch0 = (int)ch0;
        sg.nch_total = (int)tr.channels;
        sg.n_tiles = (int)((sg.f_peak_end - sg.f0 + tile_f - 1) / tile_f);
        sg.n_frames = (long long)tr.frames;
        sg.hint = (float *)(uintptr_t)m.hint_off;
        sg.filt = (const void *)(uintptr_t)it->second;  // group index, patched to its device constants
        sg.tp_rows = (void *)~(uintptr_t)0;  // no interpolator: patched to null
        sg.chunk = g.chunk;
        sg.nch_wg = (int)g_nch;
        sg.magic_nch = sg.magic_ns = sg.magic_c = 0;
        // (a last pair / triple that overlaps the set before it leaves the shared channels to that one -- if it keeps
        // at least two channels to filter itself: a workgroup with next to nothing to do runs ahead of its siblings,
        // the three no longer share their cache lines in L2 and each streams the PCM from HBM on its own -- 7 channels
        // as triples 0-2, 3-5, 4-6: 0.42 ms with channels 4 and 5 filtered twice, 0.60 ms with the third set idle)
        sg.skip_mask = 0u;
        if (strided && pi > 0 && ch0 < ch0s[pi - 1] + strided) {
          const unsigned shared = ch0s[pi - 1] + strided - ch0;
          unsigned kept = 0;
          for (unsigned cc = ch0 + shared; cc < ch0 + strided; ++cc) kept += lgd_channel_weight((int)cc, (int)tr.channels) > 0.0 ? 1u : 0u;
          if (kept >= 2) sg.skip_mask = (1u << shared) - 1u;
        }
        if (g.tp && lgd_tp_magics(g.chunk, (int)g_nch, g.tp, &sg.magic_nch, &sg.magic_ns, &sg.magic_c))
          return fail(LGD_EUNSUP, "true-peak kernel: chunk %d not divisible as needed", g.chunk);
        if (g.tp) {  // one row of candidate bits per tile and channel of this workgroup
          const long long n_tiles = sg.n_tiles;
          const long long rows = n_tiles * (long long)g_nch;
          if (rows > 0x7fffffffLL) return fail(LGD_EUNSUP, "track %u: segment too long", t);
          sg.tp_rows = (void *)(uintptr_t)(c->total_tp_rows * 1024ull);  // byte offset, patched per work set
          c->total_tp_rows += (uint64_t)((n_tiles + 7) / 8) * g_nch;        // groups of 8 tiles x channels
          g.rows_max = std::max(g.rows_max, (int)rows);
        }
        sg.peak_out = (float *)(uintptr_t)(m.peak_off + (long long)(sgi * 2ull * tr.channels));
        pair_segs[pi].push_back(sg);
        if (sb0) c->warm_bytes += (uint64_t)warm_tiles * tile_f * g_nch * (g.s16 ? 2ull : 4ull);
        sb0 += cnt;
      }
      pair_group = it->second;
      if (!strided) {
        group_segs[it->second].insert(group_segs[it->second].end(), pair_segs[pi].begin(), pair_segs[pi].end());
        pair_segs[pi].clear();
      }
    }
    if (strided) {
      // the workgroups of one segment's channel pairs read the same cache lines: emit them 8 apart
      // (workgroups b and b + 8 are dispatched to the same XCD, i.e. the same L2), 8 segments at a time
      std::vector<LgdSeg> &out = group_segs[pair_group];
      for (uint64_t base = 0; base < nseg; base += 8)
        for (size_t pi = 0; pi < ch0s.size(); ++pi)
          for (uint64_t sgi = base; sgi < std::min<uint64_t>(nseg, base + 8); ++sgi) out.push_back(pair_segs[pi][sgi]);
    }
  }
  {  // launch order: most PCM bytes first; inside a launch its groups' segments, longest first
     // (each group contiguous: lgd_tp_kernel runs per group)
    std::vector<double> bytes(c->launches.size(), 0.0);
    std::vector<double> seg_tiles(c->groups.size(), 0.0);
    for (size_t gi = 0; gi < c->groups.size(); ++gi)
      for (const LgdSeg &sg : group_segs[gi]) {
        bytes[c->groups[gi].launch] += (double)(sg.f_peak_end - sg.f0) * c->groups[gi].nch;
        seg_tiles[gi] = std::max(seg_tiles[gi], (double)(sg.n_tiles + sg.n_warm_tiles));
      }
    std::vector<size_t> order(c->launches.size());
    for (size_t li = 0; li < order.size(); ++li) order[li] = li;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return bytes[a] > bytes[b]; });
    std::vector<Launch> sorted;
    for (size_t li : order) {
      Launch L = c->launches[li];
      std::stable_sort(L.groups.begin(), L.groups.end(), [&](size_t a, size_t b) { return seg_tiles[a] > seg_tiles[b]; });
      L.seg_begin = c->segs.size();
      for (size_t gi : L.groups) {
        c->groups[gi].launch = sorted.size();
        c->groups[gi].seg_begin = c->segs.size();
        c->groups[gi].seg_count = group_segs[gi].size();
        c->segs.insert(c->segs.end(), group_segs[gi].begin(), group_segs[gi].end());
      }
      L.seg_count = c->segs.size() - L.seg_begin;
      sorted.push_back(L);
    }
    c->launches.swap(sorted);
  }
  {  // true-peak launches: the interpolating groups by kernel instance, most segments first
    std::map<std::tuple<int, int, int, int>, std::vector<size_t>> by_inst;
    for (size_t gi = 0; gi < c->groups.size(); ++gi)
      if (c->groups[gi].tp && c->groups[gi].seg_count)
        by_inst[std::make_tuple(c->groups[gi].tp_u, c->groups[gi].tp, c->groups[gi].tp_ns, c->groups[gi].s16 ? 1 : 0)].push_back(gi);
    for (const auto &kv : by_inst) {
      TpLaunch T{std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), 0, c->segs_tp.size(), 0};
      for (size_t gi : kv.second) {
        const Group &g = c->groups[gi];
        T.rows_max = std::max(T.rows_max, g.rows_max);
        c->segs_tp.insert(c->segs_tp.end(), c->segs.begin() + g.seg_begin, c->segs.begin() + g.seg_begin + g.seg_count);
      }
      T.seg_count = c->segs_tp.size() - T.seg_begin;
      c->tp_launches.push_back(T);
    }
    std::stable_sort(c->tp_launches.begin(), c->tp_launches.end(),
                     [](const TpLaunch &a, const TpLaunch &b) { return a.seg_count > b.seg_count; });
  }

  HIPCHK(hipDeviceSynchronize());  // nothing of an older plan may still be running
  c->n_sets = c->p_overlap ? ((flags & LGD_FLAG_ALBUM_PART1) ? 4 : 2) : 1;
  if (c->p_group_streams)
    for (int i = 0; i < lgd_ctx::GSTREAMS; ++i)
      if (!c->gstream[i]) HIPCHK(hipStreamCreateWithFlags(&c->gstream[i], hipStreamNonBlocking));
  if (c->p_album_slots && (uint64_t)c->p_album_slots < c->total_st)
    return fail(LGD_EINVAL, "album_slots %ld < the %llu short-term slots of this plan", c->p_album_slots,
                (unsigned long long)c->total_st);
  c->rec1_len = 4 + std::max<uint64_t>(c->total_st, (uint64_t)c->p_album_slots);
  c->cur_set = 0;
  int rc;
  if ((rc = ensure(&c->d_slices, &c->cap_slices, c->slices.size()))) return rc;
  if ((rc = ensure(&c->d_meta, &c->cap_meta, n))) return rc;
  if ((rc = ensure(&c->d_albums, &c->cap_albums, n_albums))) return rc;
  HIPCHK(hipMemcpy(c->d_albums, c->albums.data(), n_albums * sizeof(LgdAlbumMeta), hipMemcpyHostToDevice));
  std::vector<LgdRange> aranges(n_albums);
  c->ranges.resize(n);
  for (int k = 0; k < c->n_sets; ++k) {
    lgd_ctx::WorkSet &w = c->ws[k];
    w.album_pending = false;
    w.lra_base = nullptr;
    if ((rc = ensure(&w.d_E, &w.cap_E, c->total_e))) return rc;
    if ((rc = ensure(&w.d_Z, &w.cap_Z, c->total_sb))) return rc;
    if ((rc = ensure(&w.d_p1, &w.cap_p1, 4 * c->slices.size()))) return rc;
    if ((rc = ensure(&w.d_p2, &w.cap_p2, 2 * c->slices.size()))) return rc;
    if ((rc = ensure(&w.d_p2a, &w.cap_p2a, 2 * c->slices.size()))) return rc;
    if ((rc = ensure(&w.d_pmax, &w.cap_pmax, c->slices.size()))) return rc;
    if ((rc = ensure(&w.d_done, &w.cap_done, n))) return rc;
    HIPCHK(hipMemset(w.d_done, 0, (n ? n : 1) * sizeof(unsigned)));
    if ((rc = ensure(&w.d_rec1, &w.cap_rec1, (size_t)c->rec1_len))) return rc;
    HIPCHK(hipMemset(w.d_rec1, 0, (size_t)c->rec1_len * sizeof(double)));  // the padding stays 0
    w.d_st = w.d_rec1 + 4;
    if ((rc = ensure(&w.d_album, &w.cap_album, (size_t)n_albums * LGD_ALBUM_STRIDE))) return rc;
    if ((rc = ensure(&w.d_part1, &w.cap_part1, (size_t)n_albums * LGD_PART1))) return rc;
    if ((rc = ensure(&w.d_rec2, &w.cap_rec2, (size_t)n_albums * 2))) return rc;
    if ((rc = ensure(&w.d_heads, &w.cap_heads, (size_t)n_albums * 4))) return rc;
    if ((rc = ensure(&w.d_album_ranges, &w.cap_album_ranges, n_albums))) return rc;
    HIPCHK(hipMemset(w.d_album, 0, (size_t)n_albums * LGD_ALBUM_STRIDE * sizeof(double)));
    for (uint32_t a = 0; a < n_albums; ++a) {
      const LgdAlbumMeta &am = c->albums[a];
      long long slots = 0;
      for (int t = am.t0; t < am.t1; ++t) slots += c->meta[t].n_st_slots;
      aranges[a].off = am.t0 < am.t1 ? c->meta[am.t0].st_off : 0;
      aranges[a].n = slots;
      aranges[a].out = w.d_album + (size_t)a * LGD_ALBUM_STRIDE + 1;
    }
    HIPCHK(hipMemcpy(w.d_album_ranges, aranges.data(), n_albums * sizeof(LgdRange), hipMemcpyHostToDevice));
    if ((rc = ensure(&w.d_res, &w.cap_res, (size_t)n * LGR_STRIDE))) return rc;
    if ((rc = ensure(&w.d_peaks, &w.cap_peaks, c->total_peak_floats))) return rc;
    if ((rc = ensure(&w.d_segs, &w.cap_segs, c->segs.size()))) return rc;
    if ((rc = ensure(&w.d_segs_tp, &w.cap_segs_tp, c->segs_tp.size()))) return rc;
    if ((rc = ensure(&w.d_tp_rows, &w.cap_tp_rows, (size_t)c->total_tp_rows * 1024))) return rc;
    if ((rc = ensure(&w.d_hint, &w.cap_hint, (size_t)total_ch))) return rc;
    if ((rc = ensure(&w.d_ranges, &w.cap_ranges, n))) return rc;
    // the host descriptors hold offsets; each set gets its own pointers
    std::vector<LgdSeg> segs(c->segs), segs_tp(c->segs_tp);
    for (std::vector<LgdSeg> *v : {&segs, &segs_tp})
      for (LgdSeg &sg : *v) {
        sg.e_out = w.d_E + (uintptr_t)sg.e_out;
        sg.peak_out = w.d_peaks + (uintptr_t)sg.peak_out;
        sg.hint = w.d_hint + (uintptr_t)sg.hint;
        // (sg.tp_rows holds a byte offset; segments of a rate without interpolator get null)
        sg.tp_rows = sg.tp_rows == (void *)~(uintptr_t)0 ? nullptr : (void *)(w.d_tp_rows + (uintptr_t)sg.tp_rows);
        sg.filt = c->d_filt + (uintptr_t)sg.filt;
      }
    for (uint32_t t = 0; t < n; ++t) {
      c->ranges[t].off = c->meta[t].st_off;
      c->ranges[t].n = c->meta[t].n_st_slots;
      c->ranges[t].out = w.d_res + (size_t)t * LGR_STRIDE + LGR_LRA;
    }
    if (n)
      HIPCHK(hipMemcpy(w.d_ranges, c->ranges.data(), n * sizeof(LgdRange), hipMemcpyHostToDevice));
    {  // long lists -> multi-workgroup loudness-range kernels
      std::vector<int> big_tr, big_al, one{0};
      std::vector<long long> off_tr, off_al, off_one{0};
      long long need_tr = 0, need_al = 0;
      for (uint32_t t = 0; t < n; ++t)
        if (c->ranges[t].n > LGD_LRA_BIG) { big_tr.push_back((int)t); off_tr.push_back(need_tr); need_tr += 2 * c->ranges[t].n; }
      for (uint32_t a = 0; a < n_albums; ++a)
        if (aranges[a].n > LGD_LRA_BIG) { big_al.push_back((int)a); off_al.push_back(need_al); need_al += 2 * aranges[a].n; }
      // multi-GPU form: one gathered list of world x rec1_len entries; room for "album_world" ranks
      const long long need_dist = (flags & LGD_FLAG_ALBUM_PART1) ? 2ll * c->p_album_world * (long long)c->rec1_len : 0;
      w.lra_dist_cap = (flags & LGD_FLAG_ALBUM_PART1) ? (uint64_t)c->p_album_world * c->rec1_len : 0;
      w.n_big_tr = (int)big_tr.size();
      w.n_big_al = (int)big_al.size();
      const size_t nb = std::max<size_t>(std::max(big_tr.size(), big_al.size()), need_dist ? 1 : 0);
      if ((rc = ensure(&w.d_big_tr, &w.cap_big_tr, big_tr.size()))) return rc;
      if ((rc = ensure(&w.d_big_al, &w.cap_big_al, big_al.size()))) return rc;
      if ((rc = ensure(&w.d_big_one, &w.cap_big_one, 1))) return rc;
      if ((rc = ensure(&w.d_off_tr, &w.cap_off_tr, off_tr.size()))) return rc;
      if ((rc = ensure(&w.d_off_al, &w.cap_off_al, off_al.size()))) return rc;
      if ((rc = ensure(&w.d_off_one, &w.cap_off_one, 1))) return rc;
      if (nb) {
        if ((rc = ensure(&w.d_lra_hist, &w.cap_lra_hist, nb * 65536 * 8))) return rc;  // LGD_LRB_REP copies
        if ((rc = ensure(&w.d_lra_part, &w.cap_lra_part, nb * 64 * 4 + nb * 64))) return rc;  // partials + block sums
        if ((rc = ensure(&w.d_lra_picks, &w.cap_lra_picks, nb * lgd_lra_pick_bytes()))) return rc;
        if ((rc = ensure(&w.d_lra_cand, &w.cap_lra_cand, (size_t)std::max(std::max(need_tr, need_al), need_dist)))) return rc;
      }
      if (!big_tr.empty()) {
        HIPCHK(hipMemcpy(w.d_big_tr, big_tr.data(), big_tr.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(w.d_off_tr, off_tr.data(), off_tr.size() * sizeof(long long), hipMemcpyHostToDevice));
      }
      if (!big_al.empty()) {
        HIPCHK(hipMemcpy(w.d_big_al, big_al.data(), big_al.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(w.d_off_al, off_al.data(), off_al.size() * sizeof(long long), hipMemcpyHostToDevice));
      }
      HIPCHK(hipMemcpy(w.d_big_one, one.data(), sizeof(int), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(w.d_off_one, off_one.data(), sizeof(long long), hipMemcpyHostToDevice));
    }
    if (!segs.empty())
      HIPCHK(hipMemcpy(w.d_segs, segs.data(), segs.size() * sizeof(LgdSeg), hipMemcpyHostToDevice));
    if (!segs_tp.empty())
      HIPCHK(hipMemcpy(w.d_segs_tp, segs_tp.data(), segs_tp.size() * sizeof(LgdSeg), hipMemcpyHostToDevice));
  }
  if (n)
    HIPCHK(hipMemcpy(c->d_meta, c->meta.data(), n * sizeof(LgdTrackMeta), hipMemcpyHostToDevice));
  if (!c->slices.empty())
    HIPCHK(hipMemcpy(c->d_slices, c->slices.data(), c->slices.size() * sizeof(LgdSlice),
                     hipMemcpyHostToDevice));
  if (c->groups.size() > MAX_GROUPS)
    return fail(LGD_EUNSUP, "more than %zu distinct (rate, channels) pairs in one plan", MAX_GROUPS);
  for (size_t gi = 0; gi < c->groups.size(); ++gi)
    HIPCHK(hipMemcpy(c->d_filt + gi, &c->groups[gi].F, sizeof(LgdFilt), hipMemcpyHostToDevice));
  c->planned = true;
  return LGD_OK;
}

// Album stages.  Multi-GPU form (LGD_FLAG_ALBUM_PART1, one album): all1 = records 1 of all
// ranks (world * rec1_len doubles, writable; null = this rank alone), whose heads hold
// the partial sums and whose remainder becomes the short-term list.  Single-GPU form
// (LGD_FLAG_ALBUM, any number of albums): heads in their own array, the short-term
// lists are the albums' slices of this set's d_st.
static int album_stage2_on(lgd_ctx *c, lgd_ctx::WorkSet &w, double *all1, uint32_t world,
                           hipStream_t s) {
  const bool dist = (c->flags & LGD_FLAG_ALBUM_PART1) != 0;
  double *heads = dist ? (all1 ? all1 : w.d_rec1) : w.d_heads;
  if (!dist || !all1) world = 1;
  HIPCHK(lgd_launch_album_stage2(c->d_slices, (int)c->slices.size(), c->d_meta, w.d_Z, w.d_p1,
                                 w.d_p2a, c->d_albums, (int)c->albums.size(), heads, (int)world,
                                 (long long)c->rec1_len, w.d_part1, w.d_rec2, c->abs_gate,
                                 c->rel_factor, s));
  w.lra_base = dist ? heads : w.d_st;
  w.lra_n = (uint64_t)world * c->rec1_len;
  return LGD_OK;
}

static int album_stage3_on(lgd_ctx *c, lgd_ctx::WorkSet &w, const double *all2, uint32_t world,
                           hipStream_t s) {
  if (!w.lra_base) return fail(LGD_ESTATE, "album stage 3 before stage 2");
  const bool dist = (c->flags & LGD_FLAG_ALBUM_PART1) != 0;
  if (!dist || !all2) {
    all2 = w.d_rec2;
    world = 1;
  }
  const int n_albums = (int)c->albums.size();
  HIPCHK(lgd_launch_album_final(w.d_part1, all2, (int)world, n_albums, c->rel_factor, w.d_album, s));
  if (dist) {
    // (same values on every use of a plan, so a host that runs ahead of the copies is harmless)
    w.h_album_range->off = 0;
    w.h_album_range->n = (long long)w.lra_n;
    w.h_album_range->out = w.d_album + 1;
    HIPCHK(hipMemcpyAsync(w.d_album_range, w.h_album_range, sizeof(LgdRange), hipMemcpyHostToDevice, s));
    const int big = (w.lra_n > LGD_LRA_BIG && w.lra_n <= w.lra_dist_cap) ? 1 : 0;
    HIPCHK(lgd_launch_lra(w.d_album_range, 1, w.lra_base, c->minus20, w.d_big_one, big, w.d_lra_hist,
                          w.d_lra_part, w.d_lra_picks, w.d_lra_cand, w.d_off_one, 1, s));
  } else {
    HIPCHK(lgd_launch_lra(w.d_album_ranges, n_albums, w.d_st, c->minus20, w.d_big_al, w.n_big_al,
                          w.d_lra_hist, w.d_lra_part, w.d_lra_picks, w.d_lra_cand, w.d_off_al, 1, s));
  }
  return LGD_OK;
}

extern "C" int lgd_album_stage2(lgd_ctx *c, double *all_rec1, uint32_t world, void *hip_stream) {
  if (!c || !c->planned || !c->executed) return fail(LGD_ESTATE, "album stage 2 before execute");
  if (all_rec1 && world == 0) return fail(LGD_EINVAL, "album stage 2: world 0");
  return album_stage2_on(c, c->ws[c->cur_set], all_rec1, world, (hipStream_t)hip_stream);
}

extern "C" int lgd_album_stage3(lgd_ctx *c, const double *all_rec2, uint32_t world,
                                void *hip_stream) {
  if (!c || !c->planned || !c->executed) return fail(LGD_ESTATE, "album stage 3 before execute");
  if (all_rec2 && world == 0) return fail(LGD_EINVAL, "album stage 3: world 0");
  lgd_ctx::WorkSet &w = c->ws[c->cur_set];
  int rc = album_stage3_on(c, w, all_rec2, world, (hipStream_t)hip_stream);
  if (rc) return rc;
  // the next scan into this set must not start before the album of this one is done
  HIPCHK(hipEventRecord(w.ev_album, (hipStream_t)hip_stream));
  w.album_pending = true;
  return LGD_OK;
}

extern "C" int lgd_album_join(lgd_ctx *c, void *hip_stream) {
  if (!c || !c->planned || !c->executed) return fail(LGD_ESTATE, "album join before execute");
  HIPCHK(hipStreamWaitEvent((hipStream_t)hip_stream, c->ws[c->cur_set].ev_done, 0));
  return LGD_OK;
}

// Order `hip_stream` behind every scan enqueued so far (with "overlap" 1 a scan may still be
// reading its PCM on the internal stream after the caller's stream has drained).
extern "C" int lgd_join(lgd_ctx *c, void *hip_stream) {
  if (!c) return fail(LGD_EINVAL, "lgd_join: null context");
  if (!c->executed || c->n_sets < 2) return LGD_OK;  // everything ran on the caller's stream
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipEventRecord(c->ev_join, c->side));
  HIPCHK(hipStreamWaitEvent((hipStream_t)hip_stream, c->ev_join, 0));
  return LGD_OK;
}

extern "C" int lgd_execute(lgd_ctx *c, void *hip_stream) {
  if (!c || !c->planned) return fail(LGD_ESTATE, "lgd_execute before lgd_plan");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t caller = (hipStream_t)hip_stream;
  const int n = (int)c->tracks.size();
  c->last_stream = caller;
  const int k = c->executed ? (c->cur_set + 1) % c->n_sets : 0;
  lgd_ctx::WorkSet &w = c->ws[k];
  c->cur_set = k;
  // Two work sets: whole scans alternate between the caller's stream and an internal one.
  // Consecutive scans are independent (own workspace), so nothing orders them: the
  // next scan's workgroups fill the GPU as the previous scan's drain, and its small
  // gating / LRA kernels run beside the following scan -- no event packets between
  // the dominant kernels.  Results are defined after lgd_fetch (which joins both).
  // (Measured and dropped, round 3: scans alternating between two internal streams with every epilogue on a third
  // one, four work sets -- 0.282 ms per step against 0.285 for this form when all streams sat on hardware queues
  // of their own, 0.309 when two of them shared one: HIP deals a process's streams over four hardware queues, and
  // which stream lands where is not ours to choose.  What this form really does: scan k + 1, on the internal
  // stream, waits for everything enqueued before it on the caller's stream, i.e. for epilogue k - 1, and scan
  // k + 2 follows that epilogue too -- scans run in pairs that share the GPU, then their two epilogues.)
  hipStream_t s = caller;
  if (c->n_sets >= 2 && (k & 1)) {
    s = c->side;
    // everything the caller enqueued before this call (e.g. the PCM upload) comes first
    HIPCHK(hipEventRecord(w.ev_scan, caller));
    HIPCHK(hipStreamWaitEvent(s, w.ev_scan, 0));
  }
  if (w.album_pending) {  // a caller-driven album reduction still reads this set
    HIPCHK(hipStreamWaitEvent(s, w.ev_album, 0));
    w.album_pending = false;
  }
  hipEvent_t *ev = c->ev[c->n_exec % lgd_ctx::EV_RING];
  if (c->p_timing) HIPCHK(hipEventRecord(ev[0], s));
  // One scan launch per (rate, channels) group, largest first.  "group_streams" 1 sends the
  // groups of a mixed plan (C5: twelve) out on up to four streams at once so that a group's
  // workgroups fill the CUs the previous one leaves idle in its last round -- measured on C5:
  // 2.88 ms instead of 2.65 ms for the scan kernels (kernels of different LDS footprints share
  // CUs badly, and every cross-stream dependency costs its latency), hence off by default; every
  // group is sized to fill the GPU by itself instead (see the segment lengths in lgd_plan).
  const size_t ng = c->launches.size();
  const int n_side = (c->p_group_streams && ng > 1 && c->gstream[0]) ? (int)std::min<size_t>(ng - 1, lgd_ctx::GSTREAMS) : 0;
  if (n_side) {
    HIPCHK(hipEventRecord(c->ev_fork, s));
    for (int i = 0; i < n_side; ++i) HIPCHK(hipStreamWaitEvent(c->gstream[i], c->ev_fork, 0));
  }
  for (size_t k = 0; k < ng; ++k) {
    const Launch &L = c->launches[k];
    const int lane = n_side ? (int)(k % (size_t)(n_side + 1)) : 0;
    hipStream_t gs = lane == 0 ? s : c->gstream[lane - 1];
    HIPCHK(lgd_launch_scan(L.chunk, L.nch, L.tp ? 4 : 0, L.mode, L.s16, w.d_segs + L.seg_begin, (int)L.seg_count, gs));
  }
  for (int i = 0; i < n_side; ++i) {
    HIPCHK(hipEventRecord(c->ev_gjoin[i], c->gstream[i]));
    HIPCHK(hipStreamWaitEvent(s, c->ev_gjoin[i], 0));
  }
  // (the marker between the scan kernels and the true-peak kernels costs ~5 us of queue time -- an event packet
  // drains the queue -- and sits inside the bracket the roofline is quoted on: only with "timing" 2)
  if (c->p_timing >= 2) HIPCHK(hipEventRecord(ev[3], s));
  c->ev3_valid[c->n_exec % lgd_ctx::EV_RING] = c->p_timing >= 2;
  // per-channel sample peaks of the whole tracks, then the interpolator over the chunks that
  // can exceed them
  if (c->flags & LGD_FLAG_TRUE_PEAK)
    HIPCHK(lgd_launch_peak_reduce(c->d_meta, n, w.d_peaks, w.d_hint, s));
  // One true-peak launch per kernel instance (window step, interpolation factor, slab steps): chunk length,
  // channel count and constants travel with the segments.  (Sending the launches of a mixed plan out on
  // four streams at once was measured on C5: 2.563 -> 2.546 ms of kernel time, but the event packets
  // cost the pipelined step 2.40 -> 2.62 ms: not done.)
#ifndef LGD_FUSED_TP  // (experiment build: the scan kernel has evaluated the interpolator itself, make libloudscan_hip_fused.so)
  if (c->flags & LGD_FLAG_TRUE_PEAK)
    for (const TpLaunch &T : c->tp_launches)
      HIPCHK(lgd_launch_tp(T.u, T.tp, T.ns, T.s16, w.d_segs_tp + T.seg_begin, (int)T.seg_count, T.rows_max, s));
#endif
  if (c->p_timing) HIPCHK(hipEventRecord(ev[1], s));
  // gating pass 1, pass 2, result records and loudness ranges of all tracks: one launch (the long short-term
  // lists, > LGD_LRA_BIG entries, through the multi-workgroup kernels behind it)
  HIPCHK(lgd_launch_track_epilogue(c->d_slices, (int)c->slices.size(), c->d_meta, n, w.d_E, w.d_Z,
                                   w.d_st, w.d_peaks, w.d_p1, w.d_p2, w.d_pmax, w.d_res, w.d_done, c->abs_gate,
                                   c->rel_factor, c->minus20, (c->flags & LGD_FLAG_TRUE_PEAK) ? 1 : 0,
                                   w.n_big_tr > 0 ? 1 : 0, s));
  if (w.n_big_tr > 0)
    HIPCHK(lgd_launch_lra(w.d_ranges, n, w.d_st, c->minus20, w.d_big_tr, w.n_big_tr, w.d_lra_hist,
                          w.d_lra_part, w.d_lra_picks, w.d_lra_cand, w.d_off_tr, 0, s));
  c->executed = true;
  if (c->flags & (LGD_FLAG_ALBUM | LGD_FLAG_ALBUM_PART1))
    HIPCHK(lgd_launch_album_part1(w.d_res, c->d_albums, (int)c->albums.size(),
                                  (c->flags & LGD_FLAG_ALBUM_PART1) ? w.d_rec1 : w.d_heads, s));
  if (c->flags & LGD_FLAG_ALBUM) {
    int rc;
    if ((rc = album_stage2_on(c, w, nullptr, 1, s))) return rc;
    if ((rc = album_stage3_on(c, w, nullptr, 1, s))) return rc;
  }
  if (c->p_timing) {
    HIPCHK(hipEventRecord(ev[2], s));
    ++c->n_exec;
  }
  if (c->flags & LGD_FLAG_ALBUM_PART1) HIPCHK(hipEventRecord(w.ev_done, s));  // for lgd_album_join
  return LGD_OK;
}

// both streams of the last scans
static int sync_all(lgd_ctx *c) {
  HIPCHK(hipStreamSynchronize(c->last_stream));
  if (c->n_sets >= 2) HIPCHK(hipStreamSynchronize(c->side));
  for (int k = 0; k < c->n_sets; ++k)
    if (c->ws[k].album_pending) HIPCHK(hipEventSynchronize(c->ws[k].ev_album));
  return LGD_OK;
}

extern "C" int lgd_fetch(lgd_ctx *c, lgd_track_result *out, lgd_album_result *album) {
  if (!c || !c->executed) return fail(LGD_ESTATE, "lgd_fetch before lgd_execute");
  HIPCHK(hipSetDevice(c->device));
  int rc;
  if ((rc = sync_all(c))) return rc;
  const lgd_ctx::WorkSet &w = c->ws[c->cur_set];
  const size_t n = c->tracks.size();
  if (n && out) {
    std::vector<double> r(n * LGR_STRIDE);
    HIPCHK(hipMemcpy(r.data(), w.d_res, r.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t t = 0; t < n; ++t) {
      const double *p = &r[t * LGR_STRIDE];
      lgd_track_result &o = out[t];
      o.loudness = p[LGR_LOUDNESS];
      o.lra = p[LGR_LRA];
      o.peak = p[LGR_PEAK];
      o.sample_peak = p[LGR_SPEAK];
      o.true_peak = p[LGR_TPEAK];
      o.rel_threshold = p[LGR_THR];
      o.sum_abs = p[LGR_SUM_ABS];
      o.sum_rel = p[LGR_SUM_REL];
      o.n_blocks = (uint64_t)p[LGR_NBLK];
      o.n_abs = (uint64_t)p[LGR_NABS];
      o.n_rel = (uint64_t)p[LGR_NREL];
      o.n_st_blocks = (uint64_t)p[LGR_NSTBLK];
      o.n_st = (uint64_t)p[LGR_NST];
      o.max_momentary = p[LGR_MAX_M];
      o.max_shortterm = p[LGR_MAX_S];
    }
  }
  if (album) {
    if (!(c->flags & (LGD_FLAG_ALBUM | LGD_FLAG_ALBUM_PART1)))
      return fail(LGD_ESTATE, "plan was made without an album flag");
    const size_t na = c->albums.size();
    std::vector<double> ab(na * LGD_ALBUM_STRIDE);
    HIPCHK(hipMemcpy(ab.data(), w.d_album, ab.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < na; ++i) {
      const double *a = &ab[i * LGD_ALBUM_STRIDE];
      lgd_album_result &o = album[i];
      o.loudness = a[0];
      o.lra = a[1];
      o.peak = a[2];
      o.rel_threshold = a[3];
      o.sum_abs = a[4];
      o.sum_rel = a[5];
      o.n_abs = (uint64_t)a[6];
      o.n_rel = (uint64_t)a[7];
      o.n_st = (uint64_t)a[8];
      o.ranks_stage2 = (uint32_t)a[9];
      o.ranks_with_content = (uint32_t)a[10];
      o.ranks_stage3 = (uint32_t)a[11];
      o.reserved = 0;
    }
  }
  return LGD_OK;
}

extern "C" int lgd_album_record1(lgd_ctx *c, double **p, uint64_t *n_doubles) {
  if (!c || !p || !n_doubles) return fail(LGD_EINVAL, "null argument");
  if (!c->planned) return fail(LGD_ESTATE, "no plan");
  *p = c->ws[c->cur_set].d_rec1;
  *n_doubles = c->rec1_len;
  return LGD_OK;
}
extern "C" int lgd_album_record2(lgd_ctx *c, double **p) {
  if (!c || !p) return fail(LGD_EINVAL, "null argument");
  *p = c->ws[c->cur_set].d_rec2;
  return LGD_OK;
}

extern "C" int lgd_copy_subblock_energies(lgd_ctx *c, uint32_t track, double *host_out,
                                          uint64_t cap, uint64_t *n_out) {
  if (!c || !c->executed) return fail(LGD_ESTATE, "no executed plan");
  if (track >= c->tracks.size()) return fail(LGD_EINVAL, "track index too high");
  int rc;
  if ((rc = sync_all(c))) return rc;
  const LgdTrackMeta &m = c->meta[track];
  const uint64_t n = std::min<uint64_t>(cap, (uint64_t)m.n_sb);
  if (n_out) *n_out = (uint64_t)m.n_sb;
  if (n && host_out) {
    // weighted channel sum, as the gating kernel forms it
    std::vector<double> tmp((size_t)m.n_sb * m.nch);
    HIPCHK(hipMemcpy(tmp.data(), c->ws[c->cur_set].d_E + m.e_off, tmp.size() * sizeof(double),
                     hipMemcpyDeviceToHost));
    for (uint64_t j = 0; j < n; ++j) {
      double s = 0.0;
      for (int ch = 0; ch < m.nch; ++ch) {
        const double w = lgd_channel_weight(ch, m.nch);
        if (w != 0.0) s += w * tmp[(size_t)ch * m.n_sb + j];
      }
      host_out[j] = s;
    }
  }
  return LGD_OK;
}

extern "C" hipError_t lgd_launch_s16_to_f32(const short *in, float *out, size_t n, hipStream_t s);
extern "C" int lgd_convert_s16(const short *dev_in, float *dev_out, uint64_t n, void *hip_stream) {
  if (n && (!dev_in || !dev_out)) return fail(LGD_EINVAL, "lgd_convert_s16: null pointer");
  if (n) HIPCHK(lgd_launch_s16_to_f32(dev_in, dev_out, (size_t)n, (hipStream_t)hip_stream));
  return LGD_OK;
}

extern "C" int lgd_copy_channel_peaks(lgd_ctx *c, uint32_t track, double *sample_peak,
                                      double *true_peak, uint32_t cap) {
  if (!c || !c->executed) return fail(LGD_ESTATE, "lgd_copy_channel_peaks before lgd_execute");
  if (track >= c->tracks.size()) return fail(LGD_EINVAL, "track index out of range");
  const LgdTrackMeta &m = c->meta[track];
  if (cap < (uint32_t)m.nch) return fail(LGD_EINVAL, "need room for %d channels", m.nch);
  HIPCHK(hipSetDevice(c->device));
  int rc;
  if ((rc = sync_all(c))) return rc;
  std::vector<float> pk((size_t)m.n_seg * 2 * m.nch);
  if (!pk.empty())
    HIPCHK(hipMemcpy(pk.data(), c->ws[c->cur_set].d_peaks + m.peak_off, pk.size() * sizeof(float),
                     hipMemcpyDeviceToHost));
  const bool tp = (c->flags & LGD_FLAG_TRUE_PEAK) != 0;
  for (int ch = 0; ch < m.nch; ++ch) {
    double sp = 0.0, t = 0.0;
    for (int sg = 0; sg < m.n_seg; ++sg) {
      const float *pp = &pk[(size_t)sg * 2 * m.nch];
      sp = std::max(sp, (double)pp[ch]);
      t = std::max(t, (double)pp[m.nch + ch]);
    }
    if (sample_peak) sample_peak[ch] = sp;
    // ebur128_true_peak's value, max(interpolated, sample): the kernel skips interpolator
    // windows that cannot exceed the peak already found, so only this maximum is defined
    if (true_peak) true_peak[ch] = tp ? std::max(t, sp) : 0.0;
  }
  return LGD_OK;
}

extern "C" int lgd_last_kernel_ms(lgd_ctx *c, float *scan_ms, float *total_ms) {
  if (!c || !c->n_exec) return fail(LGD_ESTATE, "no executed plan");
  hipEvent_t *ev = c->ev[(c->n_exec - 1) % lgd_ctx::EV_RING];
  HIPCHK(hipEventSynchronize(ev[2]));
  if (scan_ms) HIPCHK(hipEventElapsedTime(scan_ms, ev[0], ev[1]));
  if (total_ms) HIPCHK(hipEventElapsedTime(total_ms, ev[0], ev[2]));
  return LGD_OK;
}

extern "C" int lgd_kernel_ms_stats(lgd_ctx *c, uint32_t last_n, float *scan_mean, float *scan_min,
                                   float *total_mean, uint32_t *n_used) {
  if (!c || !c->n_exec) return fail(LGD_ESTATE, "no executed plan");
  uint64_t n = std::min<uint64_t>(std::min<uint64_t>(last_n ? last_n : lgd_ctx::EV_RING,
                                                     lgd_ctx::EV_RING), c->n_exec);
  double sa = 0, ta = 0;
  float smin = 1e30f;
  for (uint64_t i = 0; i < n; ++i) {
    hipEvent_t *ev = c->ev[(c->n_exec - 1 - i) % lgd_ctx::EV_RING];
    float a = 0, b = 0;
    HIPCHK(hipEventSynchronize(ev[2]));
    HIPCHK(hipEventElapsedTime(&a, ev[0], ev[1]));
    HIPCHK(hipEventElapsedTime(&b, ev[0], ev[2]));
    sa += a; ta += b;
    smin = std::min(smin, a);
  }
  if (scan_mean) *scan_mean = (float)(sa / (double)n);
  if (scan_min) *scan_min = smin;
  if (total_mean) *total_mean = (float)(ta / (double)n);
  if (n_used) *n_used = (uint32_t)n;
  return LGD_OK;
}

// the scan kernels alone (without the true-peak kernel that lgd_kernel_ms_stats' figure includes)
extern "C" int lgd_scan_only_ms_stats(lgd_ctx *c, uint32_t last_n, float *mean, float *min_ms) {
  if (!c || !c->n_exec) return fail(LGD_ESTATE, "no executed plan");
  uint64_t n = std::min<uint64_t>(std::min<uint64_t>(last_n ? last_n : lgd_ctx::EV_RING,
                                                     lgd_ctx::EV_RING), c->n_exec);
  double sa = 0;
  float smin = 1e30f;
  for (uint64_t i = 0; i < n; ++i) {
    hipEvent_t *ev = c->ev[(c->n_exec - 1 - i) % lgd_ctx::EV_RING];
    float a = 0;
    if (!c->ev3_valid[(c->n_exec - 1 - i) % lgd_ctx::EV_RING])
      return fail(LGD_ESTATE, "the scan-kernel marker is only recorded with \"timing\" 2");
    HIPCHK(hipEventSynchronize(ev[2]));
    HIPCHK(hipEventElapsedTime(&a, ev[0], ev[3]));
    sa += a;
    smin = std::min(smin, a);
  }
  if (mean) *mean = (float)(sa / (double)n);
  if (min_ms) *min_ms = smin;
  return LGD_OK;
}

extern "C" int lgd_plan_info(lgd_ctx *c, uint64_t *n_segments, uint64_t *n_subblocks,
                             uint32_t *chunk, uint64_t *pcm_bytes, uint64_t *warm_bytes) {
  if (!c || !c->planned) return fail(LGD_ESTATE, "no plan");
  if (n_segments) *n_segments = c->segs.size();
  if (n_subblocks) *n_subblocks = c->total_sb;
  if (chunk) *chunk = c->groups.empty() ? 0u : (uint32_t)c->groups[0].chunk;
  if (pcm_bytes) *pcm_bytes = c->pcm_bytes;
  if (warm_bytes) *warm_bytes = c->warm_bytes;
  return LGD_OK;
}

#ifdef LGD_DEBUG_HWID
// placement probe build only: the chunk-maxima rows of the last scan (the scan kernel left its
// per-wave HW_ID records in them)
extern "C" int lgd_debug_rows(lgd_ctx *c, void *dst, size_t bytes) {
  if (!c || !c->executed) return -1;
  (void)hipDeviceSynchronize();
  lgd_ctx::WorkSet &w = c->ws[c->cur_set];
  return (int)hipMemcpy(dst, w.d_tp_rows, bytes, hipMemcpyDeviceToHost);
}
#endif

