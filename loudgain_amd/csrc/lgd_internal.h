// lgd_internal.h -- structures shared by the host engine and the HIP kernels.
#pragma once
#include <stdint.h>

// One unit of work for one wavefront: a run of whole 100 ms sub-blocks of one
// track (plus, for the last segment of a track, the trailing partial sub-block,
// which only feeds the peaks -- SURVEY.md A.3: leftover frames are filtered
// but never form a block).
struct LgdSeg {
  const float *pcm;      // track base (interleaved f32 -- or int16 for the S16 kernel variants --, 16-B aligned)
  long long n_floats;    // frames * channels of the whole track
  long long f0;          // first frame of this segment (multiple of s100)
  long long f_peak_end;  // tiles cover [f0, f_peak_end)
  double *e_out;         // sub-block energies of this segment: channel ch at
                         // e_out[ch * e_ch_stride + 0 .. n_sb) (unweighted sum y^2)
  float *peak_out;       // [2][nch]: sample peaks, then interpolated peaks
  int n_sb;              // whole sub-blocks in this segment
  int n_warm_tiles;      // K-filter warm-up tiles run before f0 (0 at track start)
  int e_ch_stride;       // whole sub-blocks of the track (channel stride in E)
  int ch0;               // first channel this workgroup handles (channel groups of
  int nch_total;         // streams with > 16 channels; otherwise 0 and the channel count)
  int n_tiles;           // tiles of 64 * chunk frames that cover [f0, f_peak_end)
  long long n_frames;    // frames of the whole track (n_floats / channels of the stream)
  void *tp_rows;         // chunk bounds for lgd_tp_kernel: per group of 8 tiles and channel ch of this
                         // workgroup 64 x 16 bytes at byte offset ((k / 8) * nch + ch) * 1024: lane l's
                         // 8 bf16 values (bound on every interpolator output of its chunk's frames in each
                         // tile, rounded up; a group of n tiles fills the top n slots); null without interpolator
  float *hint;           // [nch_total] this track's per-channel sample peak (lgd_peak_reduce_kernel): the
                         // bound lgd_tp_kernel prunes with
  const void *filt;      // the LgdFilt of this segment's (rate, chunk) in device memory: per segment, so that
                         // one scan launch can carry the segments of several sample rates
  // what lgd_tp_kernel needs of the segment's group, here so that its short way to the early exit does not
  // depend on a second (dependent) load: one true-peak launch carries the segments of several groups
  int chunk;             // C: frames per lane and tile of the scan kernel that wrote tp_rows
  int nch_wg;            // channels (waves) per scan workgroup = rows per tile in tp_rows
  unsigned magic_nch;    // row / nch_wg == umulhi(row, magic_nch) (0: nch_wg == 1)
  unsigned magic_ns;     // v / (chunk / U) == (v * magic_ns) >> 20 for v < 8 * (chunk / U) + 64
  unsigned magic_c;      // f / chunk == (f * magic_c) >> 20 for the frame offsets inside a row
  unsigned skip_mask;    // strided sets that overlap their neighbour: bit c = this workgroup's channel c belongs to the
                         // neighbour (no filtering, no peaks, empty true-peak records)
};

// Per-(rate, chunk) constants, passed by value as a kernel argument.
// Filter chain: y = pb(z) / pa(z) * (1 - z^-1)^2 / ra(z) x.
// Scan basis: s = (q1, alpha (q1 - beta q2), gamma p1, gamma p2), gamma = 1/dc.
struct LgdFilt {
  double ra[2];      // RLB denominator a1, a2
  double pa[2];      // shelf denominator a1, a2
  double pb[3];      // shelf numerator
  double pbn[2];     // pb1/pb0, pb2/pb0: the kernel forms y/pb0 ...
  double pb0sq;      // ... and scales each 100 ms sum by pb0^2
  double alpha, beta, inv_alpha, inv_beta;  // slope coordinate of the RLB pole pair
  double gamma, dc;  // shelf-state scaling (dc = 1 / (1 + pa1 + pa2))
  double gC[2][4];   // effect of w[0], w[1] of a chunk (C frames) on its end state, scan basis
  double P[6][16];   // transition over C * 2^j frames, j = 0..5 (scan basis, row-major;
                     // block lower triangular: P[.][2], [3], [6], [7] are zero)
  float tp[36];      // 4x: [0..11] phase 1, [12..17] phase 2 (first half), [18..29] halved (sum, difference) pairs of
                     // the mirrored phases 1 / 3; 2x: [0..11] the phase's first half (A.5).  [30..33]: the adjacent-pair
                     // bound (a2, b2, a1, b1): every output of a window is <= max(a2 S2 + b2 M, a1 S2 + b1 M), M = the
                     // window's largest |x|, S2 = the largest |x[j]| + |x[j+1]| over the pairs under the two centre taps
                     // [34]: 1 / (1 + margin) that covers the fp32 roundings of the interpolator and of this bound
  float tp_thr;      // true-peak pruning: a window whose largest |x| is <= tp_thr * (peak found so
                     // far) cannot raise the peak (tp_thr < 1 / max_phase sum |c|, rounding included)
  int tp_prune;      // 0: every window is evaluated (reference mode of the pruning tests)
  int lps;           // lanes (C-frame chunks) per 100 ms sub-block = s100 / C
  int pad;           // debug builds: floor-measurement mode bits
  int tp_dense_min;  // lgd_tp_kernel: rows with at least this many flagged chunks (of 64) are walked as a whole
  int tp_hx;         // frames of history an interpolator output reads: 11 (4x), 23 (2x), 0 (none)
  int pskip;         // bit j: P[j]'s shelf block (entries 10,11,14,15) is < 1e-19 -> skipped
  int s100;          // frames per 100 ms sub-block (generic kernel: any alignment)
};

// frames per streamed step of the scan kernel: the largest divisor of C not above 8
constexpr int lgd_unroll(int C) {
  return (C % 8 == 0) ? 8 : (C % 7 == 0) ? 7 : (C % 6 == 0) ? 6 : (C % 5 == 0) ? 5
       : (C % 4 == 0) ? 4 : (C % 3 == 0) ? 3 : (C % 2 == 0) ? 2 : 1;
}
// The gating epilogue walks the 400 ms blocks of a track in slices of this many
// blocks, one workgroup per slice (fixed size -> fixed, reproducible summation
// tree, independent of how the scan kernel was segmented).
#define LGD_SLICE 1024

struct LgdSlice {
  int track;
  int j0;             // first 400 ms block of the slice
};

struct LgdTrackMeta {
  long long e_off;    // first slot of this track in the sub-block energy array E:
                      // channel ch, sub-block j at E[e_off + ch * n_sb + j]
  long long sb_off;   // first slot of its 400 ms block energies in the Z array
  long long st_off;   // first slot in the short-term energy array
  long long peak_off; // first float of this track's [n_seg][2][nch] peak partials
  int n_sb;           // whole sub-blocks
  int n_st_slots;     // short-term blocks evaluated: n_sb >= 30 ? (n_sb-30)/10+1 : 0
  int n_seg;
  int s100;
  int nch;
  int slice_off;      // first epilogue slice of this track
  int n_slices;       // ceil((n_sb - 3) / LGD_SLICE), 0 if n_sb < 4
  int album;          // album of this track (albums are runs of consecutive tracks)
  int hint_off;       // first of this track's nch channels in the per-channel peak array (LgdSeg::hint)
  int pad;
};

// one album of a plan: tracks [t0, t1), their epilogue slices [slice0, slice1)
struct LgdAlbumMeta {
  int t0, t1;
  int slice0, slice1;
};

// one loudness-range problem: listed short-term energies st[off, off+n) -> *out
struct LgdRange {
  long long off;
  long long n;
  double *out;
};

// loudness-range lists longer than this take the multi-workgroup kernels (lgd_lra_big_*)
#define LGD_LRA_BIG 8192

// libebur128's default channel map by index: weight of channel
// `ch` of an `nch`-channel stream; 0 = EBUR128_UNUSED (peaks only, e.g. LFE).
#if defined(__HIPCC__)
#define LGD_HD __host__ __device__
#else
#define LGD_HD
#endif
static inline LGD_HD double lgd_channel_weight(int ch, int nch) {
  if (nch == 4) return ch < 2 ? 1.0 : 1.41;
  if (nch == 5) return ch < 3 ? 1.0 : 1.41;
  if (ch < 3) return 1.0;
  if (ch == 4 || ch == 5) return 1.41;
  return 0.0;
}

#define LGD_ALBUM_STRIDE 16  // doubles per album result record (12 used)
#define LGD_PART1 6          // doubles per folded album head: sum_abs, n_abs, peak, n_st, heads folded, heads with content

// per-track device result: 16 doubles (counts are integer-valued doubles)
enum {
  LGR_LOUDNESS = 0, LGR_LRA, LGR_PEAK, LGR_SPEAK, LGR_TPEAK, LGR_THR, LGR_SUM_ABS,
  LGR_SUM_REL, LGR_NBLK, LGR_NABS, LGR_NREL, LGR_NSTBLK, LGR_NST, LGR_MAX_M,
  LGR_MAX_S, LGR_SPARE, LGR_STRIDE
};
