// lgd_epilogue.hip -- gating, loudness-range and album kernels (gfx950).
//
// They replace the result queries that /root/reference/src/scan.c makes into
// libebur128: E5/E6 block lists (ebur128_calc_gating_block /
// energy_shortterm, reached from scan.c:448), E7 ebur128_loudness_global[_multiple]
// (scan.c:294,383), E8 ebur128_loudness_range[_multiple] (scan.c:297,388), E9
// ebur128_true_peak (scan.c:303,371) and the album peak loop (scan.c:359-378).
//
// Input: the 100 ms sub-block energies E[] and per-segment peaks written by
// lgd_scan_kernel.  Everything is fp64; sums run in a fixed order (fixed slice
// size, strided per-thread partials, fixed trees) so results are reproducible
// and do not depend on how the scan was segmented.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

// The epilogue kernels are a few workgroups that run beside the next scan's
// full grid (lgd_execute pipelines scans): give their waves issue priority so the
// latency-bound reductions do not crawl behind 2000 streaming waves.
#define LGD_EPI_PRIO() __builtin_amdgcn_s_setprio(3)

#include "lgd_internal.h"

#define LGD_WAVE 64
#define LGD_EPI_NT 256

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, LGD_WAVE);
  return v;
}

template <int NT>
__device__ __forceinline__ double block_sum_f64(double v, double *sh) {
  v = wave_sum_f64(v);
  const int w = threadIdx.x / LGD_WAVE, l = threadIdx.x % LGD_WAVE;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NT / LGD_WAVE; ++i) t += sh[i];
  return t;
}
template <int NT>
__device__ __forceinline__ double block_max_f64(double v, double *sh) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d, LGD_WAVE));
  const int w = threadIdx.x / LGD_WAVE, l = threadIdx.x % LGD_WAVE;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NT / LGD_WAVE; ++i) t = fmax(t, sh[i]);
  return t;
}

__device__ __forceinline__ double energy_to_loudness(double e) {
  return 10.0 * (log(e) / log(10.0)) - 0.691;
}

// ---- pass 1: 400 ms block energies (E5), 3 s block energies (E6), absolute gate
// p1[slice] = { n_abs, sum_abs, n_st, max 400 ms energy }; pmax_s[slice] = max 3 s energy on
// the 100 ms grid (windows ending inside the slice).
// (LGD_P1_LDS 1 stages the slice's sub-block energies in LDS first, one plane per weighted channel: measured no
// faster -- 13.6 us against 14 us for the 36 slices of C2's one-hour track, the kernel is a chain of a few
// dependent latencies either way -- and its 43 KB per workgroup do not fit beside the next scan's four
// workgroups per CU when scans pipeline: off.)
#ifndef LGD_P1_LDS
#define LGD_P1_LDS 0
#endif
#define LGD_P1_SPAN (LGD_SLICE + 40)  // sub-blocks a slice's blocks and windows reach: [j0, j0 + 1024 + 38)
__global__ __launch_bounds__(LGD_EPI_NT) void lgd_gate_pass1(
    const LgdSlice *__restrict__ slices, const LgdTrackMeta *__restrict__ meta,
    const double *__restrict__ E_all, double *__restrict__ Z_all, double *__restrict__ st_all,
    double *__restrict__ p1, double *__restrict__ pmax_s, double abs_gate) {
  LGD_EPI_PRIO();
  __shared__ double sh[LGD_EPI_NT / LGD_WAVE];
#if LGD_P1_LDS
  __shared__ double Es[5][LGD_P1_SPAN];
#endif
  const LgdSlice sl = slices[blockIdx.x];
  const LgdTrackMeta m = meta[sl.track];
  const double *E = E_all + m.e_off;  // channel ch, sub-block j at E[ch * n_sb + j]
  double *Z = Z_all + m.sb_off;
  const int tid = threadIdx.x;
  const int nblk = m.n_sb - 3;
  const int j1 = min(sl.j0 + LGD_SLICE, nblk);
  // channels mapped EBUR128_UNUSED (weight 0) are skipped like the reference does
  int wch[5];
  double wgt[5];
  int nw = 0;
  for (int ch = 0; ch < m.nch && nw < 5; ++ch) {
    const double w = lgd_channel_weight(ch, m.nch);
    if (w != 0.0) { wch[nw] = ch; wgt[nw] = w; ++nw; }
  }
#if LGD_P1_LDS
  for (int c = 0; c < nw; ++c) {
    const double *Ec = E + (size_t)wch[c] * m.n_sb;
    for (int i = tid; i < LGD_P1_SPAN; i += LGD_EPI_NT) {
      const int j = sl.j0 + i;
      Es[c][i] = j < m.n_sb ? Ec[j] : 0.0;
    }
  }
  __syncthreads();
#define LGD_P1_E(c_, j_) (Es[c_] + ((j_) - sl.j0))
#else
#define LGD_P1_E(c_, j_) (E + (size_t)wch[c_] * m.n_sb + (j_))
#endif
  // divide like the reference does (sum /= frames_per_block), not by a reciprocal
  const double len4 = 4.0 * (double)m.s100, len30 = 30.0 * (double)m.s100;
  double cnt = 0.0, sum = 0.0, cst = 0.0, zmax = 0.0, smax = 0.0;
  // block energy = sum_c w_c * (channel sum over the block) / frames (A.4)
  for (int j = sl.j0 + tid; j < j1; j += LGD_EPI_NT) {
    double s = 0.0;
    for (int c = 0; c < nw; ++c) {
      const double *Ec = LGD_P1_E(c, j);
      double cs = ((Ec[0] + Ec[1]) + Ec[2]) + Ec[3];
      if (wgt[c] != 1.0) cs *= wgt[c];
      s += cs;
    }
    const double zj = s / len4;
    Z[j] = zj;
    zmax = fmax(zmax, zj);
    if (zj >= abs_gate) { cnt += 1.0; sum += zj; }
  }
  // largest 3 s energy on the 100 ms grid: window jj covers sub-blocks [jj, jj + 30); the
  // sum per window is rebuilt every hop from the 30 values (exact, order-fixed)
  for (int jj = sl.j0 + tid; jj < min(sl.j0 + LGD_SLICE, m.n_sb - 29); jj += LGD_EPI_NT) {
    double s = 0.0;
    for (int c = 0; c < nw; ++c) {
      const double *p = LGD_P1_E(c, jj);
      double cs = 0.0;
#pragma unroll 6
      for (int i = 0; i < 30; ++i) cs += p[i];
      if (wgt[c] != 1.0) cs *= wgt[c];
      s += cs;
    }
    smax = fmax(smax, s / len30);
  }
  // short-term block kk ends at sub-block 10 kk + 30; it belongs to the slice
  // that holds 400 ms block 10 kk
  const int k0 = (sl.j0 + 9) / 10, k1 = min((sl.j0 + LGD_SLICE + 9) / 10, m.n_st_slots);
  for (int kk = k0 + tid; kk < k1; kk += LGD_EPI_NT) {
    double s = 0.0;
    for (int c = 0; c < nw; ++c) {
      const double *p = LGD_P1_E(c, 10 * kk);
      double cs = 0.0;
#pragma unroll 6
      for (int i = 0; i < 30; ++i) cs += p[i];
      if (wgt[c] != 1.0) cs *= wgt[c];
      s += cs;
    }
    s /= len30;
    const bool listed = s >= abs_gate;
    st_all[m.st_off + kk] = listed ? s : 0.0;  // 0.0 == not listed
    cst += listed ? 1.0 : 0.0;
  }
  cnt = block_sum_f64<LGD_EPI_NT>(cnt, sh);
  sum = block_sum_f64<LGD_EPI_NT>(sum, sh);
  cst = block_sum_f64<LGD_EPI_NT>(cst, sh);
  zmax = block_max_f64<LGD_EPI_NT>(zmax, sh);
  smax = block_max_f64<LGD_EPI_NT>(smax, sh);
  if (tid == 0) {
    double *o = p1 + 4 * (size_t)blockIdx.x;
    o[0] = cnt; o[1] = sum; o[2] = cst; o[3] = zmax;
    pmax_s[blockIdx.x] = smax;
  }
}

// ---- pass 2: relative gate (E7).  thr comes from this track's own pass-1 totals
// or, for the album pass, from the album records of all ranks (`world` records of
// `rec_stride` doubles, {sum_abs, n_abs, ...} in front; summed in rank order, so every
// workgroup and every rank gets the same bits).
// p2[slice] = { n_rel, sum_rel }
__global__ __launch_bounds__(LGD_EPI_NT) void lgd_gate_pass2(
    const LgdSlice *__restrict__ slices, const LgdTrackMeta *__restrict__ meta,
    const double *__restrict__ Z_all, const double *__restrict__ p1, double *__restrict__ p2,
    const double *__restrict__ album_rec1, int world, long long rec_stride, double abs_gate,
    double rel_factor) {
  LGD_EPI_PRIO();
  __shared__ double sh[LGD_EPI_NT / LGD_WAVE];
  const LgdSlice sl = slices[blockIdx.x];
  const LgdTrackMeta m = meta[sl.track];
  const double *Z = Z_all + m.sb_off;
  const int tid = threadIdx.x;
  const int j1 = min(sl.j0 + LGD_SLICE, m.n_sb - 3);
  double n_abs, sum_abs;
  if (album_rec1) {
    sum_abs = 0.0;
    n_abs = 0.0;
    const double *head = album_rec1 + 4 * (size_t)m.album;  // (several albums only with world == 1)
    for (int r = 0; r < world; ++r) {
      sum_abs += head[(size_t)r * rec_stride + 0];
      n_abs += head[(size_t)r * rec_stride + 1];
    }
  } else {
    double a = 0.0, b = 0.0;
    for (int i = tid; i < m.n_slices; i += LGD_EPI_NT) {
      a += p1[4 * (size_t)(m.slice_off + i) + 0];
      b += p1[4 * (size_t)(m.slice_off + i) + 1];
    }
    n_abs = block_sum_f64<LGD_EPI_NT>(a, sh);
    sum_abs = block_sum_f64<LGD_EPI_NT>(b, sh);
  }
  double thr = 0.0;
  if (n_abs > 0.0) {
    thr = sum_abs / n_abs;
    thr *= rel_factor;
  }
  double cnt = 0.0, sum = 0.0;
  for (int j = sl.j0 + tid; j < j1; j += LGD_EPI_NT) {
    const double zj = Z[j];
    if (zj >= abs_gate && zj >= thr) { cnt += 1.0; sum += zj; }
  }
  cnt = block_sum_f64<LGD_EPI_NT>(cnt, sh);
  sum = block_sum_f64<LGD_EPI_NT>(sum, sh);
  if (tid == 0) {
    p2[2 * (size_t)blockIdx.x + 0] = cnt;
    p2[2 * (size_t)blockIdx.x + 1] = sum;
  }
}

// ---- E8: loudness range of the listed short-term energies in st[off, off+n).
// Exact: libebur128 sorts and indexes; here the two order statistics are found
// by an MSB-first radix select over the IEEE bit patterns (positive doubles
// order like their bits), so no sort and no histogram quantisation.
// One workgroup per list.  The list is read ONCE, into registers (thread t holds entries t,
// t + 256, ...: up to 32 of them, LGD_LRA_BIG / 256); the 8-bit digit passes then run on
// registers and two LDS histograms.  Leading digits shared by the smallest and the largest kept
// energy are skipped (the sign / exponent byte always is).  (Before: every pass re-read the list
// from L2, 14 dependent loads per thread for the one-hour track of C2 -- 37 us of a 330 us scan,
// 67 us for a two-hour mono track.)
#define LGD_LRA_NT 256  // one wave per SIMD: fits next to two ~150-VGPR scan waves
#define LGD_LRA_RMAX (LGD_LRA_BIG / LGD_LRA_NT)

__device__ __forceinline__ double block_min_f64(double v, double *sh) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmin(v, __shfl_xor(v, d, LGD_WAVE));
  const int w = threadIdx.x / LGD_WAVE, l = threadIdx.x % LGD_WAVE;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  double t = sh[0];
#pragma unroll
  for (int i = 1; i < LGD_LRA_NT / LGD_WAVE; ++i) t = fmin(t, sh[i]);
  return t;
}

struct LgdLraShared {
  double sh[LGD_LRA_NT / LGD_WAVE];
  unsigned hist[2][256];
  unsigned long long prefix[2];
  unsigned long long rank[2];
};

// x[r] = entry tid + r * LGD_LRA_NT of the list (0.0 beyond its end; entries <= 0 are not listed
// energies and are ignored, as in the reference's histogram walk)
template <int R>
__device__ __forceinline__ void lgd_lra_in_registers(const double *__restrict__ gv, int n_list, double minus20,
                                                     double *__restrict__ out, LgdLraShared &S) {
  const int tid = threadIdx.x;
  double x[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = tid + r * LGD_LRA_NT;
    x[r] = i < n_list ? gv[i] : 0.0;
  }
  // (per-thread partials in list order with stride 256, then the fixed trees: the same order for
  // every R, so the relative threshold does not depend on the list's length class)
  double cnt = 0.0, sum = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (x[r] > 0.0) { cnt += 1.0; sum += x[r]; }
  const double n = block_sum_f64<LGD_LRA_NT>(cnt, S.sh);
  const double tot = block_sum_f64<LGD_LRA_NT>(sum, S.sh);
  if (n == 0.0) {
    if (tid == 0) *out = 0.0;
    return;
  }
  const double power = tot / n;
  const double integrated = minus20 * power;
  // keep the energies at or above the relative threshold (everything dropped sorts below them)
  cnt = 0.0;
  double lo = HUGE_VAL, hi = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (x[r] > 0.0 && !(x[r] < integrated)) {
      cnt += 1.0;
      lo = fmin(lo, x[r]);
      hi = fmax(hi, x[r]);
    } else {
      x[r] = 0.0;
    }
  }
  const double mrem = block_sum_f64<LGD_LRA_NT>(cnt, S.sh);
  if (mrem == 0.0) {
    if (tid == 0) *out = 0.0;
    return;
  }
  lo = block_min_f64(lo, S.sh);
  hi = block_max_f64<LGD_LRA_NT>(hi, S.sh);
  const unsigned long long klo = (unsigned long long)__double_as_longlong(lo),
                           khi = (unsigned long long)__double_as_longlong(hi);
  // digits (bytes, MSB first) on which every kept energy agrees
  const int first = klo == khi ? 8 : (__clzll((long long)(klo ^ khi)) >> 3);
  if (tid == 0) {
    S.rank[0] = (unsigned long long)((mrem - 1.0) * 0.95 + 0.5);
    S.rank[1] = (unsigned long long)((mrem - 1.0) * 0.1 + 0.5);
    S.prefix[0] = S.prefix[1] = first == 0 ? 0ull : (klo & (~0ull << (64 - 8 * first)));
  }
  __syncthreads();
  for (int pass = first; pass < 8; ++pass) {
    const int sh_bits = 56 - 8 * pass;
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (sh_bits + 8));
    for (int i = tid; i < 512; i += LGD_LRA_NT) S.hist[i >> 8][i & 255] = 0u;
    __syncthreads();
    const unsigned long long p0 = S.prefix[0], p1 = S.prefix[1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (x[r] > 0.0) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(x[r]);
        const unsigned dg = (unsigned)((key >> sh_bits) & 0xffu);
        if ((key & himask) == p0) atomicAdd(&S.hist[0][dg], 1u);
        if ((key & himask) == p1) atomicAdd(&S.hist[1][dg], 1u);
      }
    }
    __syncthreads();
    // waves 0 and 1 each locate their rank's digit: lane l owns bins 4l..4l+3
    if (tid < 2 * LGD_WAVE) {
      const int which = tid / LGD_WAVE, l = tid % LGD_WAVE;
      const unsigned h0 = S.hist[which][4 * l], h1 = S.hist[which][4 * l + 1],
                     h2 = S.hist[which][4 * l + 2], h3 = S.hist[which][4 * l + 3];
      unsigned incl = h0 + h1 + h2 + h3;
      const unsigned own = incl;
#pragma unroll
      for (int d = 1; d < LGD_WAVE; d <<= 1) {
        const unsigned up = __shfl_up(incl, d, LGD_WAVE);
        if (l >= d) incl += up;
      }
      const unsigned long long r = S.rank[which];
      const unsigned long long excl = incl - own;
      if (excl <= r && r < incl) {  // exactly one lane
        unsigned long long c = excl;
        int dg = 4 * l;
        if (c + h0 <= r) { c += h0; ++dg;
          if (c + h1 <= r) { c += h1; ++dg;
            if (c + h2 <= r) { c += h2; ++dg; } } }
        S.rank[which] = r - c;
        S.prefix[which] |= ((unsigned long long)dg) << sh_bits;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double h_en = __longlong_as_double((long long)S.prefix[0]);
    const double l_en = __longlong_as_double((long long)S.prefix[1]);
    *out = energy_to_loudness(h_en) - energy_to_loudness(l_en);
  }
}

// the same select for a list of any length, streamed from memory in every pass
__device__ __noinline__ void lgd_lra_streaming(const double *__restrict__ gv, long long n_list, double minus20,
                                               double *__restrict__ out, LgdLraShared &S) {
  const int tid = threadIdx.x;
  double cnt = 0.0, sum = 0.0;
  for (long long i = tid; i < n_list; i += LGD_LRA_NT) {
    const double x = gv[i];
    if (x > 0.0) { cnt += 1.0; sum += x; }
  }
  const double n = block_sum_f64<LGD_LRA_NT>(cnt, S.sh);
  const double tot = block_sum_f64<LGD_LRA_NT>(sum, S.sh);
  if (n == 0.0) {
    if (tid == 0) *out = 0.0;
    return;
  }
  const double power = tot / n;
  const double integrated = minus20 * power;
  cnt = 0.0;
  for (long long i = tid; i < n_list; i += LGD_LRA_NT) {
    const double x = gv[i];
    if (x > 0.0 && !(x < integrated)) cnt += 1.0;
  }
  const double mrem = block_sum_f64<LGD_LRA_NT>(cnt, S.sh);
  if (mrem == 0.0) {
    if (tid == 0) *out = 0.0;
    return;
  }
  if (tid == 0) {
    const unsigned long long dropped = (unsigned long long)(n - mrem);
    S.rank[0] = dropped + (unsigned long long)((mrem - 1.0) * 0.95 + 0.5);
    S.rank[1] = dropped + (unsigned long long)((mrem - 1.0) * 0.1 + 0.5);
    S.prefix[0] = S.prefix[1] = 0ull;
  }
  __syncthreads();
  for (int pass = 0; pass < 8; ++pass) {
    const int sh_bits = 56 - 8 * pass;
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (sh_bits + 8));
    for (int i = tid; i < 512; i += LGD_LRA_NT) S.hist[i >> 8][i & 255] = 0u;
    __syncthreads();
    const unsigned long long p0 = S.prefix[0], p1 = S.prefix[1];
    for (long long i = tid; i < n_list; i += LGD_LRA_NT) {
      const double x = gv[i];
      if (x > 0.0) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(x);
        const unsigned dg = (unsigned)((key >> sh_bits) & 0xffu);
        if ((key & himask) == p0) atomicAdd(&S.hist[0][dg], 1u);
        if ((key & himask) == p1) atomicAdd(&S.hist[1][dg], 1u);
      }
    }
    __syncthreads();
    if (tid < 2 * LGD_WAVE) {
      const int which = tid / LGD_WAVE, l = tid % LGD_WAVE;
      const unsigned h0 = S.hist[which][4 * l], h1 = S.hist[which][4 * l + 1],
                     h2 = S.hist[which][4 * l + 2], h3 = S.hist[which][4 * l + 3];
      unsigned long long incl = (unsigned long long)h0 + h1 + h2 + h3;
#pragma unroll
      for (int d = 1; d < LGD_WAVE; d <<= 1) {
        const unsigned long long up = __shfl_up(incl, d, LGD_WAVE);
        if (l >= d) incl += up;
      }
      const unsigned long long r = S.rank[which];
      const unsigned long long excl = incl - ((unsigned long long)h0 + h1 + h2 + h3);
      if (excl <= r && r < incl) {  // exactly one lane
        unsigned long long c = excl;
        int dg = 4 * l;
        if (c + h0 <= r) { c += h0; ++dg;
          if (c + h1 <= r) { c += h1; ++dg;
            if (c + h2 <= r) { c += h2; ++dg; } } }
        S.rank[which] = r - c;
        S.prefix[which] |= ((unsigned long long)dg) << sh_bits;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double h_en = __longlong_as_double((long long)S.prefix[0]);
    const double l_en = __longlong_as_double((long long)S.prefix[1]);
    *out = energy_to_loudness(h_en) - energy_to_loudness(l_en);
  }
}

__global__ __launch_bounds__(LGD_LRA_NT) void lgd_lra_kernel(const LgdRange *__restrict__ ranges,
                                                            const double *__restrict__ st_base,
                                                            double minus20, int skip_big) {
  LGD_EPI_PRIO();
  __shared__ LgdLraShared S;
  const LgdRange rg = ranges[blockIdx.x];
  const double *gv = st_base + rg.off;
  if (rg.n > LGD_LRA_BIG) {
    // the lgd_lra_big_* kernels take these; without their scratch (a distributed album longer
    // than the plan provided for) the list streams from L2 once per pass
    if (!skip_big) lgd_lra_streaming(gv, rg.n, minus20, rg.out, S);
    return;
  }
  const int n = (int)rg.n;
  if (n <= 1 * LGD_LRA_NT) lgd_lra_in_registers<1>(gv, n, minus20, rg.out, S);
  else if (n <= 2 * LGD_LRA_NT) lgd_lra_in_registers<2>(gv, n, minus20, rg.out, S);
  else if (n <= 4 * LGD_LRA_NT) lgd_lra_in_registers<4>(gv, n, minus20, rg.out, S);
  else if (n <= 8 * LGD_LRA_NT) lgd_lra_in_registers<8>(gv, n, minus20, rg.out, S);
  else if (n <= 16 * LGD_LRA_NT) lgd_lra_in_registers<16>(gv, n, minus20, rg.out, S);
  else lgd_lra_in_registers<LGD_LRA_RMAX>(gv, n, minus20, rg.out, S);
}

// ---- everything behind pass 1 in ONE launch (round 2: lgd_gate_pass2, lgd_track_final, lgd_lra_kernel -- three
// dependent launches of a few workgroups each behind every scan).  Workgroups [0, n_slices): pass 2 of one slice
// (E7: blocks at or above both gates); the LAST of a track's slices to finish (a counter per track, release /
// acquire fences at device scope) writes the track's result record (E7 loudness, E9 peaks, counts).
// Workgroups [n_slices, n_slices + n_tracks): the loudness range of one track's short-term list (E8) -- it only
// depends on pass 1, so it runs beside pass 2 instead of behind it (lists longer than LGD_LRA_BIG: the
// lgd_lra_big_* launches).
__global__ __launch_bounds__(LGD_EPI_NT) void lgd_track_finish_kernel(
    const LgdSlice *__restrict__ slices, int n_slices, const LgdTrackMeta *__restrict__ meta,
    const double *__restrict__ Z_all, const double *__restrict__ st_all, const float *__restrict__ peaks,
    const double *__restrict__ p1, double *__restrict__ p2, const double *__restrict__ pmax_s,
    double *__restrict__ res_all, unsigned *__restrict__ done_count, double abs_gate, double rel_factor,
    double minus20, int do_tp, int skip_big) {
  LGD_EPI_PRIO();
  __shared__ LgdLraShared S;
  __shared__ int s_last;
  double *sh = S.sh;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= n_slices) {  // ---- E8: one track's loudness range
    const int track = (int)blockIdx.x - n_slices;
    const LgdTrackMeta m = meta[track];
    const double *gv = st_all + m.st_off;
    const int n = m.n_st_slots;
    double *out = res_all + (size_t)track * LGR_STRIDE + LGR_LRA;
    if (n > LGD_LRA_BIG) {
      if (!skip_big) lgd_lra_streaming(gv, n, minus20, out, S);
      return;
    }
    if (n <= 1 * LGD_LRA_NT) lgd_lra_in_registers<1>(gv, n, minus20, out, S);
    else if (n <= 2 * LGD_LRA_NT) lgd_lra_in_registers<2>(gv, n, minus20, out, S);
    else if (n <= 4 * LGD_LRA_NT) lgd_lra_in_registers<4>(gv, n, minus20, out, S);
    else if (n <= 8 * LGD_LRA_NT) lgd_lra_in_registers<8>(gv, n, minus20, out, S);
    else if (n <= 16 * LGD_LRA_NT) lgd_lra_in_registers<16>(gv, n, minus20, out, S);
    else lgd_lra_in_registers<LGD_LRA_RMAX>(gv, n, minus20, out, S);
    return;
  }
  // ---- pass 2 of one slice; thr from the track's own pass-1 totals (strided per-thread partials, fixed trees)
  const LgdSlice sl = slices[blockIdx.x];
  const LgdTrackMeta m = meta[sl.track];
  const double *Z = Z_all + m.sb_off;
  const int j1 = min(sl.j0 + LGD_SLICE, m.n_sb - 3);
  double a = 0.0, b = 0.0, c = 0.0, mm = 0.0, ms = 0.0;
  for (int i = tid; i < m.n_slices; i += LGD_EPI_NT) {
    const size_t s = (size_t)(m.slice_off + i);
    a += p1[4 * s + 0];
    b += p1[4 * s + 1];
    c += p1[4 * s + 2];
    mm = fmax(mm, p1[4 * s + 3]);
    ms = fmax(ms, pmax_s[s]);
  }
  const double n_abs = block_sum_f64<LGD_EPI_NT>(a, sh);
  const double sum_abs = block_sum_f64<LGD_EPI_NT>(b, sh);
  double thr = 0.0;
  if (n_abs > 0.0) {
    thr = sum_abs / n_abs;
    thr *= rel_factor;
  }
  double cnt = 0.0, sum = 0.0;
  for (int j = sl.j0 + tid; j < j1; j += LGD_EPI_NT) {
    const double zj = Z[j];
    if (zj >= abs_gate && zj >= thr) { cnt += 1.0; sum += zj; }
  }
  cnt = block_sum_f64<LGD_EPI_NT>(cnt, sh);
  sum = block_sum_f64<LGD_EPI_NT>(sum, sh);
  if (tid == 0) {
    p2[2 * (size_t)blockIdx.x + 0] = cnt;
    p2[2 * (size_t)blockIdx.x + 1] = sum;
    // this slice's partials before the count; the other slices' after the count has been seen
    __threadfence();
    const unsigned old = __hip_atomic_fetch_add(&done_count[sl.track], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old + 1u == (unsigned)m.n_slices;
    if (s_last) done_count[sl.track] = 0u;  // ready for the next scan into this work set
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // ---- the track's result record
  double *res = res_all + (size_t)sl.track * LGR_STRIDE;
  mm = block_max_f64<LGD_EPI_NT>(mm, sh);
  ms = block_max_f64<LGD_EPI_NT>(ms, sh);
  const double n_st = block_sum_f64<LGD_EPI_NT>(c, sh);
  double d = 0.0, e = 0.0;
  const volatile double *p2v = p2;  // (written by other workgroups during this launch: no cached copies)
  for (int i = tid; i < m.n_slices; i += LGD_EPI_NT) {
    const size_t s = (size_t)(m.slice_off + i);
    d += p2v[2 * s + 0];
    e += p2v[2 * s + 1];
  }
  const double n_rel = block_sum_f64<LGD_EPI_NT>(d, sh);
  const double sum_rel = block_sum_f64<LGD_EPI_NT>(e, sh);
  double sp = 0.0, tp = 0.0;
  for (int i = tid; i < m.n_seg * m.nch; i += LGD_EPI_NT) {
    const int sgi = i / m.nch, ch = i % m.nch;
    const float *pp = peaks + m.peak_off + (size_t)sgi * 2 * m.nch;
    sp = fmax(sp, (double)pp[ch]);
    tp = fmax(tp, (double)pp[m.nch + ch]);
  }
  sp = block_max_f64<LGD_EPI_NT>(sp, sh);
  tp = block_max_f64<LGD_EPI_NT>(tp, sh);
  if (tid == 0) {
    res[LGR_LOUDNESS] = n_rel > 0.0 ? energy_to_loudness(sum_rel / n_rel) : -HUGE_VAL;
    res[LGR_MAX_M] = mm > 0.0 ? energy_to_loudness(mm) : -HUGE_VAL;
    res[LGR_MAX_S] = ms > 0.0 ? energy_to_loudness(ms) : -HUGE_VAL;
    res[LGR_PEAK] = do_tp ? fmax(sp, tp) : sp;
    res[LGR_SPEAK] = sp;
    // ebur128_true_peak's value: max(interpolated, sample).  (The true-peak kernel evaluates only
    // the interpolator outputs that can exceed the track's sample peak, so the interpolated
    // maximum alone is exact only where it is the larger of the two.)
    res[LGR_TPEAK] = do_tp ? fmax(sp, tp) : 0.0;
    res[LGR_THR] = thr;
    res[LGR_SUM_ABS] = sum_abs;
    res[LGR_SUM_REL] = sum_rel;
    res[LGR_NBLK] = (double)(m.n_sb >= 4 ? m.n_sb - 3 : 0);
    res[LGR_NABS] = n_abs;
    res[LGR_NREL] = n_rel;
    res[LGR_NSTBLK] = (double)m.n_st_slots;
    res[LGR_NST] = n_st;
  }
}

// ---- E8 for LONG lists (an album of hundreds of tracks: ~2.3e5 short-term energies in C4, where the
// single-workgroup select above takes ~1 ms).  Same exact result, five small launches whose
// streaming passes use LGD_LRB_WG workgroups:
//   1 count / sum the listed energies (per-workgroup partials, folded in a fixed order) and
//     clear the histogram
//   2 relative threshold (-20 LU); histogram of the KEPT energies over a 16-bit digit of their
//     bit pattern: 5 exponent bits (2^-24 .. 2^8, beyond that the end bins) + 11 mantissa bits --
//     order preserving, ~2000 bins per octave, so the bin of a rank holds a handful of entries
//   3 fold the histogram copies, block sums; one workgroup: the bins that hold the two ranks (index rule of libebur128:
//     (size_t)((m - 1) * p + 0.5)) and the ranks inside them
//   4 gather the entries of those two bins
//   5 one workgroup: exact k-th smallest inside each (radix select on the bit patterns), LRA.
#define LGD_LRB_WG 64
#define LGD_LRB_NT 256   // (lgd_lra_select: one thread per 8-bit digit)
#define LGD_LRB_BINS 65536
#define LGD_LRB_REP 8      // copies of the histogram (equal energies would otherwise queue on one counter)
struct LgdLraPick {
  double integrated;           // entries < this are dropped
  unsigned long long rank[2];  // rank inside the bin (0-based), [0] = 95 %, [1] = 10 %
  unsigned bin[2];
  unsigned cnt[2];             // gather counters
  int empty, pad;
};
__device__ __forceinline__ unsigned lgd_lra_digit(double x) {
  const long long b = __double_as_longlong(x) >> 41;       // exponent | top 11 mantissa bits
  const long long d = b - ((long long)(1023 - 24) << 11);  // 2^-24 -> 0
  return (unsigned)(d < 0 ? 0 : (d > LGD_LRB_BINS - 1 ? LGD_LRB_BINS - 1 : d));
}
// (ranges are indexed through big_idx: the big ranges of a range array)
__global__ __launch_bounds__(LGD_LRB_NT) void lgd_lra_big_pass1(const LgdRange *__restrict__ ranges,
                                                               const int *__restrict__ big_idx,
                                                               const double *__restrict__ st_base,
                                                               double *__restrict__ part,
                                                               unsigned *__restrict__ hist) {
  LGD_EPI_PRIO();
  __shared__ double sh[LGD_LRB_NT / LGD_WAVE];
  const LgdRange rg = ranges[big_idx[blockIdx.y]];
  const double *gv = st_base + rg.off;
  unsigned *h = hist + (size_t)blockIdx.y * LGD_LRB_BINS * LGD_LRB_REP;
  for (int i = blockIdx.x * LGD_LRB_NT + threadIdx.x; i < LGD_LRB_BINS * LGD_LRB_REP; i += LGD_LRB_WG * LGD_LRB_NT)
    h[i] = 0u;
  const long long per = (rg.n + LGD_LRB_WG - 1) / LGD_LRB_WG;
  const long long i0 = per * blockIdx.x, i1 = min(rg.n, i0 + per);
  double cnt = 0.0, sum = 0.0;
  for (long long i = i0 + threadIdx.x; i < i1; i += LGD_LRB_NT) {
    const double x = gv[i];
    if (x > 0.0) { cnt += 1.0; sum += x; }
  }
  cnt = block_sum_f64<LGD_LRB_NT>(cnt, sh);
  sum = block_sum_f64<LGD_LRB_NT>(sum, sh);
  if (threadIdx.x == 0) {
    double *o = part + ((size_t)blockIdx.y * LGD_LRB_WG + blockIdx.x) * 4;
    o[0] = cnt; o[1] = sum;
  }
}
__global__ __launch_bounds__(LGD_LRB_NT) void lgd_lra_big_pass2(const LgdRange *__restrict__ ranges,
                                                               const int *__restrict__ big_idx,
                                                               const double *__restrict__ st_base,
                                                               double *__restrict__ part,
                                                               unsigned *__restrict__ hist, double minus20) {
  LGD_EPI_PRIO();
  __shared__ double sh[LGD_LRB_NT / LGD_WAVE];
  const LgdRange rg = ranges[big_idx[blockIdx.y]];
  const double *gv = st_base + rg.off;
  unsigned *h = hist + ((size_t)blockIdx.y * LGD_LRB_REP + (blockIdx.x % LGD_LRB_REP)) * LGD_LRB_BINS;
  double *pp = part + (size_t)blockIdx.y * LGD_LRB_WG * 4;
  double n = 0.0, S = 0.0;
  for (int w = 0; w < LGD_LRB_WG; ++w) { n += pp[4 * w]; S += pp[4 * w + 1]; }  // same order everywhere
  double kept = 0.0;
  if (n > 0.0) {
    const double integrated = minus20 * (S / n);
    const long long per = (rg.n + LGD_LRB_WG - 1) / LGD_LRB_WG;
    const long long i0 = per * blockIdx.x, i1 = min(rg.n, i0 + per);
    for (long long i = i0 + threadIdx.x; i < i1; i += LGD_LRB_NT) {
      const double x = gv[i];
      if (x > 0.0 && !(x < integrated)) {
        kept += 1.0;
        atomicAdd(&h[lgd_lra_digit(x)], 1u);
      }
    }
  }
  kept = block_sum_f64<LGD_LRB_NT>(kept, sh);
  if (threadIdx.x == 0) pp[4 * blockIdx.x + 2] = kept;
}
// the LGD_LRB_REP copies of the histogram summed into copy 0; bsum[w] = entries in workgroup w's
// LGD_LRB_BINS / LGD_LRB_WG bins
__global__ __launch_bounds__(LGD_LRB_NT) void lgd_lra_big_fold(unsigned *__restrict__ hist,
                                                              unsigned long long *__restrict__ bsum) {
  LGD_EPI_PRIO();
  __shared__ double sh[LGD_LRB_NT / LGD_WAVE];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 *h = reinterpret_cast<u32x4 *>(hist + (size_t)blockIdx.y * LGD_LRB_BINS * LGD_LRB_REP);
  constexpr int VPW = LGD_LRB_BINS / LGD_LRB_WG / 4;  // 16-B vectors per workgroup (256)
  static_assert(VPW == LGD_LRB_NT, "one vector of 4 bins per thread");
  const int v = blockIdx.x * VPW + threadIdx.x;
  u32x4 t = h[v];
#pragma unroll
  for (int r = 1; r < LGD_LRB_REP; ++r) t += h[(size_t)r * (LGD_LRB_BINS / 4) + v];
  h[v] = t;
  // (counts are < 2^32 in all, exact in a double)
  const double tot = block_sum_f64<LGD_LRB_NT>((double)t.x + (double)t.y + (double)t.z + (double)t.w, sh);
  if (threadIdx.x == 0) bsum[(size_t)blockIdx.y * LGD_LRB_WG + blockIdx.x] = (unsigned long long)tot;
}
__global__ __launch_bounds__(LGD_LRB_NT) void lgd_lra_big_pick(const LgdRange *__restrict__ ranges,
                                                              const int *__restrict__ big_idx,
                                                              const double *__restrict__ part,
                                                              const unsigned *__restrict__ hist,
                                                              const unsigned long long *__restrict__ bsum,
                                                              LgdLraPick *__restrict__ picks, double minus20) {
  LGD_EPI_PRIO();
  __shared__ unsigned long long tsum[LGD_LRB_NT];
  __shared__ int s_blk[2];
  __shared__ unsigned long long s_before[2];
  const LgdRange rg = ranges[big_idx[blockIdx.x]];
  const unsigned *h = hist + (size_t)blockIdx.x * LGD_LRB_BINS * LGD_LRB_REP;  // folded copy 0
  const unsigned long long *bs = bsum + (size_t)blockIdx.x * LGD_LRB_WG;
  const double *pp = part + (size_t)blockIdx.x * LGD_LRB_WG * 4;
  LgdLraPick *pk = picks + blockIdx.x;
  double n = 0.0, S = 0.0, m = 0.0;
  for (int w = 0; w < LGD_LRB_WG; ++w) { n += pp[4 * w]; S += pp[4 * w + 1]; m += pp[4 * w + 2]; }
  const int tid = threadIdx.x;
  if (n == 0.0 || m == 0.0) {
    if (tid == 0) { *rg.out = 0.0; pk->empty = 1; pk->cnt[0] = pk->cnt[1] = 0u; }
    return;
  }
  const unsigned long long r[2] = {(unsigned long long)((m - 1.0) * 0.95 + 0.5),
                                   (unsigned long long)((m - 1.0) * 0.1 + 0.5)};
  if (tid == 0) {
    pk->integrated = minus20 * (S / n);
    pk->empty = 0;
    pk->cnt[0] = pk->cnt[1] = 0u;
    for (int q = 0; q < 2; ++q) {  // the 1024-bin block that holds rank q
      unsigned long long c = 0ull;
      int w = 0;
      while (w + 1 < LGD_LRB_WG && c + bs[w] <= r[q]) { c += bs[w]; ++w; }
      s_blk[q] = w;
      s_before[q] = c;
    }
  }
  __syncthreads();
  constexpr int BPB = LGD_LRB_BINS / LGD_LRB_WG;  // bins per block (1024): 4 per thread
  for (int q = 0; q < 2; ++q) {
    const int b0 = s_blk[q] * BPB + tid * 4;
    const unsigned c0 = h[b0], c1 = h[b0 + 1], c2 = h[b0 + 2], c3 = h[b0 + 3];
    __syncthreads();
    tsum[tid] = (unsigned long long)c0 + c1 + c2 + c3;
    __syncthreads();
    if (tid == 0) {
      unsigned long long run = s_before[q];
      for (int i = 0; i < LGD_LRB_NT; ++i) { const unsigned long long v = tsum[i]; tsum[i] = run; run += v; }  // exclusive
    }
    __syncthreads();
    const unsigned long long lo = tsum[tid], hi = lo + c0 + c1 + c2 + c3;
    if (lo <= r[q] && r[q] < hi) {  // exactly one thread
      unsigned long long c = lo;
      int b = b0;
      const unsigned cc[4] = {c0, c1, c2, c3};
      int j = 0;
      while (c + cc[j] <= r[q]) { c += cc[j]; ++j; ++b; }
      pk->bin[q] = (unsigned)b;
      pk->rank[q] = r[q] - c;
    }
  }
}
__global__ __launch_bounds__(LGD_LRB_NT) void lgd_lra_big_gather(const LgdRange *__restrict__ ranges,
                                                                const int *__restrict__ big_idx,
                                                                const double *__restrict__ st_base,
                                                                LgdLraPick *__restrict__ picks,
                                                                double *__restrict__ cand,
                                                                const long long *__restrict__ cand_off) {
  LGD_EPI_PRIO();
  const LgdRange rg = ranges[big_idx[blockIdx.y]];
  LgdLraPick *pk = picks + blockIdx.y;
  if (pk->empty) return;
  const double *gv = st_base + rg.off;
  const double integrated = pk->integrated;
  const unsigned b0 = pk->bin[0], b1 = pk->bin[1];
  double *c0 = cand + cand_off[blockIdx.y], *c1 = c0 + rg.n;
  const long long per = (rg.n + LGD_LRB_WG - 1) / LGD_LRB_WG;
  const long long i0 = per * blockIdx.x, i1 = min(rg.n, i0 + per);
  for (long long i = i0 + threadIdx.x; i < i1; i += LGD_LRB_NT) {
    const double x = gv[i];
    if (x > 0.0 && !(x < integrated)) {
      const unsigned d = lgd_lra_digit(x);
      if (d == b0) c0[atomicAdd(&pk->cnt[0], 1u)] = x;
      if (d == b1) c1[atomicAdd(&pk->cnt[1], 1u)] = x;
    }
  }
}
// the r-th smallest (0-based) of v[0 .. n): MSB-first radix select on the bit patterns
#define LGD_LRB_SUB 16  // copies of the digit histogram in LDS (the entries of one bin share their top digits)
#define LGD_LRB_STAGE 4096  // candidates staged in LDS (32 KiB); longer lists stream from L2
__device__ double lgd_lra_select(const double *gv, unsigned n, unsigned long long r, unsigned *hist,
                                 unsigned long long *s_state, double *stage) {
  const int tid = threadIdx.x;
  if (tid == 0) { s_state[0] = 0ull; s_state[1] = r; }
  const double *v = gv;
  if (n <= LGD_LRB_STAGE) {
    for (unsigned i = tid; i < n; i += LGD_LRB_NT) stage[i] = gv[i];
    v = stage;
  }
  __syncthreads();
  for (int pass = 0; pass < 8; ++pass) {
    const int sh_bits = 56 - 8 * pass;
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (sh_bits + 8));
    for (int i = tid; i < 256 * LGD_LRB_SUB; i += LGD_LRB_NT) hist[i] = 0u;
    __syncthreads();
    const unsigned long long prefix = s_state[0];
    for (unsigned i = tid; i < n; i += LGD_LRB_NT) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(v[i]);
      if ((key & himask) == prefix)
        atomicAdd(&hist[(tid % LGD_LRB_SUB) * 256 + (unsigned)((key >> sh_bits) & 0xffu)], 1u);
    }
    __syncthreads();
    // the digit whose cumulative count passes the rank: thread t owns digit t (LGD_LRB_NT == 256);
    // inclusive scan inside each wave, wave totals through LDS
    {
      unsigned t = 0u;
#pragma unroll
      for (int q = 0; q < LGD_LRB_SUB; ++q) t += hist[q * 256 + tid];
      unsigned incl = t;
#pragma unroll
      for (int d = 1; d < LGD_WAVE; d <<= 1) {
        const unsigned up = __shfl_up(incl, d, LGD_WAVE);
        if ((tid & (LGD_WAVE - 1)) >= d) incl += up;
      }
      __syncthreads();
      if ((tid & (LGD_WAVE - 1)) == LGD_WAVE - 1) hist[tid / LGD_WAVE] = incl;  // wave totals (hist is free now)
      __syncthreads();
      unsigned long long before = 0ull;
      for (int w = 0; w < tid / LGD_WAVE; ++w) before += hist[w];
      const unsigned long long rr = s_state[1];
      const unsigned long long hi = before + incl, lo = hi - t;
      __syncthreads();
      if (lo <= rr && rr < hi) {  // exactly one thread
        s_state[1] = rr - lo;
        s_state[0] = prefix | (((unsigned long long)tid) << sh_bits);
      }
    }
    __syncthreads();
  }
  return __longlong_as_double((long long)s_state[0]);
}
__global__ __launch_bounds__(LGD_LRB_NT) void lgd_lra_big_select(const LgdRange *__restrict__ ranges,
                                                                const int *__restrict__ big_idx,
                                                                const LgdLraPick *__restrict__ picks,
                                                                const double *__restrict__ cand,
                                                                const long long *__restrict__ cand_off) {
  LGD_EPI_PRIO();
  __shared__ unsigned hist[256 * LGD_LRB_SUB];
  __shared__ unsigned long long s_state[2];
  __shared__ double stage[LGD_LRB_STAGE];
  const LgdRange rg = ranges[big_idx[blockIdx.x]];
  const LgdLraPick pk = picks[blockIdx.x];
  if (pk.empty) return;
  const double *c0 = cand + cand_off[blockIdx.x], *c1 = c0 + rg.n;
  const double h_en = lgd_lra_select(c0, pk.cnt[0], pk.rank[0], hist, s_state, stage);
  __syncthreads();
  const double l_en = lgd_lra_select(c1, pk.cnt[1], pk.rank[1], hist, s_state, stage);
  if (threadIdx.x == 0) *rg.out = energy_to_loudness(h_en) - energy_to_loudness(l_en);
}

// ---- album stages (scan.c:359-405).  One workgroup per album of the plan. --------
// head of an album record 1 = { sum_abs, n_abs, peak, n_st } over the album's tracks
// on this rank (in the multi-GPU form the listed 3 s energies follow it in the same buffer)
__global__ __launch_bounds__(LGD_EPI_NT) void lgd_album_part1_kernel(const double *__restrict__ res,
                                                                    const LgdAlbumMeta *__restrict__ albums,
                                                                    double *__restrict__ heads) {
  LGD_EPI_PRIO();
  __shared__ double sh[LGD_EPI_NT / LGD_WAVE];
  const LgdAlbumMeta am = albums[blockIdx.x];
  double sa = 0.0, na = 0.0, pk = 0.0, ns = 0.0;
  for (int t = am.t0 + threadIdx.x; t < am.t1; t += LGD_EPI_NT) {
    const double *r = res + (size_t)t * LGR_STRIDE;
    sa += r[LGR_SUM_ABS];
    na += r[LGR_NABS];
    ns += r[LGR_NST];
    pk = fmax(pk, r[LGR_PEAK]);
  }
  sa = block_sum_f64<LGD_EPI_NT>(sa, sh);
  na = block_sum_f64<LGD_EPI_NT>(na, sh);
  ns = block_sum_f64<LGD_EPI_NT>(ns, sh);
  pk = block_max_f64<LGD_EPI_NT>(pk, sh);
  if (threadIdx.x == 0) {
    double *h = heads + 4 * (size_t)blockIdx.x;
    h[0] = sa; h[1] = na; h[2] = pk; h[3] = ns;
  }
}

// record 2 = { sum_rel, n_rel } over the album's slices on this rank (album second
// pass).  Thread 0 also folds the record-1 heads of all ranks into part1 = { sum_abs,
// n_abs, peak, n_st } (rank order) and then clears them: in the multi-GPU form the
// gathered buffer is from here on nothing but the album's short-term list (0.0 = no
// entry) for lgd_lra_kernel.
__global__ __launch_bounds__(LGD_EPI_NT) void lgd_album_part2_kernel(const double *__restrict__ p2a,
                                                                    const LgdAlbumMeta *__restrict__ albums,
                                                                    double *__restrict__ rec2,
                                                                    double *__restrict__ heads_all,
                                                                    int world, long long rec_stride,
                                                                    double *__restrict__ part1) {
  LGD_EPI_PRIO();
  __shared__ double sh[LGD_EPI_NT / LGD_WAVE];
  const LgdAlbumMeta am = albums[blockIdx.x];
  double sr = 0.0, nr = 0.0;
  for (int i = am.slice0 + threadIdx.x; i < am.slice1; i += LGD_EPI_NT) {
    nr += p2a[2 * (size_t)i + 0];
    sr += p2a[2 * (size_t)i + 1];
  }
  sr = block_sum_f64<LGD_EPI_NT>(sr, sh);
  nr = block_sum_f64<LGD_EPI_NT>(nr, sh);
  if (threadIdx.x == 0) {
    rec2[2 * (size_t)blockIdx.x + 0] = sr;
    rec2[2 * (size_t)blockIdx.x + 1] = nr;
    double sa = 0.0, na = 0.0, pk = 0.0, ns = 0.0, live = 0.0;
    for (int r = 0; r < world; ++r) {
      double *h = heads_all + (size_t)r * rec_stride + 4 * (size_t)blockIdx.x;
      sa += h[0]; na += h[1]; pk = fmax(pk, h[2]); ns += h[3];
      live += (h[1] > 0.0 || h[2] > 0.0) ? 1.0 : 0.0;  // a rank that brought blocks or a peak of its own
      h[0] = 0.0; h[1] = 0.0; h[2] = 0.0; h[3] = 0.0;
    }
    double *o = part1 + LGD_PART1 * (size_t)blockIdx.x;
    o[0] = sa; o[1] = na; o[2] = pk; o[3] = ns;
    o[4] = (double)world;  // heads folded here: what the album result reports as the ranks that took part
    o[5] = live;
  }
}

// album[a][] = { loudness, lra (lgd_lra_kernel), peak, thr, sum_abs, sum_rel, n_abs, n_rel, n_st,
//                heads folded in stage 2, heads with content, records 2 folded }
// rec2_all: the records 2 of all ranks ([rank][album][2]), summed in rank order
__global__ void lgd_album_final_kernel(const double *__restrict__ part1_all,
                                       const double *__restrict__ rec2_all, int world, int n_albums,
                                       double rel_factor, double *__restrict__ album_all) {
  LGD_EPI_PRIO();
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_albums) return;
  const double *part1 = part1_all + LGD_PART1 * (size_t)a;
  double *album = album_all + (size_t)a * LGD_ALBUM_STRIDE;
  double thr = 0.0;
  if (part1[1] > 0.0) {
    thr = part1[0] / part1[1];
    thr *= rel_factor;
  }
  double sr = 0.0, nr = 0.0;
  for (int r = 0; r < world; ++r) {
    sr += rec2_all[2 * ((size_t)r * n_albums + a) + 0];
    nr += rec2_all[2 * ((size_t)r * n_albums + a) + 1];
  }
  album[0] = nr > 0.0 ? energy_to_loudness(sr / nr) : -HUGE_VAL;
  album[2] = part1[2];
  album[3] = thr;
  album[4] = part1[0];
  album[5] = sr;
  album[6] = part1[1];
  album[7] = nr;
  album[8] = part1[3];
  album[9] = part1[4];        // record-1 heads folded in stage 2 (ranks of the exchange)
  album[10] = part1[5];       // ... of which held blocks or a peak
  album[11] = (double)world;  // records 2 folded here
}

// ------------------------------------------------------- launch wrappers ---
// skip_big: lists longer than LGD_LRA_BIG are left to the lgd_lra_big_* launches (lgd_launch_lra_big)
extern "C" hipError_t lgd_launch_track_epilogue(const LgdSlice *slices, int n_slices,
                                                const LgdTrackMeta *meta, int n_tracks,
                                                const double *E, double *Z, double *st,
                                                const float *peaks, double *p1, double *p2,
                                                double *pmax_s, double *res, unsigned *done_count,
                                                double abs_gate, double rel_factor, double minus20, int do_tp,
                                                int skip_big, hipStream_t s) {
  if (n_tracks <= 0 || n_slices <= 0) return hipSuccess;  // (every track has at least one slice)
  hipLaunchKernelGGL(lgd_gate_pass1, dim3(n_slices), dim3(LGD_EPI_NT), 0, s, slices, meta, E, Z, st, p1, pmax_s,
                     abs_gate);
  hipLaunchKernelGGL(lgd_track_finish_kernel, dim3(n_slices + n_tracks), dim3(LGD_EPI_NT), 0, s, slices, n_slices,
                     meta, Z, st, peaks, p1, p2, pmax_s, res, done_count, abs_gate, rel_factor, minus20, do_tp, skip_big);
  return hipGetLastError();
}

// n_big > 0: the ranges listed in big_idx (longer than LGD_LRA_BIG) go through the multi-workgroup
// kernels and the single-workgroup kernel skips them.  scratch: hist [n_big][8][65536] u32, part
// [n_big][64][4] f64, picks [n_big], cand [sum 2 n_i] f64 with cand_off [n_big].
extern "C" size_t lgd_lra_pick_bytes(void) { return sizeof(LgdLraPick); }
// small_too 0: only the long lists (the short ones of a plan's tracks are done by lgd_track_finish_kernel)
extern "C" hipError_t lgd_launch_lra(const void *ranges, int n_ranges, const double *st_base,
                                     double minus20, const int *big_idx, int n_big, unsigned *hist,
                                     double *part, void *picks, double *cand, const long long *cand_off,
                                     int small_too, hipStream_t s) {
  if (n_ranges <= 0) return hipSuccess;
  const LgdRange *rg = (const LgdRange *)ranges;
  if (small_too && n_big < n_ranges)
    hipLaunchKernelGGL(lgd_lra_kernel, dim3(n_ranges), dim3(LGD_LRA_NT), 0, s, rg, st_base, minus20,
                       n_big > 0 ? 1 : 0);
  if (n_big > 0) {
    hipLaunchKernelGGL(lgd_lra_big_pass1, dim3(LGD_LRB_WG, n_big), dim3(LGD_LRB_NT), 0, s, rg, big_idx, st_base,
                       part, hist);
    hipLaunchKernelGGL(lgd_lra_big_pass2, dim3(LGD_LRB_WG, n_big), dim3(LGD_LRB_NT), 0, s, rg, big_idx, st_base,
                       part, hist, minus20);
    // (block sums live behind the per-workgroup partials: part[n_big][64][4], column 3 unused -> a
    // separate region at the end of `part`: [n_big * 256 ..) reinterpreted as 64-bit counts)
    unsigned long long *bsum = reinterpret_cast<unsigned long long *>(part + (size_t)n_big * LGD_LRB_WG * 4);
    hipLaunchKernelGGL(lgd_lra_big_fold, dim3(LGD_LRB_WG, n_big), dim3(LGD_LRB_NT), 0, s, hist, bsum);
    hipLaunchKernelGGL(lgd_lra_big_pick, dim3(n_big), dim3(LGD_LRB_NT), 0, s, rg, big_idx, part, hist, bsum,
                       (LgdLraPick *)picks, minus20);
    hipLaunchKernelGGL(lgd_lra_big_gather, dim3(LGD_LRB_WG, n_big), dim3(LGD_LRB_NT), 0, s, rg, big_idx, st_base,
                       (LgdLraPick *)picks, cand, cand_off);
    hipLaunchKernelGGL(lgd_lra_big_select, dim3(n_big), dim3(LGD_LRB_NT), 0, s, rg, big_idx,
                       (const LgdLraPick *)picks, cand, cand_off);
  }
  return hipGetLastError();
}

extern "C" hipError_t lgd_launch_album_part1(const double *res, const LgdAlbumMeta *albums,
                                             int n_albums, double *heads, hipStream_t s) {
  if (n_albums <= 0) return hipSuccess;
  hipLaunchKernelGGL(lgd_album_part1_kernel, dim3(n_albums), dim3(LGD_EPI_NT), 0, s, res, albums, heads);
  return hipGetLastError();
}

extern "C" hipError_t lgd_launch_album_stage2(const LgdSlice *slices, int n_slices,
                                              const LgdTrackMeta *meta, const double *Z,
                                              const double *p1, double *p2a,
                                              const LgdAlbumMeta *albums, int n_albums,
                                              double *heads_all, int world, long long rec_stride,
                                              double *part1, double *rec2, double abs_gate,
                                              double rel_factor, hipStream_t s) {
  if (n_albums <= 0) return hipSuccess;
  if (n_slices > 0)
    hipLaunchKernelGGL(lgd_gate_pass2, dim3(n_slices), dim3(LGD_EPI_NT), 0, s, slices, meta, Z, p1, p2a,
                       (const double *)heads_all, world, rec_stride, abs_gate, rel_factor);
  hipLaunchKernelGGL(lgd_album_part2_kernel, dim3(n_albums), dim3(LGD_EPI_NT), 0, s, p2a, albums, rec2,
                     heads_all, world, rec_stride, part1);
  return hipGetLastError();
}

extern "C" hipError_t lgd_launch_album_final(const double *part1, const double *rec2_all, int world,
                                             int n_albums, double rel_factor, double *album,
                                             hipStream_t s) {
  if (n_albums <= 0) return hipSuccess;
  hipLaunchKernelGGL(lgd_album_final_kernel, dim3((n_albums + 63) / 64), dim3(64), 0, s, part1, rec2_all,
                     world, n_albums, rel_factor, album);
  return hipGetLastError();
}

// ---- PCM ingest: interleaved S16 -> f32 on the S16 grid (x / 32768, exact).
// The reference converts every decoded frame to S16 (scan.c:414,442) and
// libebur128 scales by 1/32768; uploading S16 halves the PCIe bytes.
__global__ __launch_bounds__(256) void lgd_s16_to_f32_kernel(const short *__restrict__ in,
                                                             float *__restrict__ out,
                                                             size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 8 <= n && (((uintptr_t)(in + i)) & 15) == 0 && (((uintptr_t)(out + i)) & 15) == 0) {
    const short4 a = *reinterpret_cast<const short4 *>(in + i);
    const short4 b = *reinterpret_cast<const short4 *>(in + i + 4);
    const float k = 1.0f / 32768.0f;
    *reinterpret_cast<float4 *>(out + i) = make_float4(a.x * k, a.y * k, a.z * k, a.w * k);
    *reinterpret_cast<float4 *>(out + i + 4) = make_float4(b.x * k, b.y * k, b.z * k, b.w * k);
  } else {
    for (size_t j = i; j < n && j < i + 8; ++j) out[j] = (float)in[j] * (1.0f / 32768.0f);
  }
}

extern "C" hipError_t lgd_launch_s16_to_f32(const short *in, float *out, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  const size_t nthreads = (n + 7) / 8;
  hipLaunchKernelGGL(lgd_s16_to_f32_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s,
                     in, out, n);
  return hipGetLastError();
}
