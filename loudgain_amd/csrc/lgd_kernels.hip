// lgd_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the EBU R128 scan.
//
// What they replace (all third-party libebur128 work reached from
// /root/reference/src/scan.c:448 and :294-303,:383-388):
//   lgd_scan_kernel      E3 K-weighting (K2) + sample peak (K1) + E4 true peak
//                        (K3) + the 100 ms partial sums behind E5/E6 (K4)
//   lgd_track_epilogue   E5/E6 block lists + E7 two-pass gating + E9 peaks
//   lgd_lra_kernel       E8 loudness range (exact rank selection, no sort)
//   lgd_album_*          E7/E8 "_multiple" forms + scan.c:359-378 album peak
//
// Parallelisation of the strictly sequential IIR:
// one wavefront owns a run of whole 100 ms sub-blocks.  It walks that run in
// tiles of 64 lanes x C frames; inside a tile every lane
//   A. runs the 4th-order recurrence from a ZERO state over its C frames
//      (gives the zero-state final state z_lane),
//   B. a 6-step wave scan of the affine maps  s -> A^C s + z  turns the z_lane
//      into the exact filter state at the start of every lane's chunk,
//   C. re-runs its C frames from that state, now producing y, y^2 and peaks.
// The state that enters a segment comes from `n_warm_tiles` tiles of A+B only
// over the audio just before it (the filter's memory is < 1e-26 after 300 ms).
//
// Conditioning: libebur128 runs the merged 4th-order filter in direct form II,
// whose state v = x/A(z) is ~1e5 x the signal for low-frequency content and
// whose transition powers A^k reach 2e3 (48 kHz) .. 1e5 (192 kHz): a state
// scan in that basis loses 1e-8.  Here the same transfer function is the chain
//   w[n] = x[n] - 2x[n-1] + x[n-2]      RLB numerator (1 - z^-1)^2, exact
//   q = w / ra(z)                       RLB poles
//   p = q / pa(z)                       shelf poles
//   y = pb(z) p                         shelf numerator
// (ra, pa, pb as SURVEY.md A.1; b = pb*(1,-2,1), a = pa*ra), and the carried
// state is s = (q1, alpha (q1 - beta q2), p1/dc, p2/dc): "value" and "scaled
// slope" of the nearly double RLB pole, shelf states scaled by their DC gain.
// In that basis every transition power stays <= ~10 and is block lower
// triangular (12 FMAs per scan step instead of 16).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "lgd_internal.h"

#define LGD_WAVE 64

// ---------------------------------------------------------------- helpers ---
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, LGD_WAVE);
  return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, LGD_WAVE));
  return v;
}

// ------------------------------------------------------------ scan kernel ---
// C   frames per lane (divides the 100 ms sub-block length of the rate)
// NCH 1 or 2 interleaved channels, both weight 1.0 (L / L,R)
// TP  0 = no interpolator (>= 192 kHz or disabled), 4 = 4x, 2 = 2x
template <int C, int NCH, int TP>
struct ScanCfg {
  static constexpr int HALO = (TP == 2) ? 24 : 12;     // frames kept before the tile
  static constexpr int NTAP = (TP == 2) ? 24 : 12;     // taps per non-trivial phase
  static constexpr int NPH = (TP == 4) ? 3 : (TP == 2 ? 1 : 0);
  static constexpr int TILE_F = LGD_WAVE * C;
  static constexpr int LDS_FLOATS = (TILE_F + HALO) * NCH + 4;
  static constexpr int NVEC = LDS_FLOATS / 4;
  // frames per streamed step: the largest divisor of C not above 8
  static constexpr int U = (C % 8 == 0) ? 8 : (C % 7 == 0) ? 7 : (C % 6 == 0) ? 6 : (C % 5 == 0) ? 5
                         : (C % 4 == 0) ? 4 : (C % 3 == 0) ? 3 : (C % 2 == 0) ? 2 : 1;
};

template <int C, int NCH, int TP>
__global__ __launch_bounds__(LGD_WAVE) void lgd_scan_kernel(const LgdSeg *__restrict__ segs,
                                                           const LgdFilt F) {
  using K = ScanCfg<C, NCH, TP>;
  __shared__ __attribute__((aligned(16))) float lds[K::LDS_FLOATS];

  const int lane = threadIdx.x;
  const LgdSeg sg = segs[blockIdx.x];
  const int shift = (int)((sg.f0 * NCH) & 3);
  const long long n_frames = sg.n_floats / NCH;

  const double ra1 = F.ra[0], ra2 = F.ra[1], pa1 = F.pa[0], pa2 = F.pa[1];
  const double pb0 = F.pb[0], pb1 = F.pb[1], pb2 = F.pb[2];

  double cin[NCH][4];  // wave-uniform filter state entering the tile
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int r = 0; r < 4; ++r) cin[ch][r] = 0.0;

  double acc = 0.0;       // this lane's share of sub-block `cur`
  int cur = 0;            // sub-block (relative to the segment) being summed
  long long cur_q = 0;    // chunk index (relative to f0) where `cur` starts
  float pk_s[NCH], pk_t[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) pk_s[ch] = pk_t[ch] = 0.f;

  const int n_main = (int)((sg.f_peak_end - sg.f0 + K::TILE_F - 1) / K::TILE_F);

  for (int k = -sg.n_warm_tiles; k < n_main; ++k) {
    const long long tb = sg.f0 + (long long)k * K::TILE_F;  // first frame of the tile
    // ---- stage the tile (plus HALO frames of history) through LDS ---------
    // coalesced 16-B loads; frames outside [0, n_frames) read as zero
    __syncthreads();  // previous tile's LDS reads are done
    {
      const long long g0 = (tb - K::HALO) * NCH - shift;  // float index of lds[0], 4-aligned
#pragma unroll 4
      for (int i = lane; i < K::NVEC; i += LGD_WAVE) {
        const long long g = g0 + 4LL * i;
        float4 v;
        if (g >= 0 && g + 3 < sg.n_floats) {
          v = *reinterpret_cast<const float4 *>(sg.pcm + g);
        } else {
          v.x = (g + 0 >= 0 && g + 0 < sg.n_floats) ? sg.pcm[g + 0] : 0.f;
          v.y = (g + 1 >= 0 && g + 1 < sg.n_floats) ? sg.pcm[g + 1] : 0.f;
          v.z = (g + 2 >= 0 && g + 2 < sg.n_floats) ? sg.pcm[g + 2] : 0.f;
          v.w = (g + 3 >= 0 && g + 3 < sg.n_floats) ? sg.pcm[g + 3] : 0.f;
        }
        *reinterpret_cast<float4 *>(lds + 4 * i) = v;
      }
    }
    __syncthreads();

    // this lane's chunk: frames [tb + lane*C, +C), streamed from LDS U frames at a
    // time (keeps the VGPR count low; LDS reads are ~free next to the fp64 work)
    constexpr int U = K::U;
    constexpr int HX = (TP == 0) ? 0 : (K::NTAP - 1);
    const float *chunk = lds + shift + (K::HALO + lane * C) * NCH;

    // ---- A: zero-state run of q' = x/ra, p' = q'/pa over the chunk (4 FMAs per
    // sample).  (1 - z^-1)^2 commutes with both, so second differences of the
    // last four q', p' give the zero-state (q, p) at the chunk end; q', p' stay
    // <= ~C^2 |x| here, so the differencing costs ~1e-13 |x| at most. --------
    double z[NCH][4];
    {
      double qv[NCH][4], pv[NCH][4];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int r = 0; r < 4; ++r) qv[ch][r] = pv[ch][r] = 0.0;
#pragma unroll 1
      for (int j0 = 0; j0 < C; j0 += U) {
        float xa[NCH][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if constexpr (NCH == 2) {
            const float2 t = *reinterpret_cast<const float2 *>(chunk + 2 * (j0 + u));
            xa[0][u] = t.x;
            xa[1][u] = t.y;
          } else {
            xa[0][u] = chunk[j0 + u];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) {
            double t = fma(-ra2, qv[ch][1], (double)xa[ch][u]);
            const double q0 = fma(-ra1, qv[ch][0], t);
            t = fma(-pa2, pv[ch][1], q0);
            const double p0 = fma(-pa1, pv[ch][0], t);
            qv[ch][3] = qv[ch][2]; qv[ch][2] = qv[ch][1]; qv[ch][1] = qv[ch][0]; qv[ch][0] = q0;
            pv[ch][3] = pv[ch][2]; pv[ch][2] = pv[ch][1]; pv[ch][1] = pv[ch][0]; pv[ch][0] = p0;
          }
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const double q1 = (qv[ch][0] - 2.0 * qv[ch][1]) + qv[ch][2];
        const double q2 = (qv[ch][1] - 2.0 * qv[ch][2]) + qv[ch][3];
        const double p1 = (pv[ch][0] - 2.0 * pv[ch][1]) + pv[ch][2];
        const double p2 = (pv[ch][1] - 2.0 * pv[ch][2]) + pv[ch][3];
        // the run assumed x[-1] = x[-2] = 0; the true history changes w[0] by
        // -2x[-1] + x[-2] and w[1] by x[-1] (g = their effect on the end state)
        const double xm1 = (double)chunk[-1 * NCH + ch], xm2 = (double)chunk[-2 * NCH + ch];
        const double dw0 = fma(-2.0, xm1, xm2), dw1 = xm1;
        double s0 = q1, s1 = F.alpha * fma(-F.beta, q2, q1), s2 = F.gamma * p1, s3 = F.gamma * p2;
        s0 = fma(F.g[0][0], dw0, s0); s0 = fma(F.g[1][0], dw1, s0);
        s1 = fma(F.g[0][1], dw0, s1); s1 = fma(F.g[1][1], dw1, s1);
        s2 = fma(F.g[0][2], dw0, s2); s2 = fma(F.g[1][2], dw1, s2);
        s3 = fma(F.g[0][3], dw0, s3); s3 = fma(F.g[1][3], dw1, s3);
        z[ch][0] = s0; z[ch][1] = s1; z[ch][2] = s2; z[ch][3] = s3;
      }
    }

    // ---- B: inject the carry at lane 0, then Kogge-Stone over 64 lanes.  The
    // transition is block lower triangular: rows 0,1 see columns 0,1 only. ----
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const double *P = F.P[0];
      const double t0 = fma(P[1], cin[ch][1], P[0] * cin[ch][0]);
      const double t1 = fma(P[5], cin[ch][1], P[4] * cin[ch][0]);
      const double t2 = fma(P[11], cin[ch][3], fma(P[10], cin[ch][2], fma(P[9], cin[ch][1], P[8] * cin[ch][0])));
      const double t3 = fma(P[15], cin[ch][3], fma(P[14], cin[ch][2], fma(P[13], cin[ch][1], P[12] * cin[ch][0])));
      const bool l0 = lane == 0;
      z[ch][0] += l0 ? t0 : 0.0;
      z[ch][1] += l0 ? t1 : 0.0;
      z[ch][2] += l0 ? t2 : 0.0;
      z[ch][3] += l0 ? t3 : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const int d = 1 << s;
      const double *P = F.P[s];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        double u[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) u[c] = __shfl_up(z[ch][c], d, LGD_WAVE);
        const bool on = lane >= d;
        const double t0 = fma(P[1], u[1], P[0] * u[0]);
        const double t1 = fma(P[5], u[1], P[4] * u[0]);
        const double t2 = fma(P[11], u[3], fma(P[10], u[2], fma(P[9], u[1], P[8] * u[0])));
        const double t3 = fma(P[15], u[3], fma(P[14], u[2], fma(P[13], u[1], P[12] * u[0])));
        z[ch][0] += on ? t0 : 0.0;
        z[ch][1] += on ? t1 : 0.0;
        z[ch][2] += on ? t2 : 0.0;
        z[ch][3] += on ? t3 : 0.0;
      }
    }
    // z is now the exact state at the END of each lane's chunk; the state at
    // its START is the previous lane's (lane 0: the carry).  Back to (q, p).
    double qs[NCH][2], ps[NCH][2];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      double sv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double up = __shfl_up(z[ch][r], 1, LGD_WAVE);
        sv[r] = (lane == 0) ? cin[ch][r] : up;
        cin[ch][r] = __shfl(z[ch][r], LGD_WAVE - 1, LGD_WAVE);
      }
      qs[ch][0] = sv[0];
      qs[ch][1] = fma(-F.inv_alpha, sv[1], sv[0]) * F.inv_beta;
      ps[ch][0] = sv[2] * F.dc;
      ps[ch][1] = sv[3] * F.dc;
    }

    if (k < 0) continue;  // warm-up tile: only the carry matters

    // ---- C: real run: y, energy, peaks ---------------------------------
    // frames >= n_frames were staged as zeros; their interpolator outputs do
    // not exist in the reference (it stops at the last input frame)
    const long long lane_f = tb + (long long)lane * C;
    const bool tail = (tb + K::TILE_F > n_frames);  // wave-uniform
    int nvalid = C;
    if (tail) {
      const long long rem = n_frames - lane_f;
      nvalid = rem < 0 ? 0 : (rem > C ? C : (int)rem);
    }
    double ech[NCH], xh[NCH][2];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      ech[ch] = 0.0;
      xh[ch][0] = (double)chunk[-1 * NCH + ch];
      xh[ch][1] = (double)chunk[-2 * NCH + ch];
    }
#pragma unroll 1
    for (int j0 = 0; j0 < C; j0 += U) {
      float w[NCH][U + HX];  // frames j0-HX .. j0+U-1
#pragma unroll
      for (int i = 0; i < U + HX; ++i) {
        if constexpr (NCH == 2) {
          const float2 t = *reinterpret_cast<const float2 *>(chunk + 2 * (j0 + i - HX));
          w[0][i] = t.x;
          w[1][i] = t.y;
        } else {
          w[0][i] = chunk[j0 + i - HX];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          const float xf = w[ch][HX + u];
          const double x = (double)xf;
          double t = fma(-2.0, xh[ch][0], x) + xh[ch][1];  // w[n], exact
          xh[ch][1] = xh[ch][0];
          xh[ch][0] = x;
          t = fma(-ra2, qs[ch][1], t);
          const double q0 = fma(-ra1, qs[ch][0], t);
          t = fma(-pa2, ps[ch][1], q0);
          const double p0 = fma(-pa1, ps[ch][0], t);
          double y = pb0 * p0;
          y = fma(pb1, ps[ch][0], y);
          y = fma(pb2, ps[ch][1], y);
          ech[ch] = fma(y, y, ech[ch]);
          qs[ch][1] = qs[ch][0]; qs[ch][0] = q0;
          ps[ch][1] = ps[ch][0]; ps[ch][0] = p0;
          pk_s[ch] = fmaxf(pk_s[ch], fabsf(xf));
          if constexpr (TP != 0) {
            float m = 0.f;
#pragma unroll
            for (int ph = 0; ph < K::NPH; ++ph) {
              float o = 0.f;
#pragma unroll
              for (int t2 = 0; t2 < K::NTAP; ++t2)
                o = fmaf(F.tp[ph * K::NTAP + t2], w[ch][HX + u - t2], o);
              m = fmaxf(m, fabsf(o));
            }
            if (tail) m = (j0 + u < nvalid) ? m : 0.f;
            pk_t[ch] = fmaxf(pk_t[ch], m);
          }
        }
    }
    double e = 0.0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) e = fma((double)F.w[ch], ech[ch], e);

    // ---- 100 ms sub-block sums: deterministic per-lane accumulate + wave tree
    long long rel = (long long)k * LGD_WAVE + lane - cur_q;
    for (;;) {
      const bool mine = rel >= 0 && rel < F.lps;
      acc += mine ? e : 0.0;
      if ((long long)k * LGD_WAVE + LGD_WAVE >= cur_q + F.lps) {  // `cur` ends in this tile
        const double tot = wave_sum_f64(acc);
        if (lane == 0 && cur < sg.n_sb) sg.e_out[cur] = tot;
        acc = 0.0;
        ++cur;
        cur_q += F.lps;
        rel -= F.lps;
      } else {
        break;
      }
    }
  }

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const float s = wave_max_f32(pk_s[ch]);
    const float t = wave_max_f32(pk_t[ch]);
    if (lane == 0) {
      sg.peak_out[ch] = s;
      sg.peak_out[NCH + ch] = t;
    }
  }
}

// ------------------------------------------------------- block reductions ---
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

// E5/E6/E7/E9 for one track per workgroup.
#define LGD_EPI_NT 256
__global__ __launch_bounds__(LGD_EPI_NT) void lgd_track_epilogue(
    const LgdTrackMeta *__restrict__ meta, const double *__restrict__ E_all,
    double *__restrict__ st_all, const float *__restrict__ peaks, double *__restrict__ res_all,
    double abs_gate, double rel_factor, int do_tp) {
  __shared__ double sh[LGD_EPI_NT / LGD_WAVE];
  const LgdTrackMeta m = meta[blockIdx.x];
  const double *E = E_all + m.sb_off;
  double *res = res_all + (size_t)blockIdx.x * LGR_STRIDE;
  const int tid = threadIdx.x;
  const int nblk = m.n_sb >= 4 ? m.n_sb - 3 : 0;
  // divide like the reference does (sum /= frames_per_block), not by a reciprocal
  const double len4 = 4.0 * (double)m.s100, len30 = 30.0 * (double)m.s100;

  double cnt = 0.0, sum = 0.0;
  for (int j = tid; j < nblk; j += LGD_EPI_NT) {
    const double zj = (((E[j] + E[j + 1]) + E[j + 2]) + E[j + 3]) / len4;
    if (zj >= abs_gate) { cnt += 1.0; sum += zj; }
  }
  const double n_abs = block_sum_f64<LGD_EPI_NT>(cnt, sh);
  const double sum_abs = block_sum_f64<LGD_EPI_NT>(sum, sh);
  double thr = 0.0, n_rel = 0.0, sum_rel = 0.0;
  if (n_abs > 0.0) {
    thr = sum_abs / n_abs;
    thr *= rel_factor;
    cnt = 0.0; sum = 0.0;
    for (int j = tid; j < nblk; j += LGD_EPI_NT) {
      const double zj = (((E[j] + E[j + 1]) + E[j + 2]) + E[j + 3]) / len4;
      if (zj >= abs_gate && zj >= thr) { cnt += 1.0; sum += zj; }
    }
    n_rel = block_sum_f64<LGD_EPI_NT>(cnt, sh);
    sum_rel = block_sum_f64<LGD_EPI_NT>(sum, sh);
  }
  // short-term (3 s) blocks, 1 s cadence
  cnt = 0.0;
  for (int kk = tid; kk < m.n_st_slots; kk += LGD_EPI_NT) {
    double s = 0.0;
    const double *p = E + 10 * kk;
    for (int i = 0; i < 30; ++i) s += p[i];
    s /= len30;
    const bool listed = s >= abs_gate;
    st_all[m.st_off + kk] = listed ? s : 0.0;
    cnt += listed ? 1.0 : 0.0;
  }
  const double n_st = block_sum_f64<LGD_EPI_NT>(cnt, sh);
  // peaks over segment partials (all channels)
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
    res[LGR_LOUDNESS] = n_rel > 0.0 ? 10.0 * (log(sum_rel / n_rel) / log(10.0)) - 0.691 : -HUGE_VAL;
    res[LGR_PEAK] = do_tp ? fmax(sp, tp) : sp;
    res[LGR_SPEAK] = sp;
    res[LGR_TPEAK] = do_tp ? tp : 0.0;
    res[LGR_THR] = thr;
    res[LGR_SUM_ABS] = sum_abs;
    res[LGR_SUM_REL] = sum_rel;
    res[LGR_NBLK] = (double)nblk;
    res[LGR_NABS] = n_abs;
    res[LGR_NREL] = n_rel;
    res[LGR_NSTBLK] = (double)m.n_st_slots;
    res[LGR_NST] = n_st;
  }
}

// E8: loudness range of the listed short-term energies in st[off, off+n).
// Exact: libebur128 sorts and indexes; here the two order statistics are found
// by an MSB-first radix select over the IEEE bit patterns (positive doubles
// order like their bits), so no sort and no histogram quantisation.
#define LGD_LRA_NT 256

__global__ __launch_bounds__(LGD_LRA_NT) void lgd_lra_kernel(const LgdRange *__restrict__ ranges,
                                                            const double *__restrict__ st_base,
                                                            double minus20) {
  __shared__ double sh[LGD_LRA_NT / LGD_WAVE];
  __shared__ unsigned hist[2][256];
  __shared__ unsigned long long s_prefix[2];
  __shared__ unsigned long long s_rank[2];
  const LgdRange rg = ranges[blockIdx.x];
  const double *v = st_base + rg.off;
  const int tid = threadIdx.x;

  double cnt = 0.0, sum = 0.0;
  for (long long i = tid; i < rg.n; i += LGD_LRA_NT) {
    const double x = v[i];
    if (x > 0.0) { cnt += 1.0; sum += x; }
  }
  const double n = block_sum_f64<LGD_LRA_NT>(cnt, sh);
  const double S = block_sum_f64<LGD_LRA_NT>(sum, sh);
  if (n == 0.0) {
    if (tid == 0) *rg.out = 0.0;
    return;
  }
  const double power = S / n;
  const double integrated = minus20 * power;
  cnt = 0.0;
  for (long long i = tid; i < rg.n; i += LGD_LRA_NT) {
    const double x = v[i];
    if (x > 0.0 && !(x < integrated)) cnt += 1.0;
  }
  const double mrem = block_sum_f64<LGD_LRA_NT>(cnt, sh);
  if (mrem == 0.0) {
    if (tid == 0) *rg.out = 0.0;
    return;
  }
  if (tid == 0) {
    const unsigned long long dropped = (unsigned long long)(n - mrem);
    s_rank[0] = dropped + (unsigned long long)((mrem - 1.0) * 0.95 + 0.5);
    s_rank[1] = dropped + (unsigned long long)((mrem - 1.0) * 0.1 + 0.5);
    s_prefix[0] = s_prefix[1] = 0ull;
  }
  __syncthreads();
  for (int pass = 0; pass < 8; ++pass) {
    const int sh_bits = 56 - 8 * pass;
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (sh_bits + 8));
    hist[0][tid] = 0u;
    hist[1][tid] = 0u;
    __syncthreads();
    const unsigned long long p0 = s_prefix[0], p1 = s_prefix[1];
    for (long long i = tid; i < rg.n; i += LGD_LRA_NT) {
      const double x = v[i];
      if (x > 0.0) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(x);
        const unsigned dg = (unsigned)((key >> sh_bits) & 0xffu);
        if ((key & himask) == p0) atomicAdd(&hist[0][dg], 1u);
        if ((key & himask) == p1) atomicAdd(&hist[1][dg], 1u);
      }
    }
    __syncthreads();
    if (tid < 2) {
      unsigned long long r = s_rank[tid], c = 0;
      int dg = 0;
      for (; dg < 256; ++dg) {
        const unsigned long long h = hist[tid][dg];
        if (c + h > r) break;
        c += h;
      }
      s_rank[tid] = r - c;
      s_prefix[tid] |= ((unsigned long long)dg) << sh_bits;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double h_en = __longlong_as_double((long long)s_prefix[0]);
    const double l_en = __longlong_as_double((long long)s_prefix[1]);
    const double lh = 10.0 * (log(h_en) / log(10.0)) - 0.691;
    const double ll = 10.0 * (log(l_en) / log(10.0)) - 0.691;
    *rg.out = lh - ll;
  }
}

// ---- album stages (scan.c:359-405) ---------------------------------------
// part1 = { sum_abs, n_abs, peak, n_st } over this rank's tracks
__global__ __launch_bounds__(LGD_EPI_NT) void lgd_album_part1_kernel(const double *__restrict__ res,
                                                                    int n_tracks,
                                                                    double *__restrict__ part1) {
  __shared__ double sh[LGD_EPI_NT / LGD_WAVE];
  double sa = 0.0, na = 0.0, pk = 0.0, ns = 0.0;
  for (int t = threadIdx.x; t < n_tracks; t += LGD_EPI_NT) {
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
    part1[0] = sa; part1[1] = na; part1[2] = pk; part1[3] = ns;
  }
}

// second gating pass of every track against the ALBUM relative threshold
__global__ __launch_bounds__(LGD_EPI_NT) void lgd_album_pass2_kernel(
    const LgdTrackMeta *__restrict__ meta, const double *__restrict__ E_all,
    double *__restrict__ res_all, const double *__restrict__ part1, double abs_gate,
    double rel_factor) {
  __shared__ double sh[LGD_EPI_NT / LGD_WAVE];
  const LgdTrackMeta m = meta[blockIdx.x];
  const double *E = E_all + m.sb_off;
  double *res = res_all + (size_t)blockIdx.x * LGR_STRIDE;
  const int nblk = m.n_sb >= 4 ? m.n_sb - 3 : 0;
  const double len4 = 4.0 * (double)m.s100;
  double thr = 0.0;
  if (part1[1] > 0.0) {
    thr = part1[0] / part1[1];
    thr *= rel_factor;
  }
  double cnt = 0.0, sum = 0.0;
  for (int j = threadIdx.x; j < nblk; j += LGD_EPI_NT) {
    const double zj = (((E[j] + E[j + 1]) + E[j + 2]) + E[j + 3]) / len4;
    if (zj >= abs_gate && zj >= thr) { cnt += 1.0; sum += zj; }
  }
  cnt = block_sum_f64<LGD_EPI_NT>(cnt, sh);
  sum = block_sum_f64<LGD_EPI_NT>(sum, sh);
  if (threadIdx.x == 0) {
    res[LGR_ALB_NREL] = cnt;
    res[LGR_ALB_SUM_REL] = sum;
  }
}

// part2 = { sum_rel, n_rel } over this rank's tracks
__global__ __launch_bounds__(LGD_EPI_NT) void lgd_album_part2_kernel(const double *__restrict__ res,
                                                                    int n_tracks,
                                                                    double *__restrict__ part2) {
  __shared__ double sh[LGD_EPI_NT / LGD_WAVE];
  double sr = 0.0, nr = 0.0;
  for (int t = threadIdx.x; t < n_tracks; t += LGD_EPI_NT) {
    const double *r = res + (size_t)t * LGR_STRIDE;
    sr += r[LGR_ALB_SUM_REL];
    nr += r[LGR_ALB_NREL];
  }
  sr = block_sum_f64<LGD_EPI_NT>(sr, sh);
  nr = block_sum_f64<LGD_EPI_NT>(nr, sh);
  if (threadIdx.x == 0) {
    part2[0] = sr; part2[1] = nr;
  }
}

// album[] = { loudness, lra(filled by lgd_lra_kernel), peak, thr, sum_abs, sum_rel, n_abs, n_rel, n_st }
__global__ void lgd_album_final_kernel(const double *__restrict__ part1,
                                       const double *__restrict__ part2, double rel_factor,
                                       double *__restrict__ album) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double thr = 0.0;
  if (part1[1] > 0.0) {
    thr = part1[0] / part1[1];
    thr *= rel_factor;
  }
  album[0] = part2[1] > 0.0 ? 10.0 * (log(part2[0] / part2[1]) / log(10.0)) - 0.691 : -HUGE_VAL;
  album[2] = part1[2];
  album[3] = thr;
  album[4] = part1[0];
  album[5] = part2[0];
  album[6] = part1[1];
  album[7] = part2[1];
  album[8] = part1[3];
}

// ------------------------------------------------------- launch wrappers ---
template <int C, int NCH, int TP>
static hipError_t launch_scan_t(const LgdSeg *segs, int n_seg, const LgdFilt &F, hipStream_t s) {
  hipLaunchKernelGGL((lgd_scan_kernel<C, NCH, TP>), dim3(n_seg), dim3(LGD_WAVE), 0, s, segs, F);
  return hipGetLastError();
}

template <int C>
static hipError_t launch_scan_c(int nch, int tp, const LgdSeg *segs, int n_seg, const LgdFilt &F,
                                hipStream_t s) {
  if (nch == 1) {
    if (tp == 4) return launch_scan_t<C, 1, 4>(segs, n_seg, F, s);
    if (tp == 2) return launch_scan_t<C, 1, 2>(segs, n_seg, F, s);
    return launch_scan_t<C, 1, 0>(segs, n_seg, F, s);
  }
  if (tp == 4) return launch_scan_t<C, 2, 4>(segs, n_seg, F, s);
  if (tp == 2) return launch_scan_t<C, 2, 2>(segs, n_seg, F, s);
  return launch_scan_t<C, 2, 0>(segs, n_seg, F, s);
}

// chunk lengths compiled in; the host picks one that divides the rate's s100
extern "C" const int lgd_chunk_table[] = {25, 35, 45, 49, 50, 63, 75, 0};

extern "C" hipError_t lgd_launch_scan(int chunk, int nch, int tp, const LgdSeg *segs, int n_seg,
                                      const LgdFilt *F, hipStream_t s) {
  if (n_seg <= 0) return hipSuccess;
  switch (chunk) {
    case 25: return launch_scan_c<25>(nch, tp, segs, n_seg, *F, s);
    case 35: return launch_scan_c<35>(nch, tp, segs, n_seg, *F, s);
    case 45: return launch_scan_c<45>(nch, tp, segs, n_seg, *F, s);
    case 49: return launch_scan_c<49>(nch, tp, segs, n_seg, *F, s);
    case 50: return launch_scan_c<50>(nch, tp, segs, n_seg, *F, s);
    case 63: return launch_scan_c<63>(nch, tp, segs, n_seg, *F, s);
    case 75: return launch_scan_c<75>(nch, tp, segs, n_seg, *F, s);
    default: return hipErrorInvalidValue;
  }
}

extern "C" hipError_t lgd_launch_track_epilogue(const LgdTrackMeta *meta, int n_tracks,
                                                const double *E, double *st, const float *peaks,
                                                double *res, double abs_gate, double rel_factor,
                                                int do_tp, hipStream_t s) {
  if (n_tracks <= 0) return hipSuccess;
  hipLaunchKernelGGL(lgd_track_epilogue, dim3(n_tracks), dim3(LGD_EPI_NT), 0, s, meta, E, st, peaks,
                     res, abs_gate, rel_factor, do_tp);
  return hipGetLastError();
}

extern "C" hipError_t lgd_launch_lra(const void *ranges, int n_ranges, const double *st_base,
                                     double minus20, hipStream_t s) {
  if (n_ranges <= 0) return hipSuccess;
  hipLaunchKernelGGL(lgd_lra_kernel, dim3(n_ranges), dim3(LGD_LRA_NT), 0, s,
                     (const LgdRange *)ranges, st_base, minus20);
  return hipGetLastError();
}

extern "C" hipError_t lgd_launch_album_part1(const double *res, int n_tracks, double *part1,
                                             hipStream_t s) {
  hipLaunchKernelGGL(lgd_album_part1_kernel, dim3(1), dim3(LGD_EPI_NT), 0, s, res, n_tracks, part1);
  return hipGetLastError();
}

extern "C" hipError_t lgd_launch_album_stage2(const LgdTrackMeta *meta, int n_tracks,
                                              const double *E, double *res, const double *part1,
                                              double *part2, double abs_gate, double rel_factor,
                                              hipStream_t s) {
  if (n_tracks > 0)
    hipLaunchKernelGGL(lgd_album_pass2_kernel, dim3(n_tracks), dim3(LGD_EPI_NT), 0, s, meta, E, res,
                       part1, abs_gate, rel_factor);
  hipLaunchKernelGGL(lgd_album_part2_kernel, dim3(1), dim3(LGD_EPI_NT), 0, s, res, n_tracks, part2);
  return hipGetLastError();
}

extern "C" hipError_t lgd_launch_album_final(const double *part1, const double *part2,
                                             double rel_factor, double *album, hipStream_t s) {
  hipLaunchKernelGGL(lgd_album_final_kernel, dim3(1), dim3(1), 0, s, part1, part2, rel_factor, album);
  return hipGetLastError();
}
