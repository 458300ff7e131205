// lgd_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the EBU R128 scan.
//
// What they replace (all third-party libebur128 work reached from
// /root/reference/src/scan.c:448 and :294-303,:383-388):
//   lgd_scan_kernel      E3 K-weighting (K2) + sample peak (K1) + E4 true peak
//                        (K3) + the 100 ms partial sums behind E5/E6 (K4)
// (the gating / LRA / album epilogue kernels live in lgd_epilogue.hip)
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

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Pointers that arrive inside a descriptor in memory lose their address space and
// compile to flat_* accesses, which count in lgkmcnt too: every LDS wait would
// then also wait for the prefetched tile.  These casts make them global_*.
#define LGD_GLOBAL __attribute__((address_space(1)))
typedef const f32x4 LGD_GLOBAL *gvec_ptr;
typedef const float LGD_GLOBAL *gflt_ptr;

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

  // ---- tile staging: coalesced 16-B loads -> registers -> LDS.  The NEXT tile's
  // loads are issued before the current tile is computed (software prefetch,
  // NV float4 per lane in flight); tiles touching a track edge take the guarded
  // path, where frames outside [0, n_frames) read as zero.
  constexpr int NV = (K::NVEC + LGD_WAVE - 1) / LGD_WAVE;
  f32x4 pf[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) pf[i] = (f32x4)(0.f);
  bool pf_valid = false;
#define LGD_TILE_G0(kk) ((sg.f0 + (long long)(kk) * K::TILE_F - K::HALO) * NCH - shift)
#define LGD_PREFETCH(kk)                                                                \
  do {                                                                                  \
    const long long g0_ = LGD_TILE_G0(kk); /* float index of lds[0], multiple of 4 */   \
    pf_valid = (g0_ >= 0) && (g0_ + 4LL * K::NVEC <= sg.n_floats); /* wave-uniform */   \
    if (pf_valid) {                                                                     \
      const gvec_ptr src_ = (gvec_ptr)(sg.pcm + g0_);                                   \
      _Pragma("unroll") for (int i_ = 0; i_ < NV; ++i_) {                               \
        const int idx_ = lane + LGD_WAVE * i_;                                          \
        if (i_ + 1 < NV || idx_ < K::NVEC) pf[i_] = src_[idx_];                         \
      }                                                                                 \
    }                                                                                   \
  } while (0)
  if (-sg.n_warm_tiles < n_main) LGD_PREFETCH(-sg.n_warm_tiles);

  for (int k = -sg.n_warm_tiles; k < n_main; ++k) {
    const long long tb = sg.f0 + (long long)k * K::TILE_F;  // first frame of the tile
    __syncthreads();  // previous tile's LDS reads are done
    if (pf_valid) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int idx = lane + LGD_WAVE * i;
        if (i + 1 < NV || idx < K::NVEC) *reinterpret_cast<f32x4 *>(lds + 4 * idx) = pf[i];
      }
    } else {
      const long long g0 = LGD_TILE_G0(k);
      for (int i = lane; i < K::NVEC; i += LGD_WAVE) {
        const long long g = g0 + 4LL * i;
        const gflt_ptr gp = (gflt_ptr)sg.pcm;
        f32x4 v;
        if (g >= 0 && g + 3 < sg.n_floats) {
          v = *(gvec_ptr)(sg.pcm + g);
        } else {
          v.x = (g + 0 >= 0 && g + 0 < sg.n_floats) ? gp[g + 0] : 0.f;
          v.y = (g + 1 >= 0 && g + 1 < sg.n_floats) ? gp[g + 1] : 0.f;
          v.z = (g + 2 >= 0 && g + 2 < sg.n_floats) ? gp[g + 2] : 0.f;
          v.w = (g + 3 >= 0 && g + 3 < sg.n_floats) ? gp[g + 3] : 0.f;
        }
        *reinterpret_cast<f32x4 *>(lds + 4 * i) = v;
      }
    }
    __syncthreads();
    if (k + 1 < n_main) LGD_PREFETCH(k + 1); else pf_valid = false;

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
        if (lane == 0 && cur < sg.n_sb) ((double LGD_GLOBAL *)sg.e_out)[cur] = tot;
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
      ((float LGD_GLOBAL *)sg.peak_out)[ch] = s;
      ((float LGD_GLOBAL *)sg.peak_out)[NCH + ch] = t;
    }
  }
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

