// lgd_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the EBU R128 scan.
//
// What they replace (all third-party libebur128 work reached from
// /root/reference/src/scan.c:448 and :294-303,:383-388):
//   lgd_scan_kernel        E3 K-weighting (K2) + sample peak (K1) + the 100 ms partial sums behind
//                          E5/E6 (K4) + every chunk's largest |x| for the true-peak kernel
//   lgd_peak_reduce_kernel per-channel sample peak of each track (the pruning bound)
//   lgd_tp_kernel          E4 true peak (K3): the 4x / 2x interpolator where it can exceed that bound
// (the gating / LRA / album epilogue kernels live in lgd_epilogue.hip)
//
// Parallelisation of the strictly sequential IIR:
// one wavefront owns one channel of a run of whole 100 ms sub-blocks.  It walks that run in
// tiles of 64 lanes x C frames; inside a tile every lane
//   A. runs the 4th-order recurrence from a ZERO state over its C frames
//      (gives the zero-state final state z_lane),
//   B. a 6-step wave scan of the affine maps  s -> A^C s + z  turns the z_lane
//      into the exact filter state at the start of every lane's chunk,
//   C. re-runs its C frames from that state, now producing y, y^2 and peaks.
// The sections of a tile run at different wave priorities (LGD_PRIO_*): the SIMD's other
// wave hides the latency of the fp64 dependency chains.  (The run-time-channel kernel
// additionally splits a lane's chunk into two half-chunks issued interleaved.)
// The state that enters a segment comes from `n_warm_tiles` tiles of A+B only
// over the audio just before it (default 200 ms: the filter's memory is < 1e-17 by then).
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
// tiles whose chunk maxima travel in one 16-B store per lane (8 bf16 values)
#define LGD_ROW_TILES 8

// Wave priorities per section of a tile (measured, tools/prio_sweep.sh): the latency-bound
// sections (staging, wave scan) first, and phase C above phase A -- the wave closer to
// the end of its tile wins the SIMD, so its workgroup reaches the next barrier (and issues
// the next tile's loads) sooner.  Equal priorities for A and C cost 3-5 %; an extra
// asymmetry by hardware wave slot gained nothing.
#ifndef LGD_PRIO_STAGE
#define LGD_PRIO_STAGE 3
#endif
#ifndef LGD_PRIO_SCAN
#define LGD_PRIO_SCAN 3
#endif
#ifndef LGD_PRIO_A
#define LGD_PRIO_A 1
#endif
#ifndef LGD_PRIO_C
#define LGD_PRIO_C 2
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Pointers that arrive inside a descriptor in memory lose their address space and
// compile to flat_* accesses, which count in lgkmcnt too: every LDS wait would
// then also wait for the prefetched tile.  These casts make them global_*.
#define LGD_GLOBAL __attribute__((address_space(1)))
typedef const f32x4 LGD_GLOBAL *gvec_ptr;
typedef const float LGD_GLOBAL *gflt_ptr;
// PCM element formats.  f32 (scale 1.0, ebur128_add_frames_float) or the reference's own feed, interleaved S16
// (scan.c:442-448, ebur128_add_frames_short: x / 32768).  The S16 variants widen at staging time to the INTEGER-valued
// float and carry the factor 2^-15 (2^-30 for energies) to the few values they store: every operation in between is
// linear, scaling by a power of two commutes with each rounding, so their results are bit-identical to the f32 variants'
// on the same samples -- at half the HBM traffic (2 B per sample) and one v_cvt_f32_i32 per sample.
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <bool S16> struct PcmIO {
  typedef float elem;
  typedef f32x4 vec;
  static constexpr float peak_scale = 1.f;
  static constexpr double energy_scale = 1.0;
};
template <> struct PcmIO<true> {
  typedef short elem;
  typedef s16x4 vec;
  static constexpr float peak_scale = 1.f / 32768.f;
  static constexpr double energy_scale = 1.0 / (32768.0 * 32768.0);
};
__device__ __forceinline__ f32x4 lgd_widen(const f32x4 v) { return v; }
__device__ __forceinline__ f32x4 lgd_widen(const s16x4 v) { return (f32x4){(float)v.x, (float)v.y, (float)v.z, (float)v.w}; }

// ---------------------------------------------------------------- helpers ---
// Wave-wide sums through the DPP cross-lane paths of the VALU (no LDS round trip
// as with ds_bpermute): butterfly inside each row of 16 lanes, then the row
// totals travel by row_bcast.  Fixed order -> reproducible.  Total in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_f64<0x140, 0xF>(v);  // row_mirror: every lane holds its row's sum
  v += dpp_f64<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_f64<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return __shfl(v, LGD_WAVE - 1, LGD_WAVE);
}

// two adjacent LDS dwords from two separate VGPRs.  Inline asm: the compiler does not
// count it in lgkmcnt, so the caller drains with s_waitcnt lgkmcnt(0) before the barrier.
__device__ __forceinline__ void lgd_lds_write2(float *p, float a, float b) {
  const unsigned off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)p;
  asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" : : "v"(off), "v"(a), "v"(b) : "memory");
}

// lane l <- lane l-1, lane 0 <- fill: DPP wave_shr:1 (a VALU move, no LDS round trip;
// lanes without a source keep `old`)
__device__ __forceinline__ double lgd_wave_shr1(double v, double fill) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, LGD_WAVE));
  return v;
}
// the same through the DPP paths, result wave-uniform (an SGPR); v_max_f32 drops NaNs
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_max_f32_uniform(float v) {
  v = fmaxf(v, dpp_f32<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_f32<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_f32<0x141, 0xF>(v));  // row_half_mirror
  v = fmaxf(v, dpp_f32<0x140, 0xF>(v));  // row_mirror: every lane holds its row's maximum
  v = fmaxf(v, dpp_f32<0x142, 0xA>(v));  // row_bcast:15 into rows 1 and 3
  v = fmaxf(v, dpp_f32<0x143, 0xC>(v));  // row_bcast:31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), LGD_WAVE - 1));
}
// inclusive prefix sum over the wave (Kogge-Stone inside the rows of 16, row totals by row_bcast)
__device__ __forceinline__ int wave_incl_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// ------------------------------------------------------------ scan kernel ---
// One workgroup per segment, one WAVEFRONT PER CHANNEL: the nch waves of a
// workgroup stage one interleaved tile in LDS together and each filters its own
// channel (so registers and LDS per wave do not grow with the channel count, and
// any channel count up to 16 runs the same code).
// C   frames per lane (divides the 100 ms sub-block length of the rate)
// G   channels == waves per workgroup; 0 = run-time value (frame stride in LDS
//     is then a register instead of an immediate)
// TP  0 = no interpolator (>= 192 kHz or disabled); otherwise the chunk maxima of every tile
//     are recorded for lgd_tp_kernel (instantiated as 4 for both the 4x and the 2x interpolator)
// LDS layout of a staged tile.
//  G == 0 (run-time channel count): interleaved as in memory, frame stride nch.
//  G >= 1 (compiled for 1 .. 6 and 8): PLANAR, one plane per channel, and every lane's C-frame chunk is
//  followed by PAD unused floats so that the lane stride C + PAD is odd: 64 lanes
//  reading the same chunk position then hit 32 different banks (an even stride,
//  e.g. interleaved stereo, costs 2..32-way ds_read conflicts).
//  Frame f (tile-relative, f >= -HALO) of a plane sits at
//     PO + f + floor(f / C) * PAD,   PO = HALO + PAD + (run-time alignment shift)
template <int C, int G>
struct LdsLayout {
  static constexpr bool PLANAR = (G >= 1);
  // the 16-B alignment shift of a tile (0..3 floats) is a whole number of frames for
  // 1, 2 (and 4) channels; otherwise (5.1: G = 6) the planes are addressed one frame
  // further in and the channel of a staged float is (index - shift) mod G
  static constexpr bool SHIFT_WHOLE = (G == 0) || (4 % (G ? G : 1) == 0);
  static constexpr int PAD = (PLANAR && (C % 2 == 0)) ? 1 : 0;
  static constexpr int STRIDE = C + PAD;  // lane stride inside a plane
};

template <int C, int TP>
struct ScanCfg {
  static constexpr int HALO = 12;  // frames kept before the tile (the filter needs 2)
  static constexpr int TILE_F = LGD_WAVE * C;
  // 16-B vectors per thread and tile: ceil(((TILE_F + HALO) nch + 4) / 4 / (64 nch))
  static constexpr int NV = (16 * C + HALO / 4 + 1 + 63) / 64;
  // frames per streamed step: the largest divisor of C not above 8
  static constexpr int U = lgd_unroll(C);
};

// (launch bounds: G waves per workgroup, >= 2 waves per SIMD wanted -> <= 256 VGPRs;
// the run-time-G variant must fit 16 waves -> <= 128 VGPRs, so the host gives it
// short chunks)
// WIDE (run-time-G kernel only): up to 16 waves per workgroup -> <= 128 VGPRs;
// otherwise up to 8 waves (<= 256 VGPRs, no spills).
// STR (G = 1, 2 or 3 only): the workgroup takes channels [sg.ch0, sg.ch0 + G) of a stream of sg.nch_total
// interleaved channels -- one frame per lane and load (a dword or an aligned pair), sibling
// workgroups of the other channel pairs read the same lines through L2 / Infinity Cache.  Every
// layout then runs the mono / stereo kernel's long chunks (a six-plane tile of C = 75 would not
// fit LDS twice).
template <int C, int G, int TP, bool WIDE = false, bool STR = false, bool S16 = false>
__global__ __attribute__((amdgpu_flat_work_group_size(G ? LGD_WAVE * G : 64,
                                                      G ? LGD_WAVE * G : (WIDE ? 1024 : 512)),
                          amdgpu_waves_per_eu(G ? (G > 2 ? 3 : 2) : (WIDE ? 4 : 2), 4))) void lgd_scan_kernel(
    const LgdSeg *__restrict__ segs, const int nch_rt) {
  using K = ScanCfg<C, TP>;
  using IO = PcmIO<S16>;
  typedef const typename IO::elem LGD_GLOBAL *pelem_ptr;
  typedef const typename IO::vec LGD_GLOBAL *pvec_ptr;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  // the per-(rate, chunk) constants live in constant memory: uniform loads from
  // this address space are scalar loads (s_load), so the 72 doubles of scan
  // matrices are fetched per tile instead of occupying (and spilling) SGPRs
  typedef const LgdFilt __attribute__((address_space(4))) *cfilt_ptr;
#define F (*F0)
  const int nch = G ? G : nch_rt;          // channels == waves in this workgroup
  const int tid = threadIdx.x;
  const int lane = tid & (LGD_WAVE - 1);
  // this wave's channel, as a scalar so that channel-dependent branches are uniform
  const int ch = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = LGD_WAVE * nch;
  const LgdSeg sg = segs[blockIdx.x];
  // the constants of this segment's (rate, chunk): per segment, so that one launch can carry the
  // segments of several sample rates
  const cfilt_ptr F0 = (cfilt_ptr)sg.filt;
  static_assert(!STR || G == 1 || G == 2 || G == 3 || G == 4, "strided variant: one to four channels per workgroup");
  const int shift = (STR || (G == 0 && sg.nch_total != (G ? G : nch_rt))) ? 0 : (int)((sg.f0 * nch) & 3);
  const long long n_frames = sg.n_frames;
  const int nvec = ((K::TILE_F + K::HALO) * nch + 4) >> 2;  // 16-B vectors per tile
  // frame slot of tile frame -HALO inside a plane
  using LL = LdsLayout<C, G>;
  const int slot_shift = STR ? 0 : (LL::SHIFT_WHOLE ? shift / (G ? G : 1) : 1);  // (strided staging: frame slots as they are)
  // vector i of a thread is in range for EVERY thread when this holds (compile-time
  // for the fixed-channel-count kernels: no exec masking around full vectors)
#define LGD_VEC_ALWAYS(i_) (G != 0 && (LGD_WAVE * G) * ((i_) + 1) <= (((K::TILE_F + K::HALO) * G + 4) >> 2))
  // floats per plane: HALO + shift slack + 64 padded chunks (+ tail slack), even
  // (+8: the software-pipelined reads fetch up to one step past the last chunk)
  constexpr int PLANE = (K::HALO + 4 + LL::PAD + LGD_WAVE * LL::STRIDE + 4 + 8 + 1) & ~1;
  // channel groups (streams with more than 16 channels, run-time-G kernel only):
  // this workgroup handles channels [ch0, ch0 + nch) of an nch_tot-channel stream
  const int ch0 = (G && !STR) ? 0 : sg.ch0;
  const int nch_tot = (G && !STR) ? G : sg.nch_total;
  const bool grouped = (G == 0) && (nch_tot != nch);
  // strided sets of a channel count they do not divide overlap (5 channels: triples 0-2 and 2-4): the later set
  // leaves the channels it shares to the earlier one -- its wave goes the way of a channel without loudness
  // weight (peaks only, a quarter of the work) and writes nothing but empty true-peak records.  (Leaving the tile
  // loop's body altogether -- `continue` behind the staging -- cost the OTHER waves: 7 channels 0.43 -> 0.61 ms
  // with the branch merely compiled in, no wave taking it.)
  const bool skip = STR && ((sg.skip_mask >> ch) & 1u);            // wave-uniform
  const bool filt = !skip && lgd_channel_weight(ch0 + ch, nch_tot) > 0.0;  // wave-uniform

  const double ra1 = F.ra[0], ra2 = F.ra[1], pa1 = F.pa[0], pa2 = F.pa[1];
  const double c1 = F.pbn[0], c2 = F.pbn[1];
  // (alpha, beta, gamma, 1/alpha, 1/beta, dc and pb0^2 are used once per tile: read through the
  // per-tile constants pointer where they are needed, not held in SGPRs across the tile loop)
  const int lps = F.lps;
  const int pskip = F.pskip;
  const int s100 = F.s100;
  // floor-measurement modes: 1 no global loads, 2 no arithmetic,
  // 4/8/16 skip phase A / the scan / phase C, 64 loads issued but never staged.
  // Compiled in only with -DLGD_DEBUG_MODES (make DEBUG_MODES=1).
#ifdef LGD_DEBUG_MODES
  const int dbg = F.pad;
#else
  constexpr int dbg = 0;
#endif
  double cin[4] = {0.0, 0.0, 0.0, 0.0};  // wave-uniform filter state entering the tile
  double acc = 0.0;       // this lane's share of sub-block `cur`
  int cur = 0;            // sub-block (relative to the segment) being summed
  int cur_q = 0;          // chunk index (relative to f0) where `cur` starts
  float pk_s = 0.f;
#ifdef LGD_FUSED_TP
  float pk_tp = 0.f;  // experiment: the 4x interpolator for EVERY frame while the tile is in LDS (make libloudscan_hip_fused.so, tools/variant_probe.sh)
#endif
  // (true peak: this kernel only records every chunk's largest |x|; lgd_tp_kernel decides from
  // them which interpolator outputs can matter and evaluates those)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 mc_rows = (u32x4)(0u);  // this lane's chunk bounds of the last 8 tiles (bf16, newest on top)
  float car_m = 0.f, car_p = 0.f;  // wave-uniform: largest |x| / adjacent-pair sum of the chunk in front of lane 0's
  // where this lane's next record goes: [group of 8 tiles][channel][lane] (a per-lane pointer in
  // VGPRs: the scalar form kept the base and the group index live in SGPRs through the tile loop)
  u32x4 LGD_GLOBAL *row_p = (u32x4 LGD_GLOBAL *)sg.tp_rows + (ch * LGD_WAVE + lane);
  // Mono / stereo / channel-triple / 7.1 workgroups have registers to spare (the planar three- and four-plane
  // kernels at C = 50 do not: 168 VGPRs at three waves per SIMD): the records of the first LGD_ROW_PARK groups
  // (40 tiles) wait in registers and go out behind the tile loop, so that a segment of up to 48
  // tiles (C2 / C3: 36, C4: 48) stores nothing while it runs.  (A store inside the loop is
  // retired in order with the next tile's prefetch loads: when its acknowledgement is late the
  // wave waits for it.  Measured run to run, the true-peak variant of the stereo kernel was 0 to
  // 6 % slower than the plain one with one store per 8 tiles; 5.1 as triples with true peak 0.388 ->
  // 0.367 ms, 7.1 0.399 -> 0.393 ms for 345.6 M samples.)
  constexpr int LGD_ROW_PARK = (TP != 0 && ((G >= 1 && G <= 2) || ((G == 3 || G == 4) && STR) || G == 8)) ? 5 : 0;
  u32x4 parked[LGD_ROW_PARK + 1];
#pragma unroll
  for (int i = 0; i < LGD_ROW_PARK; ++i) parked[i] = (u32x4)(0u);

  const int n_main = sg.n_tiles;
#ifdef LGD_DEBUG_HWID
  // placement probe (make libloudscan_hip_hwid.so, tools/hwid_probe.py): where this wave runs, when
  const unsigned dbg_hw = __builtin_amdgcn_s_getreg(63492 /* HW_ID */), dbg_xcc = __builtin_amdgcn_s_getreg(63508 /* XCC_ID */);
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- tile staging: coalesced 16-B loads -> registers -> LDS.  The NEXT tile's
  // loads are issued before the current tile is computed (software prefetch,
  // NV x 16 B per lane in flight); tiles touching a track edge take the guarded
  // path, where frames outside [0, n_frames) read as zero.
  typename IO::vec pf[K::NV];
#pragma unroll
  for (int i = 0; i < K::NV; ++i) pf[i] = (typename IO::vec)(0);
  bool pf_valid = false;
#define LGD_TILE_G0(kk) ((sg.f0 + (long long)(kk) * K::TILE_F - K::HALO) * nch - shift)
#define LGD_PREFETCH(kk)                                                                \
  do {                                                                                  \
    const long long g0_ = LGD_TILE_G0(kk); /* float index of lds[0], multiple of 4 */   \
    pf_valid = (g0_ >= 0) && (g0_ + 4LL * nvec <= sg.n_floats) && !grouped; /* uniform */ \
    if (dbg & 1) pf_valid = false;                                                      \
    if (pf_valid) {                                                                     \
      /* uniform base per vector (scalar adds) + one per-lane offset: no 64-bit    */   \
      /* VALU address arithmetic, no per-load branches -> the loads stay batched   */   \
      const pvec_ptr src_ = (pvec_ptr)((pelem_ptr)sg.pcm + g0_);                        \
      _Pragma("unroll") for (int i_ = 0; i_ < K::NV; ++i_) {                            \
        /* the base of every vector stays a scalar (opaque to the optimiser): SGPR-base   */ \
        /* + 32-bit lane offset addressing, no 64-bit VALU adds per load                */ \
        pvec_ptr src_i_ = src_ + nthreads * i_;                                         \
        asm volatile("" : "+s"(src_i_));                                                \
        if (LGD_VEC_ALWAYS(i_) || tid + nthreads * i_ < nvec) pf[i_] = src_i_[(unsigned)tid]; \
      }                                                                                 \
    }                                                                                   \
  } while (0)
  // ---- strided variant: one frame of this workgroup's channels per lane and load
  constexpr int NFR = K::TILE_F + K::HALO;                              // frames staged per tile
  constexpr int NVS = STR ? (NFR + LGD_WAVE * (G ? G : 1) - 1) / (LGD_WAVE * (G ? G : 1)) : 1;
  // (element-aligned vector types: a channel set starts anywhere in a frame)
  typedef typename IO::elem pel_t;
  typedef pel_t pvec2a __attribute__((ext_vector_type(2)));                           // an aligned pair: one load
  typedef const pvec2a LGD_GLOBAL *gvec2_ptr;
  typedef pel_t pvec3u __attribute__((ext_vector_type(3), aligned(sizeof(pel_t))));
  typedef const pvec3u LGD_GLOBAL *gvec3_ptr;
  typename IO::vec pfs[NVS];  // (.z only by channel triples and quads, .w only by quads); raw elements: S16 is widened at staging
  typedef pel_t pvec4u __attribute__((ext_vector_type(4), aligned(sizeof(pel_t))));
  typedef const pvec4u LGD_GLOBAL *gvec4u_ptr;
#pragma unroll
  for (int i = 0; i < NVS; ++i) pfs[i] = (typename IO::vec)(0);
  // (an aligned pair of channels is one 8-B load; odd channel counts take two dwords; a triple is
  // one 12-B load)
  const bool aligned2 = STR && G == 2 && !((nch_tot | ch0) & 1);
#define LGD_PREFETCH_STR(kk)                                                            \
  do {                                                                                  \
    const long long fb_ = sg.f0 + (long long)(kk) * K::TILE_F - K::HALO;                \
    pf_valid = (fb_ >= 0) && (fb_ + NFR <= n_frames); /* uniform */                     \
    if (dbg & 1) pf_valid = false;                                                      \
    if (pf_valid) {                                                                     \
      const pelem_ptr src_ = (pelem_ptr)sg.pcm + fb_ * nch_tot + ch0;                   \
      const unsigned lo_ = (unsigned)(tid * nch_tot);                                   \
      _Pragma("unroll") for (int i_ = 0; i_ < NVS; ++i_) {                              \
        pelem_ptr src_i_ = src_ + (long long)(nthreads * i_) * nch_tot;                 \
        asm volatile("" : "+s"(src_i_));                                                \
        if (nthreads * (i_ + 1) <= NFR || tid + nthreads * i_ < NFR) {                  \
          if (G == 4) {                                                                 \
            const pvec4u t_ = *(gvec4u_ptr)(src_i_ + lo_);                              \
            pfs[i_] = (typename IO::vec){t_.x, t_.y, t_.z, t_.w};                       \
          } else if (G == 3) {                                                          \
            const pvec3u t_ = *(gvec3_ptr)(src_i_ + lo_);                               \
            pfs[i_].x = t_.x; pfs[i_].y = t_.y; pfs[i_].z = t_.z;                       \
          } else if (G == 2 && aligned2) {                                              \
            const pvec2a t_ = *(gvec2_ptr)(src_i_ + lo_);                               \
            pfs[i_].x = t_.x; pfs[i_].y = t_.y;                                         \
          } else {                                                                      \
            pfs[i_].x = src_i_[lo_];                                                    \
            if (G == 2) pfs[i_].y = src_i_[lo_ + 1u];                                   \
          }                                                                             \
        }                                                                               \
      }                                                                                 \
    }                                                                                   \
  } while (0)
#define LGD_PREFETCH_ANY(kk) do { if constexpr (STR) LGD_PREFETCH_STR(kk); else LGD_PREFETCH(kk); } while (0)
  if (-sg.n_warm_tiles < n_main) LGD_PREFETCH_ANY(-sg.n_warm_tiles);

  for (int k = -sg.n_warm_tiles; k < n_main; ++k) {
    const long long tb = sg.f0 + (long long)k * K::TILE_F;  // first frame of the tile
    __builtin_amdgcn_s_setprio(LGD_PRIO_STAGE);  // staging: get the stores, the barrier and the next loads out first
    __syncthreads();  // every wave is done reading the previous tile
    // LDS slot of frame f of a plane (planar layouts): PO_W + f + floor(f / C) * PAD,
    // written in terms of the vector's own frame slot jj = f + HALO + shift / G
#define LGD_STORE_VEC(idx_, v_)                                                          \
    do {                                                                                 \
      if constexpr (!LL::PLANAR) {                                                       \
        *reinterpret_cast<f32x4 *>(lds + 4 * (idx_)) = (v_);                             \
      } else if constexpr (G == 1 && LL::PAD == 0) {                                     \
        *reinterpret_cast<f32x4 *>(lds + 4 * (idx_)) = (v_);                             \
      } else if constexpr (G == 2 && LL::PAD == 0) {                                     \
        /* two ds_write2_b32, each taking its two dwords from two separate VGPRs: the   */ \
        /* de-interleave costs no register shuffling (left to itself hipcc merges the   */ \
        /* stores into 64-bit ones and pays four v_mov per vector to pair the data)     */ \
        const int jj_ = 2 * (idx_);                                                      \
        lgd_lds_write2(lds + jj_, (v_).x, (v_).z);                                       \
        lgd_lds_write2(lds + PLANE + jj_, (v_).y, (v_).w);                               \
      } else {                                                                           \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                               \
          const int i_ = LL::SHIFT_WHOLE ? 4 * (idx_) + e_ : 4 * (idx_) + e_ - shift + G; \
          const int c_ = i_ % G, jj_ = i_ / G;                                           \
          /* f + C >= 0 always (HALO <= C): floor(f / C) = (f + C) / C - 1 */            \
          const int fpc_ = jj_ - K::HALO - slot_shift + C;                               \
          lds[c_ * PLANE + jj_ + LL::PAD * (int)((unsigned)fpc_ / (unsigned)C)] = (v_)[e_]; \
        }                                                                                \
      }                                                                                  \
    } while (0)
    if constexpr (STR) {
      // frame fr (0 = tile frame -HALO) of plane c sits at fr + PAD * floor((fr - HALO + C) / C)
#define LGD_STORE_FRAME(fr_, a_, b_, c_, d_)                                            \
      do {                                                                              \
        const int at_ = (fr_) + LL::PAD * (int)((unsigned)((fr_) - K::HALO + C) / (unsigned)C); \
        lds[at_] = (a_);                                                                \
        if (G >= 2) lds[PLANE + at_] = (b_);                                            \
        if (G >= 3) lds[2 * PLANE + at_] = (c_);                                        \
        if (G >= 4) lds[3 * PLANE + at_] = (d_);                                        \
      } while (0)
      if (pf_valid && !(dbg & 64)) {
#pragma unroll
        for (int i = 0; i < NVS; ++i) {
          const int fr = tid + nthreads * i;
          if (nthreads * (i + 1) <= NFR || fr < NFR)
            LGD_STORE_FRAME(fr, (float)pfs[i].x, (float)pfs[i].y, (float)pfs[i].z, (float)pfs[i].w);
        }
      } else if (!(dbg & 1) && !(dbg & 64)) {  // a tile at a track edge: frames outside the track are zero
        const pelem_ptr gp = (pelem_ptr)sg.pcm + ch0;
        for (int fr = tid; fr < NFR; fr += nthreads) {
          const long long f = tb - K::HALO + fr;
          const bool in = f >= 0 && f < n_frames;
          const float a = in ? (float)gp[f * nch_tot] : 0.f;
          const float b = (in && G >= 2) ? (float)gp[f * nch_tot + 1] : 0.f;
          const float c = (in && G >= 3) ? (float)gp[f * nch_tot + 2] : 0.f;
          const float d = (in && G >= 4) ? (float)gp[f * nch_tot + 3] : 0.f;
          LGD_STORE_FRAME(fr, a, b, c, d);
        }
      }
#undef LGD_STORE_FRAME
    } else if (pf_valid && !(dbg & 64)) {
#pragma unroll
      for (int i = 0; i < K::NV; ++i) {
        const int idx = tid + nthreads * i;
        if (LGD_VEC_ALWAYS(i) || idx < nvec) LGD_STORE_VEC(idx, lgd_widen(pf[i]));
      }
      if constexpr (LL::PLANAR && G == 2 && LL::PAD == 0)
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the asm stores above
    } else if (grouped) {
      // a channel group of a wide stream: gather [frame][ch0 .. ch0 + nch) float by float
      const pelem_ptr gp = (pelem_ptr)sg.pcm;
      const int nfl = (K::TILE_F + K::HALO) * nch;
      for (int i = tid; i < nfl; i += nthreads) {
        const int fr = i / nch, c = i - fr * nch;
        const long long f = tb - K::HALO + fr;
        lds[i] = (f >= 0 && f < n_frames) ? (float)gp[f * nch_tot + ch0 + c] : 0.f;
      }
    } else if (!(dbg & 1) && !(dbg & 64)) {
      const long long g0 = LGD_TILE_G0(k);
      const pelem_ptr gp = (pelem_ptr)sg.pcm;
      for (int i = tid; i < nvec; i += nthreads) {
        const long long g = g0 + 4LL * i;
        f32x4 v;
        if (g >= 0 && g + 3 < sg.n_floats) {
          v = lgd_widen(*(pvec_ptr)(gp + g));
        } else {
          v.x = (g + 0 >= 0 && g + 0 < sg.n_floats) ? (float)gp[g + 0] : 0.f;
          v.y = (g + 1 >= 0 && g + 1 < sg.n_floats) ? (float)gp[g + 1] : 0.f;
          v.z = (g + 2 >= 0 && g + 2 < sg.n_floats) ? (float)gp[g + 2] : 0.f;
          v.w = (g + 3 >= 0 && g + 3 < sg.n_floats) ? (float)gp[g + 3] : 0.f;
        }
        LGD_STORE_VEC(i, v);
      }
    }
#undef LGD_STORE_VEC
    __syncthreads();
    if (k + 1 < n_main) LGD_PREFETCH_ANY(k + 1); else pf_valid = false;
    // the chunk maxima of the last LGD_ROW_TILES tiles go out here
    if constexpr (TP != 0) {
      // (behind the loads just issued, and only one store per LGD_ROW_TILES tiles: a vector store
      // costs the wave ~0.4 us here whatever its width -- one per tile was 10 % of the kernel)
      // (a launch may carry segments of a rate without interpolator -- 192 kHz: tp_rows is null)
      if (k > 0 && (k & (LGD_ROW_TILES - 1)) == 0 && sg.tp_rows != nullptr) {
        const int gp = (k >> 3) - 1;  // the group just completed (wave-uniform)
        if (gp < LGD_ROW_PARK) {
#pragma unroll
          for (int i = 0; i < LGD_ROW_PARK; ++i)
            if (gp == i) parked[i] = mc_rows;
        } else {
          row_p[(size_t)gp * (nch * LGD_WAVE)] = mc_rows;
        }
      }
    }
    __builtin_amdgcn_s_setprio(LGD_PRIO_A);

    // this lane's chunk of this wave's channel: frames [tb + lane*C, +C), frame
    // stride nch floats, streamed from LDS U frames at a time
    constexpr int U = K::U;
    // this lane's chunk; LGD_X(j) = frame j of it (j < 0: history in the previous chunk)
    const float *chunk = LL::PLANAR
        ? lds + ch * PLANE + (K::HALO + slot_shift) + lane * LL::STRIDE  // + PAD folded below
        : lds + shift + (K::HALO + lane * C) * nch + ch;
#define LGD_X(j) (LL::PLANAR ? chunk[(j) + ((j) < 0 ? 0 : LL::PAD)] : chunk[(j) * (G ? G : nch)])

    // per-tile opaque copy of the constants pointer: keeps the scan-matrix loads
    // inside the tile loop (scalar cache hits) instead of hoisted into SGPRs
    cfilt_ptr Fk = F0;
    asm volatile("" : "+s"(Fk));
    if (dbg & 2) continue;
    // start state of this lane's chunk, as (q1, q2) and (p1, p2)
    double qs[2] = {0.0, 0.0}, ps[2] = {0.0, 0.0};
    // (One recurrence per lane.  Two interleaved half-chunks per lane were worth 8 % while
    // both waves of a SIMD competed at equal priority; with the phase priorities (LGD_PRIO_*)
    // the other wave hides the dependency latency and the split only cost its combine
    // steps: one stream measured -2 % at 48 kHz, -5 % at 44.1 kHz, -10 % at 22.05 kHz.)
    if (filt) {
      // ---- A: zero-state run of q' = x/ra, p' = q'/pa over the chunk (4 FMAs per
      // sample).  (1 - z^-1)^2 commutes with both filters, so second differences of the
      // last four q', p' give the zero-state (q, p) at the chunk's end; q', p' stay
      // <= ~C^2 |x| here, so the differencing costs ~1e-13 |x| at most. -------------
      double z[4];
      {
        double qv[4] = {0.0, 0.0, 0.0, 0.0}, pv[4] = {0.0, 0.0, 0.0, 0.0};
        // LDS reads run one step ahead of the arithmetic (software pipeline)
        float xa[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xa[u] = LGD_X(u);
        // drain here, so that inside the loop the only LDS reads in flight are the
        // NEXT step's (hipcc otherwise merges the pre-loop state into the loop and
        // waits for the reads it has just issued)
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        const int a_end = (dbg & 4) ? 0 : C;
#pragma unroll
        for (int j0 = 0; j0 < a_end; j0 += U) {
          float xn[U];
#pragma unroll
          for (int u = 0; u < U; ++u) xn[u] = LGD_X(j0 + U + u);  // (one step past the chunk at the end: slack)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            double t = fma(-ra2, qv[1], (double)xa[u]);
            const double q0 = fma(-ra1, qv[0], t);
            t = fma(-pa2, pv[1], q0);
            const double p0 = fma(-pa1, pv[0], t);
            qv[3] = qv[2]; qv[2] = qv[1]; qv[1] = qv[0]; qv[0] = q0;
            pv[3] = pv[2]; pv[2] = pv[1]; pv[1] = pv[0]; pv[0] = p0;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) xa[u] = xn[u];
        }
        // zero-state end state in the scan basis; the run assumed a zero input
        // history: the true x[-1], x[-2] in front of the chunk change its w[0] by
        // -2x[-1] + x[-2] and its w[1] by x[-1] (g = their effect on the end state)
        const double q1 = (qv[0] - 2.0 * qv[1]) + qv[2];
        const double q2 = (qv[1] - 2.0 * qv[2]) + qv[3];
        const double p1 = (pv[0] - 2.0 * pv[1]) + pv[2];
        const double p2 = (pv[1] - 2.0 * pv[2]) + pv[3];
        const double xm1 = (double)LGD_X(-1), xm2 = (double)LGD_X(-2);
        const double dw0 = fma(-2.0, xm1, xm2), dw1 = xm1;
        const auto *g0 = Fk->gC[0];
        const auto *g1 = Fk->gC[1];
        z[0] = fma(g1[0], dw1, fma(g0[0], dw0, q1));
        z[1] = fma(g1[1], dw1, fma(g0[1], dw0, Fk->alpha * fma(-Fk->beta, q2, q1)));
        z[2] = fma(g1[2], dw1, fma(g0[2], dw0, Fk->gamma * p1));
        z[3] = fma(g1[3], dw1, fma(g0[3], dw0, Fk->gamma * p2));
      }

      // (latency-bound section: issue priority over the SIMD's other wave, which is most
      // likely streaming through phase A or C)
      __builtin_amdgcn_s_setprio(LGD_PRIO_SCAN);
      // ---- B: inject the carry at lane 0, then Kogge-Stone over 64 lanes.  The
      // transition is block lower triangular: rows 0,1 see columns 0,1 only. ----
      if (lane == 0) {
        const auto *P = Fk->P[0];
        z[0] = fma(P[1], cin[1], fma(P[0], cin[0], z[0]));
        z[1] = fma(P[5], cin[1], fma(P[4], cin[0], z[1]));
        z[2] = fma(P[11], cin[3], fma(P[10], cin[2], fma(P[9], cin[1], fma(P[8], cin[0], z[2]))));
        z[3] = fma(P[15], cin[3], fma(P[14], cin[2], fma(P[13], cin[1], fma(P[12], cin[0], z[3]))));
      }
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        if (dbg & 8) break;
        const int d = 1 << s;
        const auto *P = Fk->P[s];
        // once the span C * 2^s exceeds the memory of the shelf poles (radius
        // ~0.85: < 1e-19 after ~300 frames) their own 2x2 block of the transition
        // is zero to double precision: those steps move and multiply half as much
        const bool shelf = !((pskip >> s) & 1);  // wave-uniform
        double u[4];
        if (s == 0) {  // distance 1: DPP
          u[0] = lgd_wave_shr1(z[0], 0.0);
          u[1] = lgd_wave_shr1(z[1], 0.0);
          u[2] = shelf ? lgd_wave_shr1(z[2], 0.0) : 0.0;
          u[3] = shelf ? lgd_wave_shr1(z[3], 0.0) : 0.0;
        } else {
          u[0] = __shfl_up(z[0], d, LGD_WAVE);
          u[1] = __shfl_up(z[1], d, LGD_WAVE);
          if (shelf) {
            u[2] = __shfl_up(z[2], d, LGD_WAVE);
            u[3] = __shfl_up(z[3], d, LGD_WAVE);
          } else {
            u[2] = u[3] = 0.0;
          }
        }
        if (lane >= d) {  // exec-masked, no cross-lane traffic inside
          z[0] = fma(P[1], u[1], fma(P[0], u[0], z[0]));
          z[1] = fma(P[5], u[1], fma(P[4], u[0], z[1]));
          double t2 = fma(P[9], u[1], fma(P[8], u[0], z[2])), t3 = fma(P[13], u[1], fma(P[12], u[0], z[3]));
          if (shelf) {
            t2 = fma(P[11], u[3], fma(P[10], u[2], t2));
            t3 = fma(P[15], u[3], fma(P[14], u[2], t3));
          }
          z[2] = t2;
          z[3] = t3;
        }
      }
      // z is now the exact state at the END of each lane's chunk; the state at
      // its START is the previous lane's (lane 0: the carry).  Back to (q, p).
      double sv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sv[r] = lgd_wave_shr1(z[r], cin[r]);
        cin[r] = __shfl(z[r], LGD_WAVE - 1, LGD_WAVE);
      }
      qs[0] = sv[0];
      qs[1] = fma(-Fk->inv_alpha, sv[1], sv[0]) * Fk->inv_beta;
      ps[0] = sv[2] * Fk->dc;
      ps[1] = sv[3] * Fk->dc;
      __builtin_amdgcn_s_setprio(LGD_PRIO_C);
    }

    if (k < 0) continue;  // warm-up tile: only the carry matters

    // ---- C: real run: y, energy, sample peak, true-peak candidates -------
    double e = 0.0, e_next = 0.0;  // e_next: generic kernel, frames past a sub-block boundary
    // generic (run-time channel count) kernel: the chunk length need not divide the
    // sub-block length; frames j < bnd of this chunk belong to sub-block sb_l, the
    // rest to sb_l + 1 (at most one boundary per chunk: C <= s100)
    int sb_l = 0, bnd = C;
    if constexpr (G == 0) {
      const unsigned fl = (unsigned)k * K::TILE_F + (unsigned)lane * C;  // chunk start, relative to f0
      sb_l = (int)(fl / (unsigned)s100);
      const unsigned rem = (unsigned)(sb_l + 1) * (unsigned)s100 - fl;
      bnd = rem < (unsigned)C ? (int)rem : C;
    }
    float mc = 0.f;  // largest |x| of this lane's chunk
    // true-peak variants: largest |x[j]| + |x[j-1]| over the chunk's frames j (the first frame pairs with the
    // last of the chunk before): what the adjacent-pair bound of the interpolator is made of (see below)
    float pp = 0.f, xlast = (TP != 0) ? fabsf(LGD_X(-1)) : 0.f;
#define LGD_STEP_PEAKS(xs, step_)                                                       \
    do {                                                                                \
      _Pragma("unroll") for (int u_ = 0; u_ < U; ++u_) {                                \
        mc = fmaxf(mc, fabsf((xs)[u_]));                                                \
        if constexpr (TP != 0) {                                                        \
          pp = fmaxf(pp, fabsf((xs)[u_]) + xlast);                                      \
          xlast = fabsf((xs)[u_]);                                                      \
        }                                                                               \
      }                                                                                 \
    } while (0)
    if (filt) {
      double xh[2], eh = 0.0, en = 0.0;
      xh[0] = (double)LGD_X(-1); xh[1] = (double)LGD_X(-2);
      // the next U frames are fetched a step ahead of the arithmetic
      float w[U];
#pragma unroll
      for (int i = 0; i < U; ++i) w[i] = LGD_X(i);
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), see phase A
      const int c_end = (dbg & 16) ? 0 : C;
#pragma unroll
      for (int j0 = 0; j0 < c_end; j0 += U) {
        float xn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xn[u] = LGD_X(j0 + U + u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double x = (double)w[u];
          double t = fma(-2.0, xh[0], x) + xh[1];  // w[n], exact
          xh[1] = xh[0];
          xh[0] = x;
          t = fma(-ra2, qs[1], t);
          const double q0 = fma(-ra1, qs[0], t);
          t = fma(-pa2, ps[1], q0);
          const double p0 = fma(-pa1, ps[0], t);
          // y / pb0 = p0 + (pb1/pb0) p1 + (pb2/pb0) p2; pb0^2 is applied per sub-block
          const double y = fma(c2, ps[1], fma(c1, ps[0], p0));
          if constexpr (G == 0) {
            const double y2 = y * y;
            const bool lo = (j0 + u) < bnd;
            eh += lo ? y2 : 0.0;
            en += lo ? 0.0 : y2;
          } else {
            eh = fma(y, y, eh);
          }
          qs[1] = qs[0]; qs[0] = q0;
          ps[1] = ps[0]; ps[0] = p0;
        }
        LGD_STEP_PEAKS(w, j0 / U);
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = xn[u];
      }
      e = eh;
      e_next = en;
    } else {
      // channel mapped EBUR128_UNUSED (e.g. LFE): no loudness, peaks only
#pragma unroll
      for (int j0 = 0; j0 < C; j0 += U) {
        float w[U];
#pragma unroll
        for (int i = 0; i < U; ++i) w[i] = LGD_X(j0 + i);
        LGD_STEP_PEAKS(w, j0 / U);
      }
    }
#undef LGD_STEP_PEAKS

    pk_s = fmaxf(pk_s, mc);
    // ---- true peak: hand lgd_tp_kernel, per chunk, a BOUND on every interpolator output of the chunk's
    // frames (one bf16 per lane and tile); with the channel's final sample peak it decides which chunks'
    // outputs can matter at all (exact pruning).  An output is sum_k c_k x[n-k] with two dominant centre
    // taps, so |y| <= a S2 + b M for M = the largest |x| it reads and S2 = the largest |x[j]| + |x[j-1]| under
    // the centre taps (lgd_engine.cpp interp_pair_bound): 1.4 M for an isolated peak, where the plain L1 M
    // is 1.86 M -- on noise that flags 40 times fewer chunks, and the flagged chunks are what the
    // true-peak kernel has to fetch again (with L1 M: 0.8 GB of the 1.4 GB of a 60 min noise track).  M and
    // S2 are taken over this chunk and the one before it (the outputs read HX frames of that; lane 0: the
    // last chunk of the previous tile; in front of the segment: the HX frames kept before the tile).
    // (Evaluating the interpolator here, with the tile still in LDS, was measured: loud passages cluster
    // in time, the workgroup that owns one then runs long after the other 999 have finished, and the kernel
    // takes as long as that workgroup.  The follow-up kernel spreads the chunks that matter over the GPU.)
    if constexpr (TP != 0) {
      if (k == 0) {  // history in front of the segment (zeros in front of the track)
        const int hx = Fk->tp_hx;
        if (hx <= K::HALO) {
          float hm = 0.f, hp = 0.f, hl = fabsf(LGD_X(-K::HALO));
#pragma unroll
          for (int j = K::HALO - 1; j >= 1; --j) {
            const float a = fabsf(LGD_X(-j));
            hm = fmaxf(hm, a);
            hp = fmaxf(hp, a + hl);
            hl = a;
          }
          car_m = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hm)));
          car_p = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hp)));
        } else {  // (2x interpolator: 23 frames of history, more than the tile keeps: nothing known)
          car_m = car_p = __builtin_inff();
        }
      }
      const float mprev = __int_as_float(__builtin_amdgcn_update_dpp(
          __float_as_int(car_m), __float_as_int(mc), 0x138, 0xf, 0xf, false));  // wave_shr:1, lane 0 <- carry
      const float pprev = __int_as_float(__builtin_amdgcn_update_dpp(
          __float_as_int(car_p), __float_as_int(pp), 0x138, 0xf, 0xf, false));
      car_m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mc), LGD_WAVE - 1));
      car_p = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pp), LGD_WAVE - 1));
      const float M = fmaxf(mc, mprev), S2 = fmaxf(pp, pprev);
      float bnd = fmaxf(fmaf(Fk->tp[30], S2, Fk->tp[31] * M), fmaf(Fk->tp[32], S2, Fk->tp[33] * M));
      if (skip) bnd = 0.f;  // (the neighbouring set owns this channel: its rows, not these, go to the true-peak kernel)
      bnd *= IO::peak_scale;  // (S16 PCM: the tile holds integer-valued samples)
      // rounded UP to bf16 (a bound may only grow), into the 8 x 16-bit shift register of the last
      // tiles (stored every 8 tiles / behind the loop)
      {
        const unsigned code = (__float_as_uint(bnd) + 0xFFFFu) >> 16;
        mc_rows.x = __builtin_amdgcn_alignbit(mc_rows.y, mc_rows.x, 16);
        mc_rows.y = __builtin_amdgcn_alignbit(mc_rows.z, mc_rows.y, 16);
        mc_rows.z = __builtin_amdgcn_alignbit(mc_rows.w, mc_rows.z, 16);
        mc_rows.w = (mc_rows.w >> 16) | (code << 16);
      }
    }
#ifdef LGD_FUSED_TP
    if constexpr (TP != 0) {
      // EXPERIMENT (never in the product build): what evaluating the interpolator inside the scan kernel costs when nothing
      // can be pruned -- every lane walks its chunk once more with a register window, as lgd_tp_kernel's dense rows do.
      // Measured (round 3, C3): the scan kernel 0.282 -> 0.547 ms on ANY material, 0.555 ms with the peak reduction --
      // against 0.58 ms for scan + follow-up kernel where nothing can be pruned and 0.30 ms on the standard material.
      if (Fk->tp_hx == 11) {
        constexpr int HXF = 11, UF = 5;
        static_assert(C % UF == 0 || true, "");
        f32x2 csd[6];
        float c2f[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { csd[i] = (f32x2){Fk->tp[18 + 2 * i], Fk->tp[19 + 2 * i]}; c2f[i] = Fk->tp[12 + i]; }
        const long long fl = tb + (long long)lane * C;
        const long long rem_ll = n_frames - fl;
        const int rem = rem_ll >= C ? C : (rem_ll < 0 ? 0 : (int)rem_ll);
        float wv[UF + HXF];
#pragma unroll
        for (int i = 0; i < HXF; ++i) wv[i] = LGD_X(i - HXF);
        if constexpr (C % UF == 0) {
#pragma unroll
          for (int st = 0; st < C / UF; ++st) {
#pragma unroll
            for (int u_ = 0; u_ < UF; ++u_) wv[HXF + u_] = LGD_X(st * UF + u_);
            f32x2 sd_[UF];
            float o2_[UF];
#pragma unroll
            for (int u_ = 0; u_ < UF; ++u_) { sd_[u_] = (f32x2){0.f, 0.f}; o2_[u_] = 0.f; }
#pragma unroll
            for (int k_ = 0; k_ < 6; ++k_) {
#pragma unroll
              for (int u_ = 0; u_ < UF; ++u_) {
                const float xa_ = wv[HXF + u_ - k_], xb_ = wv[u_ + k_];
                const f32x2 ab_ = (f32x2){xa_ + xb_, xa_ - xb_};
                sd_[u_] = __builtin_elementwise_fma(csd[k_], ab_, sd_[u_]);
                o2_[u_] = fmaf(c2f[k_], ab_.x, o2_[u_]);
              }
            }
#pragma unroll
            for (int u_ = 0; u_ < UF; ++u_) {
              const float m_ = fmaxf(fabsf(o2_[u_]), fabsf(sd_[u_].x) + fabsf(sd_[u_].y));
              pk_tp = fmaxf(pk_tp, (st * UF + u_ < rem) ? m_ : 0.f);
            }
#pragma unroll
            for (int i = 0; i < HXF; ++i) wv[i] = wv[i + UF];
          }
        }
      }
    }
#endif
#undef LGD_X

    // ---- 100 ms sub-block sums: deterministic per-lane accumulate + wave tree
    if (filt) {
      if constexpr (G != 0) {
        int rel = k * LGD_WAVE + lane - cur_q;  // (the host keeps tiles per segment < 2^24)
        for (;;) {
          const bool mine = rel >= 0 && rel < lps;
          acc += mine ? e : 0.0;
          if (k * LGD_WAVE + LGD_WAVE >= cur_q + lps) {  // `cur` ends in this tile
            const double tot = wave_sum_f64(acc) * Fk->pb0sq * IO::energy_scale;
            if (lane == 0 && cur < sg.n_sb)
              ((double LGD_GLOBAL *)sg.e_out)[(long long)(ch0 + ch) * sg.e_ch_stride + cur] = tot;
            acc = 0.0;
            ++cur;
            cur_q += lps;
            rel -= lps;
          } else {
            break;
          }
        }
      } else {
        // generic: sub-block boundaries anywhere (also inside a chunk)
        const unsigned tile_end = (unsigned)(k + 1) * K::TILE_F;  // relative to f0
        for (;;) {
          acc += (sb_l == cur ? e : 0.0) + (sb_l + 1 == cur ? e_next : 0.0);
          if (tile_end >= (unsigned)(cur + 1) * (unsigned)s100) {  // `cur` ends in this tile
            const double tot = wave_sum_f64(acc) * Fk->pb0sq * IO::energy_scale;
            if (lane == 0 && cur < sg.n_sb)
              ((double LGD_GLOBAL *)sg.e_out)[(long long)(ch0 + ch) * sg.e_ch_stride + cur] = tot;
            acc = 0.0;
            ++cur;
          } else {
            break;
          }
        }
      }
    }
  }

  if constexpr (TP != 0) {
    if (n_main > 0 && sg.tp_rows != nullptr) {
      const int g_last = (n_main - 1) >> 3;
#pragma unroll
      for (int i = 0; i < LGD_ROW_PARK; ++i)
        if (i < g_last) row_p[(size_t)i * (nch * LGD_WAVE)] = parked[i];
      row_p[(size_t)g_last * (nch * LGD_WAVE)] = mc_rows;  // the last group: n_main mod 8 tiles (8 if 0), newest on top
    }
  }
#ifdef LGD_DEBUG_HWID
  if constexpr (TP != 0) {  // the wave's record replaces the first 16 bytes of its channel's first row
    const unsigned long long dbg_t1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (lane == 0 && sg.tp_rows != nullptr)
      ((u32x4 LGD_GLOBAL *)sg.tp_rows)[ch] = (u32x4){dbg_hw, dbg_xcc, (unsigned)dbg_t0, (unsigned)dbg_t1};
  }
#endif
  if (!skip) {
    const float s = wave_max_f32(pk_s) * IO::peak_scale;
#ifdef LGD_FUSED_TP
    const float s_tp = wave_max_f32(pk_tp);
#endif
    if (lane == 0) {
      ((float LGD_GLOBAL *)sg.peak_out)[ch0 + ch] = s;
      // the interpolated peak of the segment: lgd_tp_kernel raises it (atomic max on the bits)
#ifdef LGD_FUSED_TP
      ((float LGD_GLOBAL *)sg.peak_out)[nch_tot + ch0 + ch] = TP != 0 ? s_tp : 0.f;
#else
      ((float LGD_GLOBAL *)sg.peak_out)[nch_tot + ch0 + ch] = 0.f;
#endif
    }
  }
#undef F
}

// ------------------------------------------------------- track sample peaks ---
// Per track and channel: the largest sample peak any segment found (the scan kernels'
// peak_out partials) -> hint[hint_off + ch], the bound lgd_tp_kernel prunes with.  One
// workgroup per track.  (An exchange between the running scan workgroups was measured
// instead -- polled hint words, atomic-max publishing: every variant cost the scan kernel
// 5-25 %, because all waves of a track hammer one memory channel.)
__global__ __launch_bounds__(256) void lgd_peak_reduce_kernel(const LgdTrackMeta *__restrict__ meta,
                                                             const float *__restrict__ peaks,
                                                             float *__restrict__ hint) {
  __shared__ float sh[256];
  const LgdTrackMeta m = meta[blockIdx.x];
  for (int ch = 0; ch < m.nch; ++ch) {
    float v = 0.f;
    for (int sgi = threadIdx.x; sgi < m.n_seg; sgi += 256)
      v = fmaxf(v, peaks[m.peak_off + (size_t)sgi * 2 * m.nch + ch]);
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
      if ((int)threadIdx.x < d) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + d]);
      __syncthreads();
    }
    if (threadIdx.x == 0) hint[m.hint_off + ch] = sh[0];
    __syncthreads();
  }
}
extern "C" hipError_t lgd_launch_peak_reduce(const LgdTrackMeta *meta, int n_tracks, const float *peaks,
                                             float *hint, hipStream_t s) {
  if (n_tracks <= 0) return hipSuccess;
  hipLaunchKernelGGL(lgd_peak_reduce_kernel, dim3(n_tracks), dim3(256), 0, s, meta, peaks, hint);
  return hipGetLastError();
}

// --------------------------------------------------------- true-peak kernel ---
// E4 (ebur128_check_true_peak / interp_process, reached from scan.c:448) for the chunks that
// can still raise the peak.  Row r of a segment = (tile k = r / nch, channel r mod nch): 64
// values, entry l = the largest |x| of lane l's chunk (rounded up to bf16), the C frames
// [tb + l C, + C) of the tile, tb = f0 + k * 64 C.  One wave per row; the chunks whose outputs
// could exceed the channel's final sample peak (L1 * max|x| of the chunk and the one before it) are
// "flagged".  Two ways through a row:
//   sparse (few flagged chunks): the flagged chunks are staged LGD_TP_GROUP at a time -- each with
//     the HX frames before it, eight lanes per chunk, the next group in flight while one is
//     evaluated -- and a lane takes one (chunk, step) slot of the group: U output frames from U + HX
//     staged ones, whatever chunk they belong to, so isolated chunks fill passes as well as a
//     loud passage does.  A pass is evaluated only if one of its windows can exceed the peak by the
//     ADJACENT-PAIR bound: an output is sum_k c_k x[n-k] with two dominant centre taps, so
//     |y| <= a S2 + b M (M the window's largest |x|, S2 its largest |x[j]| + |x[j+1]| under the centre
//     taps) -- 1.4 M for an isolated peak where L1 M is 1.86 M: on noise nearly every staged window
//     is dismissed by ~20 instructions instead of the ~150 of the interpolator;
//   dense (>= tp_dense_min flagged chunks: loud tones, limited music): the row is walked as a
//     whole in slabs of 64 x LP contiguous frames -- one contiguous copy global -> LDS, no per-chunk
//     addressing -- and every lane slides a register window over its LP frames: one LDS read and the
//     ~29 interpolator instructions per output frame (the sparse form re-reads 16 window values per
//     5 outputs and pays its index arithmetic per pass).  Chunks that were not flagged are evaluated
//     along: any interpolator output is a true output, the maximum cannot change.
// The non-trivial polyphase branches are evaluated in fp32 (the reference: float data, double
// accumulate, float result; difference <= 2e-7 against a bar of 1e-4):
//   4x: phases 1 and 3 mirror each other, phase 2 is symmetric: six (sum, difference) pairs of
//       window samples feed one packed FMA for (y1 + y3, y1 - y3) / 2 and one FMA for y2 --
//       24 instructions instead of 36 FMAs; max(|y1|, |y3|) = |y1 + y3| / 2 + |y1 - y3| / 2
//   2x: the one non-trivial phase is symmetric: 12 sums + 12 FMAs.
// Frames outside the track read as zero; outputs at or past the track's end do not exist in
// the reference (it stops at the last input frame) and are masked.  The wave's maximum is
// folded into the segment's interpolated peak with an atomic max on the float bits
// (non-negative floats order like their bits).
#define LGD_TP_WAVES 4
#define LGD_TP_GROUP 8     // flagged chunks staged and evaluated together (sparse rows)
#define LGD_TP_CHMAX 100   // frames of one staged chunk at most: C + HX
#define LGD_TP_AHEAD 3     // sparse rows: flagged chunks whose loads are in flight ahead of the one being looked at
// U frames per window step, TP interpolation factor, NS steps per lane of a dense slab (LP = U NS frames)
template <int U, int TP, int NS>
struct TpCfg {
  static constexpr int HX = (TP == 2) ? 23 : 11;
  static constexpr int LP = U * NS;
  static constexpr int NPRE_D = LP + 1;                       // rounds of 64 that cover 64 LP + HX slab frames
  static constexpr int BUF_D = NPRE_D * LGD_WAVE, BUF_S = LGD_TP_GROUP * LGD_TP_CHMAX;
  static constexpr int BUF = BUF_D > BUF_S ? BUF_D : BUF_S;   // floats of LDS per wave
};

// rows one wave takes: r = w + i * (waves per segment), i < LGD_TP_RPW.  The loads that decide a row's fate
// (its channel's peak, the segment's peak, the row's record) are all in flight at once, so a wave costs one
// memory latency whatever LGD_TP_RPW is -- and the kernel's fixed cost, the sweep over rows with nothing to do
// (72 000 of them in C3: 11.7 us at one row per wave), falls with the number of waves.  Consecutive rows go to
// different waves: a loud passage (consecutive tiles) is spread over a segment's waves.
// (RPW = 4 only for launches of a million rows and more -- C4's album: 5.4 M rows, 0.95 -> 0.62 ms.  With few
// rows the four rows of a wave make its life four times as long and the kernel's last, partly filled round of
// waves with it: C3's 72 000 dense rows 0.306 -> 0.329 ms; its sweep over empty rows is a latency floor, 8 us,
// that fewer waves do not lower.)
template <int U, int TP, int NS, int LGD_TP_RPW, bool S16 = false>
__global__ __launch_bounds__(LGD_WAVE * LGD_TP_WAVES) void lgd_tp_kernel(const LgdSeg *__restrict__ segs) {
  using K = TpCfg<U, TP, NS>;
  // (S16 PCM: the staged samples are integer-valued, 2^15 x the f32 variant's; thresholds are scaled up to
  // them, the wave's maximum back down -- exact, see PcmIO)
  using IO = PcmIO<S16>;
  typedef const typename IO::elem LGD_GLOBAL *pelem_ptr;
  constexpr int HX = K::HX, LP = K::LP;
  constexpr int PK = (HX - 1) / 2;  // window index of the older of the two centre-tap samples of output 0
  // per wave: the ids of the chunks of the group being evaluated, and the staged frames (a group of accepted
  // chunks: chunk c at [c * (C + HX), + C + HX), its HX frames of history first; or one slab of a dense row)
  __shared__ unsigned char chunk_of[LGD_TP_WAVES][LGD_TP_GROUP];
  __shared__ float stage[LGD_TP_WAVES][K::BUF];
  typedef const LgdFilt __attribute__((address_space(4))) *cfilt_ptr;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & (LGD_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const LgdSeg sg = segs[blockIdx.x];
  const cfilt_ptr F0 = (cfilt_ptr)sg.filt;
  const int C = sg.chunk, nch = sg.nch_wg;
  const long long tile_f = (long long)LGD_WAVE * C;
  const int n_main = sg.n_tiles;
  const int n_rows = n_main * nch;
  const int wps = (int)gridDim.y * LGD_TP_WAVES;  // waves per segment
  const int w_seg = (int)blockIdx.y * LGD_TP_WAVES + wave;
  if (w_seg >= n_rows) return;  // wave-uniform
  const int nch_tot = sg.nch_total ? sg.nch_total : nch;
  const long long n_frames = sg.n_frames;
  const int CH = C + HX, NSTEP = C / U;
  const int prune = F0->tp_prune;
  const float f_thr = F0->tp_thr, f_pthr = F0->tp[34];

  // Exact pruning.  sg.hint holds each channel's sample peak P over the whole track (lgd_peak_reduce), and
  // what ebur128_true_peak reports (E9) is max(true peak, sample peak): outputs that cannot exceed P cannot
  // change it and are not evaluated -- the result is bit-identical to evaluating everything (tp_prune 0:
  // thresholds < 0).  Two bounds, margins for the fp32 roundings included (lgd_engine.cpp):
  //   L1 * (largest |x| an output reads), L1 = largest sum |c_k| of a phase: `thr`, for whole segments;
  //   the adjacent-pair bound a S2 + b M: `pthr`, per chunk (recorded by the scan kernel) and per window.
  // Row r = (tile k = r / nch, channel r mod nch).  Its record: group g = k / 8 of the segment's tiles, 16-bit
  // slot `slot` of every lane's 16 bytes (the scan kernel shifts new tiles in at the top: a group of n tiles
  // fills slots 8 - n .. 7); entry l bounds every interpolator output of lane l's chunk.
  // A whole segment whose sample peak (exact, written by the scan kernel) stays at or below the L1 threshold
  // has no chunk that can matter -- except the first of its first tile, whose history lies in front of it.
  u32x4 rec[LGD_TP_RPW];
  float rP[LGD_TP_RPW], rS[LGD_TP_RPW];
  int rk[LGD_TP_RPW], rchan[LGD_TP_RPW], rslot[LGD_TP_RPW];
  const int g_last = (n_main - 1) >> 3;
#pragma unroll
  for (int i = 0; i < LGD_TP_RPW; ++i) {
    const int row = w_seg + i * wps;
    const bool on = row < n_rows;  // (wave-uniform)
    const int rr = on ? row : 0;
    const int k = sg.magic_nch ? (int)__umulhi((unsigned)rr, sg.magic_nch) : rr;  // row / nch
    const int ch = rr - k * nch;
    rk[i] = on ? k : -1;
    rchan[i] = sg.ch0 + ch;  // channel of the stream (channel groups of wide streams: ch0 > 0)
    const int g_ = k >> 3;
    const int n_in = g_ == g_last ? ((n_main - 1) & 7) + 1 : 8;
    rslot[i] = 8 - n_in + (k & 7);
    rP[i] = ((const float LGD_GLOBAL *)sg.hint)[rchan[i]];
    rS[i] = ((const float LGD_GLOBAL *)sg.peak_out)[rchan[i]];
    rec[i] = ((const u32x4 LGD_GLOBAL *)sg.tp_rows)[((size_t)g_ * nch + ch) * LGD_WAVE + lane];
  }
  // lane i keeps what row i needs later: the loop over the wave's rows below is not unrolled
  // (what row i needs later waits in LDS, not in lanes of VGPRs: five registers less around the loop below)
  __shared__ unsigned rowinfo[LGD_TP_WAVES][LGD_TP_RPW][5];
  bool any_row = false;
#pragma unroll
  for (int i = 0; i < LGD_TP_RPW; ++i) {
    const float thr = prune ? rP[i] * f_thr : -1.f, pthr = prune ? rP[i] * f_pthr : -1.f;
    unsigned long long m = 0ull;
    if (rk[i] >= 0 && (rk[i] == 0 || rS[i] > thr)) {
      const int slot = rslot[i];
      const unsigned w = (slot >> 1) == 0 ? rec[i].x : ((slot >> 1) == 1 ? rec[i].y : ((slot >> 1) == 2 ? rec[i].z : rec[i].w));
      const float bnd = __uint_as_float(((slot & 1) ? (w >> 16) : (w & 0xffffu)) << 16);
      m = __ballot(bnd > pthr);
    }
    any_row = any_row || m != 0ull;
    if (lane == 0) {
      unsigned *ri_ = rowinfo[wave][i];
      ri_[0] = (unsigned)m; ri_[1] = (unsigned)(m >> 32);
      ri_[2] = (unsigned)rk[i]; ri_[3] = (unsigned)rchan[i]; ri_[4] = __float_as_uint(pthr);
    }
  }
  if (!any_row) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // rowinfo: written by lane 0, read by the wave below
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float tpa[12 + 1], tpb[6 + 1];
  f32x2 tpsd[6 + 1];  // 4x: halved (sum, difference) coefficients of the mirrored phase pair
#pragma unroll
  for (int i = 0; i < (TP == 4 ? 6 : 0); ++i) tpsd[i] = (f32x2){F0->tp[18 + 2 * i], F0->tp[19 + 2 * i]};
#pragma unroll
  for (int i = 0; i < (TP == 2 ? 12 : 0); ++i) tpa[i] = F0->tp[i];
#pragma unroll
  for (int i = 0; i < (TP == 4 ? 6 : 0); ++i) tpb[i] = F0->tp[12 + i];
  (void)tpa; (void)tpb; (void)tpsd;
  const int dense_min = F0->tp_dense_min;
  const float pa2 = F0->tp[30], pb2 = F0->tp[31], pa1 = F0->tp[32], pb1 = F0->tp[33];
  float *const buf = stage[wave];
#pragma nounroll
  for (int ri = 0; ri < LGD_TP_RPW; ++ri) {
  const unsigned *ri_ = rowinfo[wave][ri];
  const unsigned long long mask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)ri_[0]) |
                                  ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)ri_[1]) << 32);
  if (mask == 0ull) continue;
  const int k = __builtin_amdgcn_readfirstlane((int)ri_[2]);
  const int chan = __builtin_amdgcn_readfirstlane((int)ri_[3]);
  const float pthr = __int_as_float(__builtin_amdgcn_readfirstlane((int)ri_[4])) * (1.f / IO::peak_scale);
  const long long tb = sg.f0 + (long long)k * tile_f;
  const pelem_ptr pcm = (pelem_ptr)sg.pcm + chan;
  const int n_chunks = __popcll(mask);
  // the interpolator over one window: wv[i] = x[n0 - HX + i] -> m_[u] = max over the non-trivial phases of
  // |y(n0 + u)|, u < U
  auto fir = [&](const float (&wv)[U + HX], float (&m_)[U]) {
    if constexpr (TP == 4) {
      f32x2 sd_[U];
      float o2_[U];
#pragma unroll
      for (int u_ = 0; u_ < U; ++u_) { sd_[u_] = (f32x2){0.f, 0.f}; o2_[u_] = 0.f; }
#pragma unroll
      for (int k_ = 0; k_ < 6; ++k_) {
        const f32x2 csd_ = tpsd[k_];
        const float c2_ = tpb[k_];
#pragma unroll
        for (int u_ = 0; u_ < U; ++u_) {
          const float xa_ = wv[HX + u_ - k_], xb_ = wv[u_ + k_];
          const f32x2 ab_ = (f32x2){xa_ + xb_, xa_ - xb_};
          sd_[u_] = __builtin_elementwise_fma(csd_, ab_, sd_[u_]);
          o2_[u_] = fmaf(c2_, ab_.x, o2_[u_]);
        }
      }
#pragma unroll
      for (int u_ = 0; u_ < U; ++u_) m_[u_] = fmaxf(fabsf(o2_[u_]), fabsf(sd_[u_].x) + fabsf(sd_[u_].y));
    } else {
      float o1_[U];
#pragma unroll
      for (int u_ = 0; u_ < U; ++u_) o1_[u_] = 0.f;
#pragma unroll
      for (int k_ = 0; k_ < 12; ++k_) {
        const float c1_ = tpa[k_];
#pragma unroll
        for (int u_ = 0; u_ < U; ++u_) o1_[u_] = fmaf(c1_, wv[HX + u_ - k_] + wv[u_ + k_], o1_[u_]);
      }
#pragma unroll
      for (int u_ = 0; u_ < U; ++u_) m_[u_] = fabsf(o1_[u_]);
    }
  };
  float pk_t = 0.f;

  if (n_chunks >= dense_min) {
    // ---------------------------------------------------------------- dense row
    // slabs of 64 x LP contiguous frames from the start of the row (the last one may reach past its end, into
    // the next tile's frames: true outputs too); slab s covers frames [s 64 LP, (s + 1) 64 LP) of the row,
    // i.e. chunks floor(s 64 LP / C) .. floor(((s + 1) 64 LP - 1) / C), and is walked if one of them is flagged
    const int n_slabs = (C + LP - 1) / LP;
    auto slab_wanted = [&](const int s) -> bool {
      const unsigned a = (unsigned)(s * (LGD_WAVE * LP)), b = a + (unsigned)(LGD_WAVE * LP - 1);
      const unsigned c0 = (a * sg.magic_c) >> 20;
      unsigned c1 = (b * sg.magic_c) >> 20;
      c1 = c1 > 63u ? 63u : c1;
      const unsigned long long span = (c1 - c0 == 63u) ? ~0ull : (((1ull << (c1 - c0 + 1u)) - 1ull) << c0);
      return (mask & span) != 0ull;
    };
    auto next_slab = [&](int s) -> int {
      while (s < n_slabs && !slab_wanted(s)) ++s;
      return s;
    };
    float pre[K::NPRE_D];
#pragma unroll
    for (int r = 0; r < K::NPRE_D; ++r) pre[r] = 0.f;  // (defined on every path: no value carried around the loop over the wave's rows)
    // element e = r * 64 + lane of slab s is frame tb + s * 64 LP - HX + e (the rounds past the slab's
    // 64 LP + HX elements are staged along, never read)
    auto fetch_slab = [&](const int s) {
      const long long f_first = tb + (long long)s * (LGD_WAVE * LP) - HX;
      if (f_first >= 0 && f_first + (long long)K::NPRE_D * LGD_WAVE <= n_frames) {  // (wave-uniform) interior
        pelem_ptr src = pcm + f_first * nch_tot;
        const unsigned lo = (unsigned)(lane * nch_tot);
#pragma unroll
        for (int r = 0; r < K::NPRE_D; ++r) {
          pelem_ptr src_r = src + (long long)(r * LGD_WAVE) * nch_tot;
          asm volatile("" : "+s"(src_r));  // scalar base + 32-bit lane offset
          pre[r] = (float)src_r[lo];
        }
      } else {
#pragma unroll
        for (int r = 0; r < K::NPRE_D; ++r) {
          const long long f = f_first + r * LGD_WAVE + lane;
          const bool in = f >= 0 && f < n_frames;
          pre[r] = in ? (float)pcm[(in ? f : 0) * nch_tot] : 0.f;
        }
      }
    };
    int s = next_slab(0);
    if (s < n_slabs) fetch_slab(s);
    while (s < n_slabs) {
      // (every lane is done reading the previous slab: the wave runs in lockstep and the fences
      // order its LDS traffic)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < K::NPRE_D; ++r) buf[r * LGD_WAVE + lane] = pre[r];
      const int s_next = next_slab(s + 1);
      if (s_next < n_slabs) fetch_slab(s_next);  // in flight while this slab is walked
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // lane l: output frames fl + [0, LP), fl = tb + s * 64 LP + l * LP; window element 0 = frame fl - HX
      // = slab element l * LP (odd lane stride: conflict-free)
      const float *w0 = buf + lane * LP;
      const long long fl = tb + (long long)s * (LGD_WAVE * LP) + (long long)lane * LP;
      const long long rem_ll = n_frames - fl;  // outputs at or past the end of the track do not exist
      const int rem = rem_ll >= LP ? LP : (rem_ll < 0 ? 0 : (int)rem_ll);
      const bool whole = __all(rem == LP);
      float wv[U + HX];
#pragma unroll
      for (int i = 0; i < HX; ++i) wv[i] = w0[i];
#pragma unroll
      for (int st = 0; st < NS; ++st) {
#pragma unroll
        for (int u_ = 0; u_ < U; ++u_) wv[HX + u_] = w0[HX + st * U + u_];
        float m_[U];
        fir(wv, m_);
        if (whole) {
#pragma unroll
          for (int u_ = 0; u_ < U; ++u_) pk_t = fmaxf(pk_t, m_[u_]);
        } else {
#pragma unroll
          for (int u_ = 0; u_ < U; ++u_) pk_t = fmaxf(pk_t, (st * U + u_ < rem) ? m_[u_] : 0.f);
        }
#pragma unroll
        for (int i = 0; i < HX; ++i) wv[i] = wv[i + U];  // (fully unrolled: renamed, not moved)
      }
      s = s_next;
    }
  } else {
    // --------------------------------------------------------------- sparse row
    // The flagged chunks one after the other, LGD_TP_AHEAD of them in flight: lane j holds frame j (and
    // j + 64) of the chunk's span [chunk start - HX, + CH) -- contiguous frames, coalesced loads -- and puts
    // them into LDS, LGD_TP_GROUP chunks at a time; then a lane takes one (chunk, step) slot of the group:
    // U output frames from U + HX staged ones, evaluated if the adjacent-pair bound of that window (its own
    // M and S2 now) can still exceed the channel's peak.
    const bool interior = tb - HX >= 0 && tb + tile_f <= n_frames;  // (wave-uniform) the row's whole span
    const int n1 = CH - LGD_WAVE;  // frames of the span's second round (C + HX <= 100, C >= 25: may be <= 0)
    float q0[LGD_TP_AHEAD], q1[LGD_TP_AHEAD];
    auto fetch_chunk = [&](const int l, float &r0, float &r1) {
      const long long f = tb + (long long)l * C - HX + lane;  // (l: wave-uniform)
      if (interior) {
        const pelem_ptr src = pcm + (tb + (long long)l * C - HX) * nch_tot;
        const unsigned lo = (unsigned)(lane * nch_tot);
        r0 = (lane < CH) ? (float)src[lo] : 0.f;
        r1 = (lane < n1) ? (float)src[lo + (unsigned)(LGD_WAVE * nch_tot)] : 0.f;
      } else {
        const bool in0 = lane < CH && f >= 0 && f < n_frames, in1 = lane < n1 && f + LGD_WAVE >= 0 && f + LGD_WAVE < n_frames;
        r0 = in0 ? (float)pcm[(in0 ? f : 0) * nch_tot] : 0.f;
        r1 = in1 ? (float)pcm[(in1 ? f + LGD_WAVE : 0) * nch_tot] : 0.f;
      }
    };
    // evaluates the n flagged chunks staged in `buf` (ids in chunk_of)
    auto run_group = [&](const int n) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // lane slot v = (chunk c of the group, step st): the U output frames c's frames [st U, + U)
      const int S = n * NSTEP;
      for (int v0 = 0; v0 < S; v0 += LGD_WAVE) {
        const int v = v0 + lane;
        const bool act = v < S;
        const int vv = act ? v : 0;
        const int c = (int)(((unsigned)vv * sg.magic_ns) >> 20);
        const int st = vv - c * NSTEP;
        const float *w0 = buf + c * CH + st * U;  // frame (st U - HX) of the chunk
        float wv[U + HX];
#pragma unroll
        for (int i = 0; i < U + HX; ++i) wv[i] = w0[i];
        // adjacent-pair bound of this window: M over all of it, S2 over the pairs the two centre taps of
        // the U outputs meet
        float wm = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < U + HX; ++i) wm = fmaxf(wm, fabsf(wv[i]));
#pragma unroll
        for (int u_ = 0; u_ < U; ++u_) s2 = fmaxf(s2, fabsf(wv[PK + u_]) + fabsf(wv[PK + 1 + u_]));
        const float bound = fmaxf(fmaf(pa2, s2, pb2 * wm), fmaf(pa1, s2, pb1 * wm));
        const bool need = act && (bound > pthr);
        if (!__any(need)) continue;  // wave-uniform: nothing in this pass can exceed the final peak
        float m_[U];
        fir(wv, m_);
        // output frames f0 + u; those at or past the end of the track do not exist
        const long long f0 = tb + (long long)chunk_of[wave][c] * C + st * U;
        const long long rem = n_frames - f0;
        const int nv = !need ? 0 : (rem < 0 ? 0 : (rem > U ? U : (int)rem));
#pragma unroll
        for (int u_ = 0; u_ < U; ++u_) pk_t = fmaxf(pk_t, (u_ < nv) ? m_[u_] : 0.f);
      }
      // (every lane is done with the group before the next one is staged over it)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
    unsigned long long todo = mask, ahead = mask;
#pragma unroll
    for (int i = 0; i < LGD_TP_AHEAD; ++i) {
      q0[i] = q1[i] = 0.f;
      if (ahead) {
        fetch_chunk(__builtin_ctzll(ahead), q0[i], q1[i]);
        ahead &= ahead - 1ull;
      }
    }
    int n_acc = 0;
    while (todo) {
      const int l = __builtin_ctzll(todo);  // (scalar)
      todo &= todo - 1ull;
      const float x0 = q0[0], x1 = q1[0];
#pragma unroll
      for (int i = 0; i + 1 < LGD_TP_AHEAD; ++i) { q0[i] = q0[i + 1]; q1[i] = q1[i + 1]; }
      if (ahead) {
        fetch_chunk(__builtin_ctzll(ahead), q0[LGD_TP_AHEAD - 1], q1[LGD_TP_AHEAD - 1]);
        ahead &= ahead - 1ull;
      }
      buf[n_acc * CH + lane] = x0;  // (lanes >= CH of the first round spill into the next slot: overwritten or unused)
      if (lane < n1) buf[n_acc * CH + LGD_WAVE + lane] = x1;
      if (lane == 0) chunk_of[wave][n_acc] = (unsigned char)l;
      if (++n_acc == LGD_TP_GROUP) {
        run_group(n_acc);
        n_acc = 0;
      }
    }
    if (n_acc) run_group(n_acc);
  }
  const float t = wave_max_f32_uniform(pk_t) * IO::peak_scale;
  if (t > 0.f && lane == 0)
    (void)__hip_atomic_fetch_max((unsigned LGD_GLOBAL *)sg.peak_out + nch_tot + chan,
                                 (unsigned)__float_as_int(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }  // rows of this wave
}

// ------------------------------------------------------- launch wrappers ---
// LDS bytes one workgroup needs (also used by the host-side planner); mirrors
// LdsLayout / PLANE in the kernel.  generic != 0: the run-time-channel-count kernel.
extern "C" size_t lgd_scan_lds_bytes(int chunk, int nch, int tp, int generic) {
  const int halo = 12;
  (void)tp;
  const bool planar = !generic;
  if (planar) {
    const int pad = (chunk % 2 == 0) ? 1 : 0;
    const int plane = (halo + 4 + pad + LGD_WAVE * (chunk + pad) + 4 + 8 + 1) & ~1;
    return (size_t)nch * plane * sizeof(float);
  }
  // (+8 frames: the software-pipelined reads fetch up to one step past the tile)
  return ((size_t)(LGD_WAVE * chunk + halo + 8) * nch + 4) * sizeof(float);
}

// Wave placement.  The workgroup dispatcher spreads the waves of one- and three-wave workgroups over
// a CU's four SIMDs evenly only if the kernel before left it in the right state: measured with
// HW_ID dumps (tools/hwid_probe.py; 2000 one-wave workgroups, 8 per CU), the mono scan kernel gets
// 2 waves on every SIMD when the previous large kernel had one-wave workgroups too (the previous
// scan), but 3 waves on ~100 SIMDs and 1 on ~150 when it follows lgd_tp_kernel's four-wave
// workgroups -- this kernel is bound by fp64 issue per SIMD, so the kernel then takes as long as its
// three-wave SIMDs: 0.334 ms instead of 0.283 ms (three channels: 4 waves on some SIMDs, +7 %).  A
// grid of empty one-wave workgroups in front restores the even placement (1024, 2048 and 4096
// workgroups measured alike; two-wave workgroups do not); 1024 cost ~3 us.  Two-, four- and eight-wave
// workgroups are placed evenly whatever ran before; six-wave workgroups (5.1) are not, with or
// without this (two per CU land 4/4/2/2 on a quarter of the CUs).
__global__ void lgd_prime_kernel() {}

template <int C, int G, int TP, bool WIDE = false, bool STR = false, bool S16 = false>
static hipError_t launch_scan_t(const LgdSeg *segs, int n_seg, int nch,
                                hipStream_t s) {
  const size_t lds_bytes = lgd_scan_lds_bytes(C, nch, TP, G == 0);
  if constexpr (G == 1 || G == 3) hipLaunchKernelGGL(lgd_prime_kernel, dim3(1024), dim3(LGD_WAVE), 0, s);
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)lgd_scan_kernel<C, G, TP, WIDE, STR, S16>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((lgd_scan_kernel<C, G, TP, WIDE, STR, S16>), dim3(n_seg), dim3(LGD_WAVE * nch), lds_bytes, s,
                     segs, nch);
  return hipGetLastError();
}

// channel pairs (or single channels) of a wider interleaved stream
template <int C, bool S16>
static hipError_t launch_scan_strided(int nch, int tp, const LgdSeg *segs, int n_seg,
                                      hipStream_t s) {
  if (nch == 1) {
    if (tp) return launch_scan_t<C, 1, 4, false, true, S16>(segs, n_seg, nch, s);
    return launch_scan_t<C, 1, 0, false, true, S16>(segs, n_seg, nch, s);
  }
  if (nch == 3) {
    if constexpr (C <= 50) {  // (three planes: the chunk lengths the three-channel kernel is built for)
      if (tp) return launch_scan_t<C, 3, 4, false, true, S16>(segs, n_seg, nch, s);
      return launch_scan_t<C, 3, 0, false, true, S16>(segs, n_seg, nch, s);
    }
    return hipErrorInvalidValue;
  }
  if (nch == 4) {  // channel quads of a wider stream (7 channels: 0-3 | 3-6)
    if constexpr (C <= 50) {
      if (tp) return launch_scan_t<C, 4, 4, false, true, S16>(segs, n_seg, nch, s);
      return launch_scan_t<C, 4, 0, false, true, S16>(segs, n_seg, nch, s);
    }
    return hipErrorInvalidValue;
  }
  if (nch != 2) return hipErrorInvalidValue;
  if (tp) return launch_scan_t<C, 2, 4, false, true, S16>(segs, n_seg, nch, s);
  return launch_scan_t<C, 2, 0, false, true, S16>(segs, n_seg, nch, s);
}

// the generic kernel (any channel count, channel groups, any sub-block alignment)
// exists for the shortest chunk only
#define LGD_GENERIC_CHUNK 25
template <int TP, bool S16>
static hipError_t launch_scan_generic(int nch, const LgdSeg *segs, int n_seg,
                                      hipStream_t s) {
  if (nch <= 8) return launch_scan_t<LGD_GENERIC_CHUNK, 0, TP, false, false, S16>(segs, n_seg, nch, s);
  return launch_scan_t<LGD_GENERIC_CHUNK, 0, TP, true, false, S16>(segs, n_seg, nch, s);
}

// (tp 2 and 4 run the same scan kernel: it only records chunk bounds, with the factor's coefficients)
template <int C, bool S16>
static hipError_t launch_scan_c(int nch, int tp, const LgdSeg *segs, int n_seg,
                                hipStream_t s) {
  if (nch == 1) {
    if (tp) return launch_scan_t<C, 1, 4, false, false, S16>(segs, n_seg, nch, s);
    return launch_scan_t<C, 1, 0, false, false, S16>(segs, n_seg, nch, s);
  }
  if (nch > 2) {  // 3 .. 6 or 8 planes per workgroup (2.1, quad, 5.0, 5.1, 7.1): the short chunks only
    if constexpr (C <= 50) {
#define LGD_DISPATCH_G(g_)                                                              \
      if (nch == g_) {                                                                  \
        if (tp) return launch_scan_t<C, g_, 4, false, false, S16>(segs, n_seg, nch, s); \
        return launch_scan_t<C, g_, 0, false, false, S16>(segs, n_seg, nch, s);         \
      }
      LGD_DISPATCH_G(3) LGD_DISPATCH_G(4) LGD_DISPATCH_G(5) LGD_DISPATCH_G(6) LGD_DISPATCH_G(8)
#undef LGD_DISPATCH_G
    }
    return hipErrorInvalidValue;
  }
  if (tp) return launch_scan_t<C, 2, 4, false, false, S16>(segs, n_seg, nch, s);
  return launch_scan_t<C, 2, 0, false, false, S16>(segs, n_seg, nch, s);
}

// The kernel instance a chunk length runs on: U = frames per window step (the scan kernel's unroll of that
// chunk), NS = steps per lane of a dense slab, LP = U NS = 15 / 21 frames: odd (conflict-free LDS stride) and
// short -- a slab of 64 LP + HX floats is ~4 / 5.5 KB per wave, so the LDS leaves eight waves per SIMD to the
// latency-bound sparse rows (LP = 25 measured: 26.9 KB per workgroup = 5 per CU, sparse rows 1.6 x slower).
extern "C" int lgd_tp_instance(int chunk, int *u_out, int *ns_out) {
  const int u = lgd_unroll(chunk);
  if (u != 5 && u != 7) return -1;
  if (u_out) *u_out = u;
  if (ns_out) *ns_out = 3;
  return 0;
}

// One launch for every segment that runs the kernel instance (U, TP, NS), whatever its chunk length and
// channel count (LgdSeg carries them).  rows_max: most rows (tiles x channels) any of the segments has.
extern "C" hipError_t lgd_launch_tp(int u, int tp, int ns, int s16, const LgdSeg *segs, int n_seg, int rows_max,
                                    hipStream_t s) {
  if (n_seg <= 0 || rows_max <= 0 || !tp) return hipSuccess;
#ifndef LGD_TP_RPW_BIG
#define LGD_TP_RPW_BIG 4
#endif
  const int rpw = (long long)n_seg * rows_max >= 1000000ll ? LGD_TP_RPW_BIG : 1;
  const dim3 grid((unsigned)n_seg, (unsigned)((rows_max + LGD_TP_WAVES * rpw - 1) / (LGD_TP_WAVES * rpw)));
  const dim3 block(LGD_WAVE * LGD_TP_WAVES);
  if ((long long)rows_max >= (1ll << 26)) return hipErrorInvalidValue;  // (row / nch by umulhi: exact below 2^26)
#define LGD_TP_CASE(u_, tp_, ns_)                                                         \
  if (u == u_ && tp == tp_ && ns == ns_) {                                                \
    if (s16) {                                                                            \
      if (rpw != 1) hipLaunchKernelGGL((lgd_tp_kernel<u_, tp_, ns_, LGD_TP_RPW_BIG, true>), grid, block, 0, s, segs); \
      else hipLaunchKernelGGL((lgd_tp_kernel<u_, tp_, ns_, 1, true>), grid, block, 0, s, segs); \
    } else if (rpw != 1) hipLaunchKernelGGL((lgd_tp_kernel<u_, tp_, ns_, LGD_TP_RPW_BIG>), grid, block, 0, s, segs); \
    else hipLaunchKernelGGL((lgd_tp_kernel<u_, tp_, ns_, 1>), grid, block, 0, s, segs);   \
    return hipGetLastError();                                                             \
  }
  LGD_TP_CASE(5, 4, 3) LGD_TP_CASE(7, 4, 3) LGD_TP_CASE(5, 2, 3) LGD_TP_CASE(7, 2, 3)
#undef LGD_TP_CASE
  return hipErrorInvalidValue;
}

// the divisions lgd_tp_kernel does by multiplication: checked by the planner per group
extern "C" int lgd_tp_magics(int chunk, int nch_wg, int tp, unsigned *magic_nch, unsigned *magic_ns, unsigned *magic_c) {
  const int u = lgd_unroll(chunk), hx = tp == 2 ? 23 : 11;
  if (chunk + hx > LGD_TP_CHMAX || chunk % u) return -1;  // (a span is at most two rounds of 64 lanes)
  // frame / chunk == (frame * magic_c) >> 20 for every frame a dense slab can start or end on
  const unsigned mc = ((1u << 20) + (unsigned)chunk - 1u) / (unsigned)chunk;
  for (unsigned x = 0; x < (unsigned)(LGD_WAVE * (chunk + 64)); ++x)
    if (((x * mc) >> 20) != x / (unsigned)chunk) return -1;
  *magic_c = mc;
  const unsigned nstep = (unsigned)(chunk / u);
  // x / d == (x * magic) >> 20 over the range the kernel divides
  const unsigned mns = ((1u << 20) + nstep - 1u) / nstep;
  for (unsigned x = 0; x < LGD_TP_GROUP * nstep + LGD_WAVE; ++x)
    if (((x * mns) >> 20) != x / nstep) return -1;
  // row / nch == umulhi(row, magic_nch) for row * (magic_nch * nch - 2^32) < 2^32; the error
  // term is < nch <= 64, so every row below 2^26 divides exactly
  *magic_nch = nch_wg == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)nch_wg - 1) / (unsigned)nch_wg);  // 0: k = row
  *magic_ns = mns;
  return 0;
}

// chunk lengths compiled in; the host picks one that divides the rate's s100
// (40, 42, 48, 60 were tried in round 1 and dropped: their unroll factors spilled registers; 70 came back once the
// once-per-tile constants had left the SGPRs: no spills at 210 - 247 VGPRs, two waves per SIMD)
extern "C" const int lgd_chunk_table[] = {25, 35, 45, 49, 50, 63, 70, 75, 0};


// F: DEVICE pointer to the group's constants
// generic: 0 = planar kernel of `nch` channels, 1 = run-time-channel kernel, 2 = channel pair /
// single channel (nch = 2 / 1) of a wider interleaved stream
extern "C" hipError_t lgd_launch_scan(int chunk, int nch, int tp, int generic, int s16, const LgdSeg *segs,
                                      int n_seg, hipStream_t s) {
  if (n_seg <= 0) return hipSuccess;
#define LGD_BY_CHUNK(fn_, ...)                                                          \
  switch (chunk) {                                                                      \
    case 25: return s16 ? fn_<25, true>(__VA_ARGS__) : fn_<25, false>(__VA_ARGS__);     \
    case 35: return s16 ? fn_<35, true>(__VA_ARGS__) : fn_<35, false>(__VA_ARGS__);     \
    case 45: return s16 ? fn_<45, true>(__VA_ARGS__) : fn_<45, false>(__VA_ARGS__);     \
    case 49: return s16 ? fn_<49, true>(__VA_ARGS__) : fn_<49, false>(__VA_ARGS__);     \
    case 50: return s16 ? fn_<50, true>(__VA_ARGS__) : fn_<50, false>(__VA_ARGS__);     \
    case 63: return s16 ? fn_<63, true>(__VA_ARGS__) : fn_<63, false>(__VA_ARGS__);     \
    case 70: return s16 ? fn_<70, true>(__VA_ARGS__) : fn_<70, false>(__VA_ARGS__);     \
    case 75: return s16 ? fn_<75, true>(__VA_ARGS__) : fn_<75, false>(__VA_ARGS__);     \
    default: return hipErrorInvalidValue;                                               \
  }
  if (generic == 2) LGD_BY_CHUNK(launch_scan_strided, nch, tp, segs, n_seg, s)
  if (nch < 1 || nch > 16) return hipErrorInvalidValue;
  if (generic) {
    if (chunk != LGD_GENERIC_CHUNK) return hipErrorInvalidValue;
    if (tp) return s16 ? launch_scan_generic<4, true>(nch, segs, n_seg, s) : launch_scan_generic<4, false>(nch, segs, n_seg, s);
    return s16 ? launch_scan_generic<0, true>(nch, segs, n_seg, s) : launch_scan_generic<0, false>(nch, segs, n_seg, s);
  }
  if (nch > 8 || nch == 7) return hipErrorInvalidValue;
  LGD_BY_CHUNK(launch_scan_c, nch, tp, segs, n_seg, s)
#undef LGD_BY_CHUNK
}
