// micro-benchmark: does v_mfma_f64_4x4x4 / 16x16x4 issue beside fp64 VALU work on gfx950, and at what cost?
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f64_coissue mfma_f64_coissue.hip && ./mfma_f64_coissue
// MODE 0: R x v_fma_f64 per step only; 1: one MFMA per step only; 2: both (the MFMA first, R independent FMAs behind it)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int MODE, int R, int BIG>
__global__ void k(double *out, int iters) {
  double d[8];
  for (int i = 0; i < 8; ++i) d[i] = 1e-3 * (threadIdx.x + i);
  double acc[4] = {0, 0, 0, 0};
  f64x4 big[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  const double a = 1e-3 * threadIdx.x, b = 1.0 + 1e-3 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (MODE != 0) {
        if (BIG) big[s & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, big[s & 1], 0, 0, 0);
        else acc[s & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[s & 3], 0, 0, 0);
      }
      if (MODE != 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[r & 7]));
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += d[i];
  for (int i = 0; i < 4; ++i) s += acc[i] + big[0][i] + big[1][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int R, int BIG>
void run(int waves_per_simd) {
  const int iters = 2000, nblk = 256, nthr = 256 * waves_per_simd;
  double *out; hipMalloc(&out, 8 * nblk * nthr);
  hipLaunchKernelGGL((k<MODE, R, BIG>), dim3(nblk), dim3(nthr), 0, 0, out, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, R, BIG>), dim3(nblk), dim3(nthr), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double steps = (double)iters * 8 * waves_per_simd;  // per SIMD
  printf("%s mode %d R=%2d waves/SIMD=%d: %.3f ms, %.2f ns per step per SIMD\n", BIG ? "16x16x4" : "4x4x4  ", MODE, R, waves_per_simd, ms,
         ms * 1e6 / steps);
  hipFree(out);
}
int main() {
  for (int w : {1, 2}) {
    run<0, 4, 0>(w); run<0, 8, 0>(w); run<0, 12, 0>(w);
    run<1, 0, 0>(w); run<2, 4, 0>(w); run<2, 8, 0>(w); run<2, 12, 0>(w);
    run<1, 0, 1>(w); run<2, 4, 1>(w); run<2, 8, 1>(w); run<2, 12, 1>(w); run<0, 16, 0>(w); run<2, 16, 1>(w);
  }
  return 0;
}
