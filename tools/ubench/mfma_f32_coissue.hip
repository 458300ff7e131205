// micro-benchmark: does v_mfma_f32_16x16x4_f32 execute beside fp32 VALU work (v_fma_f32) on gfx950?
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f32_coissue mfma_f32_coissue.hip && ./mfma_f32_coissue
// MODE 0: R x v_fma_f32 per step only; 1: one MFMA per step only; 2: both (the MFMA first, R independent FMAs behind it)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int R>
__global__ void k(float *out, int iters) {
  float d[8];
  for (int i = 0; i < 8; ++i) d[i] = 1e-3f * (threadIdx.x + i);
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const float a = 1e-3f * threadIdx.x, b = 1.0f + 1e-3f * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (MODE != 0) acc[s & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[s & 3], 0, 0, 0);
      if (MODE != 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d[r & 7]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += d[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int R>
void run(int waves_per_simd) {
  const int iters = 2000, nblk = 256, nthr = 256 * waves_per_simd;
  float *out; (void)hipMalloc(&out, 4 * nblk * nthr);
  hipLaunchKernelGGL((k<MODE, R>), dim3(nblk), dim3(nthr), 0, 0, out, iters);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, R>), dim3(nblk), dim3(nthr), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double steps = (double)iters * 8 * waves_per_simd;  // per SIMD
  printf("16x16x4 f32 mode %d R=%2d waves/SIMD=%d: %.3f ms, %.2f ns per step per SIMD\n", MODE, R, waves_per_simd, ms, ms * 1e6 / steps);
  (void)hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0, 4>(w); run<0, 6>(w); run<0, 8>(w); run<0, 12>(w);
    run<1, 0>(w); run<2, 4>(w); run<2, 6>(w); run<2, 8>(w); run<2, 12>(w);
  }
  return 0;
}
