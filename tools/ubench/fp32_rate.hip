// micro-benchmark: v_fma_f32 / v_pk_fma_f32 issue rate on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int CHAINS, int MODE>
__global__ void k(float *out, float a, float b, int iters) {
  float v[CHAINS]; f32x2 p[CHAINS];
  for (int i = 0; i < CHAINS; ++i) { v[i] = (float)(threadIdx.x + i); p[i] = (f32x2){v[i], v[i] + 1}; }
  const f32x2 a2 = {a, a}, b2 = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) {
        if (MODE == 0) v[i] = __builtin_fmaf(a, v[i], b);
        else p[i] = __builtin_elementwise_fma(a2, p[i], b2);
      }
  }
  float s = 0;
  for (int i = 0; i < CHAINS; ++i) s += v[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS, int MODE>
void run(const char *name, int waves_per_simd) {
  int iters = 4000, nblk = 256, nthr = 256 * waves_per_simd;
  float *out; (void)hipMalloc(&out, 4 * nblk * nthr);
  hipLaunchKernelGGL((k<CHAINS, MODE>), dim3(nblk), dim3(nthr), 0, 0, out, 0.999f, 0.001f, iters);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, MODE>), dim3(nblk), dim3(nthr), 0, 0, out, 0.999f, 0.001f, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double n = (double)iters * 16 * CHAINS * waves_per_simd;
  printf("%-10s chains=%2d waves/SIMD=%d: %.3f ms, ns per wave-instr per SIMD = %.3f\n", name, CHAINS, waves_per_simd, ms, ms * 1e6 / n);
  (void)hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) { run<1, 0>("fma_f32", w); run<4, 0>("fma_f32", w); run<16, 0>("fma_f32", w); }
  for (int w : {1, 2, 4}) { run<1, 1>("pk_fma_f32", w); run<4, 1>("pk_fma_f32", w); run<16, 1>("pk_fma_f32", w); }
  return 0;
}
