// micro-benchmark: true-peak FIR with v_pk_fma_f32 over sample pairs vs scalar v_fma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Coef { float c[36]; };
template <int MODE>
__global__ void k(float *out, Coef cf, int iters) {
  float w[18];
  for (int i = 0; i < 18; ++i) w[i] = (float)(threadIdx.x + i) * 1e-3f;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      float o[3][5];
      for (int p = 0; p < 3; ++p) for (int u = 0; u < 5; ++u) o[p][u] = 0.f;
#pragma unroll
      for (int t = 0; t < 12; ++t)
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const float xv = w[11 + u - t];
#pragma unroll
          for (int p = 0; p < 3; ++p) o[p][u] = __builtin_fmaf(cf.c[12 * p + t], xv, o[p][u]);
        }
      for (int p = 0; p < 3; ++p) for (int u = 0; u < 5; ++u) acc = fmaxf(acc, fabsf(o[p][u]));
    } else {
      // pairs (u, u+1) for u = 0, 2; sample 4 scalar.  even-aligned window copy we[], odd copy wo[i] = w[i+1]
      f32x2 o2[3][2]; float o1[3];
      for (int p = 0; p < 3; ++p) { o2[p][0] = (f32x2){0.f, 0.f}; o2[p][1] = (f32x2){0.f, 0.f}; o1[p] = 0.f; }
#pragma unroll
      for (int t = 0; t < 12; ++t) {
        const f32x2 x01 = {w[11 - t], w[12 - t]}, x23 = {w[13 - t], w[14 - t]};
        const float x4 = w[15 - t];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const float c = cf.c[12 * p + t];
          const f32x2 c2 = {c, c};
          o2[p][0] = __builtin_elementwise_fma(c2, x01, o2[p][0]);
          o2[p][1] = __builtin_elementwise_fma(c2, x23, o2[p][1]);
          o1[p] = __builtin_fmaf(c, x4, o1[p]);
        }
      }
      for (int p = 0; p < 3; ++p)
        acc = fmaxf(acc, fmaxf(fmaxf(fabsf(o2[p][0].x), fabsf(o2[p][0].y)), fmaxf(fmaxf(fabsf(o2[p][1].x), fabsf(o2[p][1].y)), fabsf(o1[p]))));
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) w[i] = w[i] * 0.999f + acc * 1e-9f;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE>
void run(const char *name, int waves_per_simd) {
  int iters = 4000, nblk = 256, nthr = 256 * waves_per_simd;
  float *out; (void)hipMalloc(&out, 4 * nblk * nthr);
  Coef cf; for (int i = 0; i < 36; ++i) cf.c[i] = 0.01f * (i + 1);
  hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(nthr), 0, 0, out, cf, iters);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(nthr), 0, 0, out, cf, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-10s waves/SIMD=%d: %.3f ms -> ns per 5-sample FIR block (180 FMA) per wave per SIMD = %.1f\n", name, waves_per_simd, ms,
         ms * 1e6 / ((double)iters * waves_per_simd));
  (void)hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) { run<0>("scalar", w); run<1>("packed", w); }
  return 0;
}
