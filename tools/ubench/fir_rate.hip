// micro-benchmark: the true-peak FIR inner pattern (fp32 FMA, SGPR coefficient x VGPR sample)
#include <hip/hip_runtime.h>
#include <cstdio>
struct Coef { float c[36]; };
template <int MODE>
__global__ void k(float *out, Coef cf, int iters) {
  float w[16], o[15];
  for (int i = 0; i < 16; ++i) w[i] = (float)(threadIdx.x + i) * 1e-3f;
  for (int i = 0; i < 15; ++i) o[i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 12; ++t) {
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const float xv = w[11 + u - t];
        if (MODE == 0) {            // SGPR coefficients
          o[u] = __builtin_fmaf(cf.c[t], xv, o[u]);
          o[5 + u] = __builtin_fmaf(cf.c[12 + t], xv, o[5 + u]);
          o[10 + u] = __builtin_fmaf(cf.c[24 + t], xv, o[10 + u]);
        } else {                    // VGPR x VGPR
          o[u] = __builtin_fmaf(w[t], xv, o[u]);
          o[5 + u] = __builtin_fmaf(w[(t + 1) & 15], xv, o[5 + u]);
          o[10 + u] = __builtin_fmaf(w[(t + 2) & 15], xv, o[10 + u]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = w[i] * 0.999f + o[i % 15] * 1e-6f;
  }
  float s = 0;
  for (int i = 0; i < 15; ++i) s += o[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
  double n = (double)iters * (180 + 32) * waves_per_simd;
  printf("%-10s waves/SIMD=%d: %.3f ms, ns per wave-instr per SIMD = %.3f\n", name, waves_per_simd, ms, ms * 1e6 / n);
  (void)hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) { run<0>("fir_sgpr", w); run<1>("fir_vgpr", w); }
  return 0;
}
