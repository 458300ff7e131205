// micro-benchmark: issue cost of single VALU instructions on gfx950 (independent instructions, 1 / 2 waves per SIMD)
//   hipcc -O3 --offload-arch=gfx950 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(double *out, int iters) {
  double d[8]; float f[8]; f32x2 p[8];
  for (int i = 0; i < 8; ++i) { d[i] = threadIdx.x + i; f[i] = (float)(threadIdx.x + i) * 1e-3f; p[i] = (f32x2){f[i], f[i] + 1.f}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));
        if (OP == 1) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
        if (OP == 2) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[i]));
        if (OP == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
        if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
        if (OP == 5) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
        if (OP == 6) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(f[i]));
        if (OP == 7) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[i]));
        if (OP == 8) asm volatile("v_mov_b32 %0, %0" : "+v"(f[i]));
      }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += d[i] + f[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, int waves_per_simd) {
  const int iters = 4000, nblk = 256, nthr = 256 * waves_per_simd;
  double *out; hipMalloc(&out, 8 * nblk * nthr);
  hipLaunchKernelGGL((k<OP>), dim3(nblk), dim3(nthr), 0, 0, out, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(nblk), dim3(nthr), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)iters * 64 * waves_per_simd;
  printf("%-14s waves/SIMD=%d: %.3f ms, %.2f ns per wave-instruction per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / per_simd);
  hipFree(out);
}
int main() {
  for (int w : {1, 2}) {
    run<0>("v_fma_f64", w); run<7>("v_mul_f64", w); run<2>("v_add_f64", w); run<1>("v_cvt_f64_f32", w);
    run<3>("v_fma_f32", w); run<6>("v_max3_f32", w); run<8>("v_mov_b32", w); run<4>("v_pk_fma_f32", w); run<5>("v_pk_add_f32", w);
  }
  return 0;
}
