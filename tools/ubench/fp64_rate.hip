// micro-benchmark: v_fma_f64 issue rate and dependent latency on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CHAINS, typename T>
__global__ void k(T *out, unsigned long long *cyc, T a, T b, int iters) {
  T v[CHAINS];
  for (int i = 0; i < CHAINS; ++i) v[i] = (T)(threadIdx.x + i);
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) v[i] = __builtin_fma(a, v[i], b);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  T s = 0;
  for (int i = 0; i < CHAINS; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS, typename T>
void run(const char *name, int waves_per_simd) {
  int iters = 2000;
  int nblk = 256, nthr = 256 * waves_per_simd;  // 4 SIMDs * waves each, one block per CU
  T *out; unsigned long long *cyc;
  hipMalloc(&out, sizeof(T) * nblk * nthr); hipMalloc(&cyc, 8 * nblk);
  hipLaunchKernelGGL((k<CHAINS, T>), dim3(nblk), dim3(nthr), 0, 0, out, cyc, (T)0.999, (T)0.001, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, T>), dim3(nblk), dim3(nthr), 0, 0, out, cyc, (T)0.999, (T)0.001, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nblk); hipMemcpy(h.data(), cyc, 8 * nblk, hipMemcpyDeviceToHost);
  double n_instr_per_wave = (double)iters * 16 * CHAINS;
  double total_wave_instr_per_simd = n_instr_per_wave * waves_per_simd;
  // s_memtime counts at a fixed 100 MHz on some chips; report both views
  printf("%-8s chains=%2d waves/SIMD=%d: %.3f ms, wall ns per wave-instr per SIMD = %.2f (x2.1GHz = %.2f cyc), memtime ticks/instr/wave = %.3f\n",
         name, CHAINS, waves_per_simd, ms, ms * 1e6 / total_wave_instr_per_simd,
         ms * 1e6 / total_wave_instr_per_simd * 2.1, (double)h[0] / n_instr_per_wave);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<1, double>("f64", w); run<2, double>("f64", w); run<4, double>("f64", w); run<8, double>("f64", w);
  }
  for (int w : {1, 2, 4}) { run<1, float>("f32", w); run<4, float>("f32", w); run<8, float>("f32", w); }
  return 0;
}
