// micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 in a dependent accumulator chain vs 2 / 4
// independent chains, one wave per SIMD (what sc_filter_kernel runs).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters, long long *cyc) {
  half8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  floatx16 acc[CHAINS];
  for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16 / CHAINS; u++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 16; i++) s += acc[c][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS>
void run(const char *name) {
  float *out; long long *cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 1}) {
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, 100, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 16;
    printf("%s blocks=%3d: %.2f ns/MFMA/SIMD  (%.1f s_memtime ticks per MFMA)  %.0f TFLOP/s\n", name, blocks, ms * 1e6 / n,
           (double)c / n, blocks * 4 * n * 32768.0 / (ms * 1e-3) / 1e12);
  }
}
int main() { run<1>("1 chain "); run<2>("2 chains"); run<4>("4 chains"); return 0; }
