// micro-benchmark for the stage-1 slot pattern of sc_spec2_filter_kernel (two waves per SIMD, 512 threads): how fast does a
// dependent chain of v_mfma_f32_32x32x16_f16 run when the A fragments come from an LDS ring and the reads sit
//   G = 1: between every two MFMAs        [wait, mfma, read]
//   G = 2: behind pairs of MFMAs          [wait, mfma, mfma, read, read]
//   G = 4: behind groups of four
// with WAVES = 1 or 2 waves per SIMD, CHAINS = 1 or 2 accumulators alternating, and V extra VALU (v_cvt_pk) per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/mfma_group.hip -o /tmp/mfma_group && /tmp/mfma_group
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned frag4 __attribute__((ext_vector_type(4)));
constexpr int NB = 8, NSLOT = 48;

template <int WAVES, int G, int CHAINS, int V, int DEPTH>
__global__ __launch_bounds__(256 * WAVES, WAVES) void k(float *out, int iters, long long *cyc) {
  extern __shared__ char smem[];
  for (int i = threadIdx.x; i < 16384; i += 256 * WAVES) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u;
  __syncthreads();
  half8 B[NB];
  for (int s = 0; s < NB; s++)
    for (int i = 0; i < 8; i++) B[s][i] = (_Float16)(0.001f * (s + i + (threadIdx.x & 7)));
#pragma unroll
  for (int s = 0; s < NB; s++) asm volatile("" : "+v"(B[s]));
  floatx16 acc[CHAINS];
  for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
  const unsigned addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  unsigned junk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float seed = threadIdx.x * 0.5f;
  frag4 ring[DEPTH];
#pragma unroll
  for (int t = 0; t < DEPTH; t++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[t]) : "v"(addr), "n"(1024 * (t % 4)));
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < NSLOT / G; g++) {
      // all DEPTH reads in flight; the group needs the oldest G
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(DEPTH - G));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < G; u++) {
        const int t = g * G + u;
        const half8 af = __builtin_bit_cast(half8, ring[t % DEPTH]);
        acc[t % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, B[t % NB], acc[t % CHAINS], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < G; u++) {
        const int t = g * G + u;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[t % DEPTH]) : "v"(addr), "n"(1024 * (t % 4)));
      }
#pragma unroll
      for (int v = 0; v < V * G; v++) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(junk[v % 8]) : "v"(seed), "v"(seed));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)");
  float s = 0;
  for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 16; i++) s += acc[c][i];
  for (int t = 0; t < DEPTH; t++) s += (float)ring[t][0];
  for (int v = 0; v < 8; v++) s += (float)junk[v];
  out[blockIdx.x * 256 * WAVES + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int WAVES, int G, int CHAINS, int V, int DEPTH>
void run(const char *name) {
  float *out; long long *cyc;
  hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  auto kern = k<WAVES, G, CHAINS, V, DEPTH>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256 * WAVES), 65536, 0, out, 50, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256 * WAVES), 65536, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * NSLOT;   // MFMAs per wave
  printf("%-58s %6.1f shader cycles per MFMA of one wave = %5.1f per MFMA on the SIMD   %4.0f TFLOP/s\n", name, (double)c / n, (double)c / n / WAVES,
         256.0 * 4 * WAVES * n * 32768.0 / (ms * 1e-3) / 1e12);
}
int main() {
  run<1, 1, 1, 0, 6>("1 wave/SIMD  G=1  1 chain");
  run<1, 2, 1, 0, 6>("1 wave/SIMD  G=2  1 chain");
  run<1, 3, 1, 0, 6>("1 wave/SIMD  G=3  1 chain");
  run<1, 1, 2, 0, 6>("1 wave/SIMD  G=1  2 chains");
  run<1, 1, 1, 2, 6>("1 wave/SIMD  G=1  1 chain  +2 VALU per MFMA");
  run<1, 2, 1, 2, 6>("1 wave/SIMD  G=2  1 chain  +2 VALU per MFMA");
  run<2, 1, 1, 0, 6>("2 waves/SIMD G=1  1 chain");
  run<2, 2, 1, 0, 6>("2 waves/SIMD G=2  1 chain");
  run<2, 3, 1, 0, 6>("2 waves/SIMD G=3  1 chain");
  run<2, 1, 2, 0, 6>("2 waves/SIMD G=1  2 chains");
  run<2, 1, 1, 2, 6>("2 waves/SIMD G=1  1 chain  +2 VALU per MFMA");
  run<2, 2, 1, 2, 6>("2 waves/SIMD G=2  1 chain  +2 VALU per MFMA");
  run<2, 1, 1, 4, 6>("2 waves/SIMD G=1  1 chain  +4 VALU per MFMA");
  run<2, 2, 1, 4, 6>("2 waves/SIMD G=2  1 chain  +4 VALU per MFMA");
  run<2, 1, 1, 8, 6>("2 waves/SIMD G=1  1 chain  +8 VALU per MFMA");
  return 0;
}
