// micro-benchmark of one stage-1 slot of sc_spec_filter_kernel: s_waitcnt lgkmcnt(N) ; MFMA (A fragment
// from an LDS ring, B fragment register-resident) ; ds_read_b128 of the fragment DEPTH slots ahead.
// Variants: B fragments in AGPRs or VGPRs, 1 or 2 accumulator chains; build with and without
// `-mllvm -amdgpu-mfma-vgpr-form` (accumulators in VGPRs / AGPRs).
//   hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form] tools/ubench/mfma_slot.hip -o /tmp/mfma_slot && /tmp/mfma_slot
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned frag4 __attribute__((ext_vector_type(4)));
constexpr int DEPTH = 6, NB = 16;

template <bool B_AGPR, int CHAINS, bool READS, int NVALU = 0>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters, long long *cyc) {
  extern __shared__ char smem[];
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u;  // fp16 1.0
  __syncthreads();
  half8 B[NB];
  for (int s = 0; s < NB; s++)
    for (int i = 0; i < 8; i++) B[s][i] = (_Float16)(0.001f * (s + i + (threadIdx.x & 7)));
  if (B_AGPR) {
#pragma unroll
    for (int s = 0; s < NB; s++) asm volatile("" : "+a"(B[s]));
  } else {
#pragma unroll
    for (int s = 0; s < NB; s++) asm volatile("" : "+v"(B[s]));
  }
  floatx16 acc[CHAINS];
  for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
  const unsigned addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)smem) + (threadIdx.x & 63) * 16;
  unsigned junk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float seed = threadIdx.x * 0.5f;
  frag4 ring[DEPTH];
#pragma unroll
  for (int t = 0; t < DEPTH; t++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[t]) : "v"(addr), "n"(1024 * t));
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int t = 0; t < 48; t++) {
      if (READS) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(DEPTH - 1));
      __builtin_amdgcn_sched_barrier(0);
      const half8 af = __builtin_bit_cast(half8, ring[t % DEPTH]);
      acc[t % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, B[t % NB], acc[t % CHAINS], 0, 0, 0);
      if (READS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[t % DEPTH]) : "v"(addr), "n"(1024 * (t % 32)));
#pragma unroll
      for (int v = 0; v < NVALU; v++) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(junk[v]) : "v"(seed), "v"(seed));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)");
  float s = 0;
  for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 16; i++) s += acc[c][i];
  for (int t = 0; t < DEPTH; t++) s += (float)ring[t][0];
  for (int v = 0; v < 8; v++) s += (float)junk[v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <bool B_AGPR, int CHAINS, bool READS, int NVALU = 0>
void run(const char *name) {
  float *out; long long *cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<B_AGPR, CHAINS, READS, NVALU>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<B_AGPR, CHAINS, READS, NVALU>), dim3(256), dim3(256), 65536, 0, out, 50, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<B_AGPR, CHAINS, READS, NVALU>), dim3(256), dim3(256), 65536, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 48;
  printf("%-44s %.2f ns/slot  (%.1f ticks)  %.0f TFLOP/s\n", name, ms * 1e6 / n, (double)c / n, 256.0 * 4 * n * 32768.0 / (ms * 1e-3) / 1e12);
}
int main() {
  run<true, 1, true>("B in AGPR, 1 chain, LDS reads");
  run<false, 1, true>("B in VGPR, 1 chain, LDS reads");
  run<true, 2, true>("B in AGPR, 2 chains, LDS reads");
  run<true, 1, false>("B in AGPR, 1 chain, no reads");
  run<false, 1, false>("B in VGPR, 1 chain, no reads");
  run<true, 1, true, 1>("1 chain, reads, 1 VALU per slot");
  run<true, 1, true, 2>("1 chain, reads, 2 VALU per slot");
  run<true, 1, true, 4>("1 chain, reads, 4 VALU per slot");
  run<true, 2, true, 2>("2 chains, reads, 2 VALU per slot");
  run<true, 2, true, 4>("2 chains, reads, 4 VALU per slot");
  run<true, 2, true, 6>("2 chains, reads, 6 VALU per slot");
  run<true, 3, true, 4>("3 chains, reads, 4 VALU per slot");
  return 0;
}
