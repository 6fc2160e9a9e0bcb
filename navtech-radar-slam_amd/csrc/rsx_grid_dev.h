// rsx_grid_dev.h -- what a persistent multi-workgroup kernel on gfx950 needs to hand data between its workgroups:
// a grid barrier and loads / stores that are coherent across the eight XCDs.
//
// MI355X has one L2 per XCD and they are not coherent with each other for ordinary loads and stores; a device-scope fence
// (writeback + invalidate of the whole L2) per hand-over is what made round 3's "last block done" tickets cost 40 us each.
// The forms MI355X_MICROARCH lists as valid instead, and sc_q1.hip uses since round 5:
//   * data that another workgroup will read is written with RELAXED AGENT-SCOPE atomic stores (a plain global_store with
//     sc1: write-through to memory) and read with relaxed agent-scope atomic loads (sc1: not served from the reader's L2);
//   * before a workgroup announces itself, every wave drains its stores (s_waitcnt vmcnt(0)) and the workgroup meets at
//     its own barrier;
//   * the announcement is a relaxed agent-scope fetch_add, the wait a poll of an agent-scope load.
// Data written by an EARLIER launch is read with ordinary loads (the kernel boundary made it visible).
//
// grid_sync: arrival in two levels -- workgroup b adds to group counter b % 8, the last of a group adds to the top
// counter (255 atomics on ONE word are 3.3 us at 13 ns each, 32 + 8 are 0.5) -- and every workgroup's first thread polls
// the top counter.  The counters only grow during a launch (target = members x round); grid_exit lets the LAST workgroup
// that leaves the kernel zero them, so a counter block is reusable by the next launch, whatever its grid size, without a
// memset in between.  All workgroups of the launch must be resident at the same time (grid <= what the device holds).
// If they are not -- a device with masked CUs, another persistent kernel holding a part of the chip -- a barrier would wait
// for ever and take the GPU with it.  So the poll has a watchdog: after PATIENCE the polling workgroup raises a flag, every
// workgroup that sees the flag gives up (sync returns false; workgroups that only start later see it at their first
// barrier), the kernel reports the failure through its own result and the host turns it into an error status.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rsx {
namespace grid {

constexpr int LINE = 32;       // unsigned per counter: each on its own 128-byte line
constexpr int GROUPS = 8;
constexpr int WORDS = (GROUPS + 3) * LINE;  // 8 group counters, top, departures, the give-up flag
#ifndef RSX_GRID_PATIENCE
#define RSX_GRID_PATIENCE 500000000ull  // 5 s of the 100 MHz wall clock: no barrier of these kernels waits that long
#endif
constexpr unsigned long long PATIENCE = RSX_GRID_PATIENCE;
constexpr size_t BYTES = (size_t)WORDS * 4;

template <typename T>
__device__ __forceinline__ T ld(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void st(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ldf(const float *p) { return __uint_as_float(ld(reinterpret_cast<const unsigned *>(p))); }
__device__ __forceinline__ void stf(float *p, float v) { st(reinterpret_cast<unsigned *>(p), __float_as_uint(v)); }
__device__ __forceinline__ double ldd(const double *p) {
  return __longlong_as_double((long long)ld(reinterpret_cast<const unsigned long long *>(p)));
}
__device__ __forceinline__ void std_(double *p, double v) {
  st(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v));
}

// the state a workgroup carries through a launch: which workgroup of how many it is, the round it is in
struct Member {
  unsigned *bar;   // the counter block (WORDS unsigned, zero at launch)
  unsigned b, G;   // this workgroup, workgroups that take part
  unsigned round;  // barriers passed
  bool dead;       // the launch gave up (see the watchdog): no further barrier may be entered
};

__device__ __forceinline__ unsigned group_size(unsigned G, unsigned g) { return (G - g + GROUPS - 1) / GROUPS; }

// every thread of every participating workgroup calls this the same number of times; false: the launch gave up -- the caller
// leaves (through exit) without touching another barrier
__device__ __forceinline__ bool sync(Member &m) {
  __shared__ int s_dead;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores and atomics have been performed
  __syncthreads();
  m.round++;
  if (threadIdx.x == 0) {
    int dead = 0;
    if (m.G > 1) {
      const unsigned g = m.b % GROUPS, ngroups = m.G < (unsigned)GROUPS ? m.G : (unsigned)GROUPS;
      const unsigned old = __hip_atomic_fetch_add(m.bar + g * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1u == group_size(m.G, g) * m.round)
        __hip_atomic_fetch_add(m.bar + GROUPS * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = ngroups * m.round;
      const unsigned long long t0 = wall_clock64();
      while (ld(m.bar + GROUPS * LINE) < target) {
        if (ld(m.bar + (GROUPS + 2) * LINE) != 0u) {
          dead = 1;
          break;
        }
        if (wall_clock64() - t0 > PATIENCE) {
          st(m.bar + (GROUPS + 2) * LINE, 1u);
          dead = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    s_dead = dead;
  }
  __syncthreads();
  m.dead = s_dead != 0;
  return !m.dead;
}

// once, at the end of the kernel (after the last sync): the last workgroup to leave zeroes the counters
__device__ __forceinline__ void exit(Member &m) {
  if (threadIdx.x == 0 && m.G > 1) {
    const unsigned old = __hip_atomic_fetch_add(m.bar + (GROUPS + 1) * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == m.G)
      for (int c = 0; c < GROUPS + 3; c++) st(m.bar + c * LINE, 0u);
  }
}

}  // namespace grid
}  // namespace rsx
