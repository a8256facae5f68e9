// Micro-benchmark (measurement tool, not product code; WRITTEN IN ROUND 3, NOT YET RUN): the floor of a RESIDENT pick kernel.
// One wavefront stays on the GPU and polls a pinned host word (the doorbell); the host writes a sequence number, the wavefront answers
// by writing it into a second pinned word, the host polls that.  Round trip = what launch + completion (13 of the 20 us of a
// one-request batch today) would shrink to.  Variants: payload of N bytes read by the wavefront before it answers (rows of a small batch).
// SAFETY: the kernel leaves by itself -- after `rounds` answers or after `max_spins` polls without a new doorbell value -- so a host
// that dies cannot leave a spinning kernel behind; run it under `timeout 20`.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void resident(volatile uint32_t* bell, volatile uint32_t* answer, const uint64_t* payload, uint32_t payload_words, uint32_t rounds,
                         unsigned long long max_spins) {
  uint32_t seen = 0;
  unsigned long long spins = 0;
  for (uint32_t done = 0; done < rounds;) {
    const uint32_t v = __hip_atomic_load((uint32_t*)bell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v == seen) { if (++spins > max_spins) return; __builtin_amdgcn_s_sleep(1); continue; }
    spins = 0; seen = v;
    uint64_t acc = 0;
    for (uint32_t i = threadIdx.x; i < payload_words; i += blockDim.x) acc += __hip_atomic_load(payload + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    acc = __shfl_xor((unsigned long long)acc, 1);           // (keep the loads)
    if (threadIdx.x == 0) __hip_atomic_store((uint32_t*)answer, v + (uint32_t)(acc & 0ull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    ++done;
  }
}

int main(int argc, char** argv) {
  const uint32_t rounds = argc > 1 ? (uint32_t)atoi(argv[1]) : 20000u;
  uint32_t *bell, *answer; uint64_t* payload;
  CK(hipHostMalloc((void**)&bell, 64, hipHostMallocDefault)); CK(hipHostMalloc((void**)&answer, 64, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&payload, 1 << 20, hipHostMallocDefault));
  for (size_t i = 0; i < (1 << 20) / 8; ++i) payload[i] = i;
  for (uint32_t words : {0u, 33u, 33u * 16u, 33u * 128u}) {      // nothing / 1 / 16 / 128 request rows of 264 bytes
    *bell = 0; *answer = 0;
    hipLaunchKernelGGL(resident, dim3(1), dim3(64), 0, 0, bell, answer, payload, words, rounds, 400000000ull);
    CK(hipGetLastError());
    double worst = 0, sum = 0;
    for (uint32_t r = 1; r <= rounds; ++r) {
      const auto t0 = std::chrono::steady_clock::now();
      __atomic_store_n(bell, r, __ATOMIC_RELEASE);
      while (__atomic_load_n(answer, __ATOMIC_ACQUIRE) != r) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) { printf("no answer to doorbell %u\n", r); return 2; }
      }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      sum += us; if (us > worst) worst = us;
    }
    CK(hipDeviceSynchronize());
    printf("resident wavefront, payload %6u bytes read per doorbell: round trip avg %6.2f us, worst %7.2f us over %u rounds\n", words * 8u, sum / rounds, worst, rounds);
  }
  return 0;
}
