// Measurement aid (bench.py `roofline.sustained_mfma`): what the chip sustains on the matrix pipe ALONE.  One workgroup of four
// waves per CU (one per SIMD), 256 accumulator registers per lane, nothing but back-to-back v_mfma_f32_32x32x16_bf16 on random
// bf16 operands held in registers: no LDS, no HBM, no VALU in the loop.  On MI355X this does NOT run at the 2.5 PFLOP/s of the
// data sheet (2.4 GHz): under random data the chip settles at its power limit, 1.63-1.69 GHz = 1707-1773 TFLOP/s (0.68-0.71 of peak;
// tools/probes/mfma_power_probe.hip, profiles/r03_mfma_power_probe_run65.log).  bench.py runs it next to the timed region so
// that every roofline fraction of a run can be read against the ceiling the SAME box reaches in the SAME state.
#include "common.h"

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf8;

__device__ __forceinline__ uint32_t probe_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// 8 random bf16 of magnitude 0.25 .. 4 with random mantissas and signs (full operand toggling, like activations / weights)
__device__ __forceinline__ probe_bf8 probe_frag(uint32_t seed) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t h = probe_hash(seed * 4u + i);
    const uint32_t lo = ((h & 1u) << 15) | ((125u + ((h >> 1) & 3u)) << 7) | ((h >> 3) & 0x7fu);
    const uint32_t hi = (((h >> 10) & 1u) << 15) | ((125u + ((h >> 11) & 3u)) << 7) | ((h >> 13) & 0x7fu);
    w[i] = lo | (hi << 16);
  }
  uint4 v;
  v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
  return __builtin_bit_cast(probe_bf8, v);
}

__global__ __launch_bounds__(256) void mfma_probe_kernel(float* __restrict__ out, int iters) {
  probe_bf8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = probe_frag((blockIdx.x * 256u + threadIdx.x) * 8u + i);
    b[i] = probe_frag((blockIdx.x * 256u + threadIdx.x) * 8u + 4 + i);
  }
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int m = 0; m < 16; ++m) acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m >> 2], b[m & 3], acc[m >> 2][m & 3], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;  // keeps the accumulators alive; practically never true
}
// one lane spinning on the 100 MHz constant clock (s_memrealtime): occupies a stream for `ticks` x 10 ns and nothing else of the chip
__global__ void delay_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

// Measurement aid (bench.py `simulated_scaling`, the link model): keeps `stream` busy for `microseconds` without using the chip -- the
// transfer time of an exchange over xGMI (bytes / link rate) behind the device-to-device copy that stands in for it on one GPU.
extern "C" int wan_debug_delay(double microseconds, void* stream) {
  WAN_REQUIRE(microseconds >= 0.0 && microseconds <= 5e6, "wan_debug_delay: %g us outside [0, 5 s]", microseconds);
  if (microseconds == 0.0) return 0;
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(1), 0, as_stream(stream), (long long)(microseconds * 100.0));
  WAN_LAUNCH_CHECK();
  return 0;
}

// Enqueues `iters` x 64 MFMAs per wave on every CU (iters = 40,000 is ~50 ms); *flop_out = the FLOP the launch performs.
extern "C" int wan_mfma_sustained_probe(int iters, double* flop_out, void* stream) {
  WAN_REQUIRE(iters > 0 && flop_out, "wan_mfma_sustained_probe: bad args");
  static float* sink = nullptr;
  if (sink == nullptr) WAN_CHECK_HIP(hipMalloc((void**)&sink, 256));
  int dev = 0, cus = 0;
  WAN_CHECK_HIP(hipGetDevice(&dev));
  WAN_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)cus), dim3(256), 0, as_stream(stream), sink, iters);
  WAN_LAUNCH_CHECK();
  *flop_out = (double)cus * 4.0 * (double)iters * 64.0 * 32768.0;
  return 0;
}
