// Probe: what a CU can ingest through LDS-DMA when the per-row segment of a piece is 64 B (half a cache line: the BK=32
// GEMM tiles) vs 128 B (a full line: BK=64) vs 256 B.  256 workgroups (one per CU, 128 KB of LDS each) stream the operand
// panels of a 16 x 16 grid of 256x256 GEMM tiles (Y rows ty*256.., X rows tx*256.., K = 5120 bf16) exactly like gemm256
// does (XCD-contiguous ids, 4 y-tiles per group), with a bounded 64 KB in flight per CU and no MFMA work at all.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(uint32_t voff, const u4& rsrc, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ u4 mk_rsrc(const void* base) {
  const uint64_t b = (uint64_t)base;
  u4 r; r[0] = (uint32_t)b; r[1] = (uint32_t)(b >> 32) & 0xffffu; r[2] = 0xffffffffu; r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int NX = 8; int xcd = bid % NX, idx = bid / NX; int q = nwg / NX, r = nwg % NX;
  return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// SEG = bytes per row per piece-row (64 / 128 / 256).  Per step the workgroup fetches 512 rows x SEG bytes
// (SEG*512/1024 pieces of 1 KB, split over 4 waves), into a ring holding 128 KB; in flight <= 64 KB.
template <int SEG>
__global__ __launch_bounds__(256) void probe(const char* Y, const char* X, int64_t pitch, int K2, uint64_t* out, int grid_x) {
  __shared__ __attribute__((aligned(16))) char lds[131072];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int GROUP = 4, per_group = GROUP * grid_x;
  const int gidx = wg / per_group, in_g = wg - gidx * per_group;
  const int ty = gidx * GROUP + in_g % GROUP, tx = in_g / GROUP;
  constexpr int LPR = SEG / 16;              // lanes per row
  constexpr int RPP = 64 / LPR;              // rows per piece
  constexpr int PIECES = 512 / RPP / 4;      // pieces per wave per step (4 waves)
  constexpr int STEP_BYTES = 512 * SEG;      // 32 / 64 / 128 KB
  constexpr int NSLOT = 131072 / STEP_BYTES; // ring slots
  uint32_t ofs[PIECES];
#pragma unroll
  for (int p = 0; p < PIECES; ++p) {
    const int row = (p * 4 + wave) * RPP + lane / LPR;          // 0..511: first 256 = Y rows, rest X rows
    ofs[p] = (uint32_t)((row & 255) * pitch + (lane % LPR) * 16);
  }
  const char* yb = Y + (int64_t)ty * 256 * pitch;
  const char* xb = X + (int64_t)tx * 256 * pitch;
  const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int nsteps = K2 / SEG;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < nsteps; ++s) {
    const int slot = s % NSLOT;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const bool isx = (p * 4 + wave) * RPP >= 256;
      asm volatile("s_waitcnt vmcnt(15)" ::: "memory");   // at most 16 pieces (16 KB) per wave = 64 KB per CU in flight
      dma16(ofs[p], mk_rsrc((isx ? xb : yb) + (int64_t)s * SEG), l0 + slot * STEP_BYTES + ((p * 4 + wave) * 64 + 0) * 16);
      if ((p & 7) == 7) __builtin_amdgcn_s_barrier();       // waves in loose lock-step, as the GEMM's per-k-tile barrier keeps them
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (lds[tid] == 77 && t1 == 0) out[0] = 1;
}

template <int SEG>
void run(const char* Y, const char* X, int64_t pitch, int K2, uint64_t* dbuf) {
  uint64_t h[256];
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<SEG>, dim3(256), dim3(256), 0, 0, Y, X, pitch, K2, dbuf, 16);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; uint64_t mx = 0;
    for (int i = 0; i < 256; ++i) { avg += h[i]; if (h[i] > mx) mx = h[i]; }
    avg /= 256;
    const double bytes = 512.0 * K2;   // per workgroup
    if (rep) printf("line_probe SEG=%3d B: %.3f ms  %.2f TB/s to the CUs, %.1f B/clk/CU (avg wg cycles %.0f, max %llu)\n", SEG, ms,
                    bytes * 256 / ms / 1e9, bytes / avg, avg, (unsigned long long)mx);
  }
}
int main() {
  const int K = 5120; const int64_t pitch = K * 2; const int rows = 4096;
  char *Y, *X; uint64_t* dbuf;
  hipMalloc(&Y, rows * pitch); hipMalloc(&X, rows * pitch); hipMalloc(&dbuf, 256 * 8);
  hipMemset(Y, 1, rows * pitch); hipMemset(X, 2, rows * pitch);
  run<64>(Y, X, pitch, K * 2, dbuf);
  run<128>(Y, X, pitch, K * 2, dbuf);
  run<256>(Y, X, pitch, K * 2, dbuf);
  run<64>(Y, X, pitch, K * 2, dbuf);
  return 0;
}
