// lds_valu_overlap_probe.hip -- do FP64 VALU work and LDS traffic of DIFFERENT wavefronts of a SIMD overlap on gfx950?
//
// The fused feature kernel (kernels_stft2.hip stft_feat_kernel) runs 16 wavefronts per CU whose frames alternate butterfly
// passes (FP64 VALU) with LDS exchanges; switched on one at a time the parts ADD UP to the kernel's time (round 3), PMC shows
// VALU ~46 % busy + LDS array ~42 % active, and none of round 4's rearrangements (fewer / more wavefronts, two frames per
// wavefront, priorities, LDS bytes moved to the L1) changed it.  This probe takes the kernel apart into its two ingredients:
//   A  a wavefront that only issues dependent-free FP64 FMAs                      (NV of them per round)
//   B  a wavefront that only exchanges through the LDS: ds_write_b64 x 8, ds_read_b64 x 8, wait  (NL rounds)
//   M  a wavefront that alternates: one round of A, one round of B (what a frame's passes do)
// and runs, 16 wavefronts per workgroup, one workgroup per CU:  all A | all B | half A + half B (per SIMD: 2 + 2) | all M.
// If "half A + half B" takes max(A, B) of the halves' work the pipes overlap across wavefronts and M's convoys are a
// scheduling effect; if it takes their sum they share an issue resource and the feature kernel sits at its floor.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_valu_overlap_probe.hip -o tools/bin/lds_valu_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NV = 256;  // FMAs per round of A
constexpr int NX = 4;    // exchanges per round of B (each: 8 writes + 8 reads of 8 bytes per lane)

__device__ __forceinline__ void round_valu(double (&a)[16], double x, double y)
{
#pragma unroll
  for (int r = 0; r < NV / 16; r++)
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = __builtin_fma(a[i], x, y);
}
__device__ __forceinline__ void round_lds(double* my, int lane, double (&v)[8])
{
#pragma unroll
  for (int e = 0; e < NX; e++)
  {
#pragma unroll
    for (int i = 0; i < 8; i++) my[(lane * 8 + i + (lane >> 1)) & 1023] = v[i];   // (lane-major with a skew: conflict-light)
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = my[(i * 64 + lane + 5 * e) & 1023];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// mode 0: all A; 1: all B; 2: wavefronts alternate A / B by (wave >> 2) & 1 (every SIMD gets two of each whichever way
// the hardware deals wavefronts to SIMDs: wave % 4 or wave / 4 -- wave ^ (wave >> 2) covers both); 3: all M
template <int MODE>
__global__ __launch_bounds__(1024) void probe(double* out, long long* cyc, int rounds)
{
  __shared__ double lds[16][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* my = lds[wave];
  for (int i = lane; i < 1024; i += 64) my[i] = i * 0.5;
  __syncthreads();
  double a[16], v[8];
  for (int i = 0; i < 16; i++) a[i] = 1.0 + i + lane;
  for (int i = 0; i < 8; i++) v[i] = lane + i;
  const double x = 1.0000001, y = 1e-9;
  const bool isA = MODE == 0 || (MODE == 2 && (((wave ^ (wave >> 2)) & 1) == 0));
  const bool isB = MODE == 1 || (MODE == 2 && !isA);
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; r++)
  {
    if (MODE == 3) { round_valu(a, x, y); round_lds(my, lane, v); }
    else if (MODE == 4)
    {
      // M with the data dependencies a transform has: the FMAs consume what the exchange delivered, the exchange sends what
      // the FMAs produced
#pragma unroll
      for (int i = 0; i < 8; i++) { a[i] = __builtin_fma(a[i], x, v[i]); a[i + 8] = __builtin_fma(a[i + 8], x, v[i]); }
      round_valu(a, x, y);
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = a[i] + a[i + 8];
      round_lds(my, lane, v);
    }
    else if (isA) round_valu(a, x, y);
    else if (isB) round_lds(my, lane, v);
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
  for (int i = 0; i < 16; i++) s += a[i];
  for (int i = 0; i < 8; i++) s += v[i];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char* what, int rounds)
{
  double* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 16 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(1024), 0, 0, out, cyc, rounds);   // warm
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(1024), 0, 0, out, cyc, rounds);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(256 * 16);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  long long mn = h[0], mx = h[0]; double av = 0;
  for (auto c : h) { mn = c < mn ? c : mn; mx = c > mx ? c : mx; av += c; }
  av /= h.size();
  std::printf("%-46s %8.3f ms   cycles per round and wavefront: min %7.0f  mean %7.0f  max %7.0f\n", what, ms, (double) mn / rounds, av / rounds,
              (double) mx / rounds);
  hipFree(out); hipFree(cyc);
}

int main()
{
  const int rounds = 2000;
  std::printf("per round: A = %d FP64 FMAs per wavefront (%d cycles of issue at 4 per instruction); B = %d exchanges of 8 ds_write_b64 + 8 ds_read_b64\n",
              NV, NV * 4, NX);
  run<0>("all 16 wavefronts A (VALU only)", rounds);
  run<1>("all 16 wavefronts B (LDS only)", rounds);
  run<2>("8 wavefronts A + 8 wavefronts B (2 + 2 per SIMD)", rounds);
  run<3>("all 16 wavefronts M (A then B, alternating)", rounds);
  run<4>("all 16 wavefronts M with data dependencies", rounds);
  return 0;
}
