// hbm_write_probe -- what the part sustains for streaming WRITES (and a read+write copy), next to the 5.96 TB/s
// streaming-read figure of mfma_f64_probe: the STFT phase writes 4-8x what it reads.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_write_probe.hip -o tools/bin/hbm_write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void write_k(d2* p, size_t n, double v)
{
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = d2{v, v};
}
__global__ void write8_k(double* p, size_t n, double v)
{
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = v;
}
__global__ void copy_k(const d2* __restrict__ s, d2* __restrict__ p, size_t n)
{
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = s[i];
}
__global__ void read_k(const d2* __restrict__ s, size_t n, double* out)
{
  double acc = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { d2 t = s[i]; acc += t[0] + t[1]; }
  if (acc == 12345.678) *out = acc;
}
// 16-byte pieces scattered one per 128-byte line-pair row (the bin-major store pattern): lane -> row, piece -> column
__global__ void scatter_k(d2* p, size_t rows, size_t ldq /* d2 per row */, int pieces)
{
  // each thread writes `pieces` consecutive 16-byte pieces of its own row
  for (size_t r = (size_t) blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (size_t) gridDim.x * blockDim.x)
    for (int q = 0; q < pieces; q++) p[r * ldq + q] = d2{1.0, 2.0};
}

// compute/store overlap: every wavefront alternates `nfma` FP64 FMAs (16 independent chains) with one frame's worth of
// stores, `rows` times -- the shape of one STFT frame.  mode 0: both, 1: compute only, 2: stores only.
// pattern 0: the frame-major row (16 x 8 bytes per lane, 512 contiguous bytes per instruction)
// pattern 1: the bin-major pieces (8 x 16 bytes per lane, every group of 4 lanes writes 64 contiguous bytes of another row)
// pattern 2: both.   bar: a workgroup barrier before the stores (all wavefronts of a CU store at the same moment)
// ld: each frame also LOADS 8 KB first (16 x float2 per lane) and consumes it -- a load behind the previous frame's stores
__global__ __launch_bounds__(512) void overlap_k(double* out, double* outT, const float2* in, int rows, int nfma, int mode,
                                                 int pattern, int bar, int ld, size_t rowStride)
{
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  double acc[16];
  for (int i = 0; i < 16; i++) acc[i] = 1.0 + lane * 1e-3 + i;
  double* p = out + wave * rows * rowStride + lane;
  float2 pre[16];
  if (ld == 2)
  {
    const float2* q = in + (wave * rows) * 1024 + lane;
#pragma unroll
    for (int i = 0; i < 16; i++) pre[i] = q[64 * i];
  }
  for (int r = 0; r < rows; r++)
  {
    if (ld == 1)
    {
      const float2* q = in + (wave * rows + r) * 1024 + lane;
#pragma unroll
      for (int i = 0; i < 16; i++) { const float2 v = q[64 * i]; acc[i] += (double) v.x + (double) v.y; }
    }
    if (ld == 2)
    {
      // ld == 2: the samples were requested before the previous frame's stores (below); consume them now
#pragma unroll
      for (int i = 0; i < 16; i++) acc[i] += (double) pre[i].x + (double) pre[i].y;
    }
    if (mode != 2)
      for (int k = 0; k < nfma / 16; k++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_fma(acc[i], 1.0000001, 1e-9);
    if (bar) __builtin_amdgcn_s_barrier();
    if (ld == 2)
    {
      const float2* q = in + (wave * rows + (r + 1 < rows ? r + 1 : r)) * 1024 + lane;
#pragma unroll
      for (int i = 0; i < 16; i++) pre[i] = q[64 * i];
    }
    if (mode != 1)
    {
      if (pattern != 1)
      {
#pragma unroll
        for (int i = 0; i < 16; i++) p[(size_t) r * rowStride + 64 * i] = acc[i];
      }
      if (pattern != 0)
      {
        // row f = 16 i + lane / 4 of a [1024][864]-double matrix per 108-frame group, piece lane % 4 of an 8-frame block
        double* t = outT + (wave / 2) * (size_t) 1024 * 864 + (size_t) ((wave & 1) * 54 + r) * 8;
#pragma unroll
        for (int i = 0; i < 8; i++)
          *reinterpret_cast<d2*>(t + (size_t) (128 * i + (lane >> 2) + 16 * (lane & 0)) * 864 + 2 * (lane & 3)) = d2{acc[i], acc[i + 8]};
      }
    }
  }
  if (mode == 1 && acc[3] == 12345.678) out[0] = acc[3];
}

int main()
{
  const size_t bytes = (size_t) 2 << 30;
  d2 *a, *b; double* o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 8);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, double gb, auto f) {
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    std::printf("%-44s %8.3f ms  %7.0f GB/s\n", name, ms, gb / (ms * 1e-3));
  };
  const size_t n = bytes / 16;
  for (int grid : {1024, 4096, 16384})
  {
    std::printf("grid %d x 256\n", grid);
    time("write 16 B/lane (2 GiB)", bytes / 1e9, [&] { hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, a, n, 1.0); });
    time("write 8 B/lane (2 GiB)", bytes / 1e9, [&] { hipLaunchKernelGGL(write8_k, dim3(grid), dim3(256), 0, 0, (double*) a, n * 2, 1.0); });
    time("read 16 B/lane (2 GiB)", bytes / 1e9, [&] { hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, n, o); });
    time("copy 16 B/lane (2 GiB read + 2 GiB written)", 2 * bytes / 1e9, [&] { hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n); });
  }
  time("hipMemsetAsync (2 GiB)", bytes / 1e9, [&] { hipMemsetAsync(a, 0, bytes, 0); });
  // bin-major pattern: rows of 6912 B (Tp = 864 doubles), 96 B (6 pieces) or 128 B (8 pieces) written per row visit
  for (int pieces : {1, 4, 6, 8})
  {
    const size_t ldq = 432, rows = bytes / (ldq * 16);
    char nm[96]; std::snprintf(nm, sizeof nm, "scattered %3d B per row of 6912 B", pieces * 16);
    time(nm, rows * pieces * 16 / 1e9, [&] { hipLaunchKernelGGL(scatter_k, dim3(4096), dim3(256), 0, 0, a, rows, ldq, pieces); });
  }
  {
    // 256 CUs x 8 wavefronts, 54 rows each = 110 592 rows of 8 KB (0.9 GB), 1344 FMAs per row: the bench shard's STFT
    const int rows = 54, nfma = 1344;
    const size_t rowStride = 1056;
    double *o2, *o3; float2* in2;
    hipMalloc(&o2, (size_t) 2048 * rows * rowStride * 8);
    hipMalloc(&o3, (size_t) 1024 * 1024 * 864 * 8);
    hipMalloc(&in2, (size_t) 2048 * rows * 1024 * 8);
    hipMemset(in2, 0, (size_t) 2048 * rows * 1024 * 8);
    struct V { int mode, pattern, bar, ld; };
    const V vs[] = {{1,0,0,0},{2,0,0,0},{0,0,0,0},{0,0,1,0},{2,1,0,0},{0,1,0,0},{0,1,1,0},{0,2,0,0},{0,2,1,0},{1,0,0,1},{0,0,0,1},{0,2,0,1},{0,2,1,1},{0,0,0,2},{0,2,0,2},{0,2,1,2}};
    for (const V& v : vs)
    {
      char nm[128]; std::snprintf(nm, sizeof nm, "overlap mode %d (1 compute 2 store 0 both) pattern %d barrier %d loads %d", v.mode, v.pattern, v.bar, v.ld);
      const double gb = v.mode == 1 ? 0.0 : 2048.0 * rows * 8192 / 1e9 * (v.pattern == 2 ? 2 : 1);
      time(nm, gb, [&] { hipLaunchKernelGGL(overlap_k, dim3(256), dim3(512), 0, 0, o2, o3, in2, rows, nfma, v.mode, v.pattern, v.bar, v.ld, rowStride); });
    }
  }
  return 0;
}
