// prints the lane mapping of v_mfma_f64_4x4x4_4b_f64: for every (lane of A one-hot, lane of B one-hot)
// pair that produces a non-zero, the lane of D that receives it.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* res)
{
  const int l = threadIdx.x;
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++)
    {
      double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      if (d != 0.0) res[la * 64 + lb] = l;
    }
}
int main()
{
  int* d; hipMalloc(&d, 4096 * 4); hipMemset(d, 0xff, 4096 * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; la++)
  {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; lb++) if (h[la * 64 + lb] >= 0) printf("  B%2d->D%2d", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
