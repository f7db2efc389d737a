// mfma_block_probe.hip -- the second-product block of the frame-strip kernel's H phase, verbatim (48 x
// v_mfma_f64_4x4x4_4b_f64 with the kernel's register numbers), timed in a loop on one wavefront per SIMD: is the block
// itself slower than 16 cycles per MFMA, or is it what surrounds it in the kernel?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_block_probe.hip -o tools/bin/mfma_block_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 1) void probe(long long* cyc, int iters, int mode)
{
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++)
  {
    if (mode == 1)
      asm volatile("v_fma_f64 v[166:167], v[166:167], v[166:167], v[166:167]\n\tv_fma_f64 v[168:169], v[168:169], v[168:169], v[168:169]\n\t"
                   "v_fma_f64 v[138:139], v[138:139], v[138:139], v[138:139]" ::: "v166", "v167", "v168", "v169", "v138", "v139");
    asm volatile("v_mfma_f64_4x4x4_4b_f64 a[34:35], v[166:167], a[48:49], a[34:35]\n\tv_mfma_f64_4x4x4_4b_f64 a[38:39], v[166:167], a[50:51], a[38:39]\n\tv_mfma_f64_4x4x4_4b_f64 a[42:43], v[166:167], a[56:57], a[42:43]\n\tv_mfma_f64_4x4x4_4b_f64 a[46:47], v[166:167], a[58:59], a[46:47]\n\tv_mfma_f64_4x4x4_4b_f64 a[44:45], v[168:169], a[48:49], a[44:45]\n\tv_mfma_f64_4x4x4_4b_f64 a[40:41], v[168:169], a[50:51], a[40:41]\n\tv_mfma_f64_4x4x4_4b_f64 a[36:37], v[168:169], a[56:57], a[36:37]\n\tv_mfma_f64_4x4x4_4b_f64 a[32:33], v[168:169], a[58:59], a[32:33]\n\tv_mfma_f64_4x4x4_4b_f64 a[30:31], v[170:171], a[48:49], a[30:31]\n\tv_mfma_f64_4x4x4_4b_f64 a[28:29], v[170:171], a[50:51], a[28:29]\n\tv_mfma_f64_4x4x4_4b_f64 a[26:27], v[170:171], a[56:57], a[26:27]\n\tv_mfma_f64_4x4x4_4b_f64 a[24:25], v[170:171], a[58:59], a[24:25]\n\tv_mfma_f64_4x4x4_4b_f64 a[22:23], v[172:173], a[48:49], a[22:23]\n\tv_mfma_f64_4x4x4_4b_f64 a[20:21], v[172:173], a[50:51], a[20:21]\n\tv_mfma_f64_4x4x4_4b_f64 a[18:19], v[172:173], a[56:57], a[18:19]\n\tv_mfma_f64_4x4x4_4b_f64 a[16:17], v[172:173], a[58:59], a[16:17]\n\tv_mfma_f64_4x4x4_4b_f64 a[14:15], v[174:175], a[48:49], a[14:15]\n\tv_mfma_f64_4x4x4_4b_f64 a[12:13], v[174:175], a[50:51], a[12:13]\n\tv_mfma_f64_4x4x4_4b_f64 a[10:11], v[174:175], a[56:57], a[10:11]\n\tv_mfma_f64_4x4x4_4b_f64 a[8:9], v[174:175], a[58:59], a[8:9]\n\tv_mfma_f64_4x4x4_4b_f64 a[6:7], v[176:177], a[48:49], a[6:7]\n\tv_mfma_f64_4x4x4_4b_f64 a[4:5], v[176:177], a[50:51], a[4:5]\n\tv_mfma_f64_4x4x4_4b_f64 a[2:3], v[176:177], a[56:57], a[2:3]\n\tv_mfma_f64_4x4x4_4b_f64 a[0:1], v[176:177], a[58:59], a[0:1]\n\tv_mfma_f64_4x4x4_4b_f64 a[34:35], v[138:139], a[52:53], a[34:35]\n\tv_mfma_f64_4x4x4_4b_f64 a[38:39], v[138:139], a[54:55], a[38:39]\n\tv_mfma_f64_4x4x4_4b_f64 a[42:43], v[138:139], a[60:61], a[42:43]\n\tv_mfma_f64_4x4x4_4b_f64 a[46:47], v[138:139], a[62:63], a[46:47]\n\tv_mfma_f64_4x4x4_4b_f64 a[44:45], v[140:141], a[52:53], a[44:45]\n\tv_mfma_f64_4x4x4_4b_f64 a[40:41], v[140:141], a[54:55], a[40:41]\n\tv_mfma_f64_4x4x4_4b_f64 a[36:37], v[140:141], a[60:61], a[36:37]\n\tv_mfma_f64_4x4x4_4b_f64 a[32:33], v[140:141], a[62:63], a[32:33]\n\tv_mfma_f64_4x4x4_4b_f64 a[30:31], v[142:143], a[52:53], a[30:31]\n\tv_mfma_f64_4x4x4_4b_f64 a[28:29], v[142:143], a[54:55], a[28:29]\n\tv_mfma_f64_4x4x4_4b_f64 a[26:27], v[142:143], a[60:61], a[26:27]\n\tv_mfma_f64_4x4x4_4b_f64 a[24:25], v[142:143], a[62:63], a[24:25]\n\tv_mfma_f64_4x4x4_4b_f64 a[22:23], v[144:145], a[52:53], a[22:23]\n\tv_mfma_f64_4x4x4_4b_f64 a[20:21], v[144:145], a[54:55], a[20:21]\n\tv_mfma_f64_4x4x4_4b_f64 a[18:19], v[144:145], a[60:61], a[18:19]\n\tv_mfma_f64_4x4x4_4b_f64 a[16:17], v[144:145], a[62:63], a[16:17]\n\tv_mfma_f64_4x4x4_4b_f64 a[14:15], v[162:163], a[52:53], a[14:15]\n\tv_mfma_f64_4x4x4_4b_f64 a[12:13], v[162:163], a[54:55], a[12:13]\n\tv_mfma_f64_4x4x4_4b_f64 a[10:11], v[162:163], a[60:61], a[10:11]\n\tv_mfma_f64_4x4x4_4b_f64 a[8:9], v[162:163], a[62:63], a[8:9]\n\tv_mfma_f64_4x4x4_4b_f64 a[6:7], v[164:165], a[52:53], a[6:7]\n\tv_mfma_f64_4x4x4_4b_f64 a[4:5], v[164:165], a[54:55], a[4:5]\n\tv_mfma_f64_4x4x4_4b_f64 a[2:3], v[164:165], a[60:61], a[2:3]\n\tv_mfma_f64_4x4x4_4b_f64 a[0:1], v[164:165], a[62:63], a[0:1]"
                 :
                 :
                 : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 5) cyc[0] = t1 - t0;
}
int main()
{
  long long* cyc;
  hipMalloc(&cyc, 64);
  for (int mode = 0; mode < 2; mode++)
  {
    for (int rep = 0; rep < 2; rep++)
    {
      hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, cyc, 2000, mode);
      hipDeviceSynchronize();
    }
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("mode %d: %6.2f cycles per MFMA (48 per block)\n", mode, (double) c / 2000 / 48.0);
  }
  return 0;
}
