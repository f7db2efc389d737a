// kernels_nmf4.hip -- the NMF factor update on v_mfma_f64_4x4x4_4b_f64.
//
// Why this instruction: on MI355X the four-block 4x4x4 FP64 MFMA issues at the full FP64
// matrix rate (73.5 TFLOP/s measured = 93 % of the 78.6 TFLOP/s datasheet peak, already from
// one wavefront per SIMD), whereas v_mfma_f64_16x16x4_f64 tops out at 49.6 TFLOP/s
// (profiles/r01/mfma_f64_probe.txt).  The 16x16x4 kernel in kernels_nmf.hip is kept for A/B.
//
// Lane map of v_mfma_f64_4x4x4_4b_f64 (measured with tools/mfma_layout_probe.hip):
//   lane l: x = l & 3, blk = (l >> 2) & 3, y = l >> 4
//   A: row i = x of block blk, contraction index k = y        (one f64 per lane)
//   B: col j = x of block blk, contraction index k = y
//   D: row i = y of block blk, col j = x
// i.e. D_blk[y][x] = sum_k A_blk[.][k] B_blk[k][.] for four independent 4x4 blocks.  As with the
// 16x16x4 form, the D layout of a product read as "rows = contraction, cols = rows of the next
// product" is exactly the A layout, so the quotient tile chains register-to-register:
//
//   step over 4 rows r0..r0+3 of the contraction index, 16 columns c per group (4 per block):
//     Q[r0+y][c(blk,x)]   = sum_m  MFMA( A = Mv[r0+x][M*y+m],  B = S[c(blk,x)][M*y+m] )
//     ratio               = V[r0+y][c(blk,x)] / max(Q, eps)                       (same lane)
//     out[c(blk,y)][M*x+m] += MFMA( A = ratio, B = Mv[r0+y][M*x+m] )
//
// M = Kp/4.  Each wavefront owns up to NG column groups: the stationary rows of S and the
// accumulators live in registers for the whole pass, the 4 x Kp slab of the moving factor is
// loaded once per step (in the two register distributions above) and reused by every group.
#include "fluhip_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace fluhip {

typedef double d2 __attribute__((ext_vector_type(2)));

struct Upd4Args
{
  const double* V;
  int64_t ldv, strideV;
  const double* Mv;
  int64_t strideM;
  double* S;
  int64_t strideS;
  int R, C, B;
  int nGroups;        // ceil(C / 16)
  int wavesPerBuf;    // strips per buffer
  int wgPerBuf;       // ceil(wavesPerBuf / 4)
  int nSteps;         // ceil(R / 4)
  int nsplit, stepsPerSplit;
  double* part;
  double* dpart;
  int64_t Cp;
  int xcdMap;
};

template <int N>
__device__ __forceinline__ void load_vec(double (&dst)[N], const double* p)
{
  if constexpr (N % 2 == 0)
  {
#pragma unroll
    for (int j = 0; j < N; j += 2)
    {
      d2 t = *reinterpret_cast<const d2*>(p + j);
      dst[j] = t[0];
      dst[j + 1] = t[1];
    }
  }
  else
  {
#pragma unroll
    for (int j = 0; j < N; j++) dst[j] = p[j];
  }
}

template <int M, int NG, int VAR>
__global__ __launch_bounds__(256, 1) void nmf_update4_kernel(Upd4Args a)
{
  constexpr int KP = 4 * M;
  int id = blockIdx.x;
  int buf, wg, split;
  if (a.xcdMap)
  {
    const int xcd = id & 7;
    int slot = id >> 3;
    split = slot % a.nsplit;
    slot /= a.nsplit;
    wg = slot % a.wgPerBuf;
    buf = xcd + 8 * (slot / a.wgPerBuf);
  }
  else
  {
    split = id % a.nsplit;
    wg = (id / a.nsplit) % a.wgPerBuf;
    buf = id / (a.nsplit * a.wgPerBuf);
  }
  if (buf >= a.B) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = wg * 4 + wave;
  if (strip >= a.wavesPerBuf) return;
  // groups are dealt out as evenly as possible: the first `rem` strips take one more
  const int base = a.nGroups / a.wavesPerBuf, rem = a.nGroups % a.wavesPerBuf;
  const int g0 = strip * base + min(strip, rem);
  const int ng = base + (strip < rem ? 1 : 0);
  if (ng <= 0) return;

  const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
  const double* __restrict__ V = a.V + (int64_t) buf * a.strideV;
  const double* __restrict__ Mv = a.Mv + (int64_t) buf * a.strideM;
  double* S = a.S + (int64_t) buf * a.strideS;

  // stationary operand rows and accumulators.  Groups beyond this strip's share (g >= ng) run
  // on a zero stationary operand and are never stored: keeping every group unconditional keeps
  // the whole step in one basic block, so the scheduler interleaves the NG dependent MFMA chains.
  double sb[NG][M];
  double acc[NG][M];
  int voff[NG];
#pragma unroll
  for (int g = 0; g < NG; g++)
  {
#pragma unroll
    for (int m = 0; m < M; m++) { acc[g][m] = 0.0; sb[g][m] = 0.0; }
    if (g < ng) load_vec<M>(sb[g], S + (int64_t) ((g0 + g) * 16 + 4 * blk + x) * KP + M * y);
    voff[g] = (g < ng) ? g * 16 : 0;
  }
  double dsum[M];
#pragma unroll
  for (int m = 0; m < M; m++) dsum[m] = 0.0;

  const int s0 = split * a.stepsPerSplit;
  const int s1 = min(s0 + a.stepsPerSplit, a.nSteps);
  const double* vcol = V + (int64_t) g0 * 16 + 4 * blk + x;
  const double* maBase = Mv + (int64_t) x * KP + M * y;
  const double* mbBase = Mv + (int64_t) y * KP + M * x;

  const int sLast = s1 - 1;
  auto load_ma = [&](int st, double (&ma)[M]) {
    const int64_t r0 = (int64_t) min(st, sLast) * 4; // clamped: never reads past the padded rows
    if constexpr (VAR & 16) { for (int m = 0; m < M; m++) ma[m] = 1e-3 * (st + m); return; }
    load_vec<M>(ma, maBase + r0 * KP);
  };
  auto load_mbv = [&](int st, double (&mb)[M], double (&v)[NG]) {
    const int64_t r0 = (int64_t) min(st, sLast) * 4;
    if constexpr (VAR & 16) { for (int m = 0; m < M; m++) mb[m] = 1e-3 * (st + m); }
    else load_vec<M>(mb, mbBase + r0 * KP);
    const double* vrow = vcol + (r0 + y) * a.ldv;
    if constexpr (VAR & 8) { for (int g = 0; g < NG; g++) v[g] = 1e-2 * (st + g); return; }
#pragma unroll
    for (int g = 0; g < NG; g++) v[g] = vrow[voff[g]];
  };
  // Q[r0+y][c(g,blk,x)] for one 4-row step: NG independent chains of M dependent MFMAs
  auto q_phase = [&](const double (&ma)[M], double (&q)[NG]) {
#pragma unroll
    for (int g = 0; g < NG; g++) q[g] = 0.0;
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
      for (int g = 0; g < NG; g++) q[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(ma[m], sb[g][m], q[g], 0, 0, 0);
  };
  // V / max(Q, eps): reciprocal seed + two Newton steps + one residual correction (<= 1-2 ulp;
  // Q >= eps and V >= 0 are far from the over/underflow cases IEEE division scaling is for)
  auto ratio_phase = [&](const double (&v)[NG], const double (&q)[NG], double (&ratio)[NG]) {
    if constexpr (VAR & 1)
    {
      // stage by stage across the NG independent quotients, so consecutive VALU ops never depend
      // (same arithmetic as the per-quotient form below)
      double d[NG], yv[NG];
#pragma unroll
      for (int g = 0; g < NG; g++) d[g] = q[g] > kEpsilon ? q[g] : kEpsilon;
#pragma unroll
      for (int g = 0; g < NG; g++) yv[g] = __builtin_amdgcn_rcp(d[g]);
#pragma unroll
      for (int g = 0; g < NG; g++) ratio[g] = __builtin_fma(-d[g], yv[g], 1.0);
#pragma unroll
      for (int g = 0; g < NG; g++) yv[g] = __builtin_fma(yv[g], ratio[g], yv[g]);
#pragma unroll
      for (int g = 0; g < NG; g++) ratio[g] = v[g] * yv[g];
#pragma unroll
      for (int g = 0; g < NG; g++) d[g] = __builtin_fma(-d[g], ratio[g], v[g]);
#pragma unroll
      for (int g = 0; g < NG; g++) ratio[g] = __builtin_fma(d[g], yv[g], ratio[g]);
    }
    else
    {
#pragma unroll
      for (int g = 0; g < NG; g++)
      {
        if constexpr (VAR & 4) { ratio[g] = v[g] / fmax(q[g], kEpsilon); }
        else
        {
          // clamp without the sNaN canonicalisation fmax() costs on the shared FP64 datapath
          const double d = q[g] > kEpsilon ? q[g] : kEpsilon;
          // v_rcp_f64 (~24 bits) -> one Newton step (48 bits) -> quotient -> residual correction:
          // r' = r + y (v - d r) leaves a relative error of (1 - d y)^2 ~ 2^-96, i.e. rounding only
          double yv = __builtin_amdgcn_rcp(d);
          const double e = __builtin_fma(-d, yv, 1.0);
          yv = __builtin_fma(yv, e, yv);
          const double r = v[g] * yv;
          const double e2 = __builtin_fma(-d, r, v[g]);
          ratio[g] = __builtin_fma(e2, yv, r);
        }
      }
    }
  };
  auto out_phase = [&](const double (&ratio)[NG], const double (&mb)[M]) {
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
      for (int m = 0; m < M; m++) acc[g][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(ratio[g], mb[m], acc[g][m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < M; m++) dsum[m] += mb[m];
  };

  // The quotient arithmetic (VALU) of step s is independent of the Q MFMAs of step s+1: ask the
  // scheduler to issue them alternately so the VALU work hides under the matrix pipe.
  auto interleave_valu_mfma = [&]() {
    if constexpr ((VAR & 2) == 0) return;
#pragma unroll
    for (int i = 0; i < NG * M; i++)
    {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); // two VALU
    }
  };

  // Software pipeline over the 4-row steps, two register sets (A/B), one wavefront per SIMD:
  //   iteration s:  [ ratio(s) on the VALU  ||  Q(s+1) on the matrix pipe ]  ->  out(s)
  // with ma(s+2) loaded at the top (its set was freed by Q(s)) and mb/v(s+2) loaded after
  // out(s) has consumed mb/v(s).  Every load has a full iteration of matrix work to land.
  double maA[M], mbA[M], vA[NG], qA[NG], maB[M], mbB[M], vB[NG], qB[NG], ratio[NG];
  if (s0 < s1)
  {
    load_ma(s0, maA);
    load_mbv(s0, mbA, vA);
    load_ma(s0 + 1, maB);
    load_mbv(s0 + 1, mbB, vB);
    q_phase(maA, qA);
    for (int s = s0; s < s1; s += 2)
    {
      load_ma(s + 2, maA);
      ratio_phase(vA, qA, ratio);
      q_phase(maB, qB);
      interleave_valu_mfma();
      out_phase(ratio, mbA);
      load_mbv(s + 2, mbA, vA);
      if (s + 1 < s1)
      {
        load_ma(s + 3, maB);
        ratio_phase(vB, qB, ratio);
        q_phase(maA, qA);
        interleave_valu_mfma();
        out_phase(ratio, mbB);
        load_mbv(s + 3, mbB, vB);
      }
    }
  }

  // column sums of Mv: lane holds the partial over rows == y (mod 4) for k = M*x + m
#pragma unroll
  for (int m = 0; m < M; m++)
  {
    double d = dsum[m];
    d += __shfl_xor(d, 16);
    d += __shfl_xor(d, 32);
    dsum[m] = d;
  }

  // acc[g][m] = out[col = (g0+g)*16 + 4*blk + y][k = M*x + m]
  if (a.nsplit == 1)
  {
#pragma unroll
    for (int g = 0; g < NG; g++)
    {
      if (g < ng)
      {
        const int col = (g0 + g) * 16 + 4 * blk + y;
        if (col < a.C)
        {
          double* sp = S + (int64_t) col * KP + M * x;
          double sold[M];
          load_vec<M>(sold, sp);
#pragma unroll
          for (int m = 0; m < M; m++) sp[m] = (sold[m] * acc[g][m]) / fmax(dsum[m], kEpsilon);
        }
      }
    }
  }
  else
  {
    double* part = a.part + ((int64_t) buf * a.nsplit + split) * a.Cp * KP;
#pragma unroll
    for (int g = 0; g < NG; g++)
    {
      if (g < ng)
      {
        const int col = (g0 + g) * 16 + 4 * blk + y;
        double* pp = part + (int64_t) col * KP + M * x;
#pragma unroll
        for (int m = 0; m < M; m++) pp[m] = acc[g][m];
      }
    }
    if (strip == 0 && blk == 0 && y == 0)
    {
      double* dp = a.dpart + ((int64_t) buf * a.nsplit + split) * KP + M * x;
#pragma unroll
      for (int m = 0; m < M; m++) dp[m] = dsum[m];
    }
  }
}

// defined in kernels_nmf.hip

template <int M, int NG, int VAR = 0>
static void launch4_t(const UpdateArgs& a, int wavesPerBuf, hipStream_t s)
{
  Upd4Args k;
  k.V = a.V; k.ldv = a.ldv; k.strideV = a.strideV;
  k.Mv = a.Mv; k.strideM = a.strideM;
  k.S = a.S; k.strideS = a.strideS;
  k.R = a.R; k.C = a.C; k.B = a.B;
  k.nGroups = (a.C + 15) / 16;
  k.wavesPerBuf = wavesPerBuf;
  k.wgPerBuf = (wavesPerBuf + 3) / 4;
  k.nSteps = (a.R + 3) / 4;
  k.nsplit = a.nsplit < 1 ? 1 : a.nsplit;
  k.stepsPerSplit = (k.nSteps + k.nsplit - 1) / k.nsplit;
  k.part = a.part; k.dpart = a.dpart; k.Cp = a.Cp;
  k.xcdMap = a.B >= 8 ? 1 : 0;
  const int bufs = k.xcdMap ? (int) round_up(a.B, 8) : a.B;
  const unsigned grid = (unsigned) (bufs * k.wgPerBuf * k.nsplit);
  hipLaunchKernelGGL((nmf_update4_kernel<M, NG, VAR>), dim3(grid), dim3(256), 0, s, k);
  if (k.nsplit > 1)
    launch_update_finalize(a.S, a.strideS, a.part, a.dpart, a.C, a.Kp, a.Cp, k.nsplit, a.B, s);
}

// register budget: 4*M VGPRs per group (stationary + accumulator) out of 512
static int max_groups(int M)
{
  if (M <= 8) return 9;
  if (M <= 16) return 4;
  return 2;
}

// strips (wavefronts) per buffer: fill the 1024 SIMDs in whole rounds, then as few strips as the
// register budget allows (more groups per strip = more reuse of the moving-factor slab)
int nmf_update4_waves_per_buffer(int C, int Kp, int B)
{
  const int M = Kp / 4, G = (C + 15) / 16, ngmax = max_groups(M);
  const int wmin = (G + ngmax - 1) / ngmax;
  static const int forceW = [] { const char* e = std::getenv("FLUHIP_PLAN_W"); return e ? std::atoi(e) : 0; }();
  if (forceW > 0) return forceW < wmin ? wmin : (forceW > G ? G : forceW);
  int w = wmin;
  const int simds = 1024;
  if ((int64_t) B * w >= simds)
  {
    // round the strip count up so that B*w is a multiple of the SIMD count when that is cheap
    for (int cand = wmin; cand <= wmin + 2 && cand <= G; cand++)
      if (((int64_t) B * cand) % simds == 0) { w = cand; break; }
  }
  else
  {
    const int64_t wfill = (simds + B - 1) / B; // strips per buffer that give every SIMD a wavefront
    if (wfill <= G / 3 || wfill <= wmin)
      w = (int) std::max<int64_t>(wmin, wfill);  // fill the chip by narrowing strips (>= 3 groups each)
    else
      w = std::max(wmin, (G + 2) / 3);           // few buffers: at most 3 groups per strip (G / 3 rounded DOWN made the
                                                 // widest strip 4 groups -- a quarter more work per wavefront), the contraction
                                                 // split (nsplit) supplies the rest of the parallelism
    if (w > G) w = G;
    if (w < 1) w = 1;
  }
  return w;
}

template <int M, int NG>
static void launch4_ng(const UpdateArgs& a, int w, int ng, hipStream_t s)
{
  if constexpr (NG == 1) launch4_t<M, 1>(a, w, s);
  else
  {
    if (ng >= NG)
    {
      if constexpr (M == 8 && NG >= 7)
      {
        // experiment switch (kept while tuning): FLUHIP_K4_VAR = bitmask 1 horiz-div, 2 interleave, 4 IEEE div
        static const int var = [] { const char* e = std::getenv("FLUHIP_K4_VAR"); return e ? std::atoi(e) : -1; }();
        switch (var)
        {
        case 0: launch4_t<M, NG, 0>(a, w, s); break;
        case 24: launch4_t<M, NG, 24>(a, w, s); break; // no operand loads at all: compute-side floor
        default: launch4_t<M, NG, (M * NG <= 56 ? 1 : 0)>(a, w, s); break;
        }
      }
      else
        launch4_t<M, NG, (M * NG <= 56 ? 1 : 0)>(a, w, s);
    }
    else launch4_ng<M, NG - 1>(a, w, ng, s);
  }
}

template <int M>
static void launch4_m(const UpdateArgs& a, hipStream_t s)
{
  const int w = nmf_update4_waves_per_buffer(a.C, a.Kp, a.B);
  const int G = (a.C + 15) / 16;
  const int ng = (G + w - 1) / w;
  constexpr int NGMAX = M <= 8 ? 9 : (M <= 16 ? 4 : 1);
  launch4_ng<M, NGMAX>(a, w, ng, s);
}

void launch_nmf_update4(const UpdateArgs& a, hipStream_t s)
{
  switch (a.Kp / 4)
  {
  case 4: launch4_m<4>(a, s); break;
  case 8: launch4_m<8>(a, s); break;
  case 16: launch4_m<16>(a, s); break;
  default: break;
  }
}

bool nmf_update4_supported(int Kp) { return Kp == 16 || Kp == 32 || Kp == 64; }

} // namespace fluhip
