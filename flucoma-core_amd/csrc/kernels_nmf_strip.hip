// kernels_nmf_strip.hip -- KL-NMF iteration of ONE large buffer at rank <= 16: frame strips, two launches per iteration.
//
// A single buffer (BASELINE config 2: 60 s, fft 2048, rank 16) has too few column strips for the batched kernel
// (kernels_nmf5.hip), which then splits both contractions and needs a finalize launch after each: five launches per
// iteration and V streamed twice.  Here a workgroup owns a strip of frames t (whole columns of V, all bins f):
//
//   strip kernel   the normalised W (F x 16 doubles, <= 135 KB) is staged in the LDS once per launch.  With it the H
//                  update of the strip's frames is LOCAL (the contraction runs over f, which the workgroup holds
//                  completely: ref alg/NMF.hpp:165-170), and right behind it -- with the new H still in registers --
//                  the strip's share of the NEXT W update's numerator, sum_t V/(W H) H^T over the strip's frames
//                  (alg/NMF.hpp:158-161), which goes out as one partial per workgroup.
//   reduce kernel  adds the partials in fixed order, W' <- W * num / max(den, eps) (the column normalisation of
//                  :162 stays deferred: the next strip launch normalises W while staging it).
//
// One iteration = strip + reduce; V is read once per iteration (the second phase re-reads the strip from the L2).
// Everything is deterministic (fixed summation orders, no atomics): same-seed runs are bit-identical like the
// reference's (tests/algorithms/public/TestNMF.cpp:31-39).
//
// MFMA tiles: v_mfma_f64_4x4x4_4b, lane l: x = l & 3, blk = (l >> 2) & 3, y = l >> 4; A = (row x, k y),
// B = (k y, col x), D = (row y, col x) of block blk (tools/mfma_layout_probe.hip).  A tile is 16 bins x 4 frames:
// block blk holds bins f = 32 jp + 8 blk + 2 i + e (i = row in block, e = which of the pair's two steps), so a lane's
// two steps of a pair are adjacent bins = one 16-byte load of V.  The four wavefronts of a workgroup take the bin
// pairs jp = wave, wave + 4, ... of every frame quad of the strip.
//   H phase:  Q[f][t]  = sum_k W[f][k] H[k][t]     (A = W rows, B = H)            -> D lane (x,y) = Q[f_y][t_x]
//             out[t][k] += sum_f (V/Q)[f][t] W[f][k] (A = the quotient as it lies,  B = W rows)
//   W phase:  Q[t][f]  = the same product with A and B exchanged                    -> D lane (x,y) = Q[t_y][f_x]
//             num[f][k] += sum_t (V/Q)[t][f] H[k][t]
// so in both phases the first product's result registers are the second product's A operand (no shuffles).
#include "fluhip_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace fluhip {
namespace strip {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kNQ = 6;    // frame quads of a workgroup's strip
constexpr int kEarly = 0; // bin pairs per wavefront whose V is requested before W is staged

struct StripK
{
  const double* V;
  int64_t strideV;
  int ldv;
  double* W;
  int64_t strideW;
  double* H;
  int64_t strideH;
  double* part;
  int64_t strideP;
  int psz; // doubles per workgroup partial: nPairs * 512 numerators + 16 denominators
  double* nrm;
  int F, T, K, nPairs, nq, nWG;
  int doH, doW, wPend;
  long long* dbg; // FLUHIP_STRIP_INSTR: shader-clock stamps of workgroup 0 (tools/phase_breakdown.py strip)
};

// v / d for d > 0, v >= 0 in the normal range (kernels_nmf5.hip fdiv_pos)
__device__ __forceinline__ double qdiv(double v, double d)
{
  double yv = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, yv, 1.0);
  yv = __builtin_fma(yv, e, yv);
  const double r = v * yv;
  const double res = __builtin_fma(-d, r, v);
  return __builtin_fma(res, yv, r);
}

// LDS image of W: byte offset of row f, 16-byte chunk c.  Element k = 4 m + j lives in chunk 2 j + (m >> 1), half m & 1:
// a lane's four m of one j are 32 contiguous bytes.  Two rows share a 256-byte line; the chunk position is XOR-ed with
// row bits so that both operand read patterns (16 rows x one chunk; 4 rows x 4 chunks) touch 16 different 16-byte
// bank groups.
__device__ __forceinline__ int wl_off(int f, int c)
{
  const int g = ((((c >> 1) ^ ((f >> 1) & 3)) << 2) | ((((f & 1) << 1) | (c & 1)) ^ ((f >> 3) & 3)));
  return (f >> 1) * 256 + g * 16;
}

template <int CTRL>
__device__ __forceinline__ double dppmov(double v)
{
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xffffffff), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
}

#define MFMA44(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

#define STRIP_STAMP(i)                                                                          \
  if constexpr (INSTR)                                                                           \
  {                                                                                              \
    if (g == 0 && tid == 0)                                                                      \
    {                                                                                            \
      a.dbg[2 * (i)] = (long long) __builtin_readcyclecounter();                                 \
      a.dbg[2 * (i) + 1] = (long long) wall_clock64();                                           \
    }                                                                                            \
  }

template <int NPW, bool INSTR>
__global__ __launch_bounds__(256) void nmf_strip_kernel(StripK a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
  const int g = blockIdx.x, b = blockIdx.y;
  const int wlBytes = a.nPairs * 4096;
  unsigned char* wl = smem;
  double* red = reinterpret_cast<double*>(smem + wlBytes); // [4][kNQ][4][16]; the staging statistics before that
  double* hn = red + 4 * kNQ * 4 * 16;                     // [kNQ * 4][16] the strip's new H
  double* nrmL = hn + kNQ * 64;                            // [16]
  double* denL = nrmL + 16;                                // [16] max(sum_f W[f][k], eps)

  const double* Vb = a.V + (int64_t) b * a.strideV;
  double* Wg = a.W + (int64_t) b * a.strideW;
  double* Hg = a.H + (int64_t) b * a.strideH;
  STRIP_STAMP(0)

  const int qBeg = (int) ((int64_t) g * a.nq / a.nWG), qEnd = (int) ((int64_t) (g + 1) * a.nq / a.nWG);
  const int jpLast = a.nPairs - 1;

  // V of the strip, all of it, in the H-phase view: lane (x, blk, y) holds bins 32 jp + 8 blk + 2 y + {0, 1} of frame
  // t0 + 4 q + x.  Issued right behind the loads of W (vmcnt retires in order: W must not queue behind the strip) and
  // pair-major, so that the first pairs' tiles can start while the rest of the strip is still streaming in from HBM;
  // out-of-range quads / pairs re-read a valid address and meet zero factors.
  const int nql = min(kNQ, qEnd - qBeg);
  const int t0 = 4 * qBeg;
  d2 vh[NPW][kNQ];
  double Hs[kNQ][4];

  // ---- stage W: W' / sqrt(sum W'^2) per column when the memory copy is un-normalised (alg/NMF.hpp:162) ------------
  {
    constexpr int NST = 4 * NPW; // row groups of 32
    const int cp = tid & 7, rr = tid >> 3;
    const int j0 = (2 * cp) & 3, mh = cp >> 2, ml = (cp >> 1) & 1;
    const int c0 = 2 * j0 + mh, c1 = 2 * (j0 + 1) + mh;
    d2 v[NST];
#pragma unroll
    for (int i = 0; i < NST; i++)
    {
      // no branches and no consumers here: every load goes out before the first is waited for
      v[i] = *reinterpret_cast<const d2*>(Wg + (int64_t) min(rr + 32 * i, a.F - 1) * 16 + 2 * cp);
    }
    // the strip's rows of H in the B-operand arrangement of the H phase: H[4 m + y][t0 + 4 q + x]
#pragma unroll
    for (int q = 0; q < kNQ; q++)
#pragma unroll
      for (int m = 0; m < 4; m++) Hs[q][m] = Hg[(int64_t) (t0 + 4 * min(q, nql - 1) + x) * 16 + 4 * m + y];
    __builtin_amdgcn_sched_barrier(0); // (left alone the scheduler serialises these loads to save registers)
    // the first pairs of the strip go out with W (both fit the registers while W is being staged), the rest behind it
#pragma unroll
    for (int p = 0; p < kEarly && p < NPW; p++)
#pragma unroll
      for (int q = 0; q < kNQ; q++)
        vh[p][q] = *reinterpret_cast<const d2*>(Vb + (int64_t) (t0 + 4 * min(q, nql - 1) + x) * a.ldv + 8 * blk + 2 * y +
                                                32 * min(wv + 4 * p, jpLast));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NST; i++)
      if (rr + 32 * i >= a.F) v[i] = d2{0.0, 0.0};
    double s2a = 0, s2b = 0, mxa = 0, mxb = 0, csa = 0, csb = 0;
#pragma unroll
    for (int i = 0; i < NST; i++)
    {
      s2a += v[i][0] * v[i][0];
      s2b += v[i][1] * v[i][1];
      mxa = fmax(mxa, v[i][0]);
      mxb = fmax(mxb, v[i][1]);
      csa += v[i][0];
      csb += v[i][1];
    }
    // the eight lanes of a wavefront that share a column pair (lane bits 3..5), then the four wavefronts
#pragma unroll
    for (int sh = 8; sh < 64; sh <<= 1)
    {
      s2a += __shfl_xor(s2a, sh);
      s2b += __shfl_xor(s2b, sh);
      mxa = fmax(mxa, __shfl_xor(mxa, sh));
      mxb = fmax(mxb, __shfl_xor(mxb, sh));
      csa += __shfl_xor(csa, sh);
      csb += __shfl_xor(csb, sh);
    }
    STRIP_STAMP(1)
    double* st2 = red;       // [4][16]
    double* stm = red + 64;  // [4][16]
    double* stc = red + 128; // [4][16]
    double* gm = red + 192;  // [16]
    if (lane < 8)
    {
      st2[wv * 16 + 2 * cp] = s2a;
      st2[wv * 16 + 2 * cp + 1] = s2b;
      stm[wv * 16 + 2 * cp] = mxa;
      stm[wv * 16 + 2 * cp + 1] = mxb;
      stc[wv * 16 + 2 * cp] = csa;
      stc[wv * 16 + 2 * cp + 1] = csb;
    }
    __syncthreads();
    if (tid < 16) gm[tid] = fmax(fmax(stm[tid], stm[16 + tid]), fmax(stm[32 + tid], stm[48 + tid]));
    __syncthreads();
    double gmax = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) gmax = fmax(gmax, gm[k]);
    if (tid < 16)
    {
      const double S2 = ((st2[tid] + st2[16 + tid]) + st2[32 + tid]) + st2[48 + tid];
      const double CS = ((stc[tid] + stc[16 + tid]) + stc[32 + tid]) + stc[48 + tid];
      // :162 "if (W.maxCoeff() > epsilon) W.colwise().normalize()"; padded columns keep a divisor of one
      nrmL[tid] = (a.wPend && tid < a.K && gmax > kEpsilon) ? sqrt(S2) : 1.0;
      denL[tid] = fmax(CS, kEpsilon);
    }
    __syncthreads();
    STRIP_STAMP(2)
    if (a.wPend)
    {
      const double na = nrmL[2 * cp], nb = nrmL[2 * cp + 1];
      const double ia = 1.0 / na, ib = 1.0 / nb;
      csa = 0;
      csb = 0;
#pragma unroll
      for (int i = 0; i < NST; i++)
      {
        // x / n with the reciprocal shared: quotient estimate + one residual correction
        const double ra = v[i][0] * ia, rb = v[i][1] * ib;
        v[i][0] = __builtin_fma(__builtin_fma(-na, ra, v[i][0]), ia, ra);
        v[i][1] = __builtin_fma(__builtin_fma(-nb, rb, v[i][1]), ib, rb);
        csa += v[i][0];
        csb += v[i][1];
      }
#pragma unroll
      for (int sh = 8; sh < 64; sh <<= 1)
      {
        csa += __shfl_xor(csa, sh);
        csb += __shfl_xor(csb, sh);
      }
      if (lane < 8)
      {
        stc[wv * 16 + 2 * cp] = csa;
        stc[wv * 16 + 2 * cp + 1] = csb;
      }
    }
#pragma unroll
    for (int i = 0; i < NST; i++)
      if (i < a.nPairs)
      {
        const int r = rr + 32 * i;
        *reinterpret_cast<double*>(wl + wl_off(r, c0) + ml * 8) = v[i][0];
        *reinterpret_cast<double*>(wl + wl_off(r, c1) + ml * 8) = v[i][1];
      }
    __syncthreads();
    if (a.wPend && tid < 16) denL[tid] = fmax(((stc[tid] + stc[16 + tid]) + stc[32 + tid]) + stc[48 + tid], kEpsilon);
    if (g == 0 && tid < 16) a.nrm[(int64_t) b * 16 + tid] = nrmL[tid];
    __syncthreads();
  }
  if (!a.doH && !a.doW) return;
#pragma unroll
  for (int p = kEarly; p < NPW; p++)
#pragma unroll
    for (int q = 0; q < kNQ; q++)
      vh[p][q] = *reinterpret_cast<const d2*>(Vb + (int64_t) (t0 + 4 * min(q, nql - 1) + x) * a.ldv + 8 * blk + 2 * y +
                                              32 * min(wv + 4 * p, jpLast));
  STRIP_STAMP(3)

  // per-lane LDS offsets of the two operand read patterns, [e][half]
  int offA[2][2], offB[2][2];
#pragma unroll
  for (int e = 0; e < 2; e++)
#pragma unroll
    for (int h = 0; h < 2; h++)
    {
      const int gl = ((x ^ y) << 2) | (((e << 1) | h) ^ blk);
      offA[e][h] = (4 * blk + x) * 256 + gl * 16; // rows by x, j = y
      offB[e][h] = (4 * blk + y) * 256 + gl * 16; // rows by y, j = x
    }

  double wdenAcc = 0.0;
  // one pass: the launcher sizes the grid so that no workgroup has more than kNQ quads
  if (a.doH)
  {
    // ---- H phase: the strip's frames against every bin -------------------------------------------------------------
    double out[kNQ][4];
#pragma unroll
    for (int q = 0; q < kNQ; q++)
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        if (q >= nql) Hs[q][m] = 0.0;
        out[q][m] = 0.0;
      }
    d2 wa[2][2], wb[2][2]; // operand rows of the current pair, [e][half]; the next pair's are read a pair ahead
    {
      const unsigned char* wj = wl + min(wv, jpLast) * 4096;
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
          wa[e][h] = *reinterpret_cast<const d2*>(wj + offA[e][h]);
          wb[e][h] = *reinterpret_cast<const d2*>(wj + offB[e][h]);
        }
    }
#pragma unroll
    for (int p = 0; p < NPW; p++)
    {
      const int jp = wv + 4 * p;
      __builtin_amdgcn_sched_barrier(0); // keep a pair's tiles inside the pair (register pressure)
      if (jp < a.nPairs)
      {
        d2 na[2][2], nb[2][2];
        {
          const unsigned char* wj = wl + min(jp + 4, jpLast) * 4096;
#pragma unroll
          for (int e = 0; e < 2; e++)
#pragma unroll
            for (int h = 0; h < 2; h++)
            {
              na[e][h] = *reinterpret_cast<const d2*>(wj + offA[e][h]);
              nb[e][h] = *reinterpret_cast<const d2*>(wj + offB[e][h]);
            }
        }
#pragma unroll
        for (int e = 0; e < 2; e++)
        {
          double Q[kNQ];
#pragma unroll
          for (int q = 0; q < kNQ; q++) Q[q] = 0.0;
#pragma unroll
          for (int m = 0; m < 4; m++)
#pragma unroll
            for (int q = 0; q < kNQ; q++) Q[q] = MFMA44(wa[e][m >> 1][m & 1], Hs[q][m], Q[q]);
          double R[kNQ];
#pragma unroll
          for (int q = 0; q < kNQ; q++) R[q] = qdiv(vh[p][q][e], fmax(Q[q], kEpsilon));
#pragma unroll
          for (int q = 0; q < kNQ; q++)
#pragma unroll
            for (int m = 0; m < 4; m++) out[q][m] = MFMA44(R[q], wb[e][m >> 1][m & 1], out[q][m]);
        }
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
          for (int h = 0; h < 2; h++)
          {
            wa[e][h] = na[e][h];
            wb[e][h] = nb[e][h];
          }
      }
    }
    STRIP_STAMP(4)
    // blocks of a wavefront (bins), then the four wavefronts, in fixed order
#pragma unroll
    for (int q = 0; q < kNQ; q++)
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        double v = out[q][m];
        v += dppmov<0x124>(v); // row_ror:4
        v += dppmov<0x128>(v); // row_ror:8
        if (blk == 0) red[((wv * kNQ + q) * 4 + m) * 16 + x + 4 * y] = v;
      }
    __syncthreads();
    for (int o = tid; o < kNQ * 64; o += 256)
    {
      const int q = o >> 6, rem = o & 63, yy = rem >> 4, k = rem & 15;
      double hv = 0.0;
      if (q < nql)
      {
        const int ix = (q * 4 + (k >> 2)) * 16 + (k & 3) + 4 * yy;
        const double s = ((red[ix] + red[kNQ * 64 + ix]) + red[2 * kNQ * 64 + ix]) + red[3 * kNQ * 64 + ix];
        double* hp = Hg + (int64_t) t0 * 16 + o;
        hv = *hp * s / denL[k]; // :170  H * (W^T (V / V2)) / max(W^T 1, eps)
        *hp = hv;
      }
      hn[o] = hv;
    }
  }
  else
  {
    for (int o = tid; o < kNQ * 64; o += 256) hn[o] = (o >> 6) < nql ? Hg[(int64_t) t0 * 16 + o] : 0.0;
  }
  __syncthreads();
  STRIP_STAMP(5)
  if (a.doW)
  {
    if (tid < 16)
      for (int t = 0; t < 4 * nql; t++) wdenAcc += hn[t * 16 + tid]; // :160 row sums of H, this strip's share
    // ---- W phase: the strip's share of the next W update's numerator.  The tiles are the ones the H phase used, seen
    // transposed: lane (x, blk, y) needs bins 32 jp + 8 blk + 2 x + {0, 1} of frame t0 + 4 q + y, which lane (y, blk, x)
    // holds -- one lane permutation per 32-bit half instead of a second read of V.
    double Ha[kNQ][4], Hb[kNQ][4];
#pragma unroll
    for (int q = 0; q < kNQ; q++)
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        Ha[q][m] = hn[(4 * q + x) * 16 + 4 * m + y];
        Hb[q][m] = hn[(4 * q + y) * 16 + 4 * m + x];
      }
    const int srcLane4 = 4 * (y + 4 * blk + 16 * x);
    double* P = a.part + (int64_t) b * a.strideP + (int64_t) g * a.psz;
    d2 wa[2][2];
    {
      const unsigned char* wj = wl + min(wv, jpLast) * 4096;
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int h = 0; h < 2; h++) wa[e][h] = *reinterpret_cast<const d2*>(wj + offA[e][h]);
    }
#pragma unroll
    for (int p = 0; p < NPW; p++)
    {
      const int jp = wv + 4 * p;
      __builtin_amdgcn_sched_barrier(0);
      if (jp < a.nPairs)
      {
        d2 na[2][2];
        {
          const unsigned char* wj = wl + min(jp + 4, jpLast) * 4096;
#pragma unroll
          for (int e = 0; e < 2; e++)
#pragma unroll
            for (int h = 0; h < 2; h++) na[e][h] = *reinterpret_cast<const d2*>(wj + offA[e][h]);
        }
        double vt[2][kNQ];
#pragma unroll
        for (int q = 0; q < kNQ; q++)
#pragma unroll
          for (int e = 0; e < 2; e++)
          {
            const long long bits = __double_as_longlong(vh[p][q][e]);
            const int lo = __builtin_amdgcn_ds_bpermute(srcLane4, (int) (bits & 0xffffffff));
            const int hi = __builtin_amdgcn_ds_bpermute(srcLane4, (int) (bits >> 32));
            vt[e][q] = __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
          }
#pragma unroll
        for (int e = 0; e < 2; e++)
        {
          double Q[kNQ];
#pragma unroll
          for (int q = 0; q < kNQ; q++) Q[q] = 0.0;
#pragma unroll
          for (int m = 0; m < 4; m++)
#pragma unroll
            for (int q = 0; q < kNQ; q++) Q[q] = MFMA44(Ha[q][m], wa[e][m >> 1][m & 1], Q[q]);
          double R[kNQ];
#pragma unroll
          for (int q = 0; q < kNQ; q++) R[q] = qdiv(vt[e][q], fmax(Q[q], kEpsilon));
          double num[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int q = 0; q < kNQ; q++)
#pragma unroll
            for (int m = 0; m < 4; m++) num[m] = MFMA44(R[q], Hb[q][m], num[m]);
          // complete as soon as the pair's last quad is in: out it goes (a kilobyte per store instruction)
#pragma unroll
          for (int m = 0; m < 4; m++) P[((jp * 2 + e) * 4 + m) * 64 + lane] = num[m];
        }
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
          for (int h = 0; h < 2; h++) wa[e][h] = na[e][h];
      }
    }
    STRIP_STAMP(6)
    if (tid < 16) P[a.nPairs * 512 + tid] = wdenAcc;
  }
  STRIP_STAMP(7)
}

// W'[f][k] <- W[f][k] * (sum of the numerator partials) / max(sum of the denominator partials, eps)   (alg/NMF.hpp:161)
// Element idx of a partial = ((jp * 2 + e) * 4 + m) * 64 + lane  <->  f = 32 jp + 8 blk + 2 y + e, k = 4 m + x.
constexpr int kRedWaves = 8, kRedU = 32;
__global__ __launch_bounds__(64 * kRedWaves) void nmf_strip_reduce_kernel(StripK a)
{
  __shared__ double red[kRedWaves][64];
  __shared__ double dred[32][16];
  __shared__ double den[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y;
  const int idx = blockIdx.x * 64 + lane;
  const double* P = a.part + (int64_t) b * a.strideP;
  double s = 0.0;
  for (int p0 = wv; p0 < a.nWG; p0 += kRedWaves * kRedU)
  {
    double v[kRedU];
#pragma unroll
    for (int u = 0; u < kRedU; u++)
    {
      const int p = p0 + kRedWaves * u;
      v[u] = p < a.nWG ? P[(int64_t) p * a.psz + idx] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kRedU; u++) s += v[u];
  }
  red[wv][lane] = s;
  {
    const int k = tid & 15, grp = tid >> 4; // 32 groups
    double d = 0.0;
    for (int p = grp; p < a.nWG; p += 32) d += P[(int64_t) p * a.psz + a.nPairs * 512 + k];
    dred[grp][k] = d;
  }
  __syncthreads();
  if (tid < 16)
  {
    double d = 0.0;
    for (int i = 0; i < 32; i++) d += dred[i][tid];
    den[tid] = fmax(d, kEpsilon);
  }
  __syncthreads();
  if (wv == 0)
  {
    double tot = red[0][lane];
#pragma unroll
    for (int w = 1; w < kRedWaves; w++) tot += red[w][lane];
    const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
    const int m = (idx >> 6) & 3, e = (idx >> 8) & 1, jp = idx >> 9;
    const int f = 32 * jp + 8 * blk + 2 * y + e, k = 4 * m + x;
    if (f < a.F && k < a.K)
    {
      double* wp = a.W + (int64_t) b * a.strideW + (int64_t) f * 16 + k;
      double w = *wp;
      if (a.wPend) w = w / a.nrm[(int64_t) b * 16 + k];
      *wp = w * tot / den[k];
    }
  }
}

int strip_pairs(int F) { return (F + 31) / 32; }

} // namespace strip
using namespace strip;

bool nmf_strip_supported(int F, int T, int Kp) { return Kp == 16 && F >= 1 && strip_pairs(F) <= 36 && T >= 1; }
// one pass of at most kNQ frame quads per workgroup: 256 workgroups (one per CU) while that holds, more for longer buffers
int nmf_strip_workgroups(int T)
{
  const int64_t nq = (T + 3) / 4;
  return (int) std::max<int64_t>(std::min<int64_t>(256, nq), (nq + kNQ - 1) / kNQ);
}
int64_t nmf_strip_part_doubles(int F, int T, int B)
{
  return (int64_t) B * nmf_strip_workgroups(T) * (strip_pairs(F) * 512 + 16) + 16;
}

static StripK make_k(const StripArgs& s)
{
  StripK k;
  k.V = s.V; k.strideV = s.strideV; k.ldv = (int) s.ldv;
  k.W = s.W; k.strideW = s.strideW;
  k.H = s.H; k.strideH = s.strideH;
  k.part = s.part;
  k.nPairs = strip_pairs(s.F);
  k.psz = k.nPairs * 512 + 16;
  k.nWG = nmf_strip_workgroups(s.T);
  k.strideP = (int64_t) k.nWG * k.psz;
  k.nrm = s.nrm;
  k.F = s.F; k.T = s.T; k.K = s.K;
  k.nq = (s.T + 3) / 4;
  k.doH = s.doH; k.doW = s.doW; k.wPend = s.wPend;
  // the stamps live behind the partials (nmf_strip_part_doubles leaves 16 doubles for them)
  k.dbg = reinterpret_cast<long long*>(s.part + (int64_t) s.B * k.strideP);
  return k;
}

template <int NPW, bool INSTR = false>
static void launch_strip_t(const StripK& k, int B, hipStream_t s)
{
  const size_t shmem = (size_t) k.nPairs * 4096 + (size_t) (4 * kNQ * 4 * 16 + kNQ * 64 + 32) * sizeof(double);
  auto kern = nmf_strip_kernel<NPW, INSTR>;
  (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) shmem);
  const unsigned grid = (k.doH || k.doW) ? (unsigned) k.nWG : 1u;
  hipLaunchKernelGGL(kern, dim3(grid, (unsigned) B), dim3(256), shmem, s, k);
}

void launch_nmf_strip(const StripArgs& a, hipStream_t s)
{
  const StripK k = make_k(a);
  const int npw = (k.nPairs + 3) / 4;
  if (npw <= 3) launch_strip_t<3>(k, a.B, s);
  else if (npw <= 5) launch_strip_t<5>(k, a.B, s);
  else
  {
    static const int instr = [] { const char* e = std::getenv("FLUHIP_STRIP_INSTR"); return e ? std::atoi(e) : 0; }();
    if (instr && k.doH && k.doW) launch_strip_t<9, true>(k, a.B, s);
    else launch_strip_t<9>(k, a.B, s);
  }
}

void launch_nmf_strip_reduce(const StripArgs& a, hipStream_t s)
{
  const StripK k = make_k(a);
  hipLaunchKernelGGL(nmf_strip_reduce_kernel, dim3((unsigned) (k.nPairs * 8), (unsigned) a.B), dim3(64 * kRedWaves), 0, s, k);
}

} // namespace fluhip
