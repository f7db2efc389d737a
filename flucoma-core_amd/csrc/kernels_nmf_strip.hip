// kernels_nmf_strip.hip -- KL-NMF iteration of ONE large buffer at rank <= 16: frame strips, two launches per iteration.
//
// A single buffer (BASELINE config 2: 60 s, fft 2048, rank 16) has too few column strips for the batched kernel
// (kernels_nmf5.hip), which then splits both contractions and needs a finalize launch after each: five launches per
// iteration and V streamed twice.  Here a workgroup owns a strip of frames t (whole columns of V, all bins f):
//
//   strip kernel   every wavefront keeps the rows of W it works on (F x 16 doubles altogether, <= 135 KB) in the LDS.  With them the H
//                  update of the strip's frames is LOCAL (the contraction runs over f, which the workgroup holds
//                  completely: ref alg/NMF.hpp:165-170), and right behind it -- with the new H still in registers --
//                  the strip's share of the NEXT W update's numerator, sum_t V/(W H) H^T over the strip's frames
//                  (alg/NMF.hpp:158-161), which goes out as one partial per workgroup.
//   reduce kernel  adds the partials in fixed order, W' <- W * num / max(den, eps), and leaves the column statistics
//                  of W' (sum x^2, sum x, max) as per-workgroup records: the normalisation of :162 stays deferred,
//                  every later use works with (W', norms) -- Q = W' (H / nrm), results divided by nrm.
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
#include "nmf_tile_stats.h"
#include "recip_tree.h"

#include <algorithm>
#include <cstdlib>

namespace fluhip {
namespace strip {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kNQ = 6;    // frame quads of a workgroup's strip
constexpr int kAhead = 1; // bin pairs of V a wavefront requests ahead of their use.  Requests beyond what the memory pipe takes
                          // block the issue of everything behind them: 3 pairs ahead cost 1.3 us more in the prologue and
                          // gain nothing in the loop (which is bound by its instruction stream, not by the arrival of V:
                          // profiles/r02/strip_notes.md), 5 pairs 3 us.
constexpr int kEarly = 1; // ... of which are requested before the column norms are known
constexpr bool kWriteThrough = true;
// quotients of the tile loops without / with the residual correction: see kernels_nmf5.hip (same switch)
#ifndef FLUHIP_QUOTIENT_CORRECTION
#define FLUHIP_QUOTIENT_CORRECTION 0
#endif
constexpr bool kQuotientCorrection = FLUHIP_QUOTIENT_CORRECTION != 0;
constexpr int kStatW = 12; // statistics: 12 doubles per column block (3 kinds x 4 columns), stored [kind][column][bin step]

struct StripK
{
  const double* V;
  int64_t strideV;
  int ldv;
  double* W;
  int64_t strideW;
  double* H;
  int64_t strideH;
  double* part;     // [B][nBlk / 4][nWG][4][64] numerator partials: a strip workgroup writes 2 KB runs (the four column
                    // blocks of one bin step), the four reduce workgroups of a step share one nWG x 2 KB region
  double* dpart;    // [B][nWG][16] denominator partials
  const double* statIn; // [B][nBlk][kStatW] column statistics of the W in memory
  double* statOut;      // reduce / wstats: the statistics of the W they leave behind
  double* nrm;
  int F, T, K, nPairs, nBlk, nq, nWG, qBase, qRem;
  int doH, doW, wPend;
  long long* dbg; // FLUHIP_STRIP_INSTR: shader-clock stamps of workgroup 0 (tools/strip_timing.py)
  const double* tileStat; // TILE instantiations: tile records [B][48][nRec] of the W' in memory (kernels_nmf_bintile.hip)
  int nRec;
  double* sideOut;        // doW == 2: bin F - 1's numerator partials, [B][nWG][16]
};

// LDS image of W (each wavefront keeps the bin pairs it works on): byte offset of row f, 16-byte chunk c = columns
// (2 c, 2 c + 1), inside the pair's 4 KB.  Two rows share a 256-byte line; the chunk position is XOR-ed with row bits so
// that both operand read patterns (16 rows x one chunk pair; 4 rows x 4 chunk pairs) touch 16 different 16-byte bank
// groups.
__device__ __forceinline__ int wl_off(int f, int c)
{
  const int g = ((((c >> 1) ^ ((f >> 1) & 3)) << 2) | ((((f & 1) << 1) | (c & 1)) ^ ((f >> 3) & 3)));
  return ((f >> 1) & 15) * 256 + g * 16;
}

template <int CTRL>
__device__ __forceinline__ double dppmov(double v)
{
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xffffffff), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
}

// quotients per reciprocal (recip_tree.h; round 6): 1 = a v_rcp_f64 per quotient (rounds 2 - 5), 3 / 4 / 6 = product trees
#ifndef FLUHIP_STRIP_RCP_GROUP
#define FLUHIP_STRIP_RCP_GROUP 6
#endif
constexpr int kStripRcpGroup = FLUHIP_STRIP_RCP_GROUP;

// R[q] = V[q] / max(Q[q], eps) for the kNQ tiles of one step, stage by stage: left to the scheduler the seven dependent
// operations of one quotient run back to back (register pressure), each waiting for the one before it
#define STRIP_QUOT(R, V, Q)                                                                                   \
  {                                                                                                           \
    double d_[NQ], y_[NQ], e_[NQ];                                                                         \
    _Pragma("unroll") for (int q = 0; q < NQ; q++) d_[q] = fmax(Q[q], kEpsilon);                             \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if constexpr (kStripRcpGroup > 1 && !kQuotientCorrection)                                                 \
    {                                                                                                         \
      recip_tree<NQ, kStripRcpGroup>(d_, y_);                                                                 \
    }                                                                                                         \
    else                                                                                                      \
    {                                                                                                         \
    _Pragma("unroll") for (int q = 0; q < NQ; q++) y_[q] = __builtin_amdgcn_rcp(d_[q]);                      \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    _Pragma("unroll") for (int q = 0; q < NQ; q++) e_[q] = __builtin_fma(-d_[q], y_[q], 1.0);                \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    _Pragma("unroll") for (int q = 0; q < NQ; q++) y_[q] = __builtin_fma(y_[q], e_[q], y_[q]);               \
    }                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if constexpr (kQuotientCorrection)                                                                        \
    {                                                                                                         \
      _Pragma("unroll") for (int q = 0; q < NQ; q++) e_[q] = V[q] * y_[q];                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      _Pragma("unroll") for (int q = 0; q < NQ; q++) d_[q] = __builtin_fma(-d_[q], e_[q], V[q]);             \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      _Pragma("unroll") for (int q = 0; q < NQ; q++) R[q] = __builtin_fma(d_[q], y_[q], e_[q]);              \
    }                                                                                                         \
    else                                                                                                      \
    {                                                                                                         \
      _Pragma("unroll") for (int q = 0; q < NQ; q++) R[q] = V[q] * y_[q];                                    \
    }                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  }

#define MFMA44(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

#define STRIP_STAMP(i)                                                                          \
  if constexpr (INSTR)                                                                           \
  {                                                                                              \
    if (g == 0 && tid == 0)                                                                      \
    {                                                                                            \
      a.dbg[(i)] = (long long) __builtin_readcyclecounter();                                     \
      if ((i) == 0 || (i) == 5) a.dbg[16 + (i)] = (long long) wall_clock64();                    \
    }                                                                                            \
  }

// Column statistics of W from the records the last reduce (or wstats) launch left, stat[kind][column k][step i] with
// kind = sum x^2, sum x, max x and step i = jp * 2 + e (16 bins of the column each).  Every consumer adds them in the
// same order, so every workgroup of every launch sees the same norms bit for bit.  Thread (k = tid >> 4, j = tid & 15)
// takes steps j, j + 16, ...: 16 lanes read 128 contiguous bytes.  First 256 threads of a workgroup.
struct StripStatRaw
{
  double v2[5], v1[5], vm[5];
};
// first half: the requests only (a caller with other loads to issue puts them behind these -- vmcnt retires in order --
// and consumes nothing in between)
__device__ __forceinline__ StripStatRaw strip_column_stats_load(const double* stat, int nStep, int tid)
{
  const int k = tid >> 4, j = tid & 15;
  StripStatRaw w; // nStep <= 72 (36 bin pairs): at most 5 steps per thread
#pragma unroll
  for (int u = 0; u < 5; u++)
  {
    const int i = min(j + 16 * u, nStep - 1);
    w.v2[u] = stat[(int64_t) k * nStep + i];
    w.v1[u] = stat[(int64_t) (16 + k) * nStep + i];
    w.vm[u] = stat[(int64_t) (32 + k) * nStep + i];
  }
  return w;
}
// second half: two barriers; out: nrmL[16] (1 when W is normalised), csL[16] column sums; sc = 48 doubles of scratch
__device__ __forceinline__ void strip_column_stats(const StripStatRaw& w, int nStep, int K, int wPend, double* sc, double* nrmL,
                                                   double* csL, int tid, bool active)
{
  if (active)
  {
    const int k = tid >> 4, j = tid & 15;
    double s2 = 0.0, s1 = 0.0, mx = 0.0;
#pragma unroll
    for (int u = 0; u < 5; u++)
      if (j + 16 * u < nStep)
      {
        s2 += w.v2[u];
        s1 += w.v1[u];
        mx = fmax(mx, w.vm[u]);
      }
    // the 16 lanes of a column: xor butterfly (the same tree in every lane)
#pragma unroll
    for (int sh = 1; sh < 16; sh <<= 1)
    {
      s2 += __shfl_xor(s2, sh);
      s1 += __shfl_xor(s1, sh);
      mx = fmax(mx, __shfl_xor(mx, sh));
    }
    if (j == 0)
    {
      sc[k] = s2;
      sc[16 + k] = s1;
      sc[32 + k] = mx;
    }
  }
  __syncthreads();
  if (active && tid < 16)
  {
    double gmax = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) gmax = fmax(gmax, sc[32 + k]);
    // alg/NMF.hpp:162 "if (W.maxCoeff() > epsilon) W.colwise().normalize()"; padded columns keep a divisor of one
    nrmL[tid] = (wPend && tid < K && gmax > kEpsilon) ? sqrt(sc[tid]) : 1.0;
    csL[tid] = sc[16 + tid];
  }
  __syncthreads();
}

// SIDE (round 4): every power-of-two transform has F = 32 j + 1 bins, so the last bin pair holds the Nyquist bin alone and
// costs every wavefront a whole pair slot (9 instead of 8 at fft 2048, 5 instead of 4 at fft 1024: the pairs are dealt
// round-robin and a wavefront without a pair works against zeros).  With SIDE the MFMA loops run over the pairs below it
// and bin F - 1 is a side column of the combine step: thread (frame, k) already holds H[frame][k]; sixteen lanes form
// Q_N = sum_k W'_N[k] H[frame][k] / nrm[k], the quotient V[frame][F - 1] / max(Q_N, eps) times W'_N[k] joins the numerator of
// the H update (alg/NMF.hpp:165-170), and the same with the NEW H gives the strip's share of bin F - 1's row of the next W
// update's numerator (:158-160), which leaves in the partial slots that bin has in the layout of the reduce launch.
// TILE (round 5): the W update is the bin-tiled launch of kernels_nmf_bintile.hip -- the column statistics come from ITS tile
// records, and a launch with doW == 2 leaves nothing of the next W update but bin F - 1's numerator partials (a.sideOut).
template <int NPW, int NQ, bool INSTR, bool SIDE = false, bool TILE = false>
__global__ __launch_bounds__(256) void nmf_strip_kernel(StripK a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: everything per bin pair is SALU arithmetic
  const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
  const int g = blockIdx.x, b = blockIdx.y;
  const int wlBytes = a.nPairs * 4096;
  unsigned char* wl = smem;                                 // pair jp at jp * 4096, written and read by wavefront jp & 3
  double* red = reinterpret_cast<double*>(smem + wlBytes);  // [4][kNQ][4][16]; the statistics scratch before that
  double* hn = red + 4 * kNQ * 4 * 16;                      // [kNQ * 4][16] the strip's new H
  double* nrmL = hn + kNQ * 64;                             // [16] column norms of W' (1 when W is normalised)
  double* csL = nrmL + 16;                                  // [16] column sums of W'
  unsigned char* zeroPage = reinterpret_cast<unsigned char*>(csL + 16); // 4 KB
  double* nyq = reinterpret_cast<double*>(zeroPage + 4096);             // [kNQ * 64] SIDE: bin F - 1's numerator terms per (frame, k)

  const double* Vb = a.V + (int64_t) b * a.strideV;
  const double* Wg = a.W + (int64_t) b * a.strideW;
  double* Hg = a.H + (int64_t) b * a.strideH;
  STRIP_STAMP(0)

  // quads dealt evenly (host-side quotient and remainder: a 64-bit division here is 300 instructions of cold code)
  const int qBeg = g * a.qBase + min(g, a.qRem), qEnd = qBeg + a.qBase + (g < a.qRem ? 1 : 0);
  const int jpLast = a.nPairs - 1 - (SIDE ? 1 : 0); // last pair of the MFMA loops
  const int nql = min(NQ, qEnd - qBeg);
  const int t0 = 4 * qBeg;

  // The strip's tiles of V stay in registers from their first use to their last, in the arrangement the W phase works
  // with: lane (x, blk, y) holds bins 32 jp + 8 blk + 2 x + {0, 1} of frame t0 + 4 q + y -- sixteen consecutive lanes read
  // 256 contiguous bytes of a frame (the H phase's arrangement, frames on x, made every four lanes a separate request:
  // 285 cycles of address processing per load instruction); the H phase takes its view through a lane permutation.
  // Out-of-range quads / pairs re-read a valid address and meet zero factors.
  d2 vh[NPW][NQ];
  // No load below sits behind a branch: the wait-count pass takes the path with the fewest requests at every join, and a
  // pessimistic vmcnt turns "kAhead pairs in flight" into "wait for everything".  A wavefront that has fewer pairs than
  // NPW (the pairs are dealt round-robin) repeats the last pair's addresses and works against a page of zeros instead.
  const double* vstrip = Vb + (int64_t) t0 * a.ldv; // uniform; the lane part fits 32 bits
  unsigned vofs[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) vofs[q] = (unsigned) ((4 * min(q, nql - 1) + y) * a.ldv + 8 * blk + 2 * x);
#define STRIP_LOAD_V(P)                                                                                       \
  {                                                                                                           \
    const double* vp_ = vstrip + 32 * min(wv + 4 * (P), jpLast);                                              \
    _Pragma("unroll") for (int q = 0; q < NQ; q++) vh[P][q] = *reinterpret_cast<const d2*>(vp_ + vofs[q]);   \
  }

  // A wavefront stages the rows of its own pairs: memory chunk n = lane + 64 i of the pair (row n >> 3, chunk n & 7)
  const unsigned wsrcLane = (unsigned) ((lane >> 3) * 128 + (lane & 7) * 16);
  unsigned wdst[4];
#pragma unroll
  for (int i = 0; i < 4; i++) wdst[i] = (unsigned) wl_off(8 * i + (lane >> 3), lane & 7);
#define STRIP_LOAD_W(DST, P)                                                                                  \
  {                                                                                                           \
    const unsigned char* wp_ = reinterpret_cast<const unsigned char*>(Wg) + min(wv + 4 * (P), jpLast) * 4096; \
    _Pragma("unroll") for (int i = 0; i < 4; i++)                                                             \
      DST[i] = *reinterpret_cast<const d2*>(wp_ + i * 1024 + wsrcLane);                                       \
  }
#define STRIP_STORE_W(SRC, P)                                                                                 \
  if (wv + 4 * (P) <= jpLast)                                                                                 \
  {                                                                                                           \
    unsigned char* wq_ = wl + (wv + 4 * (P)) * 4096;                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; i++) *reinterpret_cast<d2*>(wq_ + wdst[i]) = SRC[i];             \
  }
  // LDS base of a pair's operand rows: the page of zeros for a pair this wavefront does not have
#define STRIP_PAIR_BASE(P) ((wv + 4 * (P) <= jpLast) ? wl + (wv + 4 * (P)) * 4096 : zeroPage)
  d2 wt[2][4];
  d2 Hsv[NQ][2];
  STRIP_STAMP(10)
  [[maybe_unused]] StripStatRaw straw;
  [[maybe_unused]] bintile::TileStatRaw<1> traw;
  if constexpr (TILE) traw = bintile::tile_column_stats_load<1>(a.tileStat + (int64_t) b * 48 * a.nRec, a.nRec, tid);
  else straw = strip_column_stats_load(a.statIn + (int64_t) b * a.nBlk * kStatW, a.nBlk / 4, tid);
  __builtin_amdgcn_sched_barrier(0);
  STRIP_STAMP(11)
  {
    STRIP_LOAD_W(wt[0], 0)
    STRIP_LOAD_W(wt[1], 1)
    __builtin_amdgcn_sched_barrier(0);
    STRIP_STAMP(12)
    // the strip's rows of H in the B-operand arrangement of the H phase: H[4 y + m][t0 + 4 q + x]
#pragma unroll
    for (int q = 0; q < NQ; q++)
    {
      const double* hp = Hg + (int64_t) (t0 + 4 * min(q, nql - 1) + x) * 16 + 4 * y;
      Hsv[q][0] = *reinterpret_cast<const d2*>(hp);
      Hsv[q][1] = *reinterpret_cast<const d2*>(hp + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    STRIP_STAMP(13)
#pragma unroll
    for (int p = 0; p < kEarly && p < NPW; p++) STRIP_LOAD_V(p)
  }
  // the strip's old H once more, one value per thread of the combine step (o = tid, tid + 256)
  double hold[2];
#pragma unroll
  for (int i = 0; i < 2; i++) hold[i] = Hg[(int64_t) t0 * 16 + min(tid + 256 * i, nql * 64 - 1)];
  // side column: bin F - 1 of this thread's frames, and the bin's row of W' (column tid & 15)
  double vN[2] = {0.0, 0.0}, wN = 0.0;
  if constexpr (SIDE)
  {
#pragma unroll
    for (int i = 0; i < 2; i++) vN[i] = Vb[(int64_t) (t0 + (min(tid + 256 * i, nql * 64 - 1) >> 4)) * a.ldv + a.F - 1];
    wN = Wg[(int64_t) (a.F - 1) * 16 + (tid & 15)];
  }
  __builtin_amdgcn_sched_barrier(0);
  STRIP_STAMP(9)
  *reinterpret_cast<d2*>(zeroPage + tid * 16) = d2{0.0, 0.0};

  if constexpr (TILE) bintile::tile_column_stats<1>(traw, a.nRec, a.K, a.wPend, red, nrmL, csL, tid);
  else strip_column_stats(straw, a.nBlk / 4, a.K, a.wPend, red, nrmL, csL, tid, true);
  STRIP_STAMP(15)
  if (g == 0 && tid < 16) a.nrm[(int64_t) b * 16 + tid] = nrmL[tid];
  if (!a.doH && !a.doW) return;
#pragma unroll
  for (int p = kEarly; p < kAhead && p < NPW; p++) STRIP_LOAD_V(p)
  STRIP_STAMP(1)

  // per-lane LDS offsets of the two operand read patterns, [e][half]: A rows by x with columns 4 y + m, B rows by y with
  // columns 4 x + m
  int offA[2][2], offB[2][2];
#pragma unroll
  for (int e = 0; e < 2; e++)
#pragma unroll
    for (int h = 0; h < 2; h++)
    {
      const int gl = ((x ^ y) << 2) | (((e << 1) | h) ^ blk);
      offA[e][h] = (4 * blk + x) * 256 + gl * 16;
      offB[e][h] = (4 * blk + y) * 256 + gl * 16;
    }
  double rn[4]; // 1 / norm of the columns 4 y + m this lane feeds into the first product
#pragma unroll
  for (int m = 0; m < 4; m++) rn[m] = 1.0 / nrmL[4 * y + m];

  // SIDE: this thread's term of bin F - 1's row of the next W update's numerator, (V / max(W'_N (H / nrm), eps))[frame] H[frame][k]
  // with the H that the W update will see (the 16 lanes of a frame form the product; frames past the strip give 0)
  // (the quotient as the tile loops take it: reciprocal seed + one Newton step; the norms enter as reciprocals like rn[])
  auto side_quot = [](double v, double q) -> double {
    const double d = fmax(q, kEpsilon);
    double yq = __builtin_amdgcn_rcp(d);
    yq = __builtin_fma(yq, __builtin_fma(-d, yq, 1.0), yq);
    return v * yq;
  };
  const double rnk = SIDE ? 1.0 / nrmL[tid & 15] : 1.0; // this thread's column in the combine step
  auto side_share = [&](int i, double hv, bool valid, int) -> double {
    double qn = valid ? wN * (hv * rnk) : 0.0;
#pragma unroll
    for (int sh = 1; sh < 16; sh <<= 1) qn += __shfl_xor(qn, sh);
    return valid ? side_quot(vN[i], qn) * hv : 0.0;
  };
  double wdenAcc = 0.0;
  const int srcLane4 = 4 * (y + 4 * blk + 16 * x);
  if (a.doH)
  {
    // ---- H phase: the strip's frames against every bin (alg/NMF.hpp:165-170).  W stays as it is in memory,
    // W' = W diag(nrm): Q = W' (H / nrm), and the numerator and the column sums are divided by nrm at the end.
    double Hs[NQ][4], out[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        Hs[q][m] = q < nql ? Hsv[q][m >> 1][m & 1] * rn[m] : 0.0;
        out[q][m] = 0.0;
      }
    d2 wa[2][2], wb[2][2]; // operand rows of the current pair, [e][half]; the next pair's are read a pair ahead
    STRIP_STORE_W(wt[0], 0)
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
      for (int h = 0; h < 2; h++)
      {
        wa[e][h] = *reinterpret_cast<const d2*>(STRIP_PAIR_BASE(0) + offA[e][h]);
        wb[e][h] = *reinterpret_cast<const d2*>(STRIP_PAIR_BASE(0) + offB[e][h]);
      }
#pragma unroll
    for (int p = 0; p < NPW; p++)
    {
      __builtin_amdgcn_sched_barrier(0); // keep a pair's tiles inside the pair (register pressure)
      {
        // next pair's rows into the LDS, the pair after that requested, V kAhead pairs ahead; the next pair's operand
        // rows replace this pair's as soon as the product that reads them has been issued
        const bool more = p + 1 < NPW;
        const unsigned char* wj = STRIP_PAIR_BASE(p + 1);
        if (more)
        {
          STRIP_STORE_W(wt[(p + 1) & 1], p + 1)
          if (p + 2 < NPW) STRIP_LOAD_W(wt[p & 1], p + 2)
          if (p + kAhead < NPW) STRIP_LOAD_V(p + kAhead)
        }
        if (p == 2) { STRIP_STAMP(6) }
        // this phase sees the tiles transposed: lane (x, blk, y) needs what lane (y, blk, x) holds
        double vt[2][NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
          for (int e = 0; e < 2; e++)
          {
            const long long bits = __double_as_longlong(vh[p][q][e]);
            const int lo = __builtin_amdgcn_ds_bpermute(srcLane4, (int) (bits & 0xffffffff));
            const int hi = __builtin_amdgcn_ds_bpermute(srcLane4, (int) (bits >> 32));
            vt[e][q] = __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
          }
        double Q[2][NQ];
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
          for (int q = 0; q < NQ; q++) Q[e][q] = 0.0;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int e = 0; e < 2; e++)
#pragma unroll
            for (int q = 0; q < NQ; q++) Q[e][q] = MFMA44(wa[e][m >> 1][m & 1], Hs[q][m], Q[e][q]);
        __builtin_amdgcn_sched_barrier(0);
        if (p == 2) { STRIP_STAMP(7) }
        if (more)
        {
#pragma unroll
          for (int e = 0; e < 2; e++)
#pragma unroll
            for (int h = 0; h < 2; h++) wa[e][h] = *reinterpret_cast<const d2*>(wj + offA[e][h]);
        }
        double R[2][NQ];
#pragma unroll
        for (int e = 0; e < 2; e++) STRIP_QUOT(R[e], vt[e], Q[e])
        if (p == 2) { STRIP_STAMP(8) }
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
          for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int m = 0; m < 4; m++) out[q][m] = MFMA44(R[e][q], wb[e][m >> 1][m & 1], out[q][m]);
        __builtin_amdgcn_sched_barrier(0);
        if (p == 2) { STRIP_STAMP(14) }
        if (more)
        {
#pragma unroll
          for (int e = 0; e < 2; e++)
#pragma unroll
            for (int h = 0; h < 2; h++) wb[e][h] = *reinterpret_cast<const d2*>(wj + offB[e][h]);
        }
      }
    }
    STRIP_STAMP(2)
    // blocks of a wavefront (bins), then the four wavefronts, in fixed order; lane (x, y) of block 0 holds frame y,
    // column 4 x + m
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        double v = out[q][m];
        v += dppmov<0x124>(v); // row_ror:4
        v += dppmov<0x128>(v); // row_ror:8
        if (blk == 0) red[((wv * kNQ + q) * 4 + m) * 16 + x + 4 * y] = v;
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++)
    {
      const int o = tid + 256 * i;
      if (o < NQ * 64)
      {
        const int q = o >> 6, rem = o & 63, yy = rem >> 4, k = rem & 15;
        double hv = 0.0;
        if (q < nql)
        {
          const int ix = (q * 4 + (k & 3)) * 16 + (k >> 2) + 4 * yy;
          double s = ((red[ix] + red[kNQ * 64 + ix]) + red[2 * kNQ * 64 + ix]) + red[3 * kNQ * 64 + ix];
          if constexpr (SIDE)
          {
            // bin F - 1 joins the numerator behind the MFMA bins (the 16 lanes of a frame: the same butterfly in each)
            double qn = wN * (hold[i] * rnk);
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) qn += __shfl_xor(qn, sh);
            s += side_quot(vN[i], qn) * wN;
          }
          // :170  H * (W^T (V / V2)) / max(W^T 1, eps) with W = W' / nrm
          hv = hold[i] * (s / nrmL[k]) / fmax(csL[k] / nrmL[k], kEpsilon);
          Hg[(int64_t) t0 * 16 + o] = hv;
        }
        hn[o] = hv;
        if constexpr (SIDE) nyq[o] = side_share(i, hv, q < nql, k);
      }
    }
  }
  else
  {
    // no H phase in this launch: its side jobs -- the rows of W into the LDS, the strip's V -- happen here
#pragma unroll
    for (int p = 0; p < NPW; p++)
    {
      if (p >= 2) STRIP_LOAD_W(wt[p & 1], p)
      STRIP_STORE_W(wt[p & 1], p)
      if (p >= kAhead) STRIP_LOAD_V(p)
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
      if (tid + 256 * i < NQ * 64)
      {
        const bool valid = ((tid + 256 * i) >> 6) < nql;
        hn[tid + 256 * i] = valid ? hold[i] : 0.0;
        if constexpr (SIDE) nyq[tid + 256 * i] = side_share(i, hold[i], valid, tid & 15);
      }
  }
  __syncthreads();
  STRIP_STAMP(3)
  if (a.doW)
  {
    // :160 row sums of H, this strip's share: wavefront 0, lane (part = lane >> 4, k = lane & 15) adds the frames part, part + 4,
    // ..., the four parts are then added in a fixed order (a single lane per column walked 24 dependent LDS reads here)
    if (wv == 0)
    {
      const int part = lane >> 4, k = lane & 15;
      double d = 0.0;
      for (int t = part; t < 4 * nql; t += 4) d += hn[t * 16 + k];
      d += __shfl_xor(d, 16);
      d += __shfl_xor(d, 32);
      wdenAcc = d;
    }
    if constexpr (SIDE)
    {
      double sN = 0.0;
      if (wv == 1) // (beside the row sums, on another SIMD)
      {
        const int part = lane >> 4, k = lane & 15;
        for (int t = part; t < 4 * nql; t += 4) sN += nyq[t * 16 + k];
        sN += __shfl_xor(sN, 16);
        sN += __shfl_xor(sN, 32);
      }
      if (wv == 1 && lane < 16)
      {
        const int tidk = lane;
        // bin F - 1's share: its 16 values sit in blocks m = k & 3 of step (last pair, e = 0) at lane x = k >> 2
        double* dst = a.part + ((int64_t) b * a.nBlk * a.nWG + 4 * g) * 64 + ((int64_t) ((a.nPairs - 1) * 2) * a.nWG * 4 + (tidk & 3)) * 64 + (tidk >> 2);
        if constexpr (TILE) dst = a.sideOut + ((int64_t) b * a.nWG + g) * 16 + tidk; // (one partial per workgroup, strip order)
        if (kWriteThrough) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(dst), "v"(sN) : "memory");
        else *dst = sN;
      }
    }
    if constexpr (TILE) return; // (the rest of the W update is the bin-tiled launch's)
    // ---- W phase: the strip's share of the next W update's numerator (alg/NMF.hpp:158-160), on the tiles the H phase
    // used (no second read of V).
    double Ha[NQ][4], Hb[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int m = 0; m < 4; m++)
      {
        Ha[q][m] = hn[(4 * q + x) * 16 + 4 * y + m] * rn[m]; // first product: (H / nrm)[4 y + m][frame x]
        Hb[q][m] = hn[(4 * q + y) * 16 + 4 * x + m];         // second product: H[4 x + m][frame y]
      }
    double* P = a.part + ((int64_t) b * a.nBlk * a.nWG + 4 * g) * 64 + lane;
    d2 wa[2][2];
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
      for (int h = 0; h < 2; h++) wa[e][h] = *reinterpret_cast<const d2*>(STRIP_PAIR_BASE(0) + offA[e][h]);
#pragma unroll
    for (int p = 0; p < NPW; p++)
    {
      const int jp = wv + 4 * p;
      __builtin_amdgcn_sched_barrier(0);
      {
        double vt[2][NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
          for (int e = 0; e < 2; e++) vt[e][q] = vh[p][q][e];
        double Q[2][NQ];
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
          for (int q = 0; q < NQ; q++) Q[e][q] = 0.0;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int e = 0; e < 2; e++)
#pragma unroll
            for (int q = 0; q < NQ; q++) Q[e][q] = MFMA44(Ha[q][m], wa[e][m >> 1][m & 1], Q[e][q]);
        __builtin_amdgcn_sched_barrier(0);
        if (p + 1 < NPW)
        {
          const unsigned char* wj = STRIP_PAIR_BASE(p + 1);
#pragma unroll
          for (int e = 0; e < 2; e++)
#pragma unroll
            for (int h = 0; h < 2; h++) wa[e][h] = *reinterpret_cast<const d2*>(wj + offA[e][h]);
        }
        double R[2][NQ];
#pragma unroll
        for (int e = 0; e < 2; e++) STRIP_QUOT(R[e], vt[e], Q[e])
        double num[2][4];
#pragma unroll
        for (int e = 0; e < 2; e++)
#pragma unroll
          for (int m = 0; m < 4; m++) num[e][m] = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
          for (int e = 0; e < 2; e++)
#pragma unroll
            for (int m = 0; m < 4; m++) num[e][m] = MFMA44(R[e][q], Hb[q][m], num[e][m]);
        __builtin_amdgcn_sched_barrier(0);
        // complete as soon as the pair's last quad is in: out it goes (512 contiguous bytes per store instruction)
        if (jp <= jpLast)
        {
#pragma unroll
          for (int e = 0; e < 2; e++)
#pragma unroll
            for (int m = 0; m < 4; m++)
            {
              // write-through (sc0 sc1): the partials leave the L2 while the phase runs instead of at the end of the
              // launch, where 35 MB of dirty lines would stand between this kernel and the reduce launch
              double* dst = P + ((int64_t) (jp * 2 + e) * a.nWG * 4 + m) * 64;
              if (kWriteThrough) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(dst), "v"(num[e][m]) : "memory");
              else *dst = num[e][m];
            }
        }
      }
    }
    STRIP_STAMP(4)
    if (tid < 16) a.dpart[((int64_t) b * a.nWG + g) * 16 + tid] = wdenAcc;
  }
  STRIP_STAMP(5)
}

// Element `lane` of block r = (jp * 2 + e) * 4 + m  <->  f = 32 jp + 8 blk + 2 y + e, k = 4 x + m.
// Column statistics of the 64 values a wavefront holds (one block): [sum x^2, sum x, max x] of the four columns,
// 16 rows each, added over lane bits 2..5 in fixed order.
__device__ __forceinline__ void strip_block_stats(double v, double* stat, int nStep, int r, int lane)
{
  double s2 = v * v, s1 = v, mx = v;
#pragma unroll
  for (int sh = 4; sh < 64; sh <<= 1)
  {
    s2 += __shfl_xor(s2, sh);
    s1 += __shfl_xor(s1, sh);
    mx = fmax(mx, __shfl_xor(mx, sh));
  }
  if (lane < 4)
  {
    const int k = 4 * lane + (r & 3), i = r >> 2; // stat[kind][k][step]
    stat[(int64_t) k * nStep + i] = s2;
    stat[(int64_t) (16 + k) * nStep + i] = s1;
    stat[(int64_t) (32 + k) * nStep + i] = mx;
  }
}

// W'[f][k] <- (W'[f][k] / nrm[k]) * (sum of the numerator partials) / max(sum of the denominator partials, eps)
// (alg/NMF.hpp:161; the division by nrm is :162 of the previous iteration), and the statistics of the new W'.
// One workgroup per column block r (64 values), eight wavefronts taking every eighth strip.  Tried and slower on this
// part: 32 loads in flight per thread (13.4 us against 11.0), sixteen wavefronts per workgroup (15.1), one workgroup per
// bin step reading its strips' 2 KB runs as contiguous streams (23.9: a quarter of the CUs cannot pull the 35 MB).
constexpr int kRedWaves = 8, kRedU = 16;
__global__ __launch_bounds__(64 * kRedWaves) void nmf_strip_reduce_kernel(StripK a)
{
  __shared__ double red[kRedWaves][64];
  __shared__ double sc[768];
  __shared__ double nrmL[16], csL[16], den[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y, r = blockIdx.x;
  const double* P = a.part + (((int64_t) b * a.nBlk + (r & ~3)) * a.nWG + (r & 3)) * 64 + lane;
  // every request of the launch goes out before the first sum (no load behind a branch, nothing consumed in between):
  // the statistics records, the denominator partials, the numerator partials
  const StripStatRaw straw = strip_column_stats_load(a.statIn + (int64_t) b * a.nBlk * kStatW, a.nBlk / 4, tid & 255);
  const int dk = tid & 15, dgrp = tid >> 4; // 32 groups
  const double* D = a.dpart + (int64_t) b * a.nWG * 16 + dk;
  double dv[8];
#pragma unroll
  for (int u = 0; u < 8; u++) dv[u] = D[(int64_t) min(dgrp + 32 * u, a.nWG - 1) * 16];
  // the old value of the element this thread will write (first wavefront), requested now rather than after the barriers
  const int ex = lane & 3, eblk = (lane >> 2) & 3, ey = lane >> 4;
  const int em = r & 3, ee = (r >> 2) & 1, ejp = r >> 3;
  const int ef = 32 * ejp + 8 * eblk + 2 * ey + ee, ek = 4 * ex + em;
  double* wp = a.W + (int64_t) b * a.strideW + (int64_t) min(ef, a.F - 1) * 16 + ek;
  const double wold = *wp;
  double v[kRedU];
#pragma unroll
  for (int u = 0; u < kRedU; u++) v[u] = P[(int64_t) min(wv + kRedWaves * u, a.nWG - 1) * 256];   // (non-temporal loads measured 0.5 - 1 % slower here, round 6)
  __builtin_amdgcn_sched_barrier(0);
  double dsum = 0.0;
#pragma unroll
  for (int u = 0; u < 8; u++) dsum += dgrp + 32 * u < a.nWG ? dv[u] : 0.0;
  for (int p = dgrp + 256; p < a.nWG; p += 32) dsum += D[(int64_t) p * 16]; // more than 256 strips: long buffers
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < kRedU; u++) s += wv + kRedWaves * u < a.nWG ? v[u] : 0.0;
  for (int p0 = wv + kRedWaves * kRedU; p0 < a.nWG; p0 += kRedWaves * kRedU)
  {
#pragma unroll
    for (int u = 0; u < kRedU; u++) v[u] = P[(int64_t) min(p0 + kRedWaves * u, a.nWG - 1) * 256];
#pragma unroll
    for (int u = 0; u < kRedU; u++) s += p0 + kRedWaves * u < a.nWG ? v[u] : 0.0;
  }
  red[wv][lane] = s;
  strip_column_stats(straw, a.nBlk / 4, a.K, a.wPend, sc, nrmL, csL, tid, tid < 256);
  sc[(tid >> 4) * 16 + (tid & 15)] = dsum;
  __syncthreads();
  if (tid < 16)
  {
    double d = 0.0;
    for (int i = 0; i < 32; i++) d += sc[i * 16 + tid];
    den[tid] = fmax(d, kEpsilon);
  }
  __syncthreads();
  if (wv == 0)
  {
    double tot = red[0][lane];
#pragma unroll
    for (int w = 1; w < kRedWaves; w++) tot += red[w][lane];
    double wnew = 0.0;
    if (ef < a.F && ek < a.K)
    {
      wnew = (wold / nrmL[ek]) * tot / den[ek];
      *wp = wnew;
    }
    strip_block_stats(wnew, a.statOut + (int64_t) b * a.nBlk * kStatW, a.nBlk / 4, r, lane);
  }
}

// statistics records of a W that something else wrote (factor initialisation)
__global__ __launch_bounds__(64) void nmf_strip_wstats_kernel(StripK a)
{
  const int lane = threadIdx.x, b = blockIdx.y, r = blockIdx.x;
  const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
  const int m = r & 3, e = (r >> 2) & 1, jp = r >> 3;
  const int f = 32 * jp + 8 * blk + 2 * y + e, k = 4 * x + m;
  double v = 0.0;
  if (f < a.F && k < a.K) v = a.W[(int64_t) b * a.strideW + (int64_t) f * 16 + k];
  strip_block_stats(v, a.statOut + (int64_t) b * a.nBlk * kStatW, a.nBlk / 4, r, lane);
}

#ifdef FLUHIP_AB_SWITCHES // (measured slower than the fused form: compiled into the A/B build only)
// ---------------------------------------------------------------------------------------------------------------
// W update of the frame-strip schedule as its OWN launch, bin strips (round 4).
//
// The fused form above hands the W update's numerator from 256 frame strips to the reduce launch as 256 partials of the
// whole F x 16 matrix (34 MB written write-through and read back per iteration at config 2, 11.4 us of reduce launch
// behind a 10 us W phase).  Here the decomposition of the W update is turned by 90 degrees against the H update's:
// a workgroup owns PP bin pairs (96 bins: its rows of W' stay in registers as MFMA operands) and one of nSlices slices
// of the frames, whose rows of the NEW H and tiles of V it streams; the partial numerators of a bin strip are nSlices x
// 12 KB (3 MB in all), and the LAST workgroup of a strip to arrive adds them in slice order, forms
// W' <- (W' / nrm) * num / max(den, eps) (alg/NMF.hpp:158-161) and leaves the statistics records of its pairs -- no
// reduce launch.  One iteration = this launch + the strip kernel with its H phase only (which then needs no W phase,
// no partial stores, no second use of its V registers).  V is read twice per iteration instead of once (from `mag`
// both times, in the W-phase arrangement: sixteen lanes read 256 contiguous bytes of a frame).
//
// Tiles and operand roles are those of the strip kernel's W phase: a tile is 16 bins x 4 frames, lane (x, blk, y);
//   Q[t][f]   = sum_k (H / nrm)[k][t] W'[f][k]     A = H rows (frame x, k = 4 y + m), B = W rows   -> D lane = Q[t_y][f_x]
//   num[f][k] += sum_t (V / Q)[t][f] H[k][t]        A = the quotient as it lies,       B = H rows (frame y, k = 4 x + m)
// Determinism: a wavefront adds its quads in order, the four wavefronts of a workgroup and then the slices of a strip are
// added in fixed order by whoever arrives last -- the arrival order decides WHO adds, never in which order.
// Cross-workgroup visibility (MI355X guide, "inter-workgroup visibility"): partials leave write-through (sc0 sc1), every
// thread waits for its stores, one agent-scope ticket per strip; the last arriver reads the partials with sc1 loads.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kBinPP = 3; // bin pairs per workgroup

struct BinK
{
  const double* V;
  int64_t strideV;
  int ldv;
  double* W;
  int64_t strideW;
  const double* H;
  int64_t strideH;
  double* part;   // [B][nStrips][nSlices][PP * 8][64] numerator partials
  double* dpart;  // [B][nStrips][nSlices][16] row sums of H over the slice
  int* ticket;    // [B][nStrips], zero between launches
  const double* statIn;
  double* statOut;
  int F, T, K, nPairs, nBlk, nq, nStrips, nSlices, qBase, qRem, wPend;
  long long* dbg; // FLUHIP_STRIP_INSTR: 100 MHz stamps of workgroup 0 ([0..7]) and of strip 0's last arriver ([8..15])
};

// a load that bypasses this CU's L1 (global_load ... sc1): the compiler's own relaxed agent-scope load, so that its wait
// counters know about it (an inline-asm load returns into registers the compiler believes free)
__device__ __forceinline__ double load_sc1(const double* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define BIN_STAMP(i)                                                                                       \
  if constexpr (INSTR)                                                                                   \
  {                                                                                                      \
    if (tid == 0 && blockIdx.y == 0)                                                                     \
    {                                                                                                    \
      if (blockIdx.x == 0 && (i) < 8) a.dbg[(i)] = (long long) wall_clock64();                          \
      if ((i) >= 8 && strip == 0) a.dbg[(i)] = (long long) wall_clock64();                               \
    }                                                                                                    \
  }

template <int PP, bool INSTR = false>
__global__ __launch_bounds__(256) void nmf_binstrip_kernel(BinK a)
{
  __shared__ double red[4][PP * 8][64]; // the wavefronts' numerators; the statistics scratch before that
  __shared__ double nrmL[16], csL[16], denW[4][16], denL[16];
  __shared__ int lastFlag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
  const int b = blockIdx.y;
  const int strip = blockIdx.x % a.nStrips, slice = blockIdx.x / a.nStrips;
  const int jp0 = strip * PP, jpLast = a.nPairs - 1;
  const double* Vb = a.V + (int64_t) b * a.strideV;
  double* Wg = a.W + (int64_t) b * a.strideW;
  const double* Hg = a.H + (int64_t) b * a.strideH;

  BIN_STAMP(0)
  // the requests of the prologue first: statistics records, then this lane's rows of W' (B operand of the first product:
  // W'[f][4 y + m], f = 32 jp + 8 blk + 2 x + e -- 32 contiguous bytes per (pair, e))
  const StripStatRaw straw = strip_column_stats_load(a.statIn + (int64_t) b * a.nBlk * kStatW, a.nBlk / 4, tid);
  d2 wB[PP][2][2];
#pragma unroll
  for (int p = 0; p < PP; p++)
#pragma unroll
    for (int e = 0; e < 2; e++)
    {
      const double* wp = Wg + (int64_t) (32 * min(jp0 + p, jpLast) + 8 * blk + 2 * x + e) * 16 + 4 * y;
      wB[p][e][0] = *reinterpret_cast<const d2*>(wp);
      wB[p][e][1] = *reinterpret_cast<const d2*>(wp + 2);
    }
  // this wavefront's quads of the slice: q = qBeg + wv, + 4, ...
  const int qBeg = slice * a.qBase + min(slice, a.qRem), qEnd = qBeg + a.qBase + (slice < a.qRem ? 1 : 0);
  const int nMine = (qEnd - qBeg - wv + 3) >> 2; // (may be <= 0)
  const unsigned vLane = (unsigned) (y * a.ldv + 8 * blk + 2 * x);
  struct Ops
  {
    d2 ha[2], hb[2], v[PP];
  };
  auto request = [&](int q, Ops& o) {
    const int t0 = 4 * q;
    const double* ha = Hg + (int64_t) (t0 + x) * 16 + 4 * y;
    const double* hb = Hg + (int64_t) (t0 + y) * 16 + 4 * x;
    o.ha[0] = *reinterpret_cast<const d2*>(ha); o.ha[1] = *reinterpret_cast<const d2*>(ha + 2);
    o.hb[0] = *reinterpret_cast<const d2*>(hb); o.hb[1] = *reinterpret_cast<const d2*>(hb + 2);
    const double* vp = Vb + (int64_t) t0 * a.ldv + vLane;
#pragma unroll
    for (int p = 0; p < PP; p++) o.v[p] = *reinterpret_cast<const d2*>(vp + 32 * min(jp0 + p, jpLast));
  };
  // Operands are requested kDepth - 1 quads ahead: one wavefront per SIMD (a round of workgroups is one per CU) has nobody to
  // hide a memory round trip behind, and a quad is ~1 100 cycles of work against ~2 000 of latency (first version, one quad
  // ahead: 25 us for config 2's launch, most of it waiting).
  constexpr int kDepth = 3;
  Ops ops[kDepth];
  // (rows of H and V past the buffer's frames are allocated and zero: a quad index past the end re-reads the last one)
  const int qLastValid = max(qEnd - 1, qBeg);
#pragma unroll
  for (int d = 0; d < kDepth - 1; d++) request(min(qBeg + wv + 4 * d, qLastValid), ops[d]);
  __builtin_amdgcn_sched_barrier(0);

  strip_column_stats(straw, a.nBlk / 4, a.K, a.wPend, &red[0][0][0], nrmL, csL, tid, true);
  double rn[4];
#pragma unroll
  for (int m = 0; m < 4; m++) rn[m] = 1.0 / nrmL[4 * y + m];
  // pairs past the last one: zero rows (their numerators are never stored)
#pragma unroll
  for (int p = 0; p < PP; p++)
    if (jp0 + p > jpLast)
#pragma unroll
      for (int e = 0; e < 2; e++) wB[p][e][0] = wB[p][e][1] = d2{0.0, 0.0};

  double num[PP][2][4];
#pragma unroll
  for (int p = 0; p < PP; p++)
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
      for (int m = 0; m < 4; m++) num[p][e][m] = 0.0;
  double dacc[4] = {0.0, 0.0, 0.0, 0.0};

  BIN_STAMP(1)
  constexpr int NQ = PP * 2; // quotients per quad (the staging macro's width)
  auto compute = [&](const Ops& cur) {
    double Ha[4], Hb[4];
#pragma unroll
    for (int m = 0; m < 4; m++)
    {
      Ha[m] = cur.ha[m >> 1][m & 1] * rn[m];
      Hb[m] = cur.hb[m >> 1][m & 1];
      dacc[m] += Hb[m];
    }
    double Q[NQ], Vt[NQ], R[NQ];
#pragma unroll
    for (int j = 0; j < NQ; j++) { Q[j] = 0.0; Vt[j] = cur.v[j >> 1][j & 1]; }
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int j = 0; j < NQ; j++) Q[j] = MFMA44(Ha[m], wB[j >> 1][j & 1][m >> 1][m & 1], Q[j]);
    __builtin_amdgcn_sched_barrier(0);
    STRIP_QUOT(R, Vt, Q)
#pragma unroll
    for (int j = 0; j < NQ; j++)
#pragma unroll
      for (int m = 0; m < 4; m++) num[j >> 1][j & 1][m] = MFMA44(R[j], Hb[m], num[j >> 1][j & 1][m]);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int i = 0; i < nMine; i += kDepth)
  {
#pragma unroll
    for (int d = 0; d < kDepth; d++)
    {
      // (the request is unconditional -- a clamped index -- so that no load sits behind a branch; the work is)
      request(min(qBeg + wv + 4 * (i + d + kDepth - 1), qLastValid), ops[(d + kDepth - 1) % kDepth]);
      __builtin_amdgcn_sched_barrier(0);
      if (i + d < nMine) compute(ops[d]);
    }
  }

  BIN_STAMP(2)
  // ---- the four wavefronts, in fixed order ----------------------------------------------------------------------------
  // row sums of H over the wavefront's frames: lane (x, y) holds frames y of every quad, columns 4 x + m
#pragma unroll
  for (int m = 0; m < 4; m++)
  {
    double v = dacc[m];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < 4) denW[wv][4 * lane + m] = v;
  }
  __syncthreads(); // (the statistics scratch in red is free)
#pragma unroll
  for (int p = 0; p < PP; p++)
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
      for (int m = 0; m < 4; m++) red[wv][(p * 2 + e) * 4 + m][lane] = num[p][e][m];
  __syncthreads();
  const int64_t slot = ((int64_t) b * a.nStrips + strip) * a.nSlices + slice;
  double* P = a.part + slot * (PP * 8 * 64);
#pragma unroll
  for (int u = 0; u < PP * 2; u++)
  {
    const int r = wv + 4 * u; // block r of the strip, this thread's lane of it
    const double s4 = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
    double* dst = P + r * 64 + lane;
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(dst), "v"(s4) : "memory");
  }
  if (tid < 16)
  {
    const double d = ((denW[0][tid] + denW[1][tid]) + denW[2][tid]) + denW[3][tid];
    double* dst = a.dpart + slot * 16 + tid;
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(dst), "v"(d) : "memory");
  }
  BIN_STAMP(3)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  BIN_STAMP(4)
  if (tid == 0)
  {
    const int old = __hip_atomic_fetch_add(a.ticket + b * a.nStrips + strip, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lastFlag = old == a.nSlices - 1;
  }
  __syncthreads();
  BIN_STAMP(5)
  if (!lastFlag) return;
  BIN_STAMP(8)

  // ---- the last workgroup of the strip: the slices in a fixed order, the update, the statistics of the new W' -----------
  // Everything below is latency: the partials come from other XCDs' stores (sc1 loads, ~1.5 us a round trip), so every
  // round trip carries as many independent loads as the wait counter holds, 16 bytes each.
  const int64_t slot0 = ((int64_t) b * a.nStrips + strip) * a.nSlices;
  // the old values of the elements this thread will write, requested ahead of the partials (one round trip less at the end)
  double wold[PP * 2];
#pragma unroll
  for (int u = 0; u < PP * 2; u++)
  {
    const int rl = wv + 4 * u;
    const int f = 32 * min(jp0 + (rl >> 3), jpLast) + 8 * blk + 2 * y + ((rl >> 2) & 1), k = 4 * x + (rl & 3);
    wold[u] = Wg[(int64_t) min(f, a.F - 1) * 16 + k];
  }
  {
    // row sums of H: thread (k = tid & 15, g = tid >> 4) adds the slices g, g + 16, ... in that order; the 16 group sums
    // are then added in group order
    const int k = tid & 15, g = tid >> 4;
    double d = 0.0;
    for (int j0 = g; j0 < a.nSlices; j0 += 64)
    {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = load_sc1(a.dpart + (slot0 + min(j0 + 16 * u, a.nSlices - 1)) * 16 + k);
#pragma unroll
      for (int u = 0; u < 4; u++) d += j0 + 16 * u < a.nSlices ? v[u] : 0.0;
    }
    red[0][0][tid] = d; // [g][k] as tid = g * 16 + k (this workgroup's own partials left the array before the ticket)
  }
  __syncthreads();
  if (tid < 16)
  {
    double d = 0.0;
#pragma unroll
    for (int g = 0; g < 16; g++) d += red[0][0][g * 16 + tid];
    denL[tid] = fmax(d, kEpsilon);
  }
  BIN_STAMP(9)
  // numerators: the strip's PP * 8 blocks of 64 doubles = PP * 256 pairs of doubles per slice; thread tid takes the pairs
  // tid, tid + 256, ... of every slice (16-byte loads), the slices in ascending order
  const double* P0 = a.part + slot0 * (PP * 8 * 64);
  d2 tot[PP];
#pragma unroll
  for (int u = 0; u < PP; u++) tot[u] = d2{0.0, 0.0};
  constexpr int JB = 24 / PP < 1 ? 1 : 24 / PP; // slices per round trip: 2 * PP * JB eight-byte loads in flight (<= 48)
  for (int j0 = 0; j0 < a.nSlices; j0 += JB)
  {
    d2 v[JB][PP];
#pragma unroll
    for (int jj = 0; jj < JB; jj++)
#pragma unroll
      for (int u = 0; u < PP; u++)
      {
        const double* src = P0 + (int64_t) min(j0 + jj, a.nSlices - 1) * (PP * 8 * 64) + 2 * (tid + 256 * u);
        v[jj][u][0] = load_sc1(src);
        v[jj][u][1] = load_sc1(src + 1);
      }
#pragma unroll
    for (int jj = 0; jj < JB; jj++)
#pragma unroll
      for (int u = 0; u < PP; u++)
        if (j0 + jj < a.nSlices) { tot[u][0] += v[jj][u][0]; tot[u][1] += v[jj][u][1]; }
  }
  BIN_STAMP(10)
  __syncthreads(); // denL complete, red free again
  double* totL = &red[0][0][0]; // [PP * 8][64]
#pragma unroll
  for (int u = 0; u < PP; u++) *reinterpret_cast<d2*>(totL + 2 * (tid + 256 * u)) = tot[u];
  __syncthreads();
  const int nStep = a.nBlk / 4;
#pragma unroll
  for (int u = 0; u < PP * 2; u++)
  {
    const int rl = wv + 4 * u;                       // block of the strip: (pair p, e, m)
    const int p = rl >> 3, e = (rl >> 2) & 1, m = rl & 3;
    const int jp = jp0 + p;
    if (jp > jpLast) continue;                       // (wave-uniform)
    const int f = 32 * jp + 8 * blk + 2 * y + e, k = 4 * x + m;
    double wnew = 0.0;
    if (f < a.F && k < a.K)
    {
      wnew = (wold[u] / nrmL[k]) * totL[rl * 64 + lane] / denL[k];
      Wg[(int64_t) f * 16 + k] = wnew;
    }
    strip_block_stats(wnew, a.statOut + (int64_t) b * a.nBlk * kStatW, nStep, (jp * 2 + e) * 4 + m, lane);
  }
  BIN_STAMP(11)
  if (tid == 0) a.ticket[b * a.nStrips + strip] = 0; // for the next launch (kernel boundaries order it)
}

#endif // FLUHIP_AB_SWITCHES

int strip_pairs(int F) { return (F + 31) / 32; }

} // namespace strip
using namespace strip;

// LDS of a strip workgroup: the W image (4 KiB per bin pair), the H / partial staging and a page of zeros
static size_t strip_shmem(int nPairs)
{
  return (size_t) nPairs * 4096 + (size_t) (4 * kNQ * 4 * 16 + kNQ * 64 + 32) * sizeof(double) + 4096 + (size_t) kNQ * 64 * sizeof(double);
}
// the W image of all bin pairs must fit the 160 KiB of a CU next to the staging: 35 pairs, F <= 1120
bool nmf_strip_supported(int F, int T, int Kp)
{
  return Kp == 16 && F >= 1 && T >= 1 && strip_pairs(F) <= 36 && strip_shmem(strip_pairs(F)) <= (size_t) 160 * 1024;
}
// the frame-strip H update paired with the bin-tiled W update: the side-column form of the kernel (F = 32 j + 1), 3 .. 8 pair
// slots per wavefront (fft 1024 / 2048)
bool nmf_strip_tile_supported(int F, int T, int Kp)
{
  const int nPairs = strip_pairs(F);
  return nmf_strip_supported(F, T, Kp) && nPairs >= 2 && (F - 1) % 32 == 0 && (nPairs - 1 + 3) / 4 >= 3;
}
// at most kNQ frame quads per workgroup: 256 workgroups (one per CU) while that holds, more for longer buffers
int nmf_strip_workgroups(int T)
{
  const int64_t nq = (T + 3) / 4;
  return (int) std::max<int64_t>(std::min<int64_t>(256, nq), (nq + kNQ - 1) / kNQ);
}
// workspace: numerator partials, denominator partials, two generations of statistics records, 16 words of stamps
int64_t nmf_strip_part_doubles(int F, int T, int B)
{
  const int64_t nBlk = strip_pairs(F) * 8, nWG = nmf_strip_workgroups(T);
  return (int64_t) B * (nBlk * nWG * 64 + nWG * 16 + 2 * nBlk * kStatW) + 32;
}

static StripK make_k(const StripArgs& s)
{
  StripK k;
  k.V = s.V; k.strideV = s.strideV; k.ldv = (int) s.ldv;
  k.W = s.W; k.strideW = s.strideW;
  k.H = s.H; k.strideH = s.strideH;
  k.nPairs = strip_pairs(s.F);
  k.nBlk = k.nPairs * 8;
  k.nWG = nmf_strip_workgroups(s.T);
  k.part = s.part;
  k.dpart = k.part + (int64_t) s.B * k.nBlk * k.nWG * 64;
  double* stat0 = k.dpart + (int64_t) s.B * k.nWG * 16;
  double* stat1 = stat0 + (int64_t) s.B * k.nBlk * kStatW;
  k.statIn = s.statGen ? stat1 : stat0;
  k.statOut = s.statGen ? stat0 : stat1;
  k.nrm = s.nrm;
  k.F = s.F; k.T = s.T; k.K = s.K;
  k.nq = (s.T + 3) / 4;
  k.qBase = k.nq / k.nWG;
  k.qRem = k.nq % k.nWG;
  k.doH = s.doH; k.doW = s.doW; k.wPend = s.wPend;
  k.dbg = reinterpret_cast<long long*>(stat1 + (int64_t) s.B * k.nBlk * kStatW);
  k.tileStat = s.tileStat; k.nRec = s.nRec; k.sideOut = s.sideOut;
  return k;
}

template <int NPW, int NQ, bool INSTR = false, bool SIDE = false, bool TILE = false>
static void launch_strip_t(const StripK& k, int B, hipStream_t s)
{
  const size_t shmem = strip_shmem(k.nPairs); // <= 160 KiB: nmf_strip_supported() is what the planner asks
  auto kern = nmf_strip_kernel<NPW, NQ, INSTR, SIDE, TILE>;
  request_dynamic_lds(kern, (size_t) (shmem));
  const unsigned grid = (k.doH || k.doW) ? (unsigned) k.nWG : 1u;
  hipLaunchKernelGGL(kern, dim3(grid, (unsigned) B), dim3(256), shmem, s, k);
}

template <int NPW, bool SIDE, bool TILE = false>
static void launch_strip_q(const StripK& k, int B, hipStream_t s)
{
  // widest strip of the launch, in frame quads: the tile loops are built for 2, 4 or 6
  const int widest = (k.nq + k.nWG - 1) / k.nWG;
  if (widest <= 2) launch_strip_t<NPW, 2, false, SIDE, TILE>(k, B, s);
  else if (widest <= 4) launch_strip_t<NPW, 4, false, SIDE, TILE>(k, B, s);
  else
  {
    if constexpr (NPW == 9 && !SIDE)
    {
      static const int instr = [] { const char* e = fluhip::ab_getenv("FLUHIP_STRIP_INSTR"); return e ? std::atoi(e) : 0; }();
      if (instr && k.doH && k.doW) { launch_strip_t<9, kNQ, true>(k, B, s); return; }
    }
    launch_strip_t<NPW, kNQ, false, SIDE, TILE>(k, B, s);
  }
}

// the Nyquist bin as a side column of the combine step: F = 32 j + 1 (every power-of-two transform from fft 64 on)
static bool strip_side(const StripK& k)
{
  static const int off = [] { const char* e = fluhip::ab_getenv("FLUHIP_STRIP_SIDE"); return e && std::atoi(e) == 0 ? 1 : 0; }(); // A/B: 0 = all pairs in the MFMA loops
  return !off && k.nPairs >= 2 && (k.F - 1) % 32 == 0;
}

void launch_nmf_strip(const StripArgs& a, hipStream_t s)
{
  const StripK k = make_k(a);
#ifdef FLUHIP_AB_SWITCHES
  if (k.nRec > 0)
  {
    // the W update is the bin-tiled launch's (nmf_strip_tile_supported() is what the planner asked): tile records, side partials
    const int npw = (k.nPairs - 1 + 3) / 4;
    if (npw <= 4) launch_strip_q<4, true, true>(k, a.B, s);
    else launch_strip_q<8, true, true>(k, a.B, s);
    return;
  }
#endif
  if (strip_side(k))
  {
    const int npw = (k.nPairs - 1 + 3) / 4; // pair slots per wavefront without the Nyquist pair: 8 at fft 2048, 4 at fft 1024
    if (npw <= 2) launch_strip_q<2, true>(k, a.B, s);
    else if (npw <= 4) launch_strip_q<4, true>(k, a.B, s);
    else launch_strip_q<8, true>(k, a.B, s);
    return;
  }
  const int npw = (k.nPairs + 3) / 4;
  if (npw <= 3) launch_strip_q<3, false>(k, a.B, s);
  else if (npw <= 5) launch_strip_q<5, false>(k, a.B, s);
  else launch_strip_q<9, false>(k, a.B, s);
}

#ifdef FLUHIP_AB_SWITCHES
// bin-strip W update: workgroups = strips of kBinPP pairs x slices of the frames, at most one round of the chip
static void binstrip_shape(int F, int T, int* nStrips, int* nSlices)
{
  const int nPairs = strip_pairs(F), nq = (T + 3) / 4;
  *nStrips = (nPairs + kBinPP - 1) / kBinPP;
  // a slice of at least 8 quads (two per wavefront): below that the launch is all prologue and hand-off
  *nSlices = std::max(1, std::min(std::max(1, nq / 8), 256 / *nStrips));
}
int64_t nmf_binstrip_doubles(int F, int T, int B)
{
  int nStrips, nSlices;
  binstrip_shape(F, T, &nStrips, &nSlices);
  return (int64_t) B * nStrips * ((int64_t) nSlices * (kBinPP * 8 * 64 + 16) + 1) + 16;
}
// W' <- (W' / nrm) * (V / (W' (H / nrm))) H^T / rowsum(H) from the records of generation statGen, records of the new W'
// into the other generation; `work` = nmf_binstrip_doubles() of workspace whose ticket words are zero (they return to zero)
void launch_nmf_binstrip(const StripArgs& s0, double* work, hipStream_t s)
{
  const StripK k = make_k(s0);
  BinK a;
  a.V = s0.V; a.strideV = s0.strideV; a.ldv = (int) s0.ldv;
  a.W = s0.W; a.strideW = s0.strideW; a.H = s0.H; a.strideH = s0.strideH;
  a.statIn = k.statIn; a.statOut = k.statOut;
  a.F = s0.F; a.T = s0.T; a.K = s0.K; a.nPairs = k.nPairs; a.nBlk = k.nBlk; a.nq = k.nq; a.wPend = s0.wPend;
  binstrip_shape(s0.F, s0.T, &a.nStrips, &a.nSlices);
  a.qBase = a.nq / a.nSlices; a.qRem = a.nq % a.nSlices;
  a.part = work;
  a.dpart = a.part + (int64_t) s0.B * a.nStrips * a.nSlices * (kBinPP * 8 * 64);
  a.ticket = reinterpret_cast<int*>(a.dpart + (int64_t) s0.B * a.nStrips * a.nSlices * 16);
  a.dbg = k.dbg;
  static const int instr = [] { const char* e = fluhip::ab_getenv("FLUHIP_STRIP_INSTR"); return e ? std::atoi(e) : 0; }();
  if (instr)
    hipLaunchKernelGGL((nmf_binstrip_kernel<kBinPP, true>), dim3((unsigned) (a.nStrips * a.nSlices), (unsigned) s0.B), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((nmf_binstrip_kernel<kBinPP, false>), dim3((unsigned) (a.nStrips * a.nSlices), (unsigned) s0.B), dim3(256), 0, s, a);
}

#endif // FLUHIP_AB_SWITCHES

// reads the records of generation statGen, leaves those of the new W' in the other generation
void launch_nmf_strip_reduce(const StripArgs& a, hipStream_t s)
{
  const StripK k = make_k(a);
  hipLaunchKernelGGL(nmf_strip_reduce_kernel, dim3((unsigned) k.nBlk, (unsigned) a.B), dim3(64 * kRedWaves), 0, s, k);
}

// writes the records of generation statGen from the W in memory
void launch_nmf_strip_wstats(const StripArgs& a, hipStream_t s)
{
  StripK k = make_k(a);
  k.statOut = const_cast<double*>(k.statIn);
  hipLaunchKernelGGL(nmf_strip_wstats_kernel, dim3((unsigned) k.nBlk, (unsigned) a.B), dim3(64), 0, s, k);
}

} // namespace fluhip
