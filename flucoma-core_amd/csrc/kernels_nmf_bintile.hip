// kernels_nmf_bintile.hip -- W update of ONE large buffer at rank <= 16 with NO cross-workgroup reduction (round 5).
//
// alg/NMF.hpp:158-161 contracts over the frames t for every bin f:  num[f][k] = sum_t (V / max(W H, eps))[f][t] H[k][t].
// The frame-strip schedule (kernels_nmf_strip.hip) owns frames per workgroup and therefore hands this contraction from 256
// workgroups to a reduce launch as 256 partials of the whole F x 16 matrix (33.5 MB written and re-read per iteration at
// BASELINE config 2, 11 us of reduce launch); round 4's bin-strip form cut the partials tenfold and paid the same time in a
// last-arriver's chain of cross-XCD round trips.  Here the decomposition is over BINS ONLY: a workgroup owns four bins --
// one row block of v_mfma_f64_4x4x4_4b -- and ALL frames.  Its NW wavefronts split the frames, stream the bin-major rows
// of V (magT, HBM) and the rows of H (from the XCD's L2: every workgroup reads all of H, 661 KB at config 2) through
// private LDS-DMA rings, and add their numerators up through the LDS in wavefront order.  No partials in memory, no second
// launch, no ticket; the four blocks of the MFMA are four frame quads against the same four bins.
//
//   first product   Q[t][f]   = sum_k H[t][k] Wn[f][k]      A = rows of H  (lane x = frame in quad, y = k index, blk = quad)
//                                                           B = rows of Wn (lane x = bin, y = k index), stationary: Wn = W' / nrm
//                                                           D lane (x = bin, y = frame in quad)
//   quotient        R = V / max(Q, eps)                     V[bin x][frame (blk, y)] in the same lanes
//   second product  num[f][k] += sum_t R[t][f] H[t][k]      A = R as it lies (row = bin, contraction = frame in quad)
//                                                           B = rows of H (lane y = frame in quad, x = k index)
//                                                           D lane (y = bin, x = k index), summed over the blocks at the end
// Component k = 4 * (k index) + c for the four MFMAs c of a product, so that a lane's four values of a row of H / W are 32
// contiguous bytes.  The column sums of H (the denominator, :160) are taken from the second product's B operands: every
// workgroup forms them itself, in the same order, so all see the same bits.
//
// Statistics records: the deferred normalisation (:162) needs sum x^2, sum x and max x of every column of the new W'.
// A workgroup leaves ONE record for its four bins, stat[kind][k][rec]; consumers (this kernel, the strip kernel's prologue)
// add the nRec records of a column in one fixed order (tile_column_stats).  The Nyquist bin (F = 4 nRec + 1) is a side row:
// the frame-strip launch in front leaves its numerator as one partial per strip workgroup, the LAST tile workgroup adds
// them, updates the row and folds it into its record.
#include "fluhip_kernels.h"
#include "nmf_tile_stats.h"

#ifdef FLUHIP_AB_SWITCHES // (measured slower than the fused form at config 2, profiles/r05/c2_bintile.md: A/B build only)

#include <algorithm>

namespace fluhip {
namespace bintile {

typedef double d2 __attribute__((ext_vector_type(2)));

struct TileK
{
  const double* VT;      // magT [B][Fp][ldT]: row = bin, frames contiguous
  int64_t strideVT;
  int ldT;
  double* W;             // [B][Fp][16]
  int64_t strideW;
  const double* H;       // [B][Tp][16]
  int64_t strideH;
  const double* statIn;  // tile records [B][48][nRec] of the W' in memory
  double* statOut;       // ... of the W' this launch leaves
  const double* sidePart; // [B][nSideWG][16] numerator partials of bin F - 1, or null (no side row)
  int nSideWG;
  int F, K, nRec, nSteps, wPend;
  int dbg; // A/B build, FLUHIP_TILE_DBG (timing experiments, results wrong): 1 = no H refills, 2 = no V refills, 4 = no frame loop,
           // 8 = no ring fill in the prologue, 16 = one statistics record instead of all
};

#define MFMA44(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned) (size_t) (__attribute__((address_space(3))) const void*) p; }
// LDS-DMA, 16 bytes per lane to M0 + lane * 16, "scalar base + 32-bit lane offset" addressing
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned ldsAddr)
{
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(ldsAddr), "v"(voff), "s"(sbase) : "memory", "m0");
}
template <int CTRL>
__device__ __forceinline__ double dppmov(double v)
{
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xffffffff), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
}

// A wavefront walks its frames in STEPS of 16 (one 4x4x4 four-block tile: 4 bins x 4 quads of 4 frames).  Rings per
// wavefront: H in stages of one step (16 frames x 16 = 2 KB, two DMAs), V in units of two steps (4 bins x 32 frames = 1 KB,
// one DMA).  LDS images (the DMA writes lane * 16 linearly, so the arrangement is applied to the SOURCE addresses):
//   V  row r (bin) at r * 256, its 16-byte chunks rotated by (r & 1) + 8 (r >> 1): the 32 lanes of a ds_read_b64 group
//      (frames 4 blk + y, y in {0, 1} or {2, 3}, all four bins) hit 32 different 8-byte slots
//   H  frame t at t * 128, chunk c at position c ^ g(t), g(t) = bit 2 of t | bit 1 of t << 2: both operand read patterns
//      (frame by x, chunks by y; frame by y, chunks by x) touch 16 different 16-byte slots per ds_read_b128 lane group
//      (searched exhaustively over the XOR-linear maps of t; PMC: SQ_LDS_BANK_CONFLICT ~ 0)
// Why 16 wavefronts of 9 KB and not fewer with deeper rings (profiles/r05/c2_bintile.md): a step is ONE dependent chain --
// LDS reads -> first product -> sum, max, rcp, Newton, multiply -> second product -- of ~1 000 cycles for 128 cycles of matrix
// work; the first form (8 wavefronts, 32-frame stages) ran 19 us with or without its memory traffic.  Four wavefronts per
// SIMD hide each other's chains.
__device__ __forceinline__ int hswz(int t) { return ((t >> 2) & 1) | (((t >> 1) & 1) << 2); }

template <int NW, int NSV, int NSH>
__global__ __launch_bounds__(64 * NW) void nmf_bintile_kernel(TileK a)
{
  static_assert(NW == 16, "column statistics: one wavefront per column");
  static_assert(NSH == 3 && NSV == 3, "the wait counts below are written for rings of three steps / three units");
  constexpr int VSTG = 1024, HSTG = 2048;
  constexpr int WAVE_LDS = NSV * VSTG + NSH * HSTG;
  constexpr int UNR = 2 * NSV * NSH / ((2 * NSV) % NSH == 0 ? NSH : 1) / (NSH % (2 * NSV) == 0 ? 2 * NSV : 1); // steps after which slots repeat
  static_assert(UNR % 2 == 0 && UNR % NSH == 0 && UNR % (2 * NSV) == 0, "unroll");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double sc[48], nrmL[16], csL[16], denL[16], sideL[16];

  if (a.dbg & 32) return; // (timing experiment: what an empty launch of this shape costs)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
  const int rec = blockIdx.x, b = blockIdx.y;
  const int f0 = 4 * rec;
  const double* VT = a.VT + (int64_t) b * a.strideVT + (int64_t) f0 * a.ldT;
  const double* Hg = a.H + (int64_t) b * a.strideH;
  double* Wg = a.W + (int64_t) b * a.strideW;

  // this wavefront's steps (16 frames each), dealt evenly; a V unit is two of ITS steps (an odd count leaves half of the
  // last unit unused: 128 bytes of the next wavefront's frames or of the row's padding are read and ignored)
  const int base = a.nSteps / NW, rem = a.nSteps % NW;
  const int sBeg = wv * base + min(wv, rem), cnt = base + (wv < rem ? 1 : 0);
  const int lastUnit = (cnt - 1) >> 1;

  unsigned char* vring = smem + wv * WAVE_LDS;
  unsigned char* hring = vring + NSV * VSTG;
  const unsigned vringA = __builtin_amdgcn_readfirstlane(lds_addr(vring));
  const unsigned hringA = __builtin_amdgcn_readfirstlane(lds_addr(hring));

  // per-lane DMA source offsets (bytes from the stage's uniform base)
  unsigned voff;
  {
    const int r = lane >> 4, pos = lane & 15;
    const int srcc = (pos - ((r & 1) + 8 * (r >> 1))) & 15;
    voff = (unsigned) ((r * a.ldT + 2 * srcc) * 8);
  }
  unsigned hoff[2];
#pragma unroll
  for (int j = 0; j < 2; j++)
  {
    const int t = 8 * j + (lane >> 3), cp = lane & 7;
    hoff[j] = (unsigned) (t * 128 + ((cp ^ hswz(t)) * 16));
  }
  auto issue_v = [&](int unit, int slot) {
    const int u_ = min(unit, lastUnit); // past the end: a harmless re-read of the last unit
    glds16(VT + (int64_t) (sBeg + 2 * u_) * 16, voff, vringA + slot * VSTG);
  };
  auto issue_h = [&](int st, int slot) {
    const int s_ = sBeg + min(st, cnt - 1);
    const double* src = Hg + (int64_t) s_ * 16 * 16;
#pragma unroll
    for (int j = 0; j < 2; j++) glds16(src, hoff[j], hringA + slot * HSTG + j * 1024);
  };
  if (cnt > 0 && !(a.dbg & 8))
  {
    // the issue order of the steady state from the start -- an odd step issues V then H, an even step H: five requests per
    // two steps -- so that "at most 5 requests outstanding" means "step s has landed" at every wait
    static_assert(NSV == 3, "prologue order");
    issue_v(0, 0);
    issue_v(1, 1); issue_h(0, 0);
    issue_h(1, 1);
    issue_v(2, 2); issue_h(2, 2);
  }

  // ---- prologue requests: statistics records, this lane's stationary rows, the old values of the final threads ----------
  const TileStatRaw<4> straw = tile_column_stats_load<4>(a.statIn + (int64_t) b * 48 * a.nRec, (a.dbg & 16) ? 1 : a.nRec, tid);
  d2 wr[2];
  {
    const double* wp = Wg + (int64_t) (f0 + x) * 16 + 4 * y;
    wr[0] = *reinterpret_cast<const d2*>(wp);
    wr[1] = *reinterpret_cast<const d2*>(wp + 2);
  }
  const int fr = (tid >> 4) & 3, fk = tid & 15; // final threads (tid < 64): bin f0 + fr, component fk
  double wold = 0.0;
  if (tid < 64) wold = Wg[(int64_t) (f0 + fr) * 16 + fk];
  // side row (last tile workgroup): the partials of bin F - 1's numerator, thread (group tid >> 4, component tid & 15)
  const bool side = a.sidePart != nullptr && rec == a.nRec - 1;
  constexpr int SG = 4 * NW; // groups of 16 threads
  constexpr int SU = 4;      // partials per thread and round
  double sp[SU];
  double woldN = 0.0;
  if (side)
  {
    const double* P = a.sidePart + (int64_t) b * a.nSideWG * 16 + fk;
#pragma unroll
    for (int u = 0; u < SU; u++) sp[u] = P[(int64_t) min((tid >> 4) + SG * u, a.nSideWG - 1) * 16];
    if (tid < 16) woldN = Wg[(int64_t) (a.F - 1) * 16 + tid];
  }
  __builtin_amdgcn_sched_barrier(0);

  tile_column_stats<4>(straw, a.nRec, a.K, a.wPend, sc, nrmL, csL, tid);
  double wB[4]; // Wn[bin x][4 y + c]
#pragma unroll
  for (int c = 0; c < 4; c++) wB[c] = wr[c >> 1][c & 1] / nrmL[4 * y + c];

  // ---- LDS read addresses (bytes within a stage / unit) -------------------------------------------------------------------
  int offA[2], offB[2], offV[2];
  {
    const int tA = 4 * blk + x, tB = 4 * blk + y;
#pragma unroll
    for (int cc = 0; cc < 2; cc++)
    {
      offA[cc] = tA * 128 + (((2 * y + cc) ^ hswz(tA)) * 16);
      offB[cc] = tB * 128 + (((2 * x + cc) ^ hswz(tB)) * 16);
    }
#pragma unroll
    for (int h = 0; h < 2; h++) offV[h] = x * 256 + (((16 * h + 4 * blk + y) + 2 * (x & 1) + 16 * (x >> 1)) & 31) * 8;
  }

  // (every MFMA of a phase has its own accumulator: the first product's four k-chunks are summed by the VALU)
  double num[4] = {0.0, 0.0, 0.0, 0.0}, dacc[4] = {0.0, 0.0, 0.0, 0.0};
  auto step = [&](int s, int half, int slotV, int slotH) {
    // step s has landed: at most 5 younger requests are outstanding
    asm volatile("s_waitcnt vmcnt(5)" : : : "memory");
    const unsigned char* vp = vring + slotV * VSTG;
    const unsigned char* hp = hring + slotH * HSTG;
    d2 ha[2], hb[2];
#pragma unroll
    for (int cc = 0; cc < 2; cc++) ha[cc] = *reinterpret_cast<const d2*>(hp + offA[cc]);
    const double vv = *reinterpret_cast<const double*>(vp + offV[half]);
#pragma unroll
    for (int cc = 0; cc < 2; cc++) hb[cc] = *reinterpret_cast<const d2*>(hp + offB[cc]);
    double Qp[4];
#pragma unroll
    for (int c = 0; c < 4; c++) Qp[c] = MFMA44(ha[c >> 1][c & 1], wB[c], 0.0);
    __builtin_amdgcn_sched_barrier(0);
    // every read of the slots has returned: the refill goes out under the matrix work (V: behind the unit's second half)
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
    if (!(a.dbg & 2)) { if (half) issue_v((s >> 1) + NSV, slotV); }
    if (!(a.dbg & 1)) issue_h(s + NSH, slotH);
    __builtin_amdgcn_sched_barrier(0);
    const double d_ = fmax((Qp[0] + Qp[1]) + (Qp[2] + Qp[3]), kEpsilon);
    double y_ = __builtin_amdgcn_rcp(d_);
    y_ = __builtin_fma(y_, __builtin_fma(-d_, y_, 1.0), y_);
    const double R = vv * y_;
#pragma unroll
    for (int c = 0; c < 4; c++)
    {
      const double hv = hb[c >> 1][c & 1];
      num[c] = MFMA44(R, hv, num[c]);
      dacc[c] += hv;
    }
  };
  for (int s = 0; s < ((a.dbg & 4) ? 0 : cnt); s += UNR)
  {
#pragma unroll
    for (int u = 0; u < UNR; u++)
    {
      if (s + u >= cnt) break;
      step(s + u, u & 1, (u >> 1) % NSV, u % NSH);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); // the run-ahead requests have landed: the rings are free

  // ---- the wavefronts' numerators and column sums, in wavefront order ----------------------------------------------------
  double* red = reinterpret_cast<double*>(smem); // [NW][4][16] numerators, then [NW][16] column sums, then the side groups
  double* dred = red + NW * 64;
  __syncthreads(); // (every wavefront is done with its ring)
#pragma unroll
  for (int c = 0; c < 4; c++)
  {
    double v = num[c];
    v += dppmov<0x124>(v); // row_ror:4
    v += dppmov<0x128>(v); // row_ror:8: the four blocks (frame quads)
    if (blk == 0) red[(wv * 4 + c) * 16 + 4 * y + x] = v; // bin y, component 4 x + c
    double d = dacc[c];
    d += dppmov<0x124>(d);
    d += dppmov<0x128>(d);
    d += __shfl_xor(d, 16);
    d += __shfl_xor(d, 32);
    if (lane < 4) dred[wv * 16 + 4 * x + c] = d;
  }
  __syncthreads();
  if (tid < 16)
  {
    double d = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) d += dred[w * 16 + tid];
    denL[tid] = fmax(d, kEpsilon);
  }
  if (side)
  {
    // bin F - 1: the strips' partials in strip order (thread group g takes g, g + SG, ...; the groups are then added in order)
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < SU; u++) s += (tid >> 4) + SG * u < a.nSideWG ? sp[u] : 0.0;
    for (int p = (tid >> 4) + SG * SU; p < a.nSideWG; p += SG) s += a.sidePart[((int64_t) b * a.nSideWG + p) * 16 + fk];
    dred[NW * 16 + tid] = s; // [SG][16] as tid = g * 16 + k
  }
  __syncthreads();
  if (side && tid < 16)
  {
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < SG; g++) s += dred[NW * 16 + g * 16 + tid];
    double wn = 0.0;
    if (tid < a.K)
    {
      wn = (woldN / nrmL[tid]) * s / denL[tid];
      Wg[(int64_t) (a.F - 1) * 16 + tid] = wn;
    }
    sideL[tid] = wn;
  }
  if (side) __syncthreads();
  if (tid < 64)
  {
    const int c = fk & 3, xx = fk >> 2;
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) tot += red[(w * 4 + c) * 16 + 4 * fr + xx];
    double wnew = 0.0;
    if (f0 + fr < a.F && fk < a.K)
    {
      // :161  W * num / max(den, eps), W = W' / nrm (:162 of the iteration before)
      wnew = (wold / nrmL[fk]) * tot / denL[fk];
      Wg[(int64_t) (f0 + fr) * 16 + fk] = wnew;
    }
    // the record of this tile: the four bins of a column sit 16 lanes apart
    double s2 = wnew * wnew, s1 = wnew, mx = wnew;
    s2 += __shfl_xor(s2, 16); s1 += __shfl_xor(s1, 16); mx = fmax(mx, __shfl_xor(mx, 16));
    s2 += __shfl_xor(s2, 32); s1 += __shfl_xor(s1, 32); mx = fmax(mx, __shfl_xor(mx, 32));
    if (tid < 16)
    {
      if (side)
      {
        const double wn = sideL[tid];
        s2 = __builtin_fma(wn, wn, s2); s1 += wn; mx = fmax(mx, wn);
      }
      double* so = a.statOut + (int64_t) b * 48 * a.nRec;
      so[(int64_t) tid * a.nRec + rec] = s2;
      so[(int64_t) (16 + tid) * a.nRec + rec] = s1;
      so[(int64_t) (32 + tid) * a.nRec + rec] = mx;
    }
  }
}

// tile records of a W that something else wrote (factor initialisation, the normalisation at the end of a call): one
// 64-thread workgroup per tile, thread (bin r = tid >> 4, component k = tid & 15)
__global__ __launch_bounds__(64) void nmf_bintile_wstats_kernel(TileK a)
{
  const int tid = threadIdx.x, rec = blockIdx.x, b = blockIdx.y;
  const int r = tid >> 4, k = tid & 15, f = 4 * rec + r;
  const double* Wg = a.W + (int64_t) b * a.strideW;
  double v = (f < a.F && k < a.K) ? Wg[(int64_t) f * 16 + k] : 0.0;
  double s2 = v * v, s1 = v, mx = v;
  s2 += __shfl_xor(s2, 16); s1 += __shfl_xor(s1, 16); mx = fmax(mx, __shfl_xor(mx, 16));
  s2 += __shfl_xor(s2, 32); s1 += __shfl_xor(s1, 32); mx = fmax(mx, __shfl_xor(mx, 32));
  if (tid < 16)
  {
    if (a.sidePart == nullptr && rec == a.nRec - 1 && 4 * a.nRec < a.F && tid < a.K) // (flag reused: fold bin F - 1 into the last record)
    {
      const double wn = Wg[(int64_t) (a.F - 1) * 16 + tid];
      s2 = __builtin_fma(wn, wn, s2); s1 += wn; mx = fmax(mx, wn);
    }
    double* so = a.statOut + (int64_t) b * 48 * a.nRec;
    so[(int64_t) tid * a.nRec + rec] = s2;
    so[(int64_t) (16 + tid) * a.nRec + rec] = s1;
    so[(int64_t) (32 + tid) * a.nRec + rec] = mx;
  }
}

} // namespace bintile
using namespace bintile;

constexpr int kTileNW = 16, kTileNSV = 3, kTileNSH = 3;

// shapes: rank <= 16 (Kp = 16), F = 4 nRec + 1 with at most 16 * kRecU records, enough tiles to occupy the part
bool nmf_bintile_supported(int F, int T, int Kp)
{
  (void) T;
  return Kp == 16 && F >= 5 && (F - 1) % 4 == 0 && (F - 1) / 4 <= 16 * kRecU;
}
int nmf_bintile_records(int F) { return (F - 1) / 4; }
// workspace: two generations of tile records + the side partials of up to nSideWG strip workgroups
int64_t nmf_bintile_doubles(int F, int B, int nSideWG) { return (int64_t) B * (2 * 48 * nmf_bintile_records(F) + (int64_t) nSideWG * 16); }

static TileK make_tile(const BinTileArgs& s)
{
  TileK k;
  k.VT = s.VT; k.strideVT = s.strideVT; k.ldT = (int) s.ldT;
  k.W = s.W; k.strideW = s.strideW;
  k.H = s.H; k.strideH = s.strideH;
  k.nRec = nmf_bintile_records(s.F);
  double* stat0 = s.work;
  double* stat1 = stat0 + (int64_t) s.B * 48 * k.nRec;
  k.statIn = s.statGen ? stat1 : stat0;
  k.statOut = s.statGen ? stat0 : stat1;
  k.sidePart = s.sidePart; k.nSideWG = s.nSideWG;
  k.F = s.F; k.K = s.K; k.nSteps = (int) (s.ldT / 16); k.wPend = s.wPend;
  static const int dbg = [] { const char* e = fluhip::ab_getenv("FLUHIP_TILE_DBG"); return e ? std::atoi(e) : 0; }();
  k.dbg = dbg;
  return k;
}
double* nmf_bintile_side_area(const BinTileArgs& s) { return s.work + (int64_t) s.B * 2 * 48 * nmf_bintile_records(s.F); }
const double* nmf_bintile_records_ptr(const BinTileArgs& s, int gen)
{
  return s.work + (gen ? (int64_t) s.B * 48 * nmf_bintile_records(s.F) : 0);
}

// W' <- (W' / nrm) * (V / (W' / nrm H)) H^T / rowsum(H) from the records of generation statGen; the records of the new W'
// go to the other generation
void launch_nmf_bintile(const BinTileArgs& s, hipStream_t st)
{
  const TileK k = make_tile(s);
  constexpr size_t shmem = (size_t) kTileNW * (kTileNSV * 1024 + kTileNSH * 2048);
  static_assert(shmem + 1024 <= 160 * 1024, "LDS");
  auto kern = nmf_bintile_kernel<kTileNW, kTileNSV, kTileNSH>;
  request_dynamic_lds(kern, shmem);
  hipLaunchKernelGGL(kern, dim3((unsigned) k.nRec, (unsigned) s.B), dim3(64 * kTileNW), shmem, st, k);
}
// writes the records of generation statGen from the W in memory (bin F - 1 folded into the last record)
void launch_nmf_bintile_wstats(const BinTileArgs& s, hipStream_t st)
{
  TileK k = make_tile(s);
  k.statOut = const_cast<double*>(k.statIn);
  k.sidePart = nullptr;
  hipLaunchKernelGGL(nmf_bintile_wstats_kernel, dim3((unsigned) k.nRec, (unsigned) s.B), dim3(64), 0, st, k);
}

} // namespace fluhip

#endif // FLUHIP_AB_SWITCHES
