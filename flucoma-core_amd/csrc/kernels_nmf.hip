// kernels_nmf.hip -- the small gfx950 kernels around the KL-divergence NMF multiplicative updates of flucoma-core's
// algorithm::NMF (include/flucoma/algorithms/public/NMF.hpp:144-183): the fixed-order finalize of split contractions,
// column clamp / L2 normalisation (:150-153, 162), column sums, the deferred-normalisation kernels (side column, norm
// combine / apply) and the layout plumbing (scatter / gather / write-back).  The factor updates themselves are
// kernels_nmf5.hip (v_mfma_f64_4x4x4_4b + LDS-DMA, ranks up to 128), kernels_nmf_strip.hip (one large buffer at rank <= 16)
// and kernels_nmf_wide.hip (any rank above 128).  (The first fused kernel on v_mfma_f64_16x16x4 lived here through round 2:
// 35.9 - 49.6 TFLOP/s peak for that instruction against 73.5 for the 4x4x4 form, profiles/r01/mfma_f64_probe.txt.)
#include "fluhip_kernels.h"

#include <cstdlib>

#include <algorithm>

namespace fluhip {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kNormRows = 64; // rows per chunk of the column-wise passes

// split-R epilogue: fixed-order reduction of the partials, then the multiplicative step.  One block per
// (16-row chunk, buffer), thread = (split group sg of 4, row group, k): a thread adds its quarter of the nsplit
// partials in index order (eight loads in flight), the four quarter sums are combined in fixed order through
// LDS -- single-buffer problems have few rows and many splits, so the parallelism has to come from the splits.
// With deferred normalisation (UpdateArgs::nrm) the step is the one of the un-split kernel's epilogue, and
// for the W update the block leaves its column statistics (sum x^2, max) for wnorm_combine_kernel.
// Latency shape: the kernel moves a few MB through a few hundred workgroups, so what it costs is the number of
// DEPENDENT memory round trips, not bytes.  Every thread therefore issues all of its loads up front -- its share of
// the split partials for kFinBatch rows at once, the old value of S, and (the first row group) the denominator
// partials -- and only then starts adding; the first form walked its 16 rows in four dependent batches behind a
// denominator pass of its own (6.5 us per launch for 6 MB; this form: one round trip).
// ... and its partial loads are non-temporal (read once): a 10 s buffer at rank 128 93.4 -> 89.9 us per iteration, c4 x 1 the same
// (round 6; the frame-strip reduce launch measured 0.5 - 1 % SLOWER with the same change and keeps plain loads).
#ifndef FLUHIP_FIN_NT
#define FLUHIP_FIN_NT 1
#endif
// the finalize launch's results leave write-through like the update kernel's own (round 6: c4 x 1 44.58 -> 44.24 us per iteration,
// a 10 s buffer at rank 128 93.55 -> 92.5, alternating on one box; config 3 within noise).  -DFLUHIP_FIN_SC1=0: plain stores.
#ifndef FLUHIP_FIN_SC1
#define FLUHIP_FIN_SC1 1
#endif
constexpr int kFinSG = 4;      // thread groups sharing the split partials of an element
constexpr int kFinBatch = 4;   // rows per row group in flight together (a workgroup takes BATCH = 4, 8 or 16 of them, four at a time)
// (split partials per thread group: nsplit <= 64, the kernel is built for 1, 2, 4, 8 and 16)

static int fin_row_groups(int Kp) { int nrg = 256 / (Kp * kFinSG); return nrg < 1 ? 1 : nrg; }
// Rows per row group and workgroup.  Four in production.  Round 5 looked at config 3 (2 048 bins at rank 128) launch by launch:
// 512 workgroups of four rows per channel leave 512 statistics records of 2 KB for ONE workgroup per channel to add up
// (wnorm_combine_kernel<1024>: 25.8 us), so 8 / 16 rows per row group were built (FLUHIP_FIN_BATCH=8|16, A/B build) and measured:
// 128 records bring the combine to 15.1 us, but the finalize launches themselves get slower by more than that (23.7 -> 34.8
// and 11.2 -> 16.2 us: they live on their workgroup count) -- profiles/r05/c3_fin_batch.txt.  What was adopted instead is a
// pre-reduction of the records by sixteen workgroups per buffer (wnorm_prereduce_kernel below).
// A function of (C, Kp) alone: the planner sizes the statistics area with it before any launch.
static int fin_batch(int C, int Kp)
{
  (void) C; (void) Kp;
#ifdef FLUHIP_AB_SWITCHES
  static const int forced = [] { const char* e = fluhip::ab_getenv("FLUHIP_FIN_BATCH"); return e ? std::atoi(e) : 0; }();
  if (forced == 8 || forced == 16) return forced;
#endif
  return kFinBatch;
}
int update_finalize_rows(int C, int Kp) { return fin_batch(C, Kp) * fin_row_groups(Kp); }
int update_finalize_parts(int C, int Kp) { const int rows = update_finalize_rows(C, Kp); return (C + rows - 1) / rows; }

template <int PER, int BATCH = kFinBatch>
__global__ __launch_bounds__(512) void nmf_update_finalize_kernel(double* S, int64_t strideS, const double* part,
                                           const double* dpart, int C, int Kp, int64_t Cp,
                                           int nsplit, const double* nrm, int nrmMode, double* statPart, int nch,
                                           const int* splitTab)
{
  extern __shared__ double sh[]; // [kFinSG][nrg][BATCH][Kp] partial numerators, then [kFinSG][Kp] denominators; reused for the statistics
  const int chunk = blockIdx.x, buf = blockIdx.y;
  // work-list mode (ragged corpora): buffer b owns the partials [splitTab[2 b], splitTab[2 b] + splitTab[2 b + 1])
  int64_t pbase = (int64_t) buf * nsplit;
  if (splitTab)
  {
    pbase = splitTab[2 * buf];
    nsplit = splitTab[2 * buf + 1];
  }
#ifdef FLUHIP_AB_SWITCHES
  const int dbgBits = nrmMode >> 8;
  nrmMode &= 255;
#else
  constexpr int dbgBits = 0;
#endif
  const int nrg = blockDim.x / (Kp * kFinSG);
  const int k = threadIdx.x % Kp, rg = (threadIdx.x / Kp) % nrg, sg = threadIdx.x / (Kp * nrg);
  const int per = (nsplit + kFinSG - 1) / kFinSG;
  const int sb = sg * per, se = min(nsplit, sb + per);
  const int rows = BATCH * nrg;
  const int rbeg = chunk * rows, rend = min(rbeg + rows, C);
  const double* p0 = part + pbase * Cp * Kp;
  const int64_t sstride = Cp * Kp;
  double* shn = sh;                                   // [kFinSG][nrg][BATCH][Kp]
  double* shd = sh + kFinSG * nrg * BATCH * Kp;       // [kFinSG][Kp]
  // ---- the loads of this thread, issued before anything is consumed: kFinBatch rows at a time ------------------------
  double sold[BATCH];
  double dv[PER];
  double nk = 1.0;
#pragma unroll
  for (int i0 = 0; i0 < BATCH; i0 += kFinBatch)
  {
    double pv[kFinBatch][PER];
#pragma unroll
    for (int j = 0; j < kFinBatch; j++)
    {
      const int r = rbeg + (i0 + j) * nrg + rg;
      const int64_t idx = (int64_t) min(r, C - 1) * Kp + k;
#pragma unroll
      for (int u = 0; u < PER; u++)
      {
#if FLUHIP_FIN_NT
        pv[j][u] = (u < per && !(dbgBits & 2)) ? __builtin_nontemporal_load(p0 + (int64_t) min(sb + u, nsplit - 1) * sstride + idx) : 0.0;   // (read once)
#else
        pv[j][u] = (u < per && !(dbgBits & 2)) ? p0[(int64_t) min(sb + u, nsplit - 1) * sstride + idx] : 0.0;
#endif
      }
      sold[i0 + j] = (sg == 0 && !(dbgBits & 8)) ? S[(int64_t) buf * strideS + idx] : 0.0;
    }
    if (i0 == 0)   // (behind the first rows' requests, as the four-row form always had them)
    {
      const double* dp = dpart + pbase * Kp + k;
#pragma unroll
      for (int u = 0; u < PER; u++) dv[u] = (rg == 0 && u < per) ? dp[(int64_t) min(sb + u, nsplit - 1) * Kp] : 0.0;
      nk = (nrmMode && sg == 0) ? nrm[(int64_t) buf * Kp + k] : 1.0;
    }
    // ---- quarter sums in split order, combined in group order through LDS (below) ---------------------------------------
#pragma unroll
    for (int j = 0; j < kFinBatch; j++)
    {
      double num = 0.0;
#pragma unroll
      for (int u = 0; u < PER; u++)
        if (sb + u < se) num += pv[j][u];
      shn[((sg * nrg + rg) * BATCH + i0 + j) * Kp + k] = num;
    }
  }
  if (rg == 0)
  {
    double den = 0.0;
#pragma unroll
    for (int u = 0; u < PER; u++)
      if (sb + u < se) den += dv[u];
    shd[sg * Kp + k] = den;
  }
  __syncthreads();
  double ss = 0.0, mx = -INFINITY;
  if (sg == 0)
  {
    double den = shd[k];
#pragma unroll
    for (int g = 1; g < kFinSG; g++) den += shd[g * Kp + k];
    if (nrmMode == 2) den = den / nk;
    den = fmax(den, kEpsilon);
#pragma unroll
    for (int i = 0; i < BATCH; i++)
    {
      const int r = rbeg + i * nrg + rg;
      if (r < rend)
      {
        double t = shn[(rg * BATCH + i) * Kp + k];
#pragma unroll
        for (int g = 1; g < kFinSG; g++) t += shn[((g * nrg + rg) * BATCH + i) * Kp + k];
        double so = sold[i];
        if (nrmMode) so = so / nk; // W update: W = W'/nrm; H update: (H/nrm) acc == H (acc/nrm)
        const double x = (so * t) / den;
        if (!(dbgBits & 4))
        {
#if FLUHIP_FIN_SC1
          // (write-through, like the update kernel's own results: nothing dirty left for the end-of-kernel release)
          asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(S + (int64_t) buf * strideS + (int64_t) r * Kp + k), "v"(x) : "memory");
#else
          S[(int64_t) buf * strideS + (int64_t) r * Kp + k] = x;
#endif
        }
        ss += x * x;
        mx = fmax(mx, x);
      }
    }
  }
#ifdef FLUHIP_AB_SWITCHES
  if (nch < 0) return;   // (FLUHIP_FIN_DBG=1: timing experiment, no statistics -- wrong norms)
#endif
  if (!statPart) return;
  __syncthreads();
  if (sg == 0)
  {
    sh[rg * Kp + k] = ss;
    sh[(nrg + rg) * Kp + k] = mx;
  }
  __syncthreads();
  if (sg == 0 && rg == 0)
  {
    double tot = 0.0, m = -INFINITY;
    for (int j = 0; j < nrg; j++)
    {
      tot += sh[j * Kp + k];
      m = fmax(m, sh[(nrg + j) * Kp + k]);
    }
    double* q = statPart + ((int64_t) buf * nch + chunk) * 2 * Kp;
    q[k] = tot;
    q[Kp + k] = m;
  }
}

void launch_update_finalize(double* S, int64_t strideS, const double* part, const double* dpart,
                            int C, int Kp, int64_t Cp, int nsplit, int B, hipStream_t s, const double* nrm,
                            int nrmMode, double* statPart, const int* splitTab)
{
  const int nrg = fin_row_groups(Kp);
  const int nch = update_finalize_parts(C, Kp);
  const int batch = fin_batch(C, Kp);
  const size_t shmem = ((size_t) kFinSG * nrg * batch * Kp + (size_t) kFinSG * Kp) * sizeof(double);
  const int per = (nsplit + kFinSG - 1) / kFinSG;
  const dim3 grid((unsigned) nch, (unsigned) B), block((unsigned) (kFinSG * nrg * Kp));
  int nchArg = nch;
#ifdef FLUHIP_AB_SWITCHES
  // FLUHIP_FIN_DBG (timing experiments, wrong results): 1 no statistics, 2 no partial loads, 4 no result store, 8 no old-value load
  static const int dbg = [] { const char* e = fluhip::ab_getenv("FLUHIP_FIN_DBG"); return e ? std::atoi(e) : 0; }();
  if (dbg & 1) nchArg = -nch;
  nrmMode |= (dbg & 14) << 8;
#endif
#define FLUHIP_FIN_B(P, BT)                                                                                                   \
  {                                                                                                                           \
    auto kern = nmf_update_finalize_kernel<P, BT>;                                                                            \
    if (shmem > 48 * 1024) request_dynamic_lds(kern, shmem);                                                                  \
    hipLaunchKernelGGL(kern, grid, block, shmem, s, S, strideS, part, dpart, C, Kp, Cp, nsplit, nrm, nrmMode, statPart, nchArg, \
                       splitTab);                                                                                             \
  }
#ifdef FLUHIP_AB_SWITCHES
#define FLUHIP_FIN(P)                                                                                                         \
  {                                                                                                                           \
    if (batch == 16) FLUHIP_FIN_B(P, 16)                                                                                      \
    else if (batch == 8) FLUHIP_FIN_B(P, 8)                                                                                   \
    else FLUHIP_FIN_B(P, 4)                                                                                                   \
  }
#else
#define FLUHIP_FIN(P) FLUHIP_FIN_B(P, 4)
#endif
  if (per <= 1) FLUHIP_FIN(1)
  else if (per <= 2) FLUHIP_FIN(2)
  else if (per <= 4) FLUHIP_FIN(4)
  else if (per <= 8) FLUHIP_FIN(8)
  else FLUHIP_FIN(16)
#undef FLUHIP_FIN_B
#undef FLUHIP_FIN
}

// ---------------------------------------------------------------------------------------
// column L2 normalisation (Eigen colwise().normalize(): x / sqrt(sum x^2), alg/NMF.hpp:152-153,162)
// two small launches over (row chunk, buffer): per-chunk partial sums of squares and maxima,
// then a fixed-order combine + scale.  Deterministic: no atomics, the same summation tree on
// every run and for every buffer.
// ---------------------------------------------------------------------------------------

__global__ void colstats_kernel(double* Sbase, int64_t strideS, int C, int K, int Kp, int clampEps,
                                double* part, int nch, const int* rowsTab)
{
  extern __shared__ double sh[]; // [nrg][Kp] sums then [nrg][Kp] maxima
  const int chunk = blockIdx.x, b = blockIdx.y;
  if (rowsTab) C = rowsTab[b]; // ragged corpora: the buffer's own row count (rows beyond it are padding and stay zero)
  double* S = Sbase + (int64_t) b * strideS;
  const int nrg = blockDim.x / Kp;
  const int k = threadIdx.x % Kp, rg = threadIdx.x / Kp;
  const int rbeg = chunk * kNormRows, rend = min(rbeg + kNormRows, C);
  double ss = 0.0, mx = -INFINITY;
  if (k < K)
    for (int r = rbeg + rg; r < rend; r += nrg)
    {
      double x = S[(int64_t) r * Kp + k];
      if (clampEps)
      {
        x = fmax(x, kEpsilon);
        S[(int64_t) r * Kp + k] = x;
      }
      ss += x * x;
      mx = fmax(mx, x);
    }
  sh[rg * Kp + k] = ss;
  sh[(nrg + rg) * Kp + k] = mx;
  __syncthreads();
  if (rg == 0)
  {
    double tot = 0.0, m = -INFINITY;
    for (int j = 0; j < nrg; j++)
    {
      tot += sh[j * Kp + k];
      m = fmax(m, sh[(nrg + j) * Kp + k]);
    }
    double* p = part + ((int64_t) b * nch + chunk) * 2 * Kp;
    p[k] = tot;
    p[Kp + k] = m;
  }
}

__global__ void colscale_kernel(double* Sbase, int64_t strideS, int C, int K, int Kp, int checkMax,
                                const double* part, int nch, int stageInLds)
{
  extern __shared__ double sh[]; // [nch][2*Kp] partials (when they fit), then [Kp] totals + [Kp] maxima
  const int chunk = blockIdx.x, b = blockIdx.y;
  double* S = Sbase + (int64_t) b * strideS;
  const int nrg = blockDim.x / Kp;
  const int k = threadIdx.x % Kp, rg = threadIdx.x / Kp;
  const double* p = part + (int64_t) b * nch * 2 * Kp;
  double* tot = sh;
  if (stageInLds)
  {
    // all partials of this buffer in one burst of independent loads, then a fixed-order combine from LDS
    for (int i = threadIdx.x; i < nch * 2 * Kp; i += blockDim.x) sh[i] = p[i];
    __syncthreads();
    p = sh;
    tot = sh + (size_t) nch * 2 * Kp;
  }
  if (rg == 0)
  {
    double t = 0.0, m = -INFINITY;
    for (int j = 0; j < nch; j++) // same order either way
    {
      t += p[j * 2 * Kp + k];
      m = fmax(m, p[j * 2 * Kp + Kp + k]);
    }
    tot[k] = t;
    tot[Kp + k] = (k < K) ? m : -INFINITY;
  }
  __syncthreads();
  if (checkMax)
  {
    double gmax = -INFINITY;
    for (int j = 0; j < Kp; j++) gmax = fmax(gmax, tot[Kp + j]);
    if (!(gmax > kEpsilon)) return; // alg/NMF.hpp:162  if (W.maxCoeff() > epsilon)
  }
  if (k >= K) return;
  const double nrm = sqrt(tot[k]);
  const int rbeg = chunk * kNormRows, rend = min(rbeg + kNormRows, C);
  for (int r = rbeg + rg; r < rend; r += nrg) S[(int64_t) r * Kp + k] /= nrm;
}

int colnorm_scratch_doubles(int C, int Kp, int B) { return ((C + kNormRows - 1) / kNormRows) * 2 * Kp * B; }

void launch_colnorm(double* S, int64_t strideS, int C, int K, int Kp, int B, bool clampEps,
                    bool checkMax, double* scratch, hipStream_t s, const int* rowsTab)
{
  int nrg = 256 / Kp;
  if (nrg < 1) nrg = 1;
  const int threads = nrg * Kp;
  const int nch = (C + kNormRows - 1) / kNormRows;
  dim3 grid((unsigned) nch, (unsigned) B);
  hipLaunchKernelGGL(colstats_kernel, grid, dim3((unsigned) threads), (size_t) 2 * nrg * Kp * sizeof(double), s, S,
                     strideS, C, K, Kp, clampEps ? 1 : 0, scratch, nch, rowsTab);
  // very long factors (c3: 404 chunks x rank 128) do not fit the partials in LDS: combine from global there
  const bool stage = (size_t) (nch + 1) * 2 * Kp * sizeof(double) <= 48 * 1024;
  const size_t shScale = (size_t) ((stage ? nch : 0) + 1) * 2 * Kp * sizeof(double);
  hipLaunchKernelGGL(colscale_kernel, grid, dim3((unsigned) threads), shScale, s, S, strideS, C, K, Kp,
                     checkMax ? 1 : 0, scratch, nch, stage ? 1 : 0);
}

// ---------------------------------------------------------------------------------------
// column sums of a factor (denominator of the other factor's update) for the kernel forms that do not
// accumulate them themselves
// ---------------------------------------------------------------------------------------
constexpr int kSumRows = 256;

__global__ void colsum_part_kernel(const double* Mvbase, int64_t strideM, int R, int Kp, double* part, int nch)
{
  extern __shared__ double sh[]; // [nrg][Kp]
  const int chunk = blockIdx.x, b = blockIdx.y;
  const double* Mv = Mvbase + (int64_t) b * strideM;
  const int nrg = blockDim.x / Kp;
  const int k = threadIdx.x % Kp, rg = threadIdx.x / Kp;
  const int rbeg = chunk * kSumRows, rend = min(rbeg + kSumRows, R);
  double t = 0.0;
  int r = rbeg + rg;
  for (; r + 7 * nrg < rend; r += 8 * nrg) // eight independent loads in flight, summed in row order (at rank 128 a thread's 128
  {                                        // rows, one dependent load after the other, made this pre-pass 43 us on config 3)
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = Mv[(int64_t) (r + u * nrg) * Kp + k];
#pragma unroll
    for (int u = 0; u < 8; u++) t += v[u];
  }
  for (; r < rend; r += nrg) t += Mv[(int64_t) r * Kp + k];
  sh[rg * Kp + k] = t;
  __syncthreads();
  if (rg == 0)
  {
    double tot = 0.0;
    for (int j = 0; j < nrg; j++) tot += sh[j * Kp + k];
    part[((int64_t) b * nch + chunk) * Kp + k] = tot;
  }
}

// zeroSlots: that many further rows of Kp behind the sums are cleared (the other splits' denominator slots); nch == 0 with
// part = [B][Kp] sums already taken: copy them
// (recStride: doubles from one partial record to the next -- Kp for the pre-pass's own records)
__global__ void colsum_combine_kernel(const double* part, int nch, int Kp, double* out, int64_t outStride, int zeroSlots,
                                      int recStride)
{
  const int b = blockIdx.x, k = threadIdx.x;
  const double* p = part + (int64_t) b * nch * recStride + k;
  double t = 0.0;
  int j = 0;
  for (; j + 8 <= nch; j += 8) // eight independent loads in flight, summed in index order
  {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = p[(int64_t) (j + u) * recStride];
#pragma unroll
    for (int u = 0; u < 8; u++) t += v[u];
  }
  for (; j < nch; j++) t += p[(int64_t) j * recStride];
  if (nch == 0) t = part[(int64_t) b * Kp + k];
  out[(int64_t) b * outStride + k] = t;
  for (int z = 1; z <= zeroSlots; z++) out[(int64_t) b * outStride + (int64_t) z * Kp + k] = 0.0;
}

int colsum_scratch_doubles(int R, int Kp, int B) { return ((R + kSumRows - 1) / kSumRows) * Kp * B; }

void launch_colsum(const double* Mv, int64_t strideM, int R, int Kp, int B, double* out, int64_t outStride,
                   double* scratch, hipStream_t s, int zeroSlots)
{
  int nrg = 256 / Kp;
  if (nrg < 1) nrg = 1;
  const int nch = (R + kSumRows - 1) / kSumRows;
  hipLaunchKernelGGL(colsum_part_kernel, dim3((unsigned) nch, (unsigned) B), dim3((unsigned) (nrg * Kp)),
                     (size_t) nrg * Kp * sizeof(double), s, Mv, strideM, R, Kp, scratch, nch);
  hipLaunchKernelGGL(colsum_combine_kernel, dim3((unsigned) B), dim3((unsigned) Kp), 0, s, scratch, nch, Kp, out,
                     outStride, zeroSlots, Kp);
}

void launch_colsum_spread(const double* sums, int Kp, int B, double* out, int64_t outStride, int zeroSlots, hipStream_t s)
{
  hipLaunchKernelGGL(colsum_combine_kernel, dim3((unsigned) B), dim3((unsigned) Kp), 0, s, sums, 0, Kp, out, outStride,
                     zeroSlots, Kp);
}

// ---------------------------------------------------------------------------------------
// Deferred normalisation of W (UpdateArgs::nrm) -- the small kernels between the two factor updates
// ---------------------------------------------------------------------------------------
// Side column: row C-1 of S gets the factor update of kernels_nmf5.hip for a single column, its
// contraction over R cut into nsl slices, one workgroup each.  Pass 1, one thread per row r of the slice:
// ratio[r] = V[r][C-1] / max(sum_j Mv[r][j] S[C-1][j], eps).  Pass 2, thread = (row group, k):
// num_k += ratio[r] Mv[r][k], den_k += Mv[r][k] (operands loaded up front, beside the pass-1 rows, so the
// block pays one memory latency); row groups are combined in fixed order and the slice's (num, den) go
// to scratch for wnorm_combine_kernel.  No atomics: run-to-run bit-identical.
constexpr int kSideUnr = 8;    // rows of a slice per row group (held in registers)
constexpr int kSideSlices = 256; // at most; up to there a launch uses as many as keep a slice within a block's capacity (longer
                                 // buffers: a slice goes through the block in chunks)

__device__ __forceinline__ void wnorm_combine_body(double* Sbase, int64_t strideS, int C, int K, int Kp,
                                                   const double* statPart, int nParts, const double* sidePart,
                                                   int nsl, const double* wold, double* nrm, int b, double* shs, double* shm,
                                                   double* smax, double* shn = nullptr, double* shd = nullptr);
// fuse != nullptr: the LAST slice of a buffer to finish (an arrival ticket per buffer) runs the norm combine for it in
// this launch -- one launch less per iteration.  The slices' partials are published by one agent-scope release per
// workgroup and picked up behind one acquire (MI355X_MICROARCH.md, inter-workgroup visibility); the sums are taken in
// slice order whoever arrives last, so the result does not depend on the order of arrival.
struct SideFuse
{
  int* ticket;            // [B], zero between launches (the last arriver resets it)
  double* S;              // the stationary factor, writable (the side row)
  const double* statPart; // column statistics of the update launch, nParts per buffer
  int nParts, K;
  double* nrmOut;
};
template <int Kp>
__global__ __launch_bounds__(256) void side_slices_kernel(const double* Sbase, int64_t strideS, int C, SideColumn side,
                                                          const double* nrm, double* sidePart, double* wold, SideFuse fuse)
{
  typedef double d2 __attribute__((ext_vector_type(2)));
  extern __shared__ double sh[]; // [nrg][Kp] num, [nrg][Kp] den, [Kp] side row, [slice rows] quotients
  const int slice = blockIdx.x, nsl = gridDim.x, b = blockIdx.y;
  const double* S = Sbase + (int64_t) b * strideS;
  const int nrg = blockDim.x / Kp;
  const int k = threadIdx.x % Kp, rg = threadIdx.x / Kp;
  const double* Mv = side.Mv + (int64_t) b * side.strideM;
  const int RS = (side.R + nsl - 1) / nsl;
  const int r0s = slice * RS, r1s = min(r0s + RS, side.R);
  double* wsh = sh + 2 * nrg * Kp;
  double* ratio = wsh + Kp;
  // the stationary row, normalised the way the update kernel normalises its rows (S = W' / nrm)
  if (rg == 0)
  {
    const double w = S[(int64_t) (C - 1) * Kp + k] / (nrm ? nrm[(int64_t) b * Kp + k] : 1.0);
    wsh[k] = w;
    if (slice == 0) wold[(int64_t) b * Kp + k] = w;
  }
  // pass 1: HP adjacent threads per row of the chunk (one up to rank 64, two at rank 128), KH values each
  constexpr int HP = Kp > 64 ? Kp / 64 : 1, KH = Kp / HP;
  constexpr int NRG = 256 / Kp;
  constexpr int cap = Kp >= 64 ? 256 / HP : (NRG * kSideUnr < 256 ? NRG * kSideUnr : 256);   // rows a block takes at a time (side_chunk_rows)
  constexpr int UG = cap / (NRG * kSideUnr) > 1 ? cap / (NRG * kSideUnr) : 1;               // pass-2 rounds per chunk
  const int rl = (int) threadIdx.x / HP, hp = (int) threadIdx.x % HP;
  double num = 0.0, den = 0.0;
  // a slice longer than one block's capacity (long buffers: more than kSideSlices x cap rows) goes chunk by chunk; up to
  // there the loop runs once and the sums are taken in the order they always were
  for (int r0 = r0s; r0 == r0s || r0 < r1s; r0 += cap)
  {
    const int r1 = min(r0 + cap, r1s);
    double m2[kSideUnr];
#pragma unroll
    for (int u = 0; u < kSideUnr; u++)
    {
      const int r = r0 + u * nrg + rg;
      m2[u] = r < r1 ? Mv[(int64_t) r * Kp + k] : 0.0;
    }
    const int r = r0 + rl;
    double mrow[KH];
    double vr = 0.0;
    if (r < r1)
    {
      const double* m = Mv + (int64_t) r * Kp + hp * KH;
#pragma unroll
      for (int j = 0; j < KH; j += 2)
      {
        const d2 t = *reinterpret_cast<const d2*>(m + j);
        mrow[j] = t[0];
        mrow[j + 1] = t[1];
      }
      vr = side.vcol[(int64_t) b * side.strideV + r];
    }
    __syncthreads();                                   // the side row is in the LDS; the chunk before is done with `ratio`
    {
      double q0 = 0.0, q1 = 0.0;
      if (r < r1)
      {
#pragma unroll
        for (int j = 0; j < KH; j += 2)
        {
          q0 = fma(mrow[j], wsh[hp * KH + j], q0);
          q1 = fma(mrow[j + 1], wsh[hp * KH + j + 1], q1);
        }
      }
      double q = q0 + q1;
      if (HP == 2) q += __shfl_xor(q, 1);
      if (r < r1 && hp == 0) ratio[rl] = vr / fmax(q, kEpsilon);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kSideUnr; u++)
    {
      const int rr = r0 + u * nrg + rg;
      num = fma(rr < r1 ? ratio[rr - r0] : 0.0, m2[u], num);
      den += m2[u];
    }
#pragma unroll 1
    for (int g = 1; g < UG; g++)
    {
      if (r0 + g * kSideUnr * nrg >= r1) break;
#pragma unroll
      for (int u = 0; u < kSideUnr; u++)
      {
        const int rr = r0 + (g * kSideUnr + u) * nrg + rg;
        m2[u] = rr < r1 ? Mv[(int64_t) rr * Kp + k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kSideUnr; u++)
      {
        const int rr = r0 + (g * kSideUnr + u) * nrg + rg;
        num = fma(rr < r1 ? ratio[rr - r0] : 0.0, m2[u], num);
        den += m2[u];
      }
    }
  }
  sh[rg * Kp + k] = num;
  sh[(nrg + rg) * Kp + k] = den;
  __syncthreads();
  if (rg == 0)
  {
    double n = 0.0, d = 0.0;
    for (int j = 0; j < nrg; j++)
    {
      n += sh[j * Kp + k];
      d += sh[(nrg + j) * Kp + k];
    }
    double* p = sidePart + ((int64_t) b * nsl + slice) * 2 * Kp;
    p[k] = n;
    p[Kp + k] = d;
  }
  if (!fuse.ticket) return;
  __shared__ int isLast;
  __shared__ double cshs[1024], cshm[1024], csmax[128];
  __syncthreads();                                   // the slice's partial (and slice 0's wold) have been stored
  if (threadIdx.x == 0)
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int old = __hip_atomic_fetch_add(fuse.ticket + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    isLast = old == nsl - 1 ? 1 : 0;
    if (isLast)
    {
      __hip_atomic_store(fuse.ticket + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
  if (!isLast) return;
  wnorm_combine_body(fuse.S, strideS, C, fuse.K, Kp, fuse.statPart, fuse.nParts, sidePart, nsl, wold, fuse.nrmOut, b, cshs, cshm,
                     csmax);
}

// One workgroup per buffer, thread = (part group pg, k): adds the column statistics of the W update (one part per
// wavefront, or per finalize chunk when the contraction was split) -- each thread a contiguous run of parts in
// index order, the runs combined in fixed order -- and the side-column slices in slice order; writes the side
// row (S[C-1][k] = S_old[C-1][k] num_k / max(den_k, eps), not normalised like every other row of W') and the new nrm:
// alg/NMF.hpp:162  if (W.maxCoeff() > epsilon) W.colwise().normalize()  ->  nrm_k = sqrt(sum_c W'[c][k]^2), else 1.
__device__ __forceinline__ void wnorm_combine_body(double* Sbase, int64_t strideS, int C, int K, int Kp,
                                                   const double* statPart, int nParts, const double* sidePart,
                                                   int nsl, const double* wold, double* nrm, int b, double* shs, double* shm,
                                                   double* smax, double* shn, double* shd)
{
  const int npg = blockDim.x / Kp;
  const int k = threadIdx.x % Kp, pg = threadIdx.x / Kp;
  const int per = (nParts + npg - 1) / npg;
  const int p0 = pg * per, p1 = min(nParts, p0 + per);
  double t = 0.0, m = -INFINITY;
  {
    const double* p = statPart + (int64_t) b * nParts * 2 * Kp;
    int j = p0;
    for (; j + 8 <= p1; j += 8)
    {
      double vs[8], vm[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { vs[u] = p[(int64_t) (j + u) * 2 * Kp + k]; vm[u] = p[(int64_t) (j + u) * 2 * Kp + Kp + k]; }
#pragma unroll
      for (int u = 0; u < 8; u++) { t += vs[u]; m = fmax(m, vm[u]); }
    }
    for (; j < p1; j++) { t += p[(int64_t) j * 2 * Kp + k]; m = fmax(m, p[(int64_t) j * 2 * Kp + Kp + k]); }
  }
  shs[threadIdx.x] = t;
  shm[threadIdx.x] = m;
  double n = 0.0, d = 0.0, wo = 0.0;
  // (a 1024-thread block -- long factors, launch_wnorm_combine -- deals the slices over its part groups as well)
  const bool dealSlices = blockDim.x > 256;
  const int sper = dealSlices ? (nsl + npg - 1) / npg : nsl;
  const int sl0 = dealSlices ? min(nsl, pg * sper) : 0, sl1 = dealSlices ? min(nsl, sl0 + sper) : nsl;
  if (sidePart && (pg == 0 || dealSlices))
  {
    for (int j0 = sl0; j0 < sl1; j0 += 16) // sixteen slices' loads in flight at a time, summed in slice order
    {
      double pn[16], pd[16];
#pragma unroll
      for (int j = 0; j < 16; j++)
      {
        const double* p = sidePart + ((int64_t) b * nsl + min(j0 + j, nsl - 1)) * 2 * Kp;
        pn[j] = p[k];
        pd[j] = p[Kp + k];
      }
#pragma unroll
      for (int j = 0; j < 16; j++)
        if (j0 + j < sl1) { n += pn[j]; d += pd[j]; }
    }
    if (pg == 0) wo = wold[(int64_t) b * Kp + k];
  }
  if (dealSlices)
  {
    shn[threadIdx.x] = n;
    shd[threadIdx.x] = d;
  }
  __syncthreads();
  if (pg == 0)
  {
    t = shs[k];
    m = shm[k];
    for (int g = 1; g < npg; g++) { t += shs[g * Kp + k]; m = fmax(m, shm[g * Kp + k]); }
    if (dealSlices)
      for (int g = 1; g < npg; g++) { n += shn[g * Kp + k]; d += shd[g * Kp + k]; }
    if (sidePart)
    {
      const double wnew = (k < K) ? (wo * n) / fmax(d, kEpsilon) : 0.0;
      Sbase[(int64_t) b * strideS + (int64_t) (C - 1) * Kp + k] = wnew;
      t += wnew * wnew;
      m = fmax(m, wnew);
    }
    smax[k] = (k < K) ? m : -INFINITY;
  }
  __syncthreads();
  if (pg == 0)
  {
    double gmax = -INFINITY;
    for (int j = 0; j < Kp; j++) gmax = fmax(gmax, smax[j]);
    nrm[(int64_t) b * Kp + k] = (k < K && gmax > kEpsilon) ? sqrt(t) : 1.0;
  }
}

// Long factors (config 3: 512 statistics records + 256 side-column slices of 2 KB per channel, two channels): ONE workgroup
// per buffer pulling 1.5 MB through its CU took 25.8 us per iteration.  kPreGroups workgroups per buffer add up a sixteenth of
// the records each -- thread (sub, k) a contiguous run in index order, the runs in fixed order through the LDS -- and leave
// kPreGroups records per buffer in the SAME two layouts, which the 256-thread combine then reads.  No atomics, no tickets.
constexpr int kPreGroups = 16;   // (32 measured: the pre-reduction 8.9 -> 7.6 us on config 3, the combine behind it 7.1 -> 9.0: profiles/r05/c3_pre32.txt)
// colOut != nullptr: the groups also add up the rows [0, C - 1) of the new W' they are dealt (what the H update behind divides
// by: alg/NMF.hpp:169), one record of Kp sums per group; the combine adds the records and the side row.
__global__ __launch_bounds__(512) void wnorm_prereduce_kernel(int Kp, const double* statPart, int nParts, const double* sidePart,
                                                              int nsl, double* statOut, double* sideOut, const double* Sbase,
                                                              int64_t strideS, int C, double* colOut)
{
  __shared__ double sa[512], sb[512];
  const int g = blockIdx.x, b = blockIdx.y;
  const int nsub = blockDim.x / Kp, k = threadIdx.x % Kp, sub = threadIdx.x / Kp;
  auto run = [&](int n, int& j0, int& j1) {
    const int perG = (n + kPreGroups - 1) / kPreGroups;
    const int g0 = min(n, g * perG), g1 = min(n, g0 + perG);
    const int perS = (g1 - g0 + nsub - 1) / nsub;
    j0 = min(g1, g0 + sub * perS);
    j1 = min(g1, j0 + perS);
  };
  // pass 0: the statistics (sum x^2, max); pass 1: the side-column slices (numerator, denominator)
  for (int pass = 0; pass < 2; pass++)
  {
    const double* src = pass == 0 ? statPart : sidePart;
    const int n = pass == 0 ? nParts : nsl;
    if (!src) continue;
    const double* p = src + (int64_t) b * n * 2 * Kp;
    int j0, j1;
    run(n, j0, j1);
    double t = 0.0, m = pass == 0 ? -INFINITY : 0.0;
    for (int j = j0; j < j1; j += 8)
    {
      double va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; u++)
      {
        const int jj = min(j + u, j1 - 1);
        va[u] = p[(int64_t) jj * 2 * Kp + k];
        vb[u] = p[(int64_t) jj * 2 * Kp + Kp + k];
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (j + u < j1)
        {
          t += va[u];
          m = pass == 0 ? fmax(m, vb[u]) : m + vb[u];
        }
    }
    __syncthreads();
    sa[threadIdx.x] = t;
    sb[threadIdx.x] = m;
    __syncthreads();
    if (sub == 0)
    {
      for (int q = 1; q < nsub; q++)
      {
        t += sa[q * Kp + k];
        m = pass == 0 ? fmax(m, sb[q * Kp + k]) : m + sb[q * Kp + k];
      }
      double* o = (pass == 0 ? statOut : sideOut) + ((int64_t) b * kPreGroups + g) * 2 * Kp;
      o[k] = t;
      o[Kp + k] = m;
    }
  }
  if (colOut)
  {
    const double* S = Sbase + (int64_t) b * strideS;
    int j0, j1;
    run(C - 1, j0, j1);
    double t = 0.0;
    for (int j = j0; j < j1; j += 8)
    {
      double va[8];
#pragma unroll
      for (int u = 0; u < 8; u++) va[u] = S[(int64_t) min(j + u, j1 - 1) * Kp + k];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (j + u < j1) t += va[u];
    }
    __syncthreads();
    sa[threadIdx.x] = t;
    __syncthreads();
    if (sub == 0)
    {
      for (int q = 1; q < nsub; q++) t += sa[q * Kp + k];
      colOut[((int64_t) b * kPreGroups + g) * Kp + k] = t;
    }
  }
}

// cs.part != nullptr: the column sums of the new W' -- the pre-reduction's kPreGroups records per buffer plus the side row this
// workgroup has just written -- go straight into the denominator slots of the H update behind (one or two launches of it:
// slot 0 the sums, `zero` further rows of Kp cleared: the layout of launch_colsum)
struct ColsumOut
{
  const double* part = nullptr;   // [B][kPreGroups][Kp]
  double* out1 = nullptr;
  int64_t stride1 = 0;
  int zero1 = 0;
  double* out2 = nullptr;
  int64_t stride2 = 0;
  int zero2 = 0;
};
template <int THREADS>
__global__ __launch_bounds__(THREADS) void wnorm_combine_kernel(double* Sbase, int64_t strideS, int C, int K, int Kp,
                                                                const double* statPart, int nParts, const double* sidePart,
                                                                int nsl, const double* wold, double* nrm, ColsumOut cs)
{
  __shared__ double shs[THREADS], shm[THREADS], smax[128];
  __shared__ double shn[THREADS > 256 ? THREADS : 1], shd[THREADS > 256 ? THREADS : 1];
  wnorm_combine_body(Sbase, strideS, C, K, Kp, statPart, nParts, sidePart, nsl, wold, nrm, (int) blockIdx.x, shs, shm, smax, shn,
                     shd);
  if (cs.part && (int) threadIdx.x < Kp)
  {
    const int b = blockIdx.x, k = threadIdx.x;
    const double* p = cs.part + (int64_t) b * kPreGroups * Kp + k;
    double v[kPreGroups];
#pragma unroll
    for (int g = 0; g < kPreGroups; g++) v[g] = p[(int64_t) g * Kp];
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < kPreGroups; g++) t += v[g];
    t += Sbase[(int64_t) b * strideS + (int64_t) (C - 1) * Kp + k];   // (the side row: this thread wrote it in the body)
    cs.out1[(int64_t) b * cs.stride1 + k] = t;
    for (int z = 1; z <= cs.zero1; z++) cs.out1[(int64_t) b * cs.stride1 + (int64_t) z * Kp + k] = 0.0;
    if (cs.out2)
    {
      cs.out2[(int64_t) b * cs.stride2 + k] = t;
      for (int z = 1; z <= cs.zero2; z++) cs.out2[(int64_t) b * cs.stride2 + (int64_t) z * Kp + k] = 0.0;
    }
  }
}

// ---- side column and norm combine in ONE launch, one workgroup of 1024 threads per buffer (round 4) -----------------------
// For corpora (many buffers of a few hundred frames) the two launches above cost 10 + 6 us per iteration of the bench shard,
// most of it the latency of their dependent round trips: side-column slices -> partials in memory -> kernel boundary ->
// combine.  Here one workgroup does a buffer's side column over ALL frames and then its norm, with nothing in between
// leaving the CU.  Lane layout of the contraction: the Kp / 2 16-byte pieces of a row of the moving factor lie on adjacent
// lanes (coalesced loads, every row read once), FPB = 1024 / (Kp / 2) rows per pass; a row's quotient is formed by a butterfly
// of DPP exchange-adds inside its lane group (all lanes end up with the same bits: the operands of every add are the same
// two numbers in either order), its contribution to num / den stays in the lane's registers, and the FPB row groups are
// added in fixed order through the LDS at the end.  No atomics, no tickets: run-to-run bit-identical.
template <int CTRL>
__device__ __forceinline__ double side_dpp(double x)
{
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xffffffff), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
}
// sum over the PPR adjacent lanes of a row (PPR = 8 .. 64), every lane of the group receives it
template <int PPR>
__device__ __forceinline__ double side_group_sum(double q)
{
  q += side_dpp<0xB1>(q);                           // quad_perm [1,0,3,2]
  q += side_dpp<0x4E>(q);                           // quad_perm [2,3,0,1]
  q += side_dpp<0x141>(q);                          // row_half_mirror: the other quad of the half row
  if constexpr (PPR >= 16) q += side_dpp<0x140>(q); // row_mirror: the other half row
  if constexpr (PPR >= 32) q += __shfl_xor(q, 16);
  if constexpr (PPR >= 64) q += __shfl_xor(q, 32);
  return q;
}
// v / d for d >= eps: reciprocal seed, one Newton step, quotient, one residual correction (rounding error only)
__device__ __forceinline__ double side_div(double v, double d)
{
  double y = __builtin_amdgcn_rcp(d);
  y = __builtin_fma(__builtin_fma(-d, y, 1.0), y, y);
  const double r = v * y;
  return __builtin_fma(__builtin_fma(-d, r, v), y, r);
}
constexpr int kSideNormUnr = 8; // rows per lane in flight

template <int Kp>
__global__ __launch_bounds__(1024) void side_norm_kernel(double* Sbase, int64_t strideS, int C, int K, SideColumn side,
                                                         const double* statPart, int nParts, double* nrm)
{
  constexpr int PPR = Kp / 2, FPB = 1024 / PPR, NJ = 1024 / Kp; // FPB == 2 NJ
  __shared__ double wsh[Kp], smax[Kp];
  __shared__ double shN[FPB * Kp], shD[FPB * Kp];
  const int b = blockIdx.x, tid = threadIdx.x;
  double* S = Sbase + (int64_t) b * strideS;
  const double* Mv = side.Mv + (int64_t) b * side.strideM;
  const double* vcol = side.vcol + (int64_t) b * side.strideV;
  const int p = tid % PPR, g = tid / PPR;
  // Requests first, arithmetic later: the old side row, its norm and the first pass of rows leave before anything waits;
  // every further pass is requested before the one in hand is worked on, and the statistics parts the update launch left
  // (one per wavefront; summed in index order) are requested ahead of the last pass, whose arithmetic covers their latency.
  const int R = side.R;
  auto request = [&](int r0, d2 (&h)[kSideNormUnr], double (&v)[kSideNormUnr]) {
#pragma unroll
    for (int u = 0; u < kSideNormUnr; u++)
    {
      const int r = r0 + u * FPB + g;
      h[u] = d2{0.0, 0.0};
      v[u] = 0.0;
      if (r < R)
      {
        h[u] = *reinterpret_cast<const d2*>(Mv + (int64_t) r * Kp + 2 * p);
        v[u] = vcol[r];
      }
    }
  };
  double sraw = 0.0, nr = 1.0;
  if (tid < Kp)
  {
    sraw = S[(int64_t) (C - 1) * Kp + tid];
    nr = nrm[(int64_t) b * Kp + tid];
  }
  d2 h[kSideNormUnr];
  double v[kSideNormUnr];
  request(0, h, v);
  // the stationary row, normalised the way the update kernel normalises its rows (S = W' / nrm)
  if (tid < Kp) wsh[tid] = sraw / nr;
  __syncthreads();
  const double w0 = wsh[2 * p], w1 = wsh[2 * p + 1];
  double n0 = 0.0, n1 = 0.0, d0 = 0.0, d1 = 0.0;
  auto work = [&](const d2 (&h)[kSideNormUnr], const double (&v)[kSideNormUnr]) {
#pragma unroll
    for (int u = 0; u < kSideNormUnr; u++)
    {
      // (rows past the end: h = 0, v = 0 -> 0 / eps = 0, nothing added)
      const double q = side_group_sum<PPR>(__builtin_fma(h[u][0], w0, h[u][1] * w1));
      const double ratio = side_div(v[u], fmax(q, kEpsilon));
      n0 = __builtin_fma(ratio, h[u][0], n0);
      n1 = __builtin_fma(ratio, h[u][1], n1);
      d0 += h[u][0];
      d1 += h[u][1];
    }
  };
#pragma unroll 1
  for (int r0 = FPB * kSideNormUnr; r0 < R; r0 += FPB * kSideNormUnr)
  {
    d2 hn[kSideNormUnr];
    double vn[kSideNormUnr];
    request(r0, hn, vn);
    work(h, v);
#pragma unroll
    for (int u = 0; u < kSideNormUnr; u++) { h[u] = hn[u]; v[u] = vn[u]; }
  }
  const double* pp = statPart + (int64_t) b * nParts * 2 * Kp;
  double vs[8], vm[8];
#pragma unroll
  for (int u = 0; u < 8; u++)
  {
    vs[u] = 0.0;
    vm[u] = -INFINITY;
    if (tid < Kp && u < nParts) { vs[u] = pp[(int64_t) u * 2 * Kp + tid]; vm[u] = pp[(int64_t) u * 2 * Kp + Kp + tid]; }
  }
  work(h, v);
  double st = 0.0, sm = -INFINITY;
  if (tid < Kp)
  {
#pragma unroll
    for (int u = 0; u < 8; u++) { st += vs[u]; sm = fmax(sm, vm[u]); }   // (parts past nParts: + 0, max with -inf)
    for (int j = 8; j < nParts; j++) { st += pp[(int64_t) j * 2 * Kp + tid]; sm = fmax(sm, pp[(int64_t) j * 2 * Kp + Kp + tid]); }
  }
  shN[g * Kp + 2 * p] = n0; shN[g * Kp + 2 * p + 1] = n1;
  shD[g * Kp + 2 * p] = d0; shD[g * Kp + 2 * p + 1] = d1;
  __syncthreads();
  {
    // row groups j and j + NJ, then (below) the NJ sums in index order
    const int k = tid % Kp, j = tid / Kp;
    const double a = shN[j * Kp + k] + shN[(j + NJ) * Kp + k], c = shD[j * Kp + k] + shD[(j + NJ) * Kp + k];
    shN[j * Kp + k] = a;
    shD[j * Kp + k] = c;
  }
  __syncthreads();
  if (tid < Kp)
  {
    double n = 0.0, d = 0.0;
#pragma unroll
    for (int j0 = 0; j0 < NJ; j0 += 8)
    {
      double pn[8], pd[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { pn[j] = shN[(j0 + j) * Kp + tid]; pd[j] = shD[(j0 + j) * Kp + tid]; }
#pragma unroll
      for (int j = 0; j < 8; j++) { n += pn[j]; d += pd[j]; }
    }
    // the side row (not normalised, like every other row of W') and alg/NMF.hpp:162 for the whole column
    const double wnew = (tid < K) ? (wsh[tid] * n) / fmax(d, kEpsilon) : 0.0;
    S[(int64_t) (C - 1) * Kp + tid] = wnew;
    st += wnew * wnew;
    sm = fmax(sm, wnew);
    smax[tid] = (tid < K) ? sm : -INFINITY;
  }
  __syncthreads();
  if (tid < Kp)
  {
    double gmax = -INFINITY;
    for (int j = 0; j < Kp; j++) gmax = fmax(gmax, smax[j]);
    nrm[(int64_t) b * Kp + tid] = (tid < K && gmax > kEpsilon) ? sqrt(st) : 1.0;
  }
}

// ---- the side column's slices with a row per lane group (round 5) ------------------------------------------------------------
// side_slices_kernel's first pass takes a row per THREAD (two at rank 128): neighbouring lanes read 512 bytes apart, and the
// rows are read a second time for the sums -- 19.5 us per iteration for config 3's 53 MB of H (2.7 TB/s).  This is
// side_norm_kernel's lane layout on a slice of the rows: the Kp / 2 16-byte pieces of a row on adjacent lanes (one coalesced
// kilobyte per wavefront load at rank 128), the row's quotient by the DPP butterfly, its contribution to (num, den) kept in
// the lane's registers, every row read once, eight rows per lane in flight; the 1024 / (Kp / 2) row groups are added in fixed
// order through the LDS and the slice's record goes where side_slices_kernel leaves it.  Ranks 64 and 128.
template <int Kp>
__global__ __launch_bounds__(1024) void side_rows_kernel(const double* Sbase, int64_t strideS, int C, SideColumn side,
                                                        const double* nrm, double* sidePart, double* wold)
{
  // (1024 threads: 16 rows per pass at rank 128, eight passes in flight -- a slice of config 3's 101 rows is ONE round trip; with
  //  256 threads it was four dependent batches and 27.8 us)
  constexpr int PPR = Kp / 2, FPB = 1024 / PPR;
  __shared__ double wsh[Kp];
  __shared__ double shN[FPB * Kp], shD[FPB * Kp];
  const int slice = blockIdx.x, nsl = gridDim.x, b = blockIdx.y, tid = threadIdx.x;
  const double* S = Sbase + (int64_t) b * strideS;
  const double* Mv = side.Mv + (int64_t) b * side.strideM;
  const double* vcol = side.vcol + (int64_t) b * side.strideV;
  const int RS = (side.R + nsl - 1) / nsl;
  const int r0s = slice * RS, r1s = min(r0s + RS, side.R);
  const int p = tid % PPR, g = tid / PPR;
  auto request = [&](int r0, d2 (&h)[kSideNormUnr], double (&v)[kSideNormUnr]) {
#pragma unroll
    for (int u = 0; u < kSideNormUnr; u++)
    {
      const int r = r0 + u * FPB + g;
      h[u] = d2{0.0, 0.0};
      v[u] = 0.0;
      if (r < r1s)
      {
        h[u] = *reinterpret_cast<const d2*>(Mv + (int64_t) r * Kp + 2 * p);
        v[u] = vcol[r];
      }
    }
  };
  double sraw = 0.0, nr = 1.0;
  if (tid < Kp)
  {
    sraw = S[(int64_t) (C - 1) * Kp + tid];
    if (nrm) nr = nrm[(int64_t) b * Kp + tid];
  }
  d2 h[kSideNormUnr];
  double v[kSideNormUnr];
  request(r0s, h, v);
  // the stationary row, normalised the way the update kernel normalises its rows (S = W' / nrm)
  if (tid < Kp)
  {
    const double w = sraw / nr;
    wsh[tid] = w;
    if (slice == 0) wold[(int64_t) b * Kp + tid] = w;
  }
  __syncthreads();
  const double w0 = wsh[2 * p], w1 = wsh[2 * p + 1];
  double n0 = 0.0, n1 = 0.0, d0 = 0.0, d1 = 0.0;
  auto work = [&](const d2 (&h)[kSideNormUnr], const double (&v)[kSideNormUnr]) {
#pragma unroll
    for (int u = 0; u < kSideNormUnr; u++)
    {
      // (rows past the end: h = 0, v = 0 -> 0 / eps = 0, nothing added)
      const double q = side_group_sum<PPR>(__builtin_fma(h[u][0], w0, h[u][1] * w1));
      const double ratio = side_div(v[u], fmax(q, kEpsilon));
      n0 = __builtin_fma(ratio, h[u][0], n0);
      n1 = __builtin_fma(ratio, h[u][1], n1);
      d0 += h[u][0];
      d1 += h[u][1];
    }
  };
#pragma unroll 1
  for (int r0 = r0s + FPB * kSideNormUnr; r0 < r1s; r0 += FPB * kSideNormUnr)
  {
    d2 hn[kSideNormUnr];
    double vn[kSideNormUnr];
    request(r0, hn, vn);
    work(h, v);
#pragma unroll
    for (int u = 0; u < kSideNormUnr; u++) { h[u] = hn[u]; v[u] = vn[u]; }
  }
  work(h, v);
  shN[g * Kp + 2 * p] = n0; shN[g * Kp + 2 * p + 1] = n1;
  shD[g * Kp + 2 * p] = d0; shD[g * Kp + 2 * p + 1] = d1;
  __syncthreads();
  if (tid < Kp)
  {
    double n = 0.0, d = 0.0;
#pragma unroll
    for (int j = 0; j < FPB; j++) { n += shN[j * Kp + tid]; d += shD[j * Kp + tid]; }
    double* o = sidePart + ((int64_t) b * nsl + slice) * 2 * Kp;
    o[tid] = n;
    o[Kp + tid] = d;
  }
}

// W = W' / nrm in memory, nrm = 1: leaves the deferred form (after the last iteration, before anything
// outside the two update kernels reads W)
__global__ void wnorm_apply_kernel(double* Sbase, int64_t strideS, int C, int Kp, double* nrm)
{
  const int b = blockIdx.y;
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t) C * Kp)
  {
    double* p = Sbase + (int64_t) b * strideS + i;
    *p = *p / nrm[(int64_t) b * Kp + (i % Kp)];
  }
}
__global__ void fill_ones_kernel(double* p, int64_t n)
{
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 1.0;
}

// slices of the contraction per buffer: as few as keep a slice within one block's capacity (8 rows per row group in
// pass 2, HP threads per row in pass 1), 16 at least
// rows a side-column block takes at a time: what pass 1 covers with one thread (two at rank 128) per row; pass 2 holds
// kSideUnr rows per row group in registers and, from rank 64 on (4 / 2 row groups), goes through the chunk in several such
// rounds (with two groups of 128 threads and one round, long buffers went 16 rows at a time: 37 us per iteration on config 3)
static int side_chunk_rows(int Kp)
{
  const int nrg = 256 / Kp;
  return Kp >= 64 ? 256 / (Kp > 64 ? Kp / 64 : 1) : std::min(nrg * kSideUnr, 256);
}
static int side_slices_for(int R, int Kp)
{
  const int cap = side_chunk_rows(Kp);
  // FLUHIP_SIDE_SLICES=n (tests): at most n slices, so that ordinary shapes take several chunks per slice
  static const int cap2 = [] { const char* e = fluhip::ab_getenv("FLUHIP_SIDE_SLICES"); return e ? std::max(1, std::atoi(e)) : 0; }();
  if (cap2 > 0) return std::min(std::min(kSideSlices, cap2), std::max(1, (R + cap - 1) / cap));
  return std::min(kSideSlices, std::max(16, (R + cap - 1) / cap));
}
bool nmf_side_column_supported(int R, int C, int Kp)
{
  (void) R; // any length: slices beyond a block's capacity are taken in chunks
  return C % 16 == 1 && C > 16 && (Kp == 16 || Kp == 32 || Kp == 64 || Kp == 128);
}
static int64_t wnorm_scratch_base_doubles(int Kp, int B, int nStrips) { return (int64_t) B * (nStrips * 2 * Kp + kSideSlices * 2 * Kp + Kp) + B; } // + arrival tickets
// ... and behind them the pre-reduced records of long factors: [B][kPreGroups][2][Kp] statistics, as many of the side column, [B][kPreGroups][Kp] column sums
int wnorm_scratch_doubles(int Kp, int B, int nStrips) { return (int) (wnorm_scratch_base_doubles(Kp, B, nStrips) + (int64_t) B * kPreGroups * 5 * Kp); }

static_assert(kSideSlices == 4 * kSideFromHSlots, "two generations of 128 slices: 64 of partials, then the old side row");
double* wnorm_side_part(double* scratch, int Kp, int B, int nStrips, int gen)
{
  return scratch + (int64_t) B * nStrips * 2 * Kp + (int64_t) gen * B * (2 * kSideFromHSlots) * 2 * Kp;
}
double* wnorm_side_wold(double* scratch, int Kp, int B, int nStrips, int gen)
{
  return wnorm_side_part(scratch, Kp, B, nStrips, gen) + (int64_t) B * kSideFromHSlots * 2 * Kp;
}

// ---- which form the norm combine of a W update takes: ONE table (round 6) -------------------------------
// Every number the choice depends on, with where it was measured.  wnorm_combine_form() is the only reader; the launcher below
// and the planner's report (fluhip_debug_wnorm_form, tests/test_plan_table.py) both go through it.
struct WnormFormTable
{
  // side column + combine as ONE launch, one workgroup per buffer (side_norm_kernel): enough buffers to occupy the part, few
  // enough rows that a workgroup walks them in a few passes, few statistics parts (round 2: the bench shard's form before the H
  // update took both over)
  int sideNormMinBuffers = 64, sideNormMaxParts = 64;
  int64_t sideNormMaxRowDoubles = (int64_t) 64 * 2048;
  // one workgroup per buffer walks the records with 1024 threads instead of 256 from here on (config 3: 512 parts + 256
  // slices at rank 128 took 38 us with two part groups)
  int manyParts = 128, manySlices = 64;
  // sixteen workgroups per buffer pre-reduce the records (wnorm_prereduce_kernel) where they are long AND many: from ~320 KB of
  // records per buffer (round 5, config 3: 25.8 -> 6 + 9 us); where they are few kilobytes -- one 10 s buffer at rank 32: 129
  // records of 512 B -- the one-workgroup combine is the shorter chain (9.0 us for the two launches against ~6)
  int64_t bigRecordDoubles = 40000;
  // ... or wherever the H update behind would otherwise run its column-sum pre-pass (windows of a rank-128 corpus: colsum_part +
  // colsum_combine 14 us against ~6 for the pre-reduction)
  bool preForColumnSums = true;
  int maxPreRank = 512;
};
constexpr WnormFormTable kWnormForms{};

bool wnorm_side_norm_shape(int B, int nStrips, int R, int Kp)
{
  return B >= kWnormForms.sideNormMinBuffers && nStrips <= kWnormForms.sideNormMaxParts &&
         (int64_t) R * Kp <= kWnormForms.sideNormMaxRowDoubles;
}
// sidePhase as in launch_wnorm_combine; nsl = the side column's slices (0: none); wantCol = the caller gave column-sum slots
WnormForm wnorm_combine_form(int Kp, int B, int nStrips, int nsl, int sideR, int sidePhase, bool wantCol)
{
  static const bool oneLaunch = [] { const char* e = fluhip::ab_getenv("FLUHIP_SIDE_NORM"); return e ? std::atoi(e) != 0 : true; }();
  // FLUHIP_WNORM_PRE=0 (A/B build): the one-workgroup combine of rounds 3 - 4 for long factors; =2: the pre-reduction only where
  // the records are long, as in its first form
  static const int preEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_WNORM_PRE"); return e ? std::atoi(e) : 1; }();
  if (nsl > 0 && sidePhase == 0 && oneLaunch && wnorm_side_norm_shape(B, nStrips, sideR, Kp)) return WnormForm::SideNormOneLaunch;
  if (sidePhase == 1) return WnormForm::SideOnly;
  const bool many = nStrips > kWnormForms.manyParts || nsl > kWnormForms.manySlices;
  const bool bigRecords = (int64_t) (nStrips + nsl) * Kp > kWnormForms.bigRecordDoubles;
  const bool preAlways = kWnormForms.preForColumnSums && preEnv != 2;
  if (preEnv != 0 && Kp <= kWnormForms.maxPreRank && ((many && (bigRecords || wantCol)) || (wantCol && preAlways)))
    return WnormForm::PreReduce;
  return many ? WnormForm::Combine1024 : WnormForm::Combine256;
}

bool launch_wnorm_combine(double* S, int64_t strideS, int C, int K, int Kp, int B, int nStrips, double* scratch,
                          double* nrm, const SideColumn* side, hipStream_t s, int sidePhase, int sideSlices, int sideGen,
                          const WnormColsum* colsum)
{
  // sideSlices > 0 (with sidePhase 2): the side column's partials are there already, sideSlices per buffer, dense -- left by
  // the H update in front (UpdateArgs::sideOut)
  // sidePhase 0: side column launch, then the combine launch, both on s.  1: the side column launch alone (the caller runs
  // it on a second stream beside the update launch -- it reads nothing the update writes).  2: the combine launch alone.
  double* statPart = scratch;
  double* sidePart = scratch + (int64_t) B * nStrips * 2 * Kp;
  double* wold = sidePart + (int64_t) B * kSideSlices * 2 * Kp;
  if (sideGen >= 0) // (sideSlices > 0: an H update's partials, in the generation it filled)
  {
    sidePart = wnorm_side_part(scratch, Kp, B, nStrips, sideGen);
    wold = wnorm_side_wold(scratch, Kp, B, nStrips, sideGen);
  }
  int nsl = side ? (sideSlices > 0 ? sideSlices : side_slices_for(side->R, Kp)) : 0;
  const bool wantCol = colsum && colsum->out1;
  const WnormForm form = wnorm_combine_form(Kp, B, nStrips, nsl, side ? side->R : 0, sidePhase, wantCol);
  // corpora: one launch, one workgroup per buffer (side_norm_kernel).  FLUHIP_SIDE_NORM=0 (A/B build): the two launches.
  if (form == WnormForm::SideNormOneLaunch)
  {
    const dim3 grid((unsigned) B), block(1024);
    if (Kp == 16) hipLaunchKernelGGL(side_norm_kernel<16>, grid, block, 0, s, S, strideS, C, K, *side, statPart, nStrips, nrm);
    else if (Kp == 32) hipLaunchKernelGGL(side_norm_kernel<32>, grid, block, 0, s, S, strideS, C, K, *side, statPart, nStrips, nrm);
    else if (Kp == 64) hipLaunchKernelGGL(side_norm_kernel<64>, grid, block, 0, s, S, strideS, C, K, *side, statPart, nStrips, nrm);
    else hipLaunchKernelGGL(side_norm_kernel<128>, grid, block, 0, s, S, strideS, C, K, *side, statPart, nStrips, nrm);
    return false;
  }
  if (side && sidePhase != 2)
  {
    const int nrg = 256 / Kp;
    nsl = side_slices_for(side->R, Kp);
    const dim3 grid((unsigned) nsl, (unsigned) B), block((unsigned) (nrg * Kp));
    const size_t sh = (size_t) (2 * nrg * Kp + Kp + side_chunk_rows(Kp)) * sizeof(double);
    // FLUHIP_SIDE_FUSED=1: the norm combine inside this launch, by the last slice of a buffer to arrive.  Built in round 3
    // to save a launch per iteration and measured the other way round on the bench shard, same box, alternating
    // (profiles/r03/ab_side_fused.txt): 58 us between the two factor updates instead of 18, 217.9 k against 224.9 k
    // buffer-iterations/s -- 2048 small workgroups each pay an agent-scope release (and the last ones an acquire) of
    // ~1.7 us, eight deep per CU; the kernel boundary is the cheaper synchronisation here.  Off by default.
    static const bool fused = [] { const char* e = fluhip::ab_getenv("FLUHIP_SIDE_FUSED"); return e && std::atoi(e) == 1; }();
    SideFuse fz{nullptr, nullptr, nullptr, 0, 0, nullptr};
    if (fused)
      fz = SideFuse{reinterpret_cast<int*>(wold + (int64_t) B * Kp), S, statPart, nStrips, K, nrm};
    // FLUHIP_SIDE_ROWS=0 (A/B build): the row-per-thread slices kernel at ranks 64 / 128 too
    static const bool rowsForm = [] { const char* e = fluhip::ab_getenv("FLUHIP_SIDE_ROWS"); return e ? std::atoi(e) != 0 : true; }();
    if (rowsForm && !fused && Kp == 128)
      hipLaunchKernelGGL(side_rows_kernel<128>, grid, dim3(1024), 0, s, S, strideS, C, *side, nrm, sidePart, wold);
    else if (rowsForm && !fused && Kp == 64)
      hipLaunchKernelGGL(side_rows_kernel<64>, grid, dim3(1024), 0, s, S, strideS, C, *side, nrm, sidePart, wold);
    else
    if (Kp == 16) hipLaunchKernelGGL(side_slices_kernel<16>, grid, block, sh, s, S, strideS, C, *side, nrm, sidePart, wold, fz);
    else if (Kp == 32) hipLaunchKernelGGL(side_slices_kernel<32>, grid, block, sh, s, S, strideS, C, *side, nrm, sidePart, wold, fz);
    else if (Kp == 64) hipLaunchKernelGGL(side_slices_kernel<64>, grid, block, sh, s, S, strideS, C, *side, nrm, sidePart, wold, fz);
    else hipLaunchKernelGGL(side_slices_kernel<128>, grid, block, sh, s, S, strideS, C, *side, nrm, sidePart, wold, fz);
    if (fused) return false;
  }
  if (sidePhase == 1) return false;
  // one block per buffer sums nStrips statistics parts and nsl side slices per component (1024 threads where that is long), or
  // sixteen workgroups per buffer pre-reduce them first: kWnormForms
  if (form == WnormForm::PreReduce)
  {
    double* statOut = scratch + wnorm_scratch_base_doubles(Kp, B, nStrips);
    double* sideOut = statOut + (int64_t) B * kPreGroups * 2 * Kp;
    double* colPart = sideOut + (int64_t) B * kPreGroups * 2 * Kp;
    ColsumOut cs;
    if (colsum && colsum->out1)
    {
      cs.part = colPart;
      cs.out1 = colsum->out1; cs.stride1 = colsum->stride1; cs.zero1 = colsum->zero1;
      cs.out2 = colsum->out2; cs.stride2 = colsum->stride2; cs.zero2 = colsum->zero2;
    }
    hipLaunchKernelGGL(wnorm_prereduce_kernel, dim3(kPreGroups, (unsigned) B), dim3(512), 0, s, Kp, statPart, nStrips,
                       side ? sidePart : nullptr, nsl, statOut, sideOut, S, strideS, C, cs.part ? colPart : nullptr);
    hipLaunchKernelGGL(wnorm_combine_kernel<256>, dim3((unsigned) B), dim3(256), 0, s, S, strideS, C, K, Kp, statOut,
                       kPreGroups, side ? sideOut : nullptr, side ? kPreGroups : 0, wold, nrm, cs);
    return cs.part != nullptr;
  }
  else if (form == WnormForm::Combine1024)
    hipLaunchKernelGGL(wnorm_combine_kernel<1024>, dim3((unsigned) B), dim3(1024), 0, s, S, strideS, C, K, Kp, statPart,
                       nStrips, side ? sidePart : nullptr, nsl, wold, nrm, ColsumOut{});
  else
    hipLaunchKernelGGL(wnorm_combine_kernel<256>, dim3((unsigned) B), dim3(256), 0, s, S, strideS, C, K, Kp, statPart,
                       nStrips, side ? sidePart : nullptr, nsl, wold, nrm, ColsumOut{});
  return false;
}

// The side column's slices carry (numerator, denominator) per component, and the denominator of bin C - 1 of the W update IS
// the column sum of the moving factor H that every other bin divides by (alg/NMF.hpp:160): where the update launch takes its
// column sums from a pre-pass (arrays of rank 128), the side-column launch (sidePhase 1) goes IN FRONT of it and this adds
// its slices' denominators into the update's denominator slots -- the pre-pass's own sweep over H (config 3: 11 - 16 us per
// iteration) is not needed.  Same slots as launch_colsum: (buffer, split 0) the sums, zeroSlots further rows cleared.
int wnorm_side_slices(int R, int Kp) { return side_slices_for(R, Kp); }
// (one workgroup of 1024 threads per buffer, thread = (run of slices, k): 128 threads walking 256 records of 2 KB one after the
//  other -- colsum_combine_kernel with a record stride -- took 10.6 us on config 3)
__global__ __launch_bounds__(1024) void colsum_side_kernel(const double* den, int nsl, int Kp, double* out, int64_t outStride,
                                                           int zeroSlots)
{
  // (blockIdx.y: a block of KB = Kp / gridDim.y columns per workgroup -- four workgroups per buffer at rank 128)
  __shared__ double sh[1024];
  const int KB = Kp / (int) gridDim.y;
  const int b = blockIdx.x, kl = threadIdx.x % KB, pg = threadIdx.x / KB, npg = blockDim.x / KB;
  const int k = (int) blockIdx.y * KB + kl;
  const int per = (nsl + npg - 1) / npg, j0 = min(nsl, pg * per), j1 = min(nsl, j0 + per);
  const double* p = den + (int64_t) b * nsl * 2 * Kp + k;
  double t = 0.0;
  for (int j = j0; j < j1; j += 8)
  {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = p[(int64_t) min(j + u, j1 - 1) * 2 * Kp];
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (j + u < j1) t += v[u];
  }
  sh[threadIdx.x] = t;
  __syncthreads();
  if (pg == 0)
  {
    for (int g = 1; g < npg; g++) t += sh[g * KB + kl];
    out[(int64_t) b * outStride + k] = t;
    for (int z = 1; z <= zeroSlots; z++) out[(int64_t) b * outStride + (int64_t) z * Kp + k] = 0.0;
  }
}
void launch_colsum_from_side(const double* scratch, int Kp, int B, int nStrips, int nsl, double* out, int64_t outStride,
                             int zeroSlots, hipStream_t s)
{
  const double* sidePart = scratch + (int64_t) B * nStrips * 2 * Kp;   // (the layout of launch_wnorm_combine, no generation)
  const int kblocks = (Kp >= 128 && nsl >= 64) ? 4 : 1;                            // column blocks (workgroups) per buffer
  const int KB = Kp / kblocks;
  const int threads = std::max(KB, std::min(1024, KB * std::max(1, nsl / 8)));     // eight slices or more per run
  hipLaunchKernelGGL(colsum_side_kernel, dim3((unsigned) B, (unsigned) kblocks), dim3((unsigned) threads), 0, s, sidePart + Kp, nsl, Kp,
                     out, outStride, zeroSlots);
}

void launch_wnorm_apply(double* S, int64_t strideS, int C, int Kp, int B, double* nrm, hipStream_t s)
{
  const int64_t n = (int64_t) C * Kp;
  hipLaunchKernelGGL(wnorm_apply_kernel, dim3((unsigned) ((n + 255) / 256), (unsigned) B), dim3(256), 0, s, S, strideS,
                     C, Kp, nrm);
  launch_fill_ones(nrm, (int64_t) B * Kp, s);
}

// dst[c][r] = src[r][c] for a rows x cols float matrix (row strides lds / ldd): 32 x 32 tiles through the LDS
__global__ void transpose_f32_kernel(const float* src, int64_t lds_, float* dst, int64_t ldd, int rows, int64_t cols)
{
  __shared__ float tile[32][33];
  const int64_t c0 = (int64_t) blockIdx.x * 32;
  const int r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = src[(int64_t) (r0 + j) * lds_ + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < cols && r0 + tx < rows) dst[(c0 + j) * ldd + r0 + tx] = tile[tx][j];
}
void launch_transpose_f32(const float* src, int64_t lds_, float* dst, int64_t ldd, int rows, int64_t cols, hipStream_t s)
{
  // (the long dimension on grid x)
  hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned) ((cols + 31) / 32), (unsigned) ((rows + 31) / 32)), dim3(256), 0, s, src,
                     lds_, dst, ldd, rows, cols);
}

void launch_fill_ones(double* p, int64_t n, hipStream_t s)
{
  hipLaunchKernelGGL(fill_ones_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, p, n);
}

// ---------------------------------------------------------------------------------------
// layout plumbing (all tiny next to the updates)
// ---------------------------------------------------------------------------------------
__global__ void scatter_factor_kernel(const double* src, int64_t strideSrc, double* dst,
                                      int64_t strideDst, int rows, int K, int Kp, int kMajor, const int* rowsTab)
{
  const int b = blockIdx.y;
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t) rows * K) return;
  int row, k;
  if (kMajor) { k = (int) (idx / rows); row = (int) (idx % rows); }
  else { row = (int) (idx / K); k = (int) (idx % K); }
  if (rowsTab && row >= rowsTab[b]) return; // ragged corpora: rows past the buffer's own count are padding and stay zero
  dst[(int64_t) b * strideDst + (int64_t) row * Kp + k] = src[(int64_t) b * strideSrc + idx];
}

void launch_scatter_factor(const double* src, int64_t strideSrc, double* dst, int64_t strideDst,
                           int rows, int K, int Kp, int B, bool srcIsKMajor, hipStream_t s, const int* rowsTab)
{
  const int64_t total = (int64_t) rows * K;
  dim3 g((unsigned) ((total + 255) / 256), (unsigned) B);
  hipLaunchKernelGGL(scatter_factor_kernel, g, dim3(256), 0, s, src, strideSrc, dst, strideDst,
                     rows, K, Kp, srcIsKMajor ? 1 : 0, rowsTab);
}

__global__ void scatter_factor_f32_kernel(const float* src, int64_t strideSrc, double* dst,
                                          int64_t strideDst, int rows, int K, int Kp, const int* rowsTab)
{
  const int b = blockIdx.y;
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t) rows * K) return;
  const int k = (int) (idx / rows), row = (int) (idx % rows);
  if (rowsTab && row >= rowsTab[b]) return; // ragged corpora: padding rows stay zero
  dst[(int64_t) b * strideDst + (int64_t) row * Kp + k] = (double) src[(int64_t) b * strideSrc + idx];
}

void launch_scatter_factor_f32(const float* src, int64_t strideSrc, double* dst,
                               int64_t strideDst, int rows, int K, int Kp, int B, hipStream_t s, const int* rowsTab)
{
  const int64_t total = (int64_t) rows * K;
  dim3 g((unsigned) ((total + 255) / 256), (unsigned) B);
  hipLaunchKernelGGL(scatter_factor_f32_kernel, g, dim3(256), 0, s, src, strideSrc, dst,
                     strideDst, rows, K, Kp, rowsTab);
}

template <typename OutT>
__global__ void gather_w_kernel(const double* Wf, int64_t strideW, OutT* out, int64_t strideOut,
                                int F, int K, int Kp)
{
  // 32 x 32 tile transpose through LDS so both sides are coalesced
  __shared__ double tile[32][33];
  const int b = blockIdx.z;
  const int f0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 256 threads: ty 0..7
  for (int j = ty; j < 32; j += 8)
  {
    const int f = f0 + j, k = k0 + tx;
    tile[j][tx] = (f < F && k < K) ? Wf[(int64_t) b * strideW + (int64_t) f * Kp + k] : 0.0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
  {
    const int k = k0 + j, f = f0 + tx;
    if (k < K && f < F) out[(int64_t) b * strideOut + (int64_t) k * F + f] = (OutT) tile[tx][j];
  }
}

void launch_gather_w_f64(const double* Wf, int64_t strideW, double* W1, int64_t strideOut, int F,
                         int K, int Kp, int B, hipStream_t s)
{
  dim3 g((unsigned) ((F + 31) / 32), (unsigned) ((K + 31) / 32), (unsigned) B);
  hipLaunchKernelGGL(gather_w_kernel<double>, g, dim3(256), 0, s, Wf, strideW, W1, strideOut, F, K, Kp);
}

void launch_gather_w_f32(const double* Wf, int64_t strideW, float* bases, int64_t strideOut,
                         int F, int K, int Kp, int B, hipStream_t s)
{
  dim3 g((unsigned) ((F + 31) / 32), (unsigned) ((K + 31) / 32), (unsigned) B);
  hipLaunchKernelGGL(gather_w_kernel<float>, g, dim3(256), 0, s, Wf, strideW, bases, strideOut, F, K, Kp);
}

__global__ void gather_h_kernel(const double* H1, int64_t strideH, double* out, int64_t strideOut,
                                int T, int K, int Kp)
{
  const int b = blockIdx.y;
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t) T * K) return;
  const int t = (int) (idx / K), k = (int) (idx % K);
  out[(int64_t) b * strideOut + idx] = H1[(int64_t) b * strideH + (int64_t) t * Kp + k];
}

void launch_gather_h_f64(const double* H1, int64_t strideH, double* out, int64_t strideOut, int T,
                         int K, int Kp, int B, hipStream_t s)
{
  const int64_t total = (int64_t) T * K;
  dim3 g((unsigned) ((total + 255) / 256), (unsigned) B);
  hipLaunchKernelGGL(gather_h_kernel, g, dim3(256), 0, s, H1, strideH, out, strideOut, T, K, Kp);
}

// max over the valid T x K block of H1, one workgroup per buffer
__global__ void hmax_kernel(const double* H1, int64_t strideH, int T, int K, int Kp, double* out)
{
  __shared__ double sh[256];
  const double* H = H1 + (int64_t) blockIdx.x * strideH;
  double mx = -INFINITY;
  for (int64_t idx = threadIdx.x; idx < (int64_t) T * K; idx += blockDim.x)
  {
    const int t = (int) (idx / K), k = (int) (idx % K);
    mx = fmax(mx, H[(int64_t) t * Kp + k]);
  }
  sh[threadIdx.x] = mx;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1)
  {
    if ((int) threadIdx.x < st) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

__global__ void acts_kernel(const double* H1, int64_t strideH, float* acts, int64_t strideOut,
                            int T, int K, int Kp, const double* hmax)
{
  // acts[k][t] = float(H1[t][k]) * float(1/max): tile transpose for coalescing on both sides
  __shared__ double tile[32][33];
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float scale = (float) (1. / hmax[b]); // clients/nrt/NMFClient.hpp:291,298
  for (int j = ty; j < 32; j += 8)
  {
    const int t = t0 + j, k = k0 + tx;
    tile[j][tx] = (t < T && k < K) ? H1[(int64_t) b * strideH + (int64_t) t * Kp + k] : 0.0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
  {
    const int k = k0 + j, t = t0 + tx;
    if (k < K && t < T)
    {
      float x = (float) tile[tx][j];
      x *= scale;
      acts[(int64_t) b * strideOut + (int64_t) k * T + t] = x;
    }
  }
}

void launch_acts_f32(const double* H1, int64_t strideH, float* acts, int64_t strideOut, int T,
                     int K, int Kp, int B, double* scratchMax, hipStream_t s)
{
  hipLaunchKernelGGL(hmax_kernel, dim3((unsigned) B), dim3(256), 0, s, H1, strideH, T, K, Kp, scratchMax);
  dim3 g((unsigned) ((T + 31) / 32), (unsigned) ((K + 31) / 32), (unsigned) B);
  hipLaunchKernelGGL(acts_kernel, g, dim3(256), 0, s, H1, strideH, acts, strideOut, T, K, Kp, scratchMax);
}

__global__ void vhat_kernel(const double* Wf, int64_t strideW, const double* H1, int64_t strideH,
                            double* Vhat, int64_t ldV, int64_t strideV, int T, int F, int Kp)
{
  const int b = blockIdx.y;
  const int fblocks = (F + blockDim.x - 1) / blockDim.x;
  const int t = blockIdx.x / fblocks;
  const int f = (blockIdx.x % fblocks) * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const double* w = Wf + (int64_t) b * strideW + (int64_t) f * Kp;
  const double* h = H1 + (int64_t) b * strideH + (int64_t) t * Kp;
  double s = 0.0;
  for (int k = 0; k < Kp; k++) s += w[k] * h[k];
  Vhat[(int64_t) b * strideV + (int64_t) t * ldV + f] = s;
}

void launch_vhat(const double* Wf, int64_t strideW, const double* H1, int64_t strideH,
                 double* Vhat, int64_t ldV, int64_t strideV, int T, int F, int Kp, int B,
                 hipStream_t s)
{
  dim3 g((unsigned) (((F + 255) / 256) * (int64_t) T), (unsigned) B);
  hipLaunchKernelGGL(vhat_kernel, g, dim3(256), 0, s, Wf, strideW, H1, strideH, Vhat, ldV, strideV, T, F, Kp);
}

__global__ void pad_copy_kernel(const double* src, int64_t ldsrc, int64_t strideSrc, double* dst,
                                int64_t lddst, int64_t strideDst, int rows, int cols)
{
  const int b = blockIdx.y;
  const int cblocks = (cols + blockDim.x - 1) / blockDim.x;
  const int r = blockIdx.x / cblocks;
  const int c = (blockIdx.x % cblocks) * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  dst[(int64_t) b * strideDst + (int64_t) r * lddst + c] = src[(int64_t) b * strideSrc + (int64_t) r * ldsrc + c];
}

void launch_pad_copy(const double* src, int64_t ldsrc, int64_t strideSrc, double* dst,
                     int64_t lddst, int64_t strideDst, int rows, int cols, int B, hipStream_t s)
{
  dim3 g((unsigned) (((cols + 255) / 256) * (int64_t) rows), (unsigned) B);
  hipLaunchKernelGGL(pad_copy_kernel, g, dim3(256), 0, s, src, ldsrc, strideSrc, dst, lddst, strideDst, rows, cols);
}

__global__ void clamp_eps_kernel(double* p, int64_t ld, int64_t stride, int rows, int cols)
{
  const int b = blockIdx.y;
  const int cblocks = (cols + blockDim.x - 1) / blockDim.x;
  const int r = blockIdx.x / cblocks;
  const int c = (blockIdx.x % cblocks) * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  double* q = p + (int64_t) b * stride + (int64_t) r * ld + c;
  *q = fmax(*q, kEpsilon);
}

void launch_clamp_eps(double* p, int64_t ld, int64_t stride, int rows, int cols, int B, hipStream_t s)
{
  dim3 g((unsigned) (((cols + 255) / 256) * (int64_t) rows), (unsigned) B);
  hipLaunchKernelGGL(clamp_eps_kernel, g, dim3(256), 0, s, p, ld, stride, rows, cols);
}

__global__ void transpose_kernel(const double* in, int64_t ldin, int64_t strideIn, double* out,
                                 int64_t ldout, int64_t strideOut, int R, int C)
{
  __shared__ double tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
  {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < C) ? in[(int64_t) b * strideIn + (int64_t) r * ldin + c] : 0.0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
  {
    const int c = c0 + j, r = r0 + tx;
    if (c < C && r < R) out[(int64_t) b * strideOut + (int64_t) c * ldout + r] = tile[tx][j];
  }
}

void launch_transpose(const double* in, int64_t ldin, int64_t strideIn, double* out,
                      int64_t ldout, int64_t strideOut, int R, int C, int B, hipStream_t s)
{
  dim3 g((unsigned) ((C + 31) / 32), (unsigned) ((R + 31) / 32), (unsigned) B);
  hipLaunchKernelGGL(transpose_kernel, g, dim3(256), 0, s, in, ldin, strideIn, out, ldout, strideOut, R, C);
}

} // namespace fluhip
