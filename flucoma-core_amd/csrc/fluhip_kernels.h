// fluhip_kernels.h -- internal launch interface between the C-ABI layer (api_*.hip) and the
// gfx950 kernels (kernels_stft.hip, kernels_nmf.hip).  Not installed; not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>

#include "fluhip_env.h"

namespace fluhip {

constexpr double kEpsilon = 2.220446049250313e-16; // util/AlgorithmUtils.hpp:19

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Dynamic LDS above the default limit for `kern`.  A refusal is not dropped: the call leaves its error in the runtime's
// last-error slot (as does the launch that follows and cannot get its LDS), and every C-ABI entry point reads that slot behind
// its launches -- HIPCHK(ctx, hipGetLastError()) -- and returns FLUHIP_ERROR with the HIP message.
template <class Kern>
inline void request_dynamic_lds(Kern kern, size_t bytes)
{
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes) != hipSuccess)
    std::fprintf(stderr, "fluhip: %zu bytes of dynamic LDS refused for a kernel\n", bytes);
}

// ---------------------------------------------------------------------------------------
// HBM layout of one corpus (B equal-shape buffers); everything f64 unless noted.
//   audio  f32 [B][N]
//   mag    [B][Tp][Fp]   row = frame t, f contiguous      ("V col-major F x T" of alg/NMF.hpp:125)
//   magT   [B][Fp][Tp]   row = bin f,  t contiguous       (second copy, so both factor updates
//                                                          stream V with lanes along the
//                                                          contiguous axis)
//   Wf     [B][Fp][Kp]   row = bin f,  k contiguous       (W of alg/NMF.hpp, F x K)
//   H1     [B][Tp][Kp]   row = frame t, k contiguous      (H^T; == output H1 of NMF::process)
// Tp = round_up(T, 32), Fp = round_up(F, 32), Kp = round_up(K, 16); all padding is zero and
// stays zero (zero rows/cols contribute exactly 0 to every contraction).
// ---------------------------------------------------------------------------------------

struct StftArgs
{
  const float* audio;   // [B][n]  (or nullptr when audio64 is used)
  const double* audio64; // [B][n] f64 input variant
  int64_t n;            // samples per buffer
  int64_t audioStride;  // elements between buffers
  int win, fft, hop;
  int T, F;             // frames per buffer, bins
  int B;
  const double* window; // [win]
  const double* twiddle; // [fft/2] interleaved (cos, sin) of e^{-2 pi i j / fft}
  double* mag;          // [B][Tp][Fp] or nullptr
  int64_t magStride;    // per buffer
  int64_t ldMag;        // Fp
  double* spec;         // [B][T][F] interleaved complex, or nullptr
  int64_t specStride;
  double* bigScratch = nullptr; // big_fft_scratch_bytes() of workspace, needed when stft_needs_scratch(win, fft)
  int frameOffset;      // extra sample offset of frame 0 (0 for STFT::process; the buffered feature
                        // clients start (win/hop)*hop - win earlier when hop does not divide win)
  // ragged corpora (block form only): samples of every buffer (device, [B]); n / T are then those of the longest one
  // and frames past a buffer's own (n_b + hop) / hop are left zero
  const int64_t* nTab = nullptr;
};

void launch_stft(const StftArgs& a, hipStream_t s);
// block form (kernels_stft2.hip): the same transform writing the frame-major magnitudes (a.mag) AND the bin-major
// copy magT [B][*][ldMagT] in one pass; false when the shape has no block form (launch_stft + launch_transpose then)
bool launch_stft_block(const StftArgs& a, double* magT, int64_t magTStride, int64_t ldMagT, hipStream_t s);
// power-of-two fft up to 65536; sizes whose frame does not fit the LDS (above 8192) run their passes through a
// global-memory workspace the caller provides
bool stft_supported(int64_t win, int64_t fft);
bool stft_needs_scratch(int64_t win, int64_t fft);
int64_t big_fft_scratch_bytes(int64_t fft, int64_t frames, int64_t* chunkFrames);
double* launch_big_fft_passes(double* bufA, double* bufB, int nc, int fft, const double* twiddle, int nf, hipStream_t s);

// out[b][c][r] = in[b][r][c]; in [R][ldin], out [C][ldout] (valid R x C)
void launch_transpose(const double* in, int64_t ldin, int64_t strideIn, double* out,
                      int64_t ldout, int64_t strideOut, int R, int C, int B, hipStream_t s);

// One wavefront's share of a factor update in work-list mode (ragged corpora: buffers of different lengths in one
// launch).  Column groups [g0, g0 + ng) of buffer `buf`, contraction steps [s0, s1) (4 rows each); partIdx >= 0: the
// accumulators leave as partial `partIdx` of a split contraction ([Cp][Kp] doubles each) for the finalize launch,
// dIdx >= 0: this wavefront also stores the column sums of its steps as denominator partial `dIdx`; partIdx < 0: the
// wavefront covers the whole contraction and writes the result and (W update) its column statistics as part `statIdx`.
// ng == 0: an idle slot of a workgroup.
// grp: up to four consecutive wavefronts of ONE workgroup may share a strip and split its contraction between them; they
// add their accumulators through the LDS in rank order -- no partials in memory, no finalize launch -- and the leader
// (rank 0) carries partIdx / statIdx / dIdx.  grp = leader's wavefront index | rank << 4 | size << 8 | (1 << 16 when
// any group of the workgroup has more than one member: every live wavefront of it then meets at one barrier).
struct WaveDesc
{
  int buf, g0, ng, s0, s1, partIdx, statIdx, dIdx, grp, pad0, pad1, pad2;
};
// whether the staging of an intra-workgroup reduction fits the LDS next to nothing else (NG groups of M accumulators +
// M column sums per lane, four wavefronts)
inline bool nmf_update5_groups_fit(int Kp, int NG) { const int M = Kp / 4; return (NG * M + M) * 512 * 4 <= 160 * 1024; }

struct UpdateArgs
{
  const double* V;  // [B][*][ldv] rows = contraction index r, cols = c (contiguous)
  int64_t ldv, strideV;
  const double* Mv; // moving factor   [B][>=round_up(R,16)][Kp]
  int64_t strideM;
  double* S;        // stationary factor [B][>=round_up(C,32)][Kp], updated in place
  int64_t strideS;
  int R, C, B;
  int Kp;
  int Kc = 0;       // compute rank of the off-size forms (24 on arrays of rank 32; 40 / 48 / 56 on 64; 72 .. 112 in eights on 128: nmf_update5_compute_rank); 0 = Kp
  // split-R mode (single large buffer): partial numerators / denominators
  int nsplit;
  double* part;     // [B][nsplit][Cp][Kp]
  double* dpart;    // [B][nsplit][Kp]
  int64_t Cp;
  // Deferred column normalisation of W (kernels_nmf5.hip only; see "normalisation" in DESIGN.md): W stays in
  // memory as written by its update, W' = W * diag(nrm), with nrm [B][Kp] beside it.
  //   nrmMode 1 (W update, S = W'): the stationary rows are divided by nrm as they are loaded.
  //   nrmMode 2 (H update, Mv = W'): Q = W' (H / nrm)^T, numerator and denominator are divided by nrm.
  const double* nrm = nullptr;
  int nrmMode = 0;
  // W update: per-wavefront column statistics of the rows it wrote, [B][wavesPerBuf][2][Kp] (sum x^2, max x)
  double* statPart = nullptr;
  // rank 65..128: workspace of colsum_scratch_doubles() for the column sums of Mv (see launch_colsum); also
  // needs dpart of at least B * nsplit * Kp doubles
  double* colsumScratch = nullptr;
  // rank 65..128, optional: the column sums of Mv already taken by an earlier launch of the same update ([B][Kp], dense) --
  // copied into the denominator slots instead of a second pre-pass (the tail launch of a two-launch update)
  const double* colsumGiven = nullptr;
  // rank 65..128, optional: slot (buffer, split 0) of dpart holds the column sums of Mv already and the other splits' slots
  // are zero (launch_colsum_from_side, or a combine launch that left them there): the forms that take their column sums from
  // a pre-pass skip it; the forms that accumulate their own overwrite the slots as always
  bool colsumInPlace = false;
  // kernels_nmf5.hip: {launches, shader cycles, 100 MHz ticks} of one wavefront per launch, accumulated (or null)
  long long* clk = nullptr;
  // kernels_nmf5.hip work-list mode: listWGs workgroups of 4 wavefronts, list[4 * listWGs]; listNG = the widest strip;
  // listPartial: the list's wavefronts write split partials (the caller finalizes)
  const WaveDesc* list = nullptr;
  int listWGs = 0, listNG = 0, listPartial = 0;
  // > 0: strips (wavefronts) per buffer instead of the planner's choice for a.B buffers (windows of a larger corpus keep
  // the schedule of one round)
  int stripsOverride = 0;
  // H update of a corpus whose W update keeps its last bin as a side column (SideColumn): leave, beside the new H, every
  // wavefront's share of that side column's contraction for the W update that follows -- sideOut [B][strips][2][Kp] in the
  // layout of the side-column slices (launch_wnorm_combine), sideWold [B][Kp] the old side row W'[R-1] / nrm.
  double* sideOut = nullptr;
  double* sideWold = nullptr;
  // ... and, when the norm combine of the W update in front has not been launched (launch_wnorm_combine), do it in the
  // prologue of every wavefront: cmbStat [B][cmbParts][2][Kp] the W update's column statistics, cmbSide [B][cmbSlices][2][Kp]
  // and cmbWold [B][Kp] the side column's partials / old side row (an earlier launch's sideOut / sideWold), cmbK the rank;
  // the new norms go to cmbNrmOut [B][Kp] (= nrm, which the launch then does not read), the new side row to row R - 1 of
  // cmbRowOut (= Mv).  Return value of launch_nmf_update5: bit 0 side partials left, bit 1 norm combine done.
  const double* cmbStat = nullptr;
  const double* cmbSide = nullptr;
  const double* cmbWold = nullptr;
  double* cmbNrmOut = nullptr;
  double* cmbRowOut = nullptr;
  int cmbParts = 0, cmbSlices = 0, cmbK = 0;
  // Column sums of the moving factor from the neighbouring launches instead of accumulators in the loop (round 6; rank 32,
  // the two-launch iteration of a corpus).  W update: colIn = the side-column partials the H update in front left
  // ([B][colInN][2][Kp], = cmbSide of the H update behind: their denominators are the column sums of H over each wavefront's
  // frames), colOut [B][strips][Kp] takes the column sums of the rows of W' this launch writes.  H update (norm form):
  // colIn = that colOut, colInN = the W update's strips.  Return value of launch_nmf_update5: bit 2 = the launch took its
  // column sums from colIn (and, W update, filled colOut).
  const double* colIn = nullptr;
  double* colOut = nullptr;
  int colInN = 0;
  bool dryRun = false; // nothing is launched: the return value says what a launch with these arguments would do
};

// out[b * outStride + k] = sum_r Mv[b][r][k], r < R: 256-row partials, then a fixed-order combine
int colsum_scratch_doubles(int R, int Kp, int B);
// zeroSlots: rows of Kp behind each buffer's sums that are cleared along with it (out[b * outStride + (1 + z) * Kp + k])
void launch_colsum(const double* Mv, int64_t strideM, int R, int Kp, int B, double* out, int64_t outStride,
                   double* scratch, hipStream_t s, int zeroSlots = 0);
// the same output from sums [B][Kp] taken earlier
void launch_colsum_spread(const double* sums, int Kp, int B, double* out, int64_t outStride, int zeroSlots, hipStream_t s);

// S[c][k] <- S[c][k] * (sum_r (V[r][c] / max(sum_j Mv[r][j] S[c][j], eps)) * Mv[r][k])
//                    / max(sum_r Mv[r][k], eps)
// alg/NMF.hpp:158-161 with (V, Mv, S) = (mag, H1, Wf) and :165-170 with (magT, Wf, H1).
// split-contraction epilogue shared by the kernel forms: S <- S * (sum of the numerator partials) / max(sum of the
// denominator partials, eps), with the deferred-normalisation arithmetic and (statPart) per-chunk column
// statistics [B][update_finalize_parts(C, Kp)][2][Kp] when asked for
void launch_update_finalize(double* S, int64_t strideS, const double* part, const double* dpart, int C, int Kp,
                            int64_t Cp, int nsplit, int B, hipStream_t s, const double* nrm = nullptr,
                            int nrmMode = 0, double* statPart = nullptr, const int* splitTab = nullptr);
// splitTab (device, [B][2] ints, work-list mode): first partial and number of partials of every buffer; `nsplit` is
// then the largest count
int update_finalize_parts(int C, int Kp);
int nmf_update5_waves_per_buffer(int C, int Kp, int B);          // the planner's strips per buffer
int launch_nmf_update5(const UpdateArgs& a, hipStream_t s);     // v_mfma_f64_4x4x4_4b + LDS-DMA operand streaming
// any rank (used above Kp = 128): un-fused, over a materialised ratio matrix (kernels_nmf_wide.hip)
void launch_nmf_update_wide(const UpdateArgs& a, double* scratch, hipStream_t s);
int64_t nmf_update_wide_scratch_doubles(int R, int C, int Kp, int B);
bool nmf_update5_supported(int Kp);

// per column k < K of S [C][Kp]: optional clamp to eps, then (if !checkMax or max(S) > eps)
// divide the column by its L2 norm.  alg/NMF.hpp:150-153 (init) and :162 (after W update).
// rowsTab (device, [B] ints; ragged corpora): the valid rows of every buffer -- the clamp touches only those
void launch_colnorm(double* S, int64_t strideS, int C, int K, int Kp, int B, bool clampEps,
                    bool checkMax, double* scratch, hipStream_t s, const int* rowsTab = nullptr);
int colnorm_scratch_doubles(int C, int Kp, int B);

// ---- deferred normalisation of W (UpdateArgs::nrm) ------------------------------------------------
// Side column: every power-of-two FFT has F = 16 m + 1 bins, so in the W update the Nyquist bin would cost
// each wavefront of the MFMA kernel a whole extra 16-column group (9 instead of 8 at fft 2048).  With a
// side column the update kernel is launched on the first C-1 columns and row C-1 of S is updated by
// launch_wnorm_combine (same formula, scalar FMAs, fixed summation order).
struct SideColumn
{
  const double* vcol; // the R values V[:, C-1] of buffer 0, contiguous (a row of the transposed copy)
  int64_t strideV;    // buffer stride of vcol
  const double* Mv;   // moving factor [B][>=R][Kp] as it was during the update
  int64_t strideM;
  int R;
};
bool nmf_side_column_supported(int R, int C, int Kp);
// after a W update launched with UpdateArgs::statPart = scratch and nStrips wavefronts per buffer:
// [side column ->] new nrm [B][Kp] (1 where alg/NMF.hpp:162 would skip the normalisation)
int wnorm_scratch_doubles(int Kp, int B, int nStrips);
// the side-column launch's slices per buffer (sidePhase 0 / 1 of launch_wnorm_combine), and the column sums of the moving
// factor from those slices' denominators into the update launch's denominator slots (see kernels_nmf.hip)
int wnorm_side_slices(int R, int Kp);
// the shapes whose side column and norm combine are ONE launch (side_norm_kernel, sidePhase 0): they keep that launch
bool wnorm_side_norm_shape(int B, int nStrips, int R, int Kp);
void launch_colsum_from_side(const double* scratch, int Kp, int B, int nStrips, int nsl, double* out, int64_t outStride,
                             int zeroSlots, hipStream_t s);
// colsum (long factors only: the pre-reduced combine): the column sums of the new W' -- what the H update behind divides by --
// written by the combine launch into that update's denominator slots (slot 0 the sums, `zero` further rows of Kp cleared: the
// layout of launch_colsum), for one or two launches of it; launch_wnorm_combine returns true when it did
struct WnormColsum
{
  double* out1 = nullptr;
  int64_t stride1 = 0;
  int zero1 = 0;
  double* out2 = nullptr;
  int64_t stride2 = 0;
  int zero2 = 0;
};
bool launch_wnorm_combine(double* S, int64_t strideS, int C, int K, int Kp, int B, int nStrips, double* scratch,
                          double* nrm, const SideColumn* side, hipStream_t s, int sidePhase = 0, int sideSlices = 0, int sideGen = -1,
                          const WnormColsum* colsum = nullptr);
// Where the side-column partials / the old side row of an H update with UpdateArgs::sideOut live inside `scratch`: two
// areas (gen 0 / 1) inside the slice region, up to 64 slices per buffer each -- an H update that does the norm combine reads
// one generation in its prologue while its own epilogues fill the other.
constexpr int kSideFromHSlots = 64; // slices of side-column partials per buffer and generation (H update's SIDEQ epilogue: one per strip)
double* wnorm_side_part(double* scratch, int Kp, int B, int nStrips, int gen);
// the form the norm combine takes for a shape (kernels_nmf.hip kWnormForms: the one table of its thresholds)
enum class WnormForm : int { SideNormOneLaunch = 0, SideOnly = 1, PreReduce = 2, Combine1024 = 3, Combine256 = 4 };
WnormForm wnorm_combine_form(int Kp, int B, int nStrips, int nsl, int sideR, int sidePhase, bool wantCol);
double* wnorm_side_wold(double* scratch, int Kp, int B, int nStrips, int gen);
// S = S / nrm in memory, nrm = 1
void launch_wnorm_apply(double* S, int64_t strideS, int C, int Kp, int B, double* nrm, hipStream_t s);
void launch_fill_ones(double* p, int64_t n, hipStream_t s);
int nmf_update5_compute_rank(int K, int Kp);  // the off-size forms' compute rank for rank K on arrays of rank Kp (UpdateArgs::Kc): 24 | 40 / 48 / 56 | 72 .. 112 in eights, else Kp
int nmf_update5_strips(int C, int Kp, int B); // wavefronts per buffer launch_nmf_update5 uses (nsplit == 1)
int nmf_update5_max_groups(int Kp);           // widest strip (16-column groups) a wavefront can hold at this rank

// Strip schedule of one large buffer at rank <= 16 (kernels_nmf_strip.hip): a workgroup owns a strip of frames, does
// their H update (alg/NMF.hpp:165-170) against the W it staged (normalised while staging when wPend) and, behind it,
// the strip's share of the next W update's numerator (:158-160); the reduce launch forms W' (:161, un-normalised).
struct StripArgs
{
  const double* V;  // mag [B][Tp][ldv]
  int64_t strideV, ldv;
  double* W;        // [B][Fp][16]
  int64_t strideW;
  double* H;        // [B][Tp][16]
  int64_t strideH;
  double* part;     // nmf_strip_part_doubles() of workspace
  double* nrm;      // [B][16]: written by the strip launch (the column norms it worked with, 1 when !wPend)
  int statGen;      // which of the two generations of column-statistics records describes the W in memory: written by
                    // launch_nmf_strip_wstats (same generation) or by the reduce launch (into the other one)
  int F, T, K, B;
  int doH, doW;     // both 0: only nrm is produced
  int wPend;        // W in memory is the un-normalised W' of the last reduce launch
  // round 5, the W update as a bin-tiled launch (kernels_nmf_bintile.hip): the column statistics then come from its tile records
  // (tileStat: generation statGen of [B][48][nRec], nRec > 0) instead of the reduce launch's, and doW == 2 leaves only the
  // Nyquist bin's numerator partials, one per workgroup, in sideOut [B][workgroups][16]
  const double* tileStat = nullptr;
  int nRec = 0;
  double* sideOut = nullptr;
};
bool nmf_strip_supported(int F, int T, int Kp);
bool nmf_strip_tile_supported(int F, int T, int Kp); // ... with the bin-tiled W update (StripArgs::tileStat)
int nmf_strip_workgroups(int T);
int64_t nmf_strip_part_doubles(int F, int T, int B);
void launch_nmf_strip(const StripArgs& a, hipStream_t s);
void launch_nmf_strip_reduce(const StripArgs& a, hipStream_t s);
void launch_nmf_strip_wstats(const StripArgs& a, hipStream_t s);
// Bin-tiled W update of the same schedule (kernels_nmf_bintile.hip, round 5): a workgroup owns four bins and all frames, no
// partials of the numerator in memory, no reduce launch.  alg/NMF.hpp:158-161; :162 stays deferred (tile records).
struct BinTileArgs
{
  const double* VT;  // magT [B][Fp][ldT]
  int64_t strideVT, ldT;
  double* W;         // [B][Fp][16]
  int64_t strideW;
  const double* H;   // [B][Tp][16]
  int64_t strideH;
  double* work;      // nmf_bintile_doubles(): two generations of tile records, then the side partials
  const double* sidePart; // the side partials the strip launch left (nmf_bintile_side_area()), or null: no Nyquist side row
  int nSideWG;
  int statGen;       // generation of records that describes the W in memory; the launch writes the other one
  int F, K, B, wPend;
};
bool nmf_bintile_supported(int F, int T, int Kp);
int nmf_bintile_records(int F);
int64_t nmf_bintile_doubles(int F, int B, int nSideWG);
double* nmf_bintile_side_area(const BinTileArgs& s);
const double* nmf_bintile_records_ptr(const BinTileArgs& s, int gen);
void launch_nmf_bintile(const BinTileArgs& s, hipStream_t st);
void launch_nmf_bintile_wstats(const BinTileArgs& s, hipStream_t st);
// the W update of the frame-strip schedule as its own launch over bin strips (no numerator partials of the whole matrix, no
// reduce launch); `work`: nmf_binstrip_doubles() doubles, zeroed once
int64_t nmf_binstrip_doubles(int F, int T, int B);
void launch_nmf_binstrip(const StripArgs& a, double* work, hipStream_t s);

// dst[b][row][k] = src[b or 0][k*rows + row] (colMajorSrc) or src[row*K + k]
// rowsTab (device, [B] ints; ragged corpora): only rows < rowsTab[b] of buffer b are written
void launch_scatter_factor(const double* src, int64_t strideSrc, double* dst, int64_t strideDst,
                           int rows, int K, int Kp, int B, bool srcIsKMajor, hipStream_t s, const int* rowsTab = nullptr);
// f32 seed variant: src[b][k][rows] floats (BufferAdaptor channel-major)
void launch_scatter_factor_f32(const float* src, int64_t strideSrc, double* dst,
                               int64_t strideDst, int rows, int K, int Kp, int B, hipStream_t s, const int* rowsTab = nullptr);

// W1[b][k][f] = Wf[b][f][k]  (f64 and f32 flavours; f32 == clients/nrt/NMFClient.hpp:281-282)
void launch_gather_w_f64(const double* Wf, int64_t strideW, double* W1, int64_t strideOut, int F,
                         int K, int Kp, int B, hipStream_t s);
void launch_gather_w_f32(const double* Wf, int64_t strideW, float* bases, int64_t strideOut,
                         int F, int K, int Kp, int B, hipStream_t s);
// H1out[b][t][k] = H1[b][t][k] without the Kp padding
void launch_gather_h_f64(const double* H1, int64_t strideH, double* out, int64_t strideOut, int T,
                         int K, int Kp, int B, hipStream_t s);
// acts[b][k][t] = float(H1[t][k]) * float(1 / max H1)   (clients/nrt/NMFClient.hpp:289-298)
void launch_acts_f32(const double* H1, int64_t strideH, float* acts, int64_t strideOut, int T,
                     int K, int Kp, int B, double* scratchMax, hipStream_t s);
// Vhat[b][t][f] = sum_k Wf[f][k] H1[t][k]   (alg/NMF.hpp:182)
void launch_vhat(const double* Wf, int64_t strideW, const double* H1, int64_t strideH,
                 double* Vhat, int64_t ldV, int64_t strideV, int T, int F, int Kp, int B,
                 hipStream_t s);
// dst[b][t][f] (ld) = src[b][t*ldsrc + f] : strided host-layout copy into the padded layout
// p[b][r][c] = max(p[b][r][c], eps) over the valid rows x cols only (the padding stays zero)
void launch_clamp_eps(double* p, int64_t ld, int64_t stride, int rows, int cols, int B, hipStream_t s);
void launch_pad_copy(const double* src, int64_t ldsrc, int64_t strideSrc, double* dst,
                     int64_t lddst, int64_t strideDst, int rows, int cols, int B, hipStream_t s);

// MelBands / MFCC over magnitudes
struct FeatArgs
{
  const double* mag;    // [B][*][ldMag]
  int64_t magStride, ldMag;
  int T, F, B, win;
  const double* filtT;  // [F][bandsPad], bin-major mel filter bank, zero padded
  int nBands, bandsPad;
  // the same filter bank restricted to each band's support (a triangle is non-zero on a few dozen bins of the F):
  // band b covers bins bandLo[b] .. bandLo[b] + maxLen - 1 with weights wpack[i][b] (i-major, zero past the support)
  const int* bandLo;    // [bandsPad]
  const double* wpack;  // [maxLen][bandsPad]
  int maxLen;
  int magNorm, usePower, logOutput;
  const double* dct;    // [nDct][nBands] or nullptr (MelBands output)
  int nDct, startCoeff;
  int nOut;             // features per frame written (nBands, or nCoefs)
  float* out;           // [B][nOut][T]
};
void launch_features(const FeatArgs& a, hipStream_t s);
// fused form (kernels_stft2.hip): STFT -> mel bands [-> DCT] without the magnitudes leaving the chip.  up / dn / slot
// [64 * stft_features_bins_per_lane(fft)]: rising- and falling-edge weight of every bin and the interval boundary a
// bin closes (or -1); false when the shape has no fused form
struct StftArgs;
bool launch_stft_features(const StftArgs& a, const FeatArgs& f, const double* up, const double* dn, const short* slot,
                          hipStream_t s);
int stft_features_bins_per_lane(int fft);

// resynthesis: masked inverse STFT of component k with overlap-add
struct ResynthArgs
{
  const double* spec;   // [T][F] interleaved complex
  const double* Wf;     // [Fp][Kp]
  const double* H1;     // [Tp][Kp]
  const double* Vhat;   // [T][ldV]
  int64_t ldV;
  int Kp, k;
  int win, fft, hop, T, F;
  const double* window;
  const double* twiddle;
  double* frames;       // scratch [T][win] windowed, scaled inverse frames
  double* out;          // [n] f64 overlap-added, normalised, trimmed
  float* out32;         // [n] or nullptr
  int64_t n;
  int64_t outStride = 0; // samples between the components' outputs (0: n)
  int64_t trim;         // leading samples dropped: win/2 for ISTFT::process, `padding` for BufSTFT
  int nComp = 1;        // components k .. k + nComp - 1 in one launch: frames [nComp][T][win], out / out32 [nComp][n]
  double* bigScratch = nullptr; // set (big_fft_scratch_bytes(fft, T)) when stft_needs_scratch(win, fft): global-memory passes
};
// Wf == nullptr: no ratio mask (plain inverse STFT of `spec`)
void launch_resynth(const ResynthArgs& a, hipStream_t s);
// dst[c][r] = src[r][c], floats (rows x cols, row strides lds / ldd)
void launch_transpose_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int rows, int64_t cols, hipStream_t s);

// Batched resynthesis (kernels_stft2.hip, resynth_seq_kernel): every component of every buffer of a corpus in one launch.
// A wavefront owns (buffer, component, a run of hop slots) and walks the frames that cover the run IN ORDER: masked
// spectrum -> inverse real transform in registers (the forward passes of the STFT kernels on the conjugate) -> window ->
// added to 2 win doubles of running overlap-add state held in registers; after each frame the oldest `hop` samples are
// final (divided by the window-power normaliser) and leave as floats, the state shifts by hop.  No frames in memory,
// no second pass, and a sample's frames are added in increasing frame order like the reference's sequential overlap-add.
struct ResynthBatchArgs
{
  const double* spec;    // [B][T][F] interleaved complex
  int64_t specStride;
  const double* mult;    // [B][T][F]  1 / max(Vhat, eps)   (launch_resynth_mult)
  int64_t multStride;
  const double* Wt;      // [B][Kp][F]  W transposed (component-major rows)
  int64_t wtStride;
  const double* H1;      // [B][Tp][Kp]
  int64_t hStride;
  int Kp, K;
  int win, fft, hop, T, F, B;
  const double* window;  // [max(win, fft)], zero past win
  const double* twiddle; // [fft/2] e^{-2 pi i m / fft}
  const double* nrmTab;  // [hop] normaliser of a position covered by win / hop frames (resynth_normaliser_table)
  float* out32;          // [B][K][outStride]
  int64_t n, outStride;
  int64_t trim;
  const int64_t* nTab = nullptr; // ragged corpora: samples per buffer (T, n are then the longest buffer's)
};
bool resynth_batch_supported(int win, int fft, int hop);
void launch_resynth_normaliser(const double* window, int win, int hop, double* tab /*[hop]*/, hipStream_t s);
bool launch_resynth_batch(const ResynthBatchArgs& a, hipStream_t s);
// Wt[b][k][f] = W[f][k] (Kp x F per buffer) and mult[b][t][f] = 1 / max(sum_k W[f][k] H[t][k], eps) (alg/NMF.hpp:33-42 estimate's V-hat,
// alg/RatioMask.hpp:39-41)
void launch_resynth_mult(const double* Wf, int64_t strideW, const double* H1, int64_t strideH, double* Wt, double* mult,
                         int T, int F, int K, int Kp, int B, hipStream_t s);

// BufSTFT plumbing: spec [T][F] c128 <-> float magnitude / phase, bin-major [F][T]
void launch_spec_to_magphase(const double* spec, int T, int F, float* mag, float* phase, hipStream_t s);
void launch_polar_to_spec(const float* mag, const float* phase, int T, int F, double* spec, hipStream_t s);

// one-sided Jacobi SVD of the n x T matrix whose rows are G's (kernels_svd.hip): G <- S V^T row by row (unsorted),
// Jt <- U^T, norms <- singular values in row order; returns the number of sweeps or -1
int launch_jacobi_svd(double* G, int64_t ldg, int n, int T, double* Jt, double* norms, unsigned* flag, int maxSweeps,
                      hipStream_t s);

} // namespace fluhip
