// kernels_nmf5.hip -- NMF factor update, v_mfma_f64_4x4x4_4b_f64 + LDS-DMA operand streaming.
//
// One kernel serves both factor updates (alg/NMF.hpp:158-161 and :165-170 are the same contraction with the factor
// roles swapped; the register tiling and the measured lane map of the 4x4x4 MFMA are in DESIGN.md section 3).  What changes is how operands reach the wavefront: both streamed operands --
// the 4-row slab of V under the wavefront's column strip and the 4 x Kp slab of the moving factor
// -- are copied HBM/L2 -> LDS by `global_load_lds_dwordx4` (no VGPRs, asynchronous, counted on
// vmcnt) into a private per-wavefront ring of NS stages, NS-1 steps ahead of their use, and read
// back with ds_read just before the MFMAs that consume them.  That removes the two VGPR prefetch
// sets of the register-staged kernel (room for the stage-wise quotient at 9 groups) and makes the
// prefetch depth a matter of LDS, not registers.  Rings are private to a wavefront, so there are
// no workgroup barriers: ordering is the issuing wavefront's own counted `s_waitcnt vmcnt(N)`.
//
// LDS images
//   V stage : [4 rows][NG*16 doubles], linear copy of the strip          (ds_read_b64, conflict free)
//   Mv stage: [4 rows][Kp doubles], 16-byte chunks of row r rotated by r  (pos = (c + r) mod Kp/2)
//             so that the two register distributions the MFMAs need -- (row x, chunks of y) and
//             (row y, chunks of x) -- both hit distinct banks within a ds_read_b128 lane group.
//             The rotation is applied on the per-lane *source* address; the LDS destination of an
//             LDS-DMA is always base + lane*16.
//
// Pipeline forms (template MODE; the launcher picks, FLUHIP_K5_MODE overrides for A/B):
//   0  reads and ring refill grouped between the phases                      (first LDS-DMA form; NG = 1 strips)
//   1  reads of step s+1 and refill of slot s interleaved into step s's MFMAs, second operand set   (Kp <= 32)
//   2  the same with one operand set refilled in place behind its consumers                          (Kp >= 64)
// Other template switches: DS = 0 takes the column sums of Mv from a pre-pass (Kp = 128 has no registers for
// them), INSTR = 1 adds cycle counters and a timeline (tools/phase_breakdown.py), WPS = 2 is the two-wavefronts-
// per-SIMD experiment.  Deferred column normalisation (UpdateArgs::nrm), per-wavefront column statistics, the
// LDS-staged result stores and the split-contraction partial stores live in the common prologue / epilogue.
// DESIGN.md section 3 and profiles/r01/update_kernel_notes.md carry the measurements behind each of these.
#include "fluhip_kernels.h"
#include "recip_tree.h"

#include <cstdlib>

namespace fluhip {

typedef double d2 __attribute__((ext_vector_type(2)));

struct Upd5Args
{
  const double* V;
  int64_t ldv, strideV;
  const double* Mv;
  int64_t strideM;
  double* S;
  int64_t strideS;
  int R, C, B;
  int nGroups, wavesPerBuf, wgPerBuf, nSteps, nsplit, stepsPerSplit;
  double* part;
  double* dpart;
  int64_t Cp;
  int xcdMap;
  const double* nrm;
  int nrmMode;
  double* statPart;
  long long* clk; // {launches, shader cycles, 100 MHz ticks} of wavefront 0 of workgroup 0, accumulated per launch (or null)
  // work-list mode (ragged corpora, fluhip_kernels.h WaveDesc): wavefront w of workgroup i takes list[4 i + w]; the uniform
  // mapping above (wavesPerBuf, nsplit, stepsPerSplit, xcdMap) is then unused
  const WaveDesc* list;
  // SIDEQ instantiations (H update): per-wavefront partials of the NEXT W update's side column, [B][wavesPerBuf][2][KP], and the
  // normalised old side row [B][KP] (UpdateArgs::sideOut / sideWold)
  double* sidePart;
  double* sideWold;
  // SIDEQ & 2: the norm combine of the W update in front, done by every wavefront in its prologue (UpdateArgs::cmb*)
  const double* cmbStat;
  const double* cmbSide;
  const double* cmbWold;
  double* cmbNrmOut;
  double* cmbRowOut;
  int cmbParts, cmbSlices, cmbK;
  // DS == 2 (round 6): the column sums of Mv come from what the launch IN FRONT left, and this launch leaves the column sums
  // of the rows it writes for the launch BEHIND -- no accumulators in the loop (UpdateArgs::colIn / colOut).
  //   W update: colIn = the side-column partials of the H update in front, [B][colInN][2][KPM] (their denominators ARE the
  //             column sums of the new H over each wavefront's frames); colOut [B][wavesPerBuf][KPM]
  //   H update (NORMQ): colIn = the colOut of the W update in front, [B][colInN][KPM]; the side row its prologue forms is added
  const double* colIn;
  double* colOut;
  int colInN;
};

// The quotients V / max(Q, eps) of the hot loop: v_rcp_f64 (2^29 ulp, i.e. ~23 bits) -> one Newton step (2^-46) -> product.
// Relative error <= ~1.4e-14 per quotient; the device factors stay where they were against the oracle (1e-14 after 200
// iterations, tools/parity_levels.py: the difference is dominated by the summation orders).  -DFLUHIP_QUOTIENT_CORRECTION=1
// adds the residual correction (error ~2^-96 before the final rounding: IEEE division for all practical purposes) at 2 of 7
// FP64 operations per element: 93 of the 2 470 cycles of a 4-row step, 3 % of the job.
#ifndef FLUHIP_QUOTIENT_CORRECTION
#define FLUHIP_QUOTIENT_CORRECTION 0
#endif
constexpr bool kQuotientCorrection = FLUHIP_QUOTIENT_CORRECTION != 0;
#ifndef FLUHIP_SHARED_RECIPROCAL
#define FLUHIP_SHARED_RECIPROCAL 1
#endif
constexpr bool kSharedReciprocal = FLUHIP_SHARED_RECIPROCAL != 0;
// quotients per reciprocal of the hot loop: 2 = pairs (rounds 2 - 5), 4 / 8 = a product tree (ratio_phase)
#ifndef FLUHIP_RCP_GROUP
#define FLUHIP_RCP_GROUP 4
#endif
constexpr int kRcpGroup = FLUHIP_RCP_GROUP;
#ifndef FLUHIP_QV_M10
#define FLUHIP_QV_M10 1
#endif
static_assert(kRcpGroup == 2 || kRcpGroup == 4 || kRcpGroup == 8, "quotients per reciprocal");
// The results of a launch leave with write-through (sc1) stores: with plain stores the kernel ends on tens of MB of
// dirty L2 lines that the end-of-kernel release has to write back before the next launch may start
// (MI355X_MICROARCH.md "publish-large": 8.2 vs 3.0 us for 64 KB per workgroup).  -DFLUHIP_EPILOGUE_SC1=0: plain stores.
#ifndef FLUHIP_EPILOGUE_SC1
#define FLUHIP_EPILOGUE_SC1 1
#endif
// THE TWO WAIT STATES BEHIND THE STORE ARE LOAD-BEARING (round 5; DESIGN section 3 "The store hazard behind the inline asm").
// gfx940-class parts read the data registers of a store of MORE than 64 bits over several cycles after issue, and a VALU
// instruction that writes one of them within two wait states corrupts what is stored.  The compiler's hazard recognizer
// (GCNHazardRecognizer, the "VMEM store data" hazard) inserts those wait states behind the stores IT emits, but it does not look
// inside an inline-asm statement -- so when the register allocator reuses one of the four data registers for whatever comes
// next (one instantiation did: <8 x 4 components, 8 groups, in-place pipeline> restored a spilled lane index into the second
// of them with v_accvgpr_read_b32 right behind the asm), lanes of the store leave with the NEW value in that dword: the high
// halves of four components in the first four columns of every strip, run-to-run different -- the "MODE 2 race" of rounds
// 3 and 4.  Found and proven at the instruction level (tools/m2_isa_probe.cpp: 45 000 wrong entries per launch without,
// none with `s_nop 1` patched behind the stores of the unchanged compiler output); tools/isa_store_hazard.py audits the
// shipped code objects for the pattern.
__device__ __forceinline__ void store_result16(double* p, double __attribute__((ext_vector_type(2))) t)
{
#if FLUHIP_EPILOGUE_SC1 == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#elif FLUHIP_EPILOGUE_SC1 == 3
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#elif FLUHIP_EPILOGUE_SC1 == 4
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#elif FLUHIP_EPILOGUE_SC1
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#else
  *reinterpret_cast<double __attribute__((ext_vector_type(2)))*>(p) = t;
#endif
}

// v / d for d > 0, v >= 0 in the normal range: v_rcp_f64 -> one Newton step -> quotient -> residual correction
// (error ~2^-96 before the final rounding; exact when d == 1)
__device__ __forceinline__ double fdiv_pos(double v, double d)
{
  double y = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, y, 1.0);
  y = __builtin_fma(y, e, y);
  const double r = v * y;
  const double res = __builtin_fma(-d, r, v);
  return __builtin_fma(res, y, r);
}

#define FLUHIP_GLDS(src, dst)                                                                     \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (src),          \
                                   (__attribute__((address_space(3))) void*) (dst), 16, 0, 0)

// LDS-DMA in the "scalar base + 32-bit lane offset" form, spelled out: left to the compiler the builtin
// takes a 64-bit vector address, i.e. one v_lshl_add_u64 per copy -- a VALU instruction in the middle of
// the MFMA stream costs ~14 cycles there (tools/mfma_coissue_probe.hip).
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned ldsAddr)
{
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :
               : "s"(ldsAddr), "v"(voff), "s"(sbase)
               : "memory", "m0");
}
// The V slab's copies are NON-TEMPORAL (round 6): a launch reads its layout of V once -- 905 MB on the bench shard, 3.5 x the
// 256 MB Infinity Cache -- and with the default policy that stream displaces the factor matrices (61 MB) every launch re-reads
// from the launch before: the stationary rows of the prologue, the moving factor's slabs.  `nt` on the slab copies of the loop:
// bench shard 546.6 -> 534.9 us per iteration alternating on one box, and the part clocks 1.2 % HIGHER (profiles/r06/v_nt.txt).
// -DFLUHIP_V_NT=0: the default policy.
// Split-contraction partials are staged through the LDS and stored write-through in whole lines, like the results (round 6):
// c4 x 1 47.6 -> 44.6 us per iteration, a 10 s buffer at rank 128 101.1 -> 93.5, config 3 1 831 -> 1 807, alternating on one box
// (profiles/r06/partials_staged.txt).  -DFLUHIP_PARTIAL_STAGED=0: plain 16-byte stores from the MFMA layout.
#ifndef FLUHIP_PARTIAL_STAGED
#define FLUHIP_PARTIAL_STAGED 1
#endif
#ifndef FLUHIP_V_NT
#define FLUHIP_V_NT 1
#endif
__device__ __forceinline__ void glds16_v(const void* sbase, unsigned voff, unsigned ldsAddr)
{
#if FLUHIP_V_NT
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt"
               :
               : "s"(ldsAddr), "v"(voff), "s"(sbase)
               : "memory", "m0");
#else
  glds16(sbase, voff, ldsAddr);
#endif
}
__device__ __forceinline__ unsigned lds_addr(const void* p)
{
  return (unsigned) (size_t) (__attribute__((address_space(3))) const void*) p;
}

template <int N>
__device__ __forceinline__ void load_vec5(double (&dst)[N], const double* p)
{
#pragma unroll
  for (int j = 0; j < N; j += 2)
  {
    d2 t = *reinterpret_cast<const d2*>(p + j);
    dst[j] = t[0];
    dst[j + 1] = t[1];
  }
}
// the stationary rows.  Non-temporal loads (-DFLUHIP_STATIONARY_NT=1) were tried in round 6: the bench shard the same (534.6 /
// 539.3 against 534.5 / 535.6 us per iteration), a 10 s buffer at rank 128 90.1 -> 96.3 (the pieces of a split contraction read
// the same rows).  Not adopted.
#ifndef FLUHIP_STATIONARY_NT
#define FLUHIP_STATIONARY_NT 0
#endif
template <int N>
__device__ __forceinline__ void load_stationary5(double (&dst)[N], const double* p)
{
#if FLUHIP_STATIONARY_NT
#pragma unroll
  for (int j = 0; j < N; j += 2)
  {
    d2 t = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + j));
    dst[j] = t[0];
    dst[j + 1] = t[1];
  }
#else
  load_vec5<N>(dst, p);
#endif
}

// quad_perm exchange of a double (the four lanes x = 0 .. 3 of a result row)
template <int CTRL>
__device__ __forceinline__ double quad_dpp(double v)
{
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xffffffff), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
}

// SIDEQ (round 4; H update of a corpus whose W update keeps bin C_w - 1 as a side column): the epilogue, which holds the new
// rows of H in registers, also forms this wavefront's share of the side column's contraction for the W update that follows --
// q_t = sum_k H[t][k] w_k with w = W'[R-1] / nrm, V[R-1][t] / max(q_t, eps), num_k, den_k over its frames -- so that no launch
// has to read H again for it (side_slices_kernel: 10 us per iteration of the bench shard, all of it a second pass over H).
// SIDEQ & 2 (H update right behind a W update whose norm combine is still due): every wavefront forms the new norms and
// the new side row of W' itself, in its prologue, from the column statistics the W update's wavefronts left and the side-column
// partials of the H update before that -- a few KB per wavefront, the same sums in the same order in all of them -- instead
// of a launch in between (5.7 us + a kernel boundary per iteration of the bench shard).  The wavefronts of a buffer all store
// the (identical) side row; strip 0 stores the norms.  Nothing in this launch reads either back from memory except each
// wavefront's own last DMA stage (row R - 1 of W'), 270 us behind its own store.
// KPM (round 5): the rank the ARRAYS are laid out for (row stride of S and Mv, of the per-buffer vectors and of the statistics /
// partial records), KPM >= KP = 4 M the rank the kernel computes.  KPM > KP serves the off-size ranks: 33 .. 48 compute twelve
// MFMAs per product on the layout of rank 64 (every helper kernel keeps its rank-64 form; components 48 .. 63 are zero
// and are neither read nor written here), 65 .. 96 twenty-four on the layout of rank 128 -- three quarters of the matrix work
// the padded form pays (clients/nrt/NMFClient.hpp:68: `components` is any integer >= 1).
template <int M, int NG, int NS, int WPS, int INSTR = 0, int MODE = 0, int DS = 1, int LIST = 0, int SIDEQ = 0, int KPM = 4 * M>
__global__ __launch_bounds__(256 * WPS, WPS) void nmf_update5_kernel(Upd5Args a)
{
  constexpr int KP = 4 * M;
  static_assert(KPM >= KP && KPM % 16 == 0, "array rank");
  constexpr int SPR = KP / 2;                      // 16-byte chunks per Mv row
  constexpr int NJV = (32 * NG + 63) / 64;         // DMA instructions per V stage
  constexpr int NJM = (4 * SPR + 63) / 64;         // DMA instructions per Mv stage
  constexpr int VSTAGE = NJV * 1024;               // bytes
  constexpr int MSTAGE = NJM * 1024;
  constexpr int WAVE_LDS = NS * (VSTAGE + MSTAGE);
  constexpr int IPS = NJV + NJM;                   // vmcnt events per stage
  // work-list mode: a wavefront's LDS region also stages its accumulators for the intra-workgroup reduction
  // DS == 2: no column-sum accumulators and the first product's results in VGPRs (QV below) -- the two go together: the
  // sixteen registers of the accumulators are what the VGPR form of the first product needs
  // (... and every form of the two-operand-set pipeline that has the registers for it WITH its accumulators: rank 16 at any
  //  strip width, the off-size forms of ranks 17 .. 24 (up to eight groups) and 33 .. 40, rank 32 up to six -- none of them spills, probed
  //  instantiation by instantiation; the same MFMAs in the same order, bit for bit the same results)
  constexpr bool QV = MODE == 1 && (DS == 2 || (DS == 1 && INSTR == 0 && (M == 4 || (M == 6 && NG <= 8) || (M == 8 && NG <= 6) || (FLUHIP_QV_M10 && M == 10 && NG <= 4))));
  constexpr int STG_BYTES = (NG * M + M) * 512;
  constexpr int WAVE_REGION = (LIST != 0 && STG_BYTES > WAVE_LDS && 4 * STG_BYTES <= 160 * 1024) ? STG_BYTES : WAVE_LDS;

  extern __shared__ __attribute__((aligned(16))) char lds[];

  int id = blockIdx.x;
  int buf, wg, split;
  // work-list mode is a separate instantiation (LIST): the uniform form below keeps its statement order -- and with it its
  // register allocation, which sits at the limit of the file -- exactly as it was
  [[maybe_unused]] const WaveDesc* wd = nullptr;
  if constexpr (LIST != 0)
  {
    wd = a.list + ((int64_t) id * 4 * WPS + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    buf = __builtin_amdgcn_readfirstlane(wd->buf);
    wg = 0; split = 0;
  }
  else
  {
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
  }
  long long tEntry = 0;
  if constexpr (INSTR) tEntry = (long long) __builtin_amdgcn_s_memrealtime();
  // box-invariant cost of a launch: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) of one wavefront that
  // lives as long as the launch does (one wavefront per SIMD, one round); their ratio is the clock the part sustained
  long long clkC0 = 0, clkR0 = 0;
  const bool stamp = a.clk != nullptr && blockIdx.x == 0;
  if (stamp)
  {
    clkC0 = (long long) __builtin_readcyclecounter();
    clkR0 = (long long) __builtin_amdgcn_s_memrealtime();
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int WPB = 4 * WPS; // wavefronts per workgroup
  const int strip = wg * WPB + wave;
  int g0 = 0, ng = 0;
  if constexpr (LIST != 0)
  {
    g0 = __builtin_amdgcn_readfirstlane(wd->g0);
    ng = __builtin_amdgcn_readfirstlane(wd->ng);
  }
  else
  {
  if (strip >= a.wavesPerBuf) return;
  // Column groups are dealt out as evenly as possible.  With two wavefronts per SIMD (WPS == 2)
  // wavefronts w and w+4 of a workgroup share a SIMD, so the "one extra group" strips are spread
  // over distinct SIMD pairs first: the per-SIMD load stays within one group of the mean.
  const int base = a.nGroups / a.wavesPerBuf, rem = a.nGroups % a.wavesPerBuf;
  {
    const int nPairs = (a.wavesPerBuf + WPS - 1) / WPS;
    int acc0 = 0;
    for (int st = 0; st <= strip; st++)
    {
      const int wgi = st / WPB, w = st % WPB;
      const int pair = wgi * 4 + (w & 3), member = w >> 2;
      // extras e = pair + nPairs * member  (e < rem)
      const int extra = (pair + nPairs * member) < rem ? 1 : 0;
      const int cnt = base + extra;
      if (st == strip) { g0 = acc0; ng = cnt; }
      acc0 += cnt;
    }
  }
  }
  if (ng <= 0) return;

  const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
  const double* __restrict__ V = a.V + (int64_t) buf * a.strideV;
  const double* __restrict__ Mv = a.Mv + (int64_t) buf * a.strideM;
  double* S = a.S + (int64_t) buf * a.strideS;

  char* vring = lds + wave * WAVE_REGION;
  char* mring = vring + NS * VSTAGE;

  // ---- per-lane DMA source offsets (fixed for the whole pass) ---------------------------------
  // V stage chunk i = 64 j + lane: row = i / (8 NG), 16-byte column chunk cc = i % (8 NG); chunks of
  // groups this strip does not own (tail strips) and the slack of the last instruction re-read
  // valid data of the strip instead of running off the row.
  // byte offsets relative to a wave-uniform base, so the DMA takes "scalar base + 32-bit lane offset"
  // addressing and no 64-bit vector address arithmetic is left in the loop
  unsigned voffs[NJV];
#pragma unroll
  for (int j = 0; j < NJV; j++)
  {
    int i = 64 * j + lane;
    int row = i / (8 * NG), cc = i % (8 * NG);
    if (row > 3) { row = 3; }
    cc = min(cc, 8 * ng - 1);
    voffs[j] = (unsigned) ((row * a.ldv + cc * 2) * 8);
  }
  unsigned moffs[NJM];
#pragma unroll
  for (int j = 0; j < NJM; j++)
  {
    int p = 64 * j + lane;
    int row = p / SPR, pos = p % SPR;
    if (row > 3) row = 3;
    const int cs = (pos - row + SPR) % SPR;          // undo the per-row rotation ...
    const int cx = cs ^ (((cs >> 4) & 3) << 2);        // ... and the XOR of slot bits 2-3 with bits 4-5
    const int c = cx < SPR ? cx : cs;                  // (rows that end inside a block of 16 chunks -- the off-size forms M = 10, 14, 20, 28: a chunk whose partner lies past the row stays where it is)
    moffs[j] = (unsigned) ((row * KPM + c * 2) * 8);    // slot pos of row `row` holds chunk c
  }

  int s0, s1;
  if constexpr (LIST != 0)
  {
    s0 = __builtin_amdgcn_readfirstlane(wd->s0);
    s1 = __builtin_amdgcn_readfirstlane(wd->s1);
  }
  else
  {
    s0 = split * a.stepsPerSplit;
    s1 = min(s0 + a.stepsPerSplit, a.nSteps);
  }
  const int sLast = s1 - 1;

  auto issue_stage = [&](int st) {
    const int sc = min(st, sLast); // past the end: harmless re-read of the last step
    const int slot = st % NS;
    const char* vsrc = reinterpret_cast<const char*>(V + (int64_t) sc * 4 * a.ldv + (int64_t) g0 * 16); // uniform
    const char* msrc = reinterpret_cast<const char*>(Mv + (int64_t) sc * 4 * KPM);                        // uniform
#pragma unroll
    for (int j = 0; j < NJV; j++) FLUHIP_GLDS(vsrc + voffs[j], vring + slot * VSTAGE + j * 1024);
#pragma unroll
    for (int j = 0; j < NJM; j++) FLUHIP_GLDS(msrc + moffs[j], mring + slot * MSTAGE + j * 1024);
  };

  // ring fill of the overlapped pipeline (slots numbered from s0), in two parts: the first two stages go out
  // ahead of the stationary rows -- the prologue needs them first -- the rest behind them, so that the
  // stationary rows do not queue behind 6 stages of streaming data at every wavefront at once
  auto fill_slots = [&](int t0, int t1) {
    if (s0 < s1)
    {
#pragma unroll
      for (int t = t0; t < t1; t++)
      {
        const int sc = min(s0 + t, sLast);
        const char* vsrc = reinterpret_cast<const char*>(V + (int64_t) sc * 4 * a.ldv + (int64_t) g0 * 16);
        const char* msrc = reinterpret_cast<const char*>(Mv + (int64_t) sc * 4 * KPM);
#pragma unroll
        for (int j = 0; j < NJV; j++) FLUHIP_GLDS(vsrc + voffs[j], vring + t * VSTAGE + j * 1024);
#pragma unroll
        for (int j = 0; j < NJM; j++) FLUHIP_GLDS(msrc + moffs[j], mring + t * MSTAGE + j * 1024);
      }
    }
  };
  long long tP1 = 0, tP2 = 0, tP3 = 0;
  if constexpr (INSTR) tP1 = (long long) __builtin_amdgcn_s_memrealtime();
  // SIDEQ: what the epilogue needs from memory -- row R - 1 of the moving factor (the W update's side row) and the side bin's
  // magnitudes of this strip's frames -- is copied into the LDS behind the rings by the first DMAs of the launch: the oldest
  // requests, landed long before the loop ends (fetched in the epilogue they cost the launch two memory latencies, 4.5 us)
  constexpr int NJSV = SIDEQ ? (NG * 128 + 1023) / 1024 : 0;
  constexpr bool NORMQ = (SIDEQ & 2) != 0;
  [[maybe_unused]] char* sideL = nullptr;
  // NORMQ: the requests of the norm combine leave first (lane k < KP holds component k)
  [[maybe_unused]] double cs[8], cm[8], cn[8], cd[8], cwo = 0.0, cc[8];
  static_assert(DS != 2 || NORMQ || SIDEQ == 0, "DS == 2: the W update's plain form or the H update's norm form");
  if constexpr (DS == 2)
  {
    // the column sums of Mv from the partials of the launch in front: lane k < KP takes component k (requests first, sums below)
#pragma unroll
    for (int u = 0; u < 8; u++) cc[u] = 0.0;
    if (lane < KP)
    {
      // (W update: the denominators of the side-column partials, [n][2][KPM] -- H update: the W update's column partials, [n][KPM])
      const int64_t rec = NORMQ ? KPM : 2 * KPM;
      const double* cp = a.colIn + (int64_t) buf * a.colInN * rec + (NORMQ ? 0 : KPM) + lane;
#pragma unroll
      for (int u = 0; u < 8; u++) if (u < a.colInN) cc[u] = cp[(int64_t) u * rec];
    }
  }
  if constexpr (NORMQ)
  {
#pragma unroll
    for (int u = 0; u < 8; u++) { cs[u] = 0.0; cm[u] = -INFINITY; cn[u] = 0.0; cd[u] = 0.0; }
    if (lane < KP)
    {
      const double* sp = a.cmbStat + (int64_t) buf * a.cmbParts * 2 * KPM + lane;
      const double* qp = a.cmbSide + (int64_t) buf * a.cmbSlices * 2 * KPM + lane;
      cwo = a.cmbWold[(int64_t) buf * KPM + lane];
#pragma unroll
      for (int u = 0; u < 8; u++)
      {
        if (u < a.cmbParts) { cs[u] = sp[(int64_t) u * 2 * KPM]; cm[u] = sp[(int64_t) u * 2 * KPM + KPM]; }
        if (u < a.cmbSlices) { cn[u] = qp[(int64_t) u * 2 * KPM]; cd[u] = qp[(int64_t) u * 2 * KPM + KPM]; }
      }
    }
  }
  constexpr int SIDE_BYTES = SIDEQ ? (1 + NJSV) * 1024 : 0;
  constexpr int EXTRA_BYTES = SIDE_BYTES + (DS == 2 ? 512 : 0);      // per wavefront, behind the rings
  [[maybe_unused]] double* colL = reinterpret_cast<double*>(lds + WPB * WAVE_REGION + wave * EXTRA_BYTES + SIDE_BYTES);
  if constexpr (SIDEQ)
  {
    sideL = lds + WPB * WAVE_REGION + wave * EXTRA_BYTES;
    if constexpr (!NORMQ)
    {
      const char* wsrc = reinterpret_cast<const char*>(Mv + (int64_t) (a.R - 1) * KPM) + min(lane, SPR - 1) * 16;
      FLUHIP_GLDS(wsrc, sideL);
    }
#pragma unroll
    for (int j = 0; j < NJSV; j++)
    {
      const int i = min(64 * j + lane, 8 * ng - 1);
      const char* vsrc = reinterpret_cast<const char*>(V + (int64_t) (a.R - 1) * a.ldv + (int64_t) g0 * 16) + i * 16;
      FLUHIP_GLDS(vsrc, sideL + 1024 + j * 1024);
    }
  }
  if constexpr (MODE >= 1) fill_slots(0, 2);

  // ---- stationary operand + accumulators ------------------------------------------------------
  double sb[NG][M];
  double acc[NG][M];
#pragma unroll
  for (int g = 0; g < NG; g++)
  {
#pragma unroll
    for (int m = 0; m < M; m++) { acc[g][m] = 0.0; sb[g][m] = 0.0; }
    if (g < ng) load_stationary5<M>(sb[g], S + (int64_t) ((g0 + g) * 16 + 4 * blk + x) * KPM + M * y);
  }
  [[maybe_unused]] double colSum = 0.0;
  if constexpr (DS == 2)
  {
    // in index order, like every other combine of per-wavefront records (the same sum in every wavefront of the buffer)
#pragma unroll
    for (int u = 0; u < 8; u++) colSum += cc[u];
    if (lane < KP)
    {
      const int64_t rec = NORMQ ? KPM : 2 * KPM;
      const double* cp = a.colIn + (int64_t) buf * a.colInN * rec + (NORMQ ? 0 : KPM) + lane;
      for (int j = 8; j < a.colInN; j++) colSum += cp[(int64_t) j * rec];
    }
    if constexpr (!NORMQ)
    {
      if (lane < KP) colL[lane] = colSum;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if constexpr (NORMQ)
  {
    // sums in index order (parts / slices beyond eight: a second, dependent round -- corpora of few buffers per chip)
    double t = 0.0, mv = -INFINITY, n = 0.0, d = 0.0;
#pragma unroll
    for (int u = 0; u < 8; u++) { t += cs[u]; mv = fmax(mv, cm[u]); n += cn[u]; d += cd[u]; }
    if (lane < KP)
    {
      const double* sp = a.cmbStat + (int64_t) buf * a.cmbParts * 2 * KPM + lane;
      const double* qp = a.cmbSide + (int64_t) buf * a.cmbSlices * 2 * KPM + lane;
      for (int j = 8; j < a.cmbParts; j++) { t += sp[(int64_t) j * 2 * KPM]; mv = fmax(mv, sp[(int64_t) j * 2 * KPM + KPM]); }
      for (int j = 8; j < a.cmbSlices; j++) { n += qp[(int64_t) j * 2 * KPM]; d += qp[(int64_t) j * 2 * KPM + KPM]; }
    }
    // the side row (not normalised, like every other row of W') and alg/NMF.hpp:162 for the whole column
    const bool liveK = lane < a.cmbK;
    const double wnew = liveK ? (cwo * n) / fmax(d, kEpsilon) : 0.0;
    t = __builtin_fma(wnew, wnew, t);
    mv = liveK ? fmax(mv, wnew) : -INFINITY;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) mv = fmax(mv, __shfl_xor(mv, off));
    const double nv = (liveK && mv > kEpsilon) ? sqrt(t) : 1.0;
    double* ldsW = reinterpret_cast<double*>(sideL);
    if (lane < KP)
    {
      ldsW[lane] = wnew;
      ldsW[64 + lane] = nv;
      if constexpr (DS == 2) colL[lane] = colSum + wnew;   // sum_f W'[f][k]: the rows the W update wrote + the side row formed here
      a.cmbRowOut[(int64_t) buf * a.strideM + (int64_t) (a.R - 1) * KPM + lane] = wnew;
      if (strip == 0) a.cmbNrmOut[(int64_t) buf * KPM + lane] = nv;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (a.nrmMode)
  {
    // deferred normalisation: W update -> the stationary rows are W' / nrm = W; H update -> Q = W' (H / nrm)^T
    double nr[M];
    if constexpr (NORMQ)
    {
      const double* ldsN = reinterpret_cast<const double*>(sideL) + 64;
#pragma unroll
      for (int m = 0; m < M; m++) nr[m] = ldsN[M * y + m];
    }
    else
    load_vec5<M>(nr, a.nrm + (int64_t) buf * KPM + M * y);
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
      for (int m = 0; m < M; m++) sb[g][m] = fdiv_pos(sb[g][m], nr[m]);
  }
  if constexpr (INSTR)
  {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::"v"(sb[0][0]), "v"(sb[NG - 1][M - 1])); // the stationary rows are in registers here
    tP2 = (long long) __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (MODE >= 1) fill_slots(2, NS);
  double dsum[M];
#pragma unroll
  for (int m = 0; m < M; m++) dsum[m] = 0.0;

  // ---- LDS read addresses (bytes within a stage) ------------------------------------------------
  // ma: row x, chunks M/2*y .. ; mb: row y, chunks M/2*x ..  Chunk c of row r sits at slot
  // ((c ^ (((c >> 4) & 3) << 2)) + r) % SPR: the rotation separates the 4 rows, the XOR separates
  // chunks 16 apart (rows longer than 256 B), so both distributions are conflict free per lane group.
  auto slot_of = [](int c, int r) { const int cx = c ^ (((c >> 4) & 3) << 2); return ((cx < SPR ? cx : c) + r) % SPR; };
  int maOff[M / 2], mbOff[M / 2];
#pragma unroll
  for (int j = 0; j < M / 2; j++)
  {
    maOff[j] = x * (KP * 8) + slot_of((M / 2) * y + j, x) * 16;
    mbOff[j] = y * (KP * 8) + slot_of((M / 2) * x + j, y) * 16;
  }
  const int vOff = y * (NG * 128) + (4 * blk + x) * 8;

  auto read_ma = [&](int st, double (&ma)[M]) {
    const char* mp = mring + (st % NS) * MSTAGE;
#pragma unroll
    for (int j = 0; j < M / 2; j++)
    {
      d2 t = *reinterpret_cast<const d2*>(mp + maOff[j]);
      ma[2 * j] = t[0];
      ma[2 * j + 1] = t[1];
    }
  };
  auto read_v = [&](int st, double (&v)[NG]) {
    const char* vp = vring + (st % NS) * VSTAGE + vOff;
#pragma unroll
    for (int g = 0; g < NG; g++) v[g] = *reinterpret_cast<const double*>(vp + g * 128);
  };
  auto read_mb = [&](int st, double (&mb)[M]) {
    const char* mp = mring + (st % NS) * MSTAGE;
#pragma unroll
    for (int j = 0; j < M / 2; j++)
    {
      d2 t = *reinterpret_cast<const d2*>(mp + mbOff[j]);
      mb[2 * j] = t[0];
      mb[2 * j + 1] = t[1];
    }
  };

  // Q for one 4-row step.  A dependent v_mfma_f64_4x4x4 cannot issue back to back on its own
  // accumulator, so each group's contraction over m is split into P interleaved partial chains
  // (>= 8 independent accumulators in flight, the count at which the probe reaches 73 TFLOP/s).
  constexpr int P0 = (NG >= 5) ? 1 : (NG >= 3 ? 2 : (NG >= 2 ? 4 : 8));
  // (M = 12: four chains where eight do not divide; the other off-size forms -- M = 6, 10, 14, 20, 28 -- take P0 chains of
  //  unequal length: qp[g][m % P])
  constexpr int P = (M % P0 == 0 || M < P0) ? P0 : (M == 12 ? P0 / 2 : P0);
  auto q_phase = [&](const double (&ma)[M], double (&q)[NG]) {
    constexpr int PP = (P <= M) ? P : M;
    double qp[NG][PP];
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
      for (int p = 0; p < PP; p++) qp[g][p] = 0.0;
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
      for (int g = 0; g < NG; g++)
        qp[g][m % PP] = __builtin_amdgcn_mfma_f64_4x4x4f64(ma[m], sb[g][m], qp[g][m % PP], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < NG; g++)
    {
      double t = qp[g][0];
#pragma unroll
      for (int p = 1; p < PP; p++) t += qp[g][p];
      q[g] = t;
    }
  };
  // V / max(Q, eps), stage by stage across the NG independent quotients:
  // v_rcp_f64 (~23 bits) -> one Newton step -> quotient [-> residual correction, kQuotientCorrection]
  auto ratio_phase = [&](const double (&v)[NG], const double (&q)[NG], double (&ratio)[NG]) {
    double d[NG], yv[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) d[g] = q[g] > kEpsilon ? q[g] : kEpsilon;
    if constexpr (kQuotientCorrection || !kSharedReciprocal)
    {
#pragma unroll
      for (int g = 0; g < NG; g++) yv[g] = __builtin_amdgcn_rcp(d[g]);
#pragma unroll
      for (int g = 0; g < NG; g++) ratio[g] = __builtin_fma(-d[g], yv[g], 1.0);
#pragma unroll
      for (int g = 0; g < NG; g++) yv[g] = __builtin_fma(yv[g], ratio[g], yv[g]);
    }
    else if constexpr (kRcpGroup > 2)
    {
      // one reciprocal per group of up to kRcpGroup quotients (round 6, recip_tree.h): NG = 7 / 8 cost 27 / 30 operation slots
      // where the pairs below cost 33 / 36
      recip_tree<NG, kRcpGroup>(d, yv);
    }
    else
    {
      // one reciprocal per TWO quotients: r = 1 / (a b) refined once, 1 / a = r b, 1 / b = r a (v_rcp_f64 is a quarter-rate
      // instruction; a b stays far inside the range: eps^2 .. max(Q)^2)
      constexpr int NP = NG / 2;
      double pr[NP > 0 ? NP : 1], rr[NP > 0 ? NP : 1], ee[NP > 0 ? NP : 1];
#pragma unroll
      for (int h = 0; h < NP; h++) pr[h] = d[2 * h] * d[2 * h + 1];
#pragma unroll
      for (int h = 0; h < NP; h++) rr[h] = __builtin_amdgcn_rcp(pr[h]);
      if constexpr (NG & 1) yv[NG - 1] = __builtin_amdgcn_rcp(d[NG - 1]);
#pragma unroll
      for (int h = 0; h < NP; h++) ee[h] = __builtin_fma(-pr[h], rr[h], 1.0);
      if constexpr (NG & 1) ratio[NG - 1] = __builtin_fma(-d[NG - 1], yv[NG - 1], 1.0);
#pragma unroll
      for (int h = 0; h < NP; h++) rr[h] = __builtin_fma(rr[h], ee[h], rr[h]);
      if constexpr (NG & 1) yv[NG - 1] = __builtin_fma(yv[NG - 1], ratio[NG - 1], yv[NG - 1]);
#pragma unroll
      for (int h = 0; h < NP; h++)
      {
        yv[2 * h] = rr[h] * d[2 * h + 1];
        yv[2 * h + 1] = rr[h] * d[2 * h];
      }
    }
#pragma unroll
    for (int g = 0; g < NG; g++) ratio[g] = v[g] * yv[g];
    if constexpr (kQuotientCorrection)
    {
#pragma unroll
      for (int g = 0; g < NG; g++) d[g] = __builtin_fma(-d[g], ratio[g], v[g]);
#pragma unroll
      for (int g = 0; g < NG; g++) ratio[g] = __builtin_fma(d[g], yv[g], ratio[g]);
    }
  };
  auto out_phase = [&](const double (&ratio)[NG], const double (&mb)[M]) {
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
      for (int m = 0; m < M; m++) acc[g][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(ratio[g], mb[m], acc[g][m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < M; m++) if constexpr (DS == 1) dsum[m] += mb[m];
  };

  // ---- pipeline -------------------------------------------------------------------------------
  // stages s0 .. s0+NS-1 in flight; iteration s consumes stage s (mb, v) and stage s+1 (ma), then
  // refills the slot of stage s with stage s+NS.  Completion is in issue order, so "stage s+1 has
  // landed" == at most (NS-2) stages outstanding.
  if constexpr (MODE == 2)
  {
    // ---- overlapped form, one operand register set refilled in place -------------------------------
    // As MODE 1 below, but without the second operand set: the Q-phase walks m upwards and every 16-byte
    // chunk of ma (step s+1) is dead once its MFMAs have issued, so the same registers take the chunk of
    // step s+2 right behind them; the out-phase runs m-outer for the same reason and refills mb with step
    // s+1; v of step s+1 is read at the head of the Q-phase (v(s) died in the quotient block).  The reads
    // still have a whole step to land.  The column sums of Mv move to the head of the step, into the VALU
    // block, where mb(s) is complete.  This is what lets Kp = 64 / 128 (M = 16 / 32) run overlapped.
    if (s0 < s1)
    {
      static_assert(NS >= 4, "overlapped pipeline: ring depth >= 4");
      const unsigned vringA = __builtin_amdgcn_readfirstlane(lds_addr(vring));
      const unsigned mringA = __builtin_amdgcn_readfirstlane(lds_addr(mring));
      const char* vAddr = vring + vOff;
      const char* maAddr[M / 2];
      const char* mbAddr[M / 2];
#pragma unroll
      for (int j = 0; j < M / 2; j++) { maAddr[j] = mring + maOff[j]; mbAddr[j] = mring + mbOff[j]; }
      double v[NG], ma[M], mb[M], qA[NG], qB[NG];
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * IPS) : "memory");
#pragma unroll
      for (int j = 0; j < M / 2; j++)
      {
        d2 t = *reinterpret_cast<const d2*>(maAddr[j]);
        ma[2 * j] = t[0]; ma[2 * j + 1] = t[1];
      }
      q_phase(ma, qA);
#pragma unroll
      for (int g = 0; g < NG; g++) v[g] = *reinterpret_cast<const double*>(vAddr + g * 128);
#pragma unroll
      for (int j = 0; j < M / 2; j++)
      {
        d2 t = *reinterpret_cast<const d2*>(mbAddr[j]);
        mb[2 * j] = t[0]; mb[2 * j + 1] = t[1];
        d2 t2 = *reinterpret_cast<const d2*>(maAddr[j] + MSTAGE);
        ma[2 * j] = t2[0]; ma[2 * j + 1] = t2[1];
      }
      constexpr int NMF = M * NG;
      constexpr int DMASTEP = NMF / IPS > 0 ? NMF / IPS : 1;
      constexpr int VSTEP = (NMF / 2) / NG > 0 ? (NMF / 2) / NG : 1; // v reads over the first half of the Q-phase
      constexpr int QREADS = NG + M / 2;                             // ds_reads issued in the Q-phase
      auto half = [&](int s, int u, int u1, int u2, const double (&qc)[NG], double (&qn)[NG]) {
        double ratio[NG];
        // stages s+1, s+2 landed.  (Refills past the last step re-read the last step's rows -- L2 hits; skipping
        // them was measured slower both ways: a branch per DMA splits the MFMA stream into basic blocks, and
        // issuing them under EXEC = 0 stalls on every EXEC write.)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * IPS) : "memory");
        ratio_phase(v, qc, ratio);
#pragma unroll
        for (int m = 0; m < M; m++) if constexpr (DS == 1) dsum[m] += mb[m];
        __builtin_amdgcn_sched_barrier(0);
        {
          constexpr int PP = (P <= M) ? P : M;
          double qp[NG][PP];
#pragma unroll
          for (int g = 0; g < NG; g++)
#pragma unroll
            for (int p = 0; p < PP; p++) qp[g][p] = 0.0;
#pragma unroll
          for (int m = 0; m < M; m++)
#pragma unroll
            for (int g = 0; g < NG; g++)
            {
              qp[g][m % PP] = __builtin_amdgcn_mfma_f64_4x4x4f64(ma[m], sb[g][m], qp[g][m % PP], 0, 0, 0);
              const int i = m * NG + g;
              if (i % VSTEP == VSTEP - 1 && i / VSTEP < NG)
              {
                const int r = i / VSTEP;
                v[r] = *reinterpret_cast<const double*>(vAddr + u1 * VSTAGE + r * 128);
                __builtin_amdgcn_sched_barrier(0);
              }
              if ((m & 1) && g == NG - 1) // chunk m/2 of ma(s+1) is spent: its registers take ma(s+2)
              {
                __builtin_amdgcn_sched_barrier(0);
                d2 t = *reinterpret_cast<const d2*>(maAddr[m / 2] + u2 * MSTAGE);
                ma[m - 1] = t[0];
                ma[m] = t[1];
                __builtin_amdgcn_sched_barrier(0);
              }
            }
#pragma unroll
          for (int g = 0; g < NG; g++)
          {
            double t = qp[g][0];
#pragma unroll
            for (int p = 1; p < PP; p++) t += qp[g][p];
            qn[g] = t;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          const int sc = min(s + NS, sLast);
          const char* vsrc = reinterpret_cast<const char*>(V + (int64_t) sc * 4 * a.ldv + (int64_t) g0 * 16);
          const char* msrc = reinterpret_cast<const char*>(Mv + (int64_t) sc * 4 * KPM);
          // every ds_read of slot u was issued in earlier steps; the Q-phase above issued QREADS newer ones
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(QREADS < 15 ? QREADS : 15) : "memory");
#pragma unroll
          for (int m = 0; m < M; m++)
#pragma unroll
            for (int g = 0; g < NG; g++)
            {
              acc[g][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(ratio[g], mb[m], acc[g][m], 0, 0, 0);
              const int i = m * NG + g;
              if (i % DMASTEP == DMASTEP / 2 && i / DMASTEP < IPS)
              {
                const int j = i / DMASTEP;
                if (j < NJV) glds16_v(vsrc, voffs[j], vringA + u * VSTAGE + j * 1024);
                else glds16(msrc, moffs[j - NJV], mringA + u * MSTAGE + (j - NJV) * 1024);
                __builtin_amdgcn_sched_barrier(0);
              }
              if ((m & 1) && g == NG - 1) // chunk m/2 of mb(s) is spent: its registers take mb(s+1)
              {
                __builtin_amdgcn_sched_barrier(0);
                d2 t = *reinterpret_cast<const d2*>(mbAddr[m / 2] + u1 * MSTAGE);
                mb[m - 1] = t[0];
                mb[m] = t[1];
                __builtin_amdgcn_sched_barrier(0);
              }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      constexpr int UNR = (NS % 2 == 0) ? NS : 2 * NS; // slots and the qA/qB roles repeat together
      for (int s = s0; s < s1; s += UNR)
      {
#pragma unroll
        for (int w = 0; w < UNR; w++)
        {
          if (s + w >= s1) break;
          const int u = w % NS;
          if (w % 2 == 0) half(s + w, u, (u + 1) % NS, (u + 2) % NS, qA, qB);
          else half(s + w, u, (u + 1) % NS, (u + 2) % NS, qB, qA);
        }
      }
      // the last step's column sums were taken at its head; mb now holds a step past the end
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  else
  if constexpr (MODE == 1)
  {
    // ---- overlapped form ------------------------------------------------------------------------
    // Measured on the device (tools/mfma_coissue_probe.hip): inside a stream of independent 4x4x4 f64
    // MFMAs a ds_read costs ~5 cycles of issue, a 16-byte-per-lane VMEM instruction ~30, and ANY VALU
    // instruction ~14 (pipe switch) -- so the operand reads of step s+1 and the DMA refill of slot s are
    // spread between step s's MFMAs, into a second operand register set, with every LDS address a
    // loop-invariant VGPR plus an immediate: the loop is unrolled over the ring so that slot numbers are
    // compile-time constants and no address arithmetic is left on the VALU.  Slots are numbered from s0.
    if (s0 < s1)
    {
      static_assert(NS >= 4 && NS % 2 == 0, "overlapped pipeline: even ring depth >= 4");
      const unsigned vringA = __builtin_amdgcn_readfirstlane(lds_addr(vring));
      const unsigned mringA = __builtin_amdgcn_readfirstlane(lds_addr(mring));
      // loop-invariant per-lane LDS addresses
      const char* vAddr = vring + vOff;
      const char* maAddr[M / 2];
      const char* mbAddr[M / 2];
#pragma unroll
      for (int j = 0; j < M / 2; j++) { maAddr[j] = mring + maOff[j]; mbAddr[j] = mring + mbOff[j]; }

      double vX[NG], maX[M], mbX[M], vY[NG], maY[M], mbY[M], qA[NG], qB[NG];
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * IPS) : "memory");
      if constexpr (INSTR) tP3 = (long long) __builtin_amdgcn_s_memrealtime();
      auto read_set = [&](int slotV, int slotA, double (&v)[NG], double (&ma)[M], double (&mb)[M], bool wantV) {
        if (wantV)
        {
#pragma unroll
          for (int g = 0; g < NG; g++) v[g] = *reinterpret_cast<const double*>(vAddr + slotV * VSTAGE + g * 128);
#pragma unroll
          for (int j = 0; j < M / 2; j++)
          {
            d2 t = *reinterpret_cast<const d2*>(mbAddr[j] + slotV * MSTAGE);
            mb[2 * j] = t[0]; mb[2 * j + 1] = t[1];
          }
        }
#pragma unroll
        for (int j = 0; j < M / 2; j++)
        {
          d2 t = *reinterpret_cast<const d2*>(maAddr[j] + slotA * MSTAGE);
          ma[2 * j] = t[0]; ma[2 * j + 1] = t[1];
        }
      };
      read_set(0, 0, vX, maX, mbX, false);
      q_phase(maX, qA);
      read_set(0, 1, vX, maX, mbX, true);

      constexpr int NVR = QV ? (NG + 1) / 2 : NG;   // read slots of the V slab (QV forms: two groups to an instruction)
      constexpr int NRD = NVR + M;                  // ds_read instructions per operand set
      constexpr int NMF = M * NG;                   // MFMAs per phase
      constexpr int RDSTEP = (NMF / NRD) >= 2 ? 2 : 1;
      constexpr int DMASTEP = NMF / IPS > 0 ? NMF / IPS : 1;
      long long tWait = 0, tRatio = 0, tQ = 0, tOut = 0, tReal = 0;
      auto tick = [&]() -> long long {
        if constexpr (INSTR) { __builtin_amdgcn_sched_barrier(0); long long c = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); return c; }
        return 0;
      };
      long long tLoop0 = 0;
      if constexpr (INSTR) { tLoop0 = (long long) __builtin_amdgcn_s_memrealtime(); tReal = -tLoop0; }
      // step s sits in slot u; reads stage s+1 (slot u1) and the ma part of stage s+2 (slot u2); refills slot u
      auto half = [&](int s, int u, int u1, int u2, const double (&v)[NG], const double (&ma)[M], const double (&mb)[M],
                      double (&vn)[NG], double (&man)[M], double (&mbn)[M], const double (&qc)[NG],
                      double (&qn)[NG]) {
        double ratio[NG];
        constexpr int PP = (P <= M) ? P : M;
        // LATE: the asm chains' sums wait until the second product has gone by (below)
        constexpr bool LATE = QV && PP > 1;
        double qp[NG][PP];
        const long long c0 = tick();
        // stages s+1, s+2 landed.  (Refills past the last step re-read the last step's rows -- L2 hits; skipping
        // them was measured slower both ways: a branch per DMA splits the MFMA stream into basic blocks, and
        // issuing them under EXEC = 0 stalls on every EXEC write.)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 3) * IPS) : "memory");
        const long long c1 = tick();
        ratio_phase(v, qc, ratio);
        __builtin_amdgcn_sched_barrier(0);
        const long long c2 = tick();
        {
#pragma unroll
          for (int g = 0; g < NG; g++)
#pragma unroll
            for (int p = 0; p < PP; p++) qp[g][p] = 0.0;
#pragma unroll
          for (int m = 0; m < M; m++)
#pragma unroll
            for (int g = 0; g < NG; g++)
            {
              if constexpr (QV)
              {
                // The first product with VGPR results (round 6).  Every MFMA the compiler selects in a function that needs
                // AGPRs at all gets an AGPR destination, and the quotient block then opens with 2 NG v_accvgpr_read_b32 to
                // fetch Q -- which only the VALU ever reads.  Spelled in asm the accumulate chain stays in VGPRs (ACC_CD = 0:
                // srcC and vdst are one register class per instruction).  Hazards the compiler cannot see inside the asm, by
                // construction: a chain's links are NG >= 2 MFMAs apart (DMFMA 4x4 -> overlapped SrcC: 4 wait states), the
                // VALU reads Q a whole second product (NG M MFMAs) later, the operands come from ds_reads behind an
                // s_waitcnt and from registers last written in the prologue.
                if (m < PP) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=&v"(qp[g][m % PP]) : "v"(ma[m]), "v"(sb[g][m]));
                else asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(qp[g][m % PP]) : "v"(ma[m]), "v"(sb[g][m]));
              }
              else
              qp[g][m % PP] = __builtin_amdgcn_mfma_f64_4x4x4f64(ma[m], sb[g][m], qp[g][m % PP], 0, 0, 0);
              const int i = m * NG + g;
              if (i % RDSTEP == RDSTEP - 1 && i / RDSTEP < NRD)
              {
                const int r = i / RDSTEP;
                if (r < NVR)
                {
                  if constexpr (QV)
                  {
                    // two groups' magnitudes per instruction (ds_read2_b64: the slab's columns of a lane lie 128 bytes apart)
                    vn[2 * r] = *reinterpret_cast<const double*>(vAddr + u1 * VSTAGE + (2 * r) * 128);
                    if (2 * r + 1 < NG) vn[2 * r + 1] = *reinterpret_cast<const double*>(vAddr + u1 * VSTAGE + (2 * r + 1) * 128);
                  }
                  else vn[r] = *reinterpret_cast<const double*>(vAddr + u1 * VSTAGE + r * 128);
                }
                else if (r < NVR + M / 2)
                {
                  d2 t = *reinterpret_cast<const d2*>(mbAddr[r - NVR] + u1 * MSTAGE);
                  mbn[2 * (r - NVR)] = t[0];
                  mbn[2 * (r - NVR) + 1] = t[1];
                }
                else
                {
                  d2 t = *reinterpret_cast<const d2*>(maAddr[r - NVR - M / 2] + u2 * MSTAGE);
                  man[2 * (r - NVR - M / 2)] = t[0];
                  man[2 * (r - NVR - M / 2) + 1] = t[1];
                }
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          if constexpr (!LATE)
          {
#pragma unroll
            for (int g = 0; g < NG; g++)
            {
              double t = qp[g][0];
#pragma unroll
              for (int p = 1; p < PP; p++) t += qp[g][p];
              qn[g] = t;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        const long long c3 = tick();
        {
          const int sc = min(s + NS, sLast);
          const char* vsrc = reinterpret_cast<const char*>(V + (int64_t) sc * 4 * a.ldv + (int64_t) g0 * 16);
          const char* msrc = reinterpret_cast<const char*>(Mv + (int64_t) sc * 4 * KPM);
#pragma unroll
          for (int g = 0; g < NG; g++)
#pragma unroll
            for (int m = 0; m < M; m++)
            {
              acc[g][m] = __builtin_amdgcn_mfma_f64_4x4x4f64(ratio[g], mb[m], acc[g][m], 0, 0, 0);
              const int i = g * M + m;
              if (i % DMASTEP == DMASTEP / 2 && i / DMASTEP < IPS)
              {
                const int j = i / DMASTEP;
                if (j < NJV) glds16_v(vsrc, voffs[j], vringA + u * VSTAGE + j * 1024);
                else glds16(msrc, moffs[j - NJV], mringA + u * MSTAGE + (j - NJV) * 1024);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
#pragma unroll
          for (int m = 0; m < M; m++) if constexpr (DS == 1) dsum[m] += mb[m];
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LATE)
        {
          // THE PLACE OF THESE SUMS IS LOAD-BEARING (round 6, session 2).  With more than one partial chain per group (NG <= 4) the
          // VALU adds the chains up -- and a v_add_f64 that reads the destination of a double-precision MFMA still in flight reads
          // the OLD register: the hazard recognizer pads the MFMAs it selects itself (DMFMA 4x4 write -> VALU read: 6 wait
          // states), not the inside of an asm statement, and the scheduler may hoist the adds above the last asm MFMA as well.
          // With the sums right behind the first product <10, 2> did both (`v_mfma v[126:127] ...; s_nop 0; v_add_f64 ..,
          // v[126:127]`): every W update of a rank 33 .. 40 corpus with two column groups per strip came out 6e-2 wrong
          // (tests/test_gpu_random_shapes.py r1_B20_K40).  Here the whole second product (NG M >= 8 MFMAs) lies between the chains and
          // their sums: the scheduling barrier above keeps the adds below it, the empty statements (volatile, like the MFMAs:
          // they keep their order) tie them to the chain registers -- no wait states to pay (behind the first product they
          // cost the narrow-strip forms 4 %), and the next step's quotient block is where the sums are wanted.
          // tools/isa_mfma_valu_hazard.py audits the shipped code objects for the pattern (CPU test).
#pragma unroll
          for (int g = 0; g < NG; g++)
#pragma unroll
            for (int p = 0; p < PP; p++) asm volatile("" : "+v"(qp[g][p]));
#pragma unroll
          for (int g = 0; g < NG; g++)
          {
            double t = qp[g][0];
#pragma unroll
            for (int p = 1; p < PP; p++) t += qp[g][p];
            qn[g] = t;
          }
        }
        const long long c4 = tick();
        if constexpr (INSTR) { tWait += c1 - c0; tRatio += c2 - c1; tQ += c3 - c2; tOut += c4 - c3; }
      };
      for (int s = s0; s < s1; s += NS)
      {
#pragma unroll
        for (int u = 0; u < NS; u++)
        {
          if (s + u >= s1) break;
          if (u % 2 == 0) half(s + u, u, (u + 1) % NS, (u + 2) % NS, vX, maX, mbX, vY, maY, mbY, qA, qB);
          else half(s + u, u, (u + 1) % NS, (u + 2) % NS, vY, maY, mbY, vX, maX, mbX, qB, qA);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (INSTR)
      {
        const long long tLoop1 = (long long) __builtin_amdgcn_s_memrealtime();
        tReal += tLoop1;
        if (blockIdx.x == 17 && threadIdx.x == 0 && a.dpart)
        {
          long long* o = reinterpret_cast<long long*>(a.dpart);
          o[0] = tWait; o[1] = 0; o[2] = tRatio; o[3] = tQ; o[4] = tOut; o[5] = 0; o[6] = s1 - s0; o[7] = tReal;
        }
        // timeline samples (100 MHz ticks): entry, loop start, loop end of wavefront 0 of a few workgroups
        if (threadIdx.x == 0 && a.dpart && (blockIdx.x & 63) == 17)
        {
          long long* o = reinterpret_cast<long long*>(a.dpart) + 16 + 4 * (blockIdx.x >> 6);
          o[0] = tEntry; o[1] = tLoop0; o[2] = tLoop1;
          if (blockIdx.x == 17) { long long* q = reinterpret_cast<long long*>(a.dpart) + 12; q[0] = tP1; q[1] = tP2; q[2] = tP3; }
        }
      }
    }
  }
  else
  if (s0 < s1)
  {
#pragma unroll
    for (int t = 0; t < NS; t++) issue_stage(s0 + t);
    double ma[M], mb[M], v[NG], qA[NG], qB[NG], ratio[NG];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * IPS) : "memory");
    read_ma(s0, ma);
    q_phase(ma, qA);
    long long tWait = 0, tRead = 0, tRatio = 0, tQ = 0, tOut = 0, tDma = 0;
    auto tick = [&]() -> long long {
      if constexpr (INSTR) { __builtin_amdgcn_sched_barrier(0); long long c = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); return c; }
      return 0;
    };
    // Each half-iteration: wait for stage s+1 -> LDS reads of step s (and ma of s+1) -> as soon as they
    // have landed in registers the slot of stage s is free, so its refill (stage s+NS) is issued
    // right away and the DMA instructions can overlap the matrix work that follows.
    for (int s = s0; s < s1; s += 2)
    {
      long long c0 = tick();
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * IPS) : "memory");
      long long c1 = tick();
      read_v(s, v);
      read_ma(s + 1, ma);
      read_mb(s, mb);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // every ds_read of stage s has returned
      long long c2 = tick();
      issue_stage(s + NS);
      long long c3 = tick();
      ratio_phase(v, qA, ratio);
      long long c4 = tick();
      q_phase(ma, qB);
      long long c5 = tick();
      out_phase(ratio, mb);
      long long c6 = tick();
      if constexpr (INSTR) { tWait += c1 - c0; tRead += c2 - c1; tDma += c3 - c2; tRatio += c4 - c3; tQ += c5 - c4; tOut += c6 - c5; }
      if (s + 1 < s1)
      {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * IPS) : "memory");
        read_v(s + 1, v);
        read_ma(s + 2, ma);
        read_mb(s + 1, mb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_stage(s + 1 + NS);
        ratio_phase(v, qB, ratio);
        q_phase(ma, qA);
        out_phase(ratio, mb);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // drain the run-ahead DMAs before the LDS is released
    if constexpr (INSTR)
    {
      if (blockIdx.x == 17 && threadIdx.x == 0 && a.dpart)
      {
        long long* o = reinterpret_cast<long long*>(a.dpart);
        o[0] = tWait; o[1] = tRead; o[2] = tRatio; o[3] = tQ; o[4] = tOut; o[5] = tDma; o[6] = (s1 - s0 + 1) / 2;
      }
    }
  }

  // where the results go: split partials (a finalize launch forms the result) or the result itself + column statistics
  [[maybe_unused]] int lPart = -1, lStat = 0, lD = -1;
  if constexpr (LIST != 0 && 4 * STG_BYTES <= 160 * 1024)
  {
    // ---- intra-workgroup reduction: the wavefronts that split this strip's contraction add up through the LDS ----------
    // (every DMA of the ring has landed: vmcnt(0) above; the staging slot is the wavefront's own region)
    const int grp = __builtin_amdgcn_readfirstlane(wd->grp);
    if (grp >> 16)
    {
      const int leader = grp & 15, rank = (grp >> 4) & 15, gsize = (grp >> 8) & 15;
      double* mine = reinterpret_cast<double*>(lds + wave * WAVE_REGION);
      if (rank > 0)
      {
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
          for (int m = 0; m < M; m++) mine[(g * M + m) * 64 + lane] = acc[g][m];
#pragma unroll
        for (int m = 0; m < M; m++) mine[(NG * M + m) * 64 + lane] = dsum[m];
      }
      __syncthreads();
      if (rank > 0) return;
      for (int r = 1; r < gsize; r++) // rank order: the sum does not depend on which wavefront finished first
      {
        const double* src = reinterpret_cast<const double*>(lds + (leader + r) * WAVE_REGION);
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
          for (int m = 0; m < M; m++) acc[g][m] += src[(g * M + m) * 64 + lane];
#pragma unroll
        for (int m = 0; m < M; m++) dsum[m] += src[(NG * M + m) * 64 + lane];
      }
    }
  }
  if constexpr (LIST != 0)
  {
    lPart = __builtin_amdgcn_readfirstlane(wd->partIdx);
    lStat = __builtin_amdgcn_readfirstlane(wd->statIdx);
    lD = __builtin_amdgcn_readfirstlane(wd->dIdx);
  }
  auto whole = [&]() -> bool { if constexpr (LIST != 0) return lPart < 0; else return a.nsplit == 1; };
  if constexpr (DS == 1)
  {
#pragma unroll
    for (int m = 0; m < M; m++)
    {
      double d = dsum[m];
      d += __shfl_xor(d, 16);
      d += __shfl_xor(d, 32);
      dsum[m] = d;
    }
  }
  else if constexpr (DS == 2)
  {
    // the prologue's sum of the partials the launch in front left (un-split launches only: the launcher sees to it)
#pragma unroll
    for (int m = 0; m < M; m++) dsum[m] = colL[M * x + m];
  }
  else if (whole())
  {
    // DS == 0: the column sums of Mv were taken by launch_colsum into slot (buf, split 0) of dpart (the wide
    // ranks have no registers to spare for M accumulators every wavefront would hold identically)
    load_vec5<M>(dsum, a.dpart + (int64_t) buf * KPM + M * x);
  }

  if (whole())
  {
    double nrE[M], ss[M], mx[M];
    [[maybe_unused]] double csum[M];     // DS == 2, W update: column sums of the rows this wavefront writes (colOut)
    constexpr bool COLOUT = DS == 2 && !NORMQ;
    [[maybe_unused]] double nrL = 1.0;   // SIDEQ: the norm of component `lane` (the old side row leaves normalised)
    if constexpr (NORMQ)
    {
      const double* ldsN = reinterpret_cast<const double*>(sideL) + 64;
#pragma unroll
      for (int m = 0; m < M; m++) nrE[m] = ldsN[M * x + m];
      if (lane < KP) nrL = ldsN[lane];
    }
    else
    {
      if (a.nrmMode) load_vec5<M>(nrE, a.nrm + (int64_t) buf * KPM + M * x);
      if constexpr (SIDEQ) if (strip == 0 && lane < KP) nrL = a.nrm[(int64_t) buf * KPM + lane];
    }
    if (a.nrmMode == 2)
    {
#pragma unroll
      for (int m = 0; m < M; m++) dsum[m] = fdiv_pos(dsum[m], nrE[m]); // sum_r W[r][k] = (sum_r W'[r][k]) / nrm_k
    }
    // the M denominators are shared by every column of the strip: one reciprocal + Newton step each, then
    // quotient + residual correction per element (as in fdiv_pos)
    double dd[M], dy[M];
#pragma unroll
    for (int m = 0; m < M; m++)
    {
      dd[m] = fmax(dsum[m], kEpsilon);
      double y0 = __builtin_amdgcn_rcp(dd[m]);
      dy[m] = __builtin_fma(y0, __builtin_fma(-dd[m], y0, 1.0), y0);
      ss[m] = 0.0;
      mx[m] = -INFINITY;
      if constexpr (COLOUT) csum[m] = 0.0;
    }
    // S_old comes from the stationary registers, not from memory: sb holds S (already divided by nrm when the
    // normalisation is deferred) as (col 4 blk + x, k = M y + m); the result layout is (col 4 blk + y,
    // k = M x + m) -- the same block with x and y exchanged, one lane permutation.  At the end of the loop
    // all 1024 wavefronts stand here at once, so every byte not moved is time: re-reading S was a third
    // of the epilogue's traffic.  (H update with deferred normalisation: (H/nrm) acc == H (acc/nrm) up
    // to rounding, so the divided rows serve there too.)
    const int srcLane = y + 4 * blk + 16 * x;
    // The results leave through LDS: in the MFMA result layout a lane holds 64 bytes of a row and a store
    // instruction would write four 16-byte pieces 64 bytes apart per row -- quarter-filled write requests,
    // measured at ~1.7 TB/s with every wavefront storing at once.  A group's 16 rows are contiguous in memory
    // (16 x Kp doubles), so they are laid out in the wavefront's (now idle) ring with the rows 16 bytes apart in
    // bank phase, read back linearly and stored as whole kilobytes per instruction.
    constexpr int ROWB = KP * 8 + 16;
    constexpr int GRPB = 16 * ROWB;
    constexpr int GB = (WAVE_LDS / GRPB) >= NG ? NG : (WAVE_LDS / GRPB);
    static_assert(GB >= 1, "result staging does not fit the ring");
    constexpr int CPR = KP / 2;            // 16-byte chunks per row
    constexpr int NST = 16 * CPR / 64;     // store instructions per group
    char* stg = lds + wave * WAVE_REGION;
    [[maybe_unused]] double wsd[M], numS[M], denS[M], vsd[NG];
    if constexpr (SIDEQ)
    {
      static_assert(KP <= 64 && LIST == 0, "side-column partials: one lane per component at the end");
      // w_k = W'[R-1][k] / nrm_k for this lane's k = M x + m (nrE holds those norms), the side bin's magnitudes of this lane's frames
      const double* wl = reinterpret_cast<const double*>(sideL);
      const double* vl = reinterpret_cast<const double*>(sideL + 1024);
#pragma unroll
      for (int m = 0; m < M; m++) { wsd[m] = fdiv_pos(wl[M * x + m], nrE[m]); numS[m] = 0.0; denS[m] = 0.0; }
#pragma unroll
      for (int g = 0; g < NG; g++)
      {
        const int col = (g0 + g) * 16 + 4 * blk + y;
        vsd[g] = (g < ng && col < a.C) ? vl[g * 16 + 4 * blk + y] : 0.0;
      }
    }
#pragma unroll
    for (int gb = 0; gb < NG; gb += GB)
    {
#pragma unroll
      for (int g = gb; g < gb + GB && g < NG; g++)
      {
        const int col = (g0 + g) * 16 + 4 * blk + y;
        const bool live = g < ng && col < a.C;
        char* row = stg + (g - gb) * GRPB + (4 * blk + y) * ROWB + (M * x) * 8;
        [[maybe_unused]] double rr[M];
#pragma unroll
        for (int m = 0; m < M; m += 2)
        {
          double r2[2];
#pragma unroll
          for (int e = 0; e < 2; e++)
          {
            const double so = __shfl(sb[g][m + e], srcLane);
            const double v = so * acc[g][m + e];
            const double r0 = v * dy[m + e];
            const double r = __builtin_fma(__builtin_fma(-dd[m + e], r0, v), dy[m + e], r0);
            r2[e] = r;
            if constexpr (SIDEQ) rr[m + e] = r;
            if (live)
            {
              ss[m + e] = __builtin_fma(r, r, ss[m + e]);
              mx[m + e] = fmax(mx[m + e], r);
              if constexpr (COLOUT) csum[m + e] += r;
            }
          }
          *reinterpret_cast<d2*>(row + m * 8) = d2{r2[0], r2[1]};
        }
        if constexpr (SIDEQ)
        {
          // the four lanes x = 0 .. 3 of a frame hold its KP new activations: the dot product with w closes over the quad
          double qs = 0.0;
#pragma unroll
          for (int m = 0; m < M; m++) qs = __builtin_fma(rr[m], wsd[m], qs);
          qs += quad_dpp<0xB1>(qs);
          qs += quad_dpp<0x4E>(qs);
          const double ratio = live ? fdiv_pos(vsd[g], fmax(qs, kEpsilon)) : 0.0;
#pragma unroll
          for (int m = 0; m < M; m++)
          {
            numS[m] = __builtin_fma(ratio, rr[m], numS[m]);
            denS[m] += live ? rr[m] : 0.0;
          }
        }
      }
#pragma unroll
      for (int g = gb; g < gb + GB && g < NG; g++)
      {
        if (g < ng)
        {
#pragma unroll
          for (int j = 0; j < NST; j++)
          {
            const int c = 64 * j + lane;
            const int r = c / CPR, piece = c % CPR;
            const d2 t = *reinterpret_cast<const d2*>(stg + (g - gb) * GRPB + r * ROWB + piece * 16);
            const int col = (g0 + g) * 16 + r;
            if (col < a.C) store_result16(S + (int64_t) col * KPM + piece * 2, t);
          }
        }
      }
    }
    if constexpr (SIDEQ)
    {
      // the 16 lanes (blk, y) of equal x hold partial sums of the same components: added in fixed order through the (idle)
      // ring, one lane per component; the wavefront's (num, den) go where side_slices_kernel would have left a slice's
      double* red = reinterpret_cast<double*>(stg);          // [2][16][KP]
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the staged results have been read back
      const int jr = 4 * blk + y;
#pragma unroll
      for (int m = 0; m < M; m++)
      {
        red[jr * KP + M * x + m] = numS[m];
        red[(16 + jr) * KP + M * x + m] = denS[m];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane < KP)
      {
        double pn[16], pd[16];
#pragma unroll
        for (int j = 0; j < 16; j++) { pn[j] = red[j * KP + lane]; pd[j] = red[(16 + j) * KP + lane]; }
        double n = 0.0, d = 0.0;
#pragma unroll
        for (int j = 0; j < 16; j++) { n += pn[j]; d += pd[j]; }
        double* sp = a.sidePart + ((int64_t) buf * a.wavesPerBuf + strip) * 2 * KPM;
        sp[lane] = n;
        sp[KPM + lane] = d;
        if (strip == 0) a.sideWold[(int64_t) buf * KPM + lane] = reinterpret_cast<const double*>(sideL)[lane] / nrL;
      }
    }
    if (a.statPart)
    {
      // column statistics of the rows this wavefront wrote: lanes of equal x hold the same k = M x + m;
      // fixed butterfly over (blk, y), then lanes 0..3 store
#pragma unroll
      for (int m = 0; m < M; m++)
      {
        double t = ss[m], u = mx[m];
#pragma unroll
        for (int off = 4; off < 64; off <<= 1)
        {
          t += __shfl_xor(t, off);
          u = fmax(u, __shfl_xor(u, off));
        }
        ss[m] = t;
        mx[m] = u;
      }
      if (lane < 4)
      {
        double* sp = a.statPart + (LIST ? (int64_t) lStat : ((int64_t) buf * a.wavesPerBuf + strip)) * 2 * KPM + M * x;
#pragma unroll
        for (int m = 0; m < M; m++) { sp[m] = ss[m]; sp[KPM + m] = mx[m]; }
      }
    }
    if constexpr (COLOUT)
    {
      // the same fixed butterfly over (blk, y); lanes 0 .. 3 store their M components
#pragma unroll
      for (int m = 0; m < M; m++)
      {
        double t = csum[m];
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) t += __shfl_xor(t, off);
        csum[m] = t;
      }
      if (lane < 4)
      {
        double* cp = a.colOut + ((int64_t) buf * a.wavesPerBuf + strip) * KPM + M * x;
#pragma unroll
        for (int m = 0; m < M; m++) cp[m] = csum[m];
      }
    }
  }
  else
  {
    double* part = a.part + (LIST ? (int64_t) lPart : ((int64_t) buf * a.nsplit + split)) * a.Cp * KPM;
#if FLUHIP_PARTIAL_STAGED
    // The partials leave like the results (round 6): laid out in the wavefront's idle ring, read back linearly and stored a
    // kilobyte per instruction, write-through -- whole lines per request instead of 64 sixteen-byte pieces 256 bytes apart,
    // and nothing left dirty in the L2 for the end-of-kernel release to write back in front of the finalize launch.
    {
      constexpr int ROWB = KP * 8 + 16;
      constexpr int GRPB = 16 * ROWB;
      constexpr int GB = (WAVE_LDS / GRPB) >= NG ? NG : (WAVE_LDS / GRPB);
      static_assert(GB >= 1, "partial staging does not fit the ring");
      constexpr int CPR = KP / 2, NST = 16 * CPR / 64;
      char* stg = lds + wave * WAVE_REGION;
#pragma unroll
      for (int gb = 0; gb < NG; gb += GB)
      {
#pragma unroll
        for (int g = gb; g < gb + GB && g < NG; g++)
        {
          char* row = stg + (g - gb) * GRPB + (4 * blk + y) * ROWB + (M * x) * 8;
#pragma unroll
          for (int m = 0; m < M; m += 2) *reinterpret_cast<d2*>(row + m * 8) = d2{acc[g][m], acc[g][m + 1]};
        }
#pragma unroll
        for (int g = gb; g < gb + GB && g < NG; g++)
        {
          if (g < ng)
          {
#pragma unroll
            for (int j = 0; j < NST; j++)
            {
              const int c = 64 * j + lane;
              const int r = c / CPR, piece = c % CPR;
              const d2 t = *reinterpret_cast<const d2*>(stg + (g - gb) * GRPB + r * ROWB + piece * 16);
              store_result16(part + (int64_t) ((g0 + g) * 16 + r) * KPM + piece * 2, t);
            }
          }
        }
      }
    }
#else
#pragma unroll
    for (int g = 0; g < NG; g++)
    {
      if (g < ng)
      {
        const int col = (g0 + g) * 16 + 4 * blk + y;
        double* pp = part + (int64_t) col * KPM + M * x;
        // (the form of rounds 1 - 5.  Written through piece by piece -- 16 bytes per lane 256 bytes apart -- it measured slower
        //  still: profiles/r06/partials_write_through.txt)
#pragma unroll
        for (int m = 0; m < M; m++) pp[m] = acc[g][m];
      }
    }
#endif
    if (DS && (LIST ? lD >= 0 : strip == 0) && blk == 0 && y == 0)
    {
      double* dp = a.dpart + (LIST ? (int64_t) lD : ((int64_t) buf * a.nsplit + split)) * KPM + M * x;
#pragma unroll
      for (int m = 0; m < M; m++) dp[m] = dsum[m];
    }
  }
  if (stamp && threadIdx.x == 0)
  {
    __builtin_amdgcn_s_waitcnt(0); // the results have left
    const long long c1 = (long long) __builtin_readcyclecounter(), r1 = (long long) __builtin_amdgcn_s_memrealtime();
    a.clk[0] += 1;
    a.clk[1] += c1 - clkC0;
    a.clk[2] += r1 - clkR0;
  }
  if constexpr (INSTR)
  {
    __builtin_amdgcn_s_waitcnt(0);
    if (threadIdx.x == 0 && a.dpart && (blockIdx.x & 63) == 17)
      reinterpret_cast<long long*>(a.dpart)[16 + 4 * (blockIdx.x >> 6) + 3] = (long long) __builtin_amdgcn_s_memrealtime();
  }
}


// work-list mode: the LIST instantiation of the same kernel, one workgroup per four descriptors
template <int M, int NG, int NS, int MODE, int KPM = 4 * M>
static void launch5_list(const UpdateArgs& a, hipStream_t s)
{
  Upd5Args k{};
  k.V = a.V; k.ldv = a.ldv; k.strideV = a.strideV;
  k.Mv = a.Mv; k.strideM = a.strideM;
  k.S = a.S; k.strideS = a.strideS;
  k.R = a.R; k.C = a.C; k.B = a.B;
  k.nGroups = (a.C + 15) / 16;
  k.wavesPerBuf = 1; k.wgPerBuf = 1; k.nSteps = (a.R + 3) / 4; k.nsplit = 1; k.stepsPerSplit = k.nSteps;
  k.part = a.part; k.dpart = a.dpart; k.Cp = a.Cp;
  k.nrm = a.nrm; k.nrmMode = a.nrmMode; k.statPart = a.listPartial ? nullptr : a.statPart;
  k.clk = a.clk; k.xcdMap = 0; k.list = a.list;
  constexpr int KP = 4 * M, SPR = KP / 2;
  constexpr int NJV = (32 * NG + 63) / 64, NJM = (4 * SPR + 63) / 64;
  constexpr size_t ring = (size_t) NS * (NJV + NJM) * 1024, stg = (size_t) (NG * M + M) * 512;
  constexpr size_t shmem = 4 * ((stg > ring && 4 * stg <= 160 * 1024) ? stg : ring);
  static_assert(shmem <= 160 * 1024, "LDS");
  auto kern = nmf_update5_kernel<M, NG, NS, 1, 0, MODE, 1, 1, 0, KPM>;
  request_dynamic_lds(kern, (size_t) (shmem));
  hipLaunchKernelGGL(kern, dim3((unsigned) a.listWGs), dim3(256), shmem, s, k);
}

template <int M, int NG, int NS, int WPS, int INSTR = 0, int MODE = 0, int DS = 1, int SIDEQ = 0, int KPM = 4 * M>
static void launch5_t(const UpdateArgs& a, int wavesPerBuf, hipStream_t s)
{
  Upd5Args k;
  if (a.dryRun) return;
  k.sidePart = SIDEQ ? a.sideOut : nullptr; k.sideWold = SIDEQ ? a.sideWold : nullptr;
  k.cmbStat = a.cmbStat; k.cmbSide = a.cmbSide; k.cmbWold = a.cmbWold; k.cmbNrmOut = a.cmbNrmOut; k.cmbRowOut = a.cmbRowOut;
  k.cmbParts = a.cmbParts; k.cmbSlices = a.cmbSlices; k.cmbK = a.cmbK;
  k.colIn = a.colIn; k.colOut = a.colOut; k.colInN = a.colInN;
  k.V = a.V; k.ldv = a.ldv; k.strideV = a.strideV;
  k.Mv = a.Mv; k.strideM = a.strideM;
  k.S = a.S; k.strideS = a.strideS;
  k.R = a.R; k.C = a.C; k.B = a.B;
  k.nGroups = (a.C + 15) / 16;
  k.wavesPerBuf = wavesPerBuf;
  k.wgPerBuf = (wavesPerBuf + 4 * WPS - 1) / (4 * WPS);
  k.nSteps = (a.R + 3) / 4;
  k.nsplit = a.nsplit < 1 ? 1 : a.nsplit;
  k.stepsPerSplit = (k.nSteps + k.nsplit - 1) / k.nsplit;
  k.part = a.part; k.dpart = a.dpart; k.Cp = a.Cp;
  // split contraction: the stationary rows are normalised on load all the same; the rest of the deferred form
  // (epilogue arithmetic, statistics) is the finalize kernel's
  k.nrm = a.nrm; k.nrmMode = a.nrmMode; k.statPart = a.nsplit > 1 ? nullptr : a.statPart;
  k.clk = a.clk;
  k.xcdMap = a.B >= 8 ? 1 : 0;
  k.list = nullptr;
  const int bufs = k.xcdMap ? (int) round_up(a.B, 8) : a.B;
  const unsigned grid = (unsigned) (bufs * k.wgPerBuf * k.nsplit);
  constexpr int KP = 4 * M, SPR = KP / 2;
  constexpr int NJV = (32 * NG + 63) / 64, NJM = (4 * SPR + 63) / 64;
  constexpr size_t shmem = (size_t) 4 * WPS * ((NS * (NJV + NJM) + (SIDEQ ? 1 + (NG * 128 + 1023) / 1024 : 0)) * 1024 + (DS == 2 ? 512 : 0));
  static_assert(shmem <= 160 * 1024, "LDS ring does not fit");
  auto kern = nmf_update5_kernel<M, NG, NS, WPS, INSTR, MODE, DS, 0, SIDEQ, KPM>;
  request_dynamic_lds(kern, (size_t) (shmem));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * WPS), shmem, s, k);
  if (k.nsplit > 1)
    launch_update_finalize(a.S, a.strideS, a.part, a.dpart, a.C, a.Kp, a.Cp, k.nsplit, a.B, s, a.nrm, a.nrmMode,
                           a.statPart);
}

#ifndef FLUHIP_K5_OFFSIZE_TU
static int max_groups(int M)
{
  if (M <= 8) return 9;
  if (M <= 16) return 4; // (M = 12: five groups are the most without spills; four keep the strips of rank 64 and its whole rounds)
  return 2;              // (M = 24: two, as at M = 32)
}

// strips (wavefronts) per buffer: fill the 1024 SIMDs in whole rounds, then as few strips as the
// register budget allows (more groups per strip = more reuse of the moving-factor slab)
int nmf_update5_waves_per_buffer(int C, int Kp, int B)
{
  const int M = Kp / 4, G = (C + 15) / 16, ngmax = max_groups(M);
  const int wmin = (G + ngmax - 1) / ngmax;
  static const int forceW = [] { const char* e = fluhip::ab_getenv("FLUHIP_PLAN_W"); return e ? std::atoi(e) : 0; }();
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


#endif // !FLUHIP_K5_OFFSIZE_TU

// ring depth bounded by the 160 KiB of LDS: 4*WPS wavefronts x NS x (V + Mv stage)
template <int M, int NG, int WPS>
constexpr int ring_depth()
{
  constexpr int NJ = (32 * NG + 63) / 64 + (4 * (2 * M) + 63) / 64;
  constexpr int fit = 160 / (4 * WPS * NJ);
  return fit >= 6 ? 6 : (fit < 3 ? 3 : fit);
}

#ifndef FLUHIP_K5_OFFSIZE_TU
// bit 0: the launch also left the side-column partials UpdateArgs::sideOut asks for; bit 1: it did the norm combine
// UpdateArgs::cmb* describes (the SIDEQ instantiations: ranks up to 64, the production pipeline form of the rank, strips of
// two groups or more, deferred normalisation, un-split)
template <int M, int NG, int WPS>
static int launch5_ng(const UpdateArgs& a, int w, int ng, hipStream_t s)
{
  if constexpr (NG == 1) launch5_t<M, 1, ring_depth<M, 1, WPS>(), WPS>(a, w, s);
  else
  {
    // (w <= kSideFromHSlots: the partials are laid out [B][strips][2][Kp] in an area that holds kSideFromHSlots slices per
    //  buffer and generation, wnorm_side_part -- long buffers take more strips than that and keep the side-column launch)
    const bool sideq = a.sideOut && a.sideWold && a.nrmMode == 2 && a.nsplit <= 1 && w <= kSideFromHSlots;
    const bool normq = sideq && a.cmbStat && a.cmbSide && a.cmbWold && a.cmbNrmOut && a.cmbRowOut && (a.R + 3) / 4 > 12;
    (void) normq;
    if (ng >= NG)
    {
      constexpr int NS = ring_depth<M, NG, WPS>();
      if constexpr (((M == 8 && (NG == 9 || NG == 8)) || (M == 4 && NG == 3)) && WPS == 1)
      {
        // FLUHIP_K5_INSTR=1: per-phase s_memtime breakdown of one wavefront (tools/phase_breakdown.py)
        static const int instr = [] { const char* e = fluhip::ab_getenv("FLUHIP_K5_INSTR"); return e ? std::atoi(e) : 0; }();
        static const int imode = [] { const char* e = fluhip::ab_getenv("FLUHIP_K5_MODE"); return e ? std::atoi(e) : 1; }();
        if (instr && imode == 1) { launch5_t<M, NG, NS, WPS, 1, 1>(a, w, s); return 0; }
        if (instr) { launch5_t<M, NG, NS, WPS, 1>(a, w, s); return 0; }
      }
      if constexpr (WPS == 1 && NS >= 4 && NS % 2 == 0)
      {
        // FLUHIP_K5_MODE: 0 = grouped reads/refill (the first LDS-DMA form), 1 = overlapped with a second
        // operand set, 2 = overlapped with one set refilled in place (the only overlapped form that fits
        // M = 32); default: 2 for M >= 16, 1 below
        static const int mode = [] { const char* e = fluhip::ab_getenv("FLUHIP_K5_MODE"); return e ? std::atoi(e) : -1; }();
        // (rounds 3 - 4 honoured mode 2 from M = 16 on only: forced onto M = 8 it came out wrong on a 128-buffer corpus and was
        //  taken for a race of the in-place refill.  Round 5 found the cause elsewhere -- the store hazard behind the inline asm
        //  of store_result16, which that one instantiation's register allocation happened to expose -- and with it repaired the
        //  form is right at every rank; FLUHIP_K5_MODE_ANY is no longer needed and is accepted for old scripts.)
        const int eff = mode >= 0 ? mode : (M >= 16 ? 2 : 1);
        if constexpr (M == 32)
        {
          // rank 65..128: the in-place overlapped form only fits without the M column-sum accumulators;
          // they are taken once per launch by a small pre-pass instead (DS = 0)
          if (eff == 2 && a.colsumScratch)
          {
            const int ns = a.nsplit < 1 ? 1 : a.nsplit;
            // slot (buffer, split 0) of dpart takes the sums, the other splits' slots zero (the finalize adds them up)
            if (a.colsumInPlace) { /* the sums are in their slots already */ }
            else if (a.colsumGiven) launch_colsum_spread(a.colsumGiven, a.Kp, a.B, a.dpart, (int64_t) ns * a.Kp, ns - 1, s);
            else launch_colsum(a.Mv, a.strideM, a.R, a.Kp, a.B, a.dpart, (int64_t) ns * a.Kp, a.colsumScratch, s, ns - 1);
            launch5_t<M, NG, NS, WPS, 0, 2, 0>(a, w, s);
            return 0;
          }
        }
        else if (eff == 2)
        {
          if constexpr (M == 16)
          {
            if (normq) { launch5_t<M, NG, NS, WPS, 0, 2, 1, 3>(a, w, s); return 3; }
            if (sideq) { launch5_t<M, NG, NS, WPS, 0, 2, 1, 1>(a, w, s); return 1; }
          }
          launch5_t<M, NG, NS, WPS, 0, 2>(a, w, s);
          return 0;
        }
        if constexpr (M <= 16)
          if (eff == 1)
          {
            if constexpr (M <= 8)
            {
              // (M = 8 at nine groups per strip sits at 512 registers: the prologue of the norm form would spill 14 of them)
              if constexpr (!(M == 8 && NG == 9))
              {
                // round 6, rank 32: no column-sum accumulators in the loop and the first product's results in VGPRs when the
                // W update in front left the column sums of its rows (UpdateArgs::colIn; return bit 2)
                // (eight groups per strip: the norm form's prologue and the VGPR first product together spill 66 registers)
                if constexpr (M == 8 && NG <= 7)
                  if (normq && a.colIn && a.colInN > 0) { launch5_t<M, NG, NS, WPS, 0, 1, 2, 3>(a, w, s); return 7; }
                if (normq) { launch5_t<M, NG, NS, WPS, 0, 1, 1, 3>(a, w, s); return 3; }
              }
              if (sideq) { launch5_t<M, NG, NS, WPS, 0, 1, 1, 1>(a, w, s); return 1; }
              // ... and the W update between two such H updates: its column sums are the denominators of the side-column
              // partials the H update in front left; it leaves the column sums of the rows it writes (colOut)
              if constexpr (M == 8 && NG <= 8)
                if (a.colIn && a.colInN > 0 && a.colOut && a.nsplit <= 1 && a.nrmMode == 1 && a.statPart)
                {
                  launch5_t<M, NG, NS, WPS, 0, 1, 2, 0>(a, w, s);
                  return 4;
                }
            }
            launch5_t<M, NG, NS, WPS, 0, 1>(a, w, s);
            return 0;
          }
      }
      launch5_t<M, NG, NS, WPS>(a, w, s);
    }
    else return launch5_ng<M, NG - 1, WPS>(a, w, ng, s);
  }
  return 0;
}

#endif // !FLUHIP_K5_OFFSIZE_TU

#ifdef FLUHIP_K5_OFFSIZE_TU
// Off-size ranks (round 5): fewer MFMAs per product than the rank the arrays are laid out for (KPM) --
//   arrays of rank 32 : M = 6 (ranks 17 .. 24)                          arrays of rank 64 : M = 10 / 12 / 14 (33 .. 40 / 48 / 56)
//   arrays of rank 128: M = 18, 20 .. 28 (65 .. 72, .. 80, .. 112)
// One pipeline form each: the two-operand-set form up to M = 14, with the side column of the next W update and the norm
// combine riding in the H launch as at ranks 32 / 64; the in-place form with its column sums
// from M = 18 on (28 with the column sums from the pre-pass, as 32).  They are compiled as their own translation unit (kernels_nmf5_off.hip includes this file with
// FLUHIP_K5_OFFSIZE_TU defined) so that the two halves of the instantiation list build side by side.
template <int M, int NG, int KPM>
static int launch5_off_ng(const UpdateArgs& a, int w, int ng, hipStream_t s)
{
  if constexpr (NG == 1) { launch5_t<M, 1, ring_depth<M, 1, 1>(), 1, 0, 0, 1, 0, KPM>(a, w, s); return 0; }
  else
  {
    if (ng >= NG)
    {
      constexpr int NS = ring_depth<M, NG, 1>();
      static_assert(NS >= 4 && NS % 2 == 0, "overlapped pipeline: even ring depth >= 4");
      if constexpr (M <= 14)
      {
        constexpr int MODE = 1;   // (two operand sets fit up to M = 14 at four groups -- 460 of 512 registers -- and measure 3 % ahead of the in-place form there: rank 56 51.6 -> 50.1 ms per 50 iterations of the bench corpus)
        const bool sideq = a.sideOut && a.sideWold && a.nrmMode == 2 && a.nsplit <= 1 && w <= kSideFromHSlots;
        const bool normq = sideq && a.cmbStat && a.cmbSide && a.cmbWold && a.cmbNrmOut && a.cmbRowOut && (a.R + 3) / 4 > 12;
        if (normq) { launch5_t<M, NG, NS, 1, 0, MODE, 1, 3, KPM>(a, w, s); return 3; }
        if (sideq) { launch5_t<M, NG, NS, 1, 0, MODE, 1, 1, KPM>(a, w, s); return 1; }
        launch5_t<M, NG, NS, 1, 0, MODE, 1, 0, KPM>(a, w, s);
      }
      else if constexpr (M >= 28)
      {
        // as at M = 32: the in-place form fits without the M column-sum accumulators only; a pre-pass takes them
        if (a.colsumScratch)
        {
          const int ns = a.nsplit < 1 ? 1 : a.nsplit;
          if (a.colsumInPlace) { /* the sums are in their slots already */ }
          else if (a.colsumGiven) launch_colsum_spread(a.colsumGiven, a.Kp, a.B, a.dpart, (int64_t) ns * a.Kp, ns - 1, s);
          else launch_colsum(a.Mv, a.strideM, a.R, a.Kp, a.B, a.dpart, (int64_t) ns * a.Kp, a.colsumScratch, s, ns - 1);
          launch5_t<M, NG, NS, 1, 0, 2, 0, 0, KPM>(a, w, s);
        }
        else launch5_t<M, NG, NS, 1, 0, 0, 1, 0, KPM>(a, w, s);
      }
      else launch5_t<M, NG, NS, 1, 0, 2, 1, 0, KPM>(a, w, s);
      return 0;
    }
    return launch5_off_ng<M, NG - 1, KPM>(a, w, ng, s);
  }
}
#endif // FLUHIP_K5_OFFSIZE_TU

// the instantiation whose strip width is the list's widest; one pipeline form per rank class: overlapped with two operand
// sets up to rank 32, refilled in place at rank 64, grouped at rank 128 (the in-place form has no room for the column sums
// there, and the pre-pass that replaces them deals its slots per equal-length buffer)
template <int M, int NG, int KPM = 4 * M>
static void launch5_list_ng(const UpdateArgs& a, int ng, hipStream_t s)
{
  if constexpr (NG == 1) launch5_list<M, 1, ring_depth<M, 1, 1>(), 0, KPM>(a, s);
  else
  {
    if (ng >= NG)
    {
      constexpr int NS = ring_depth<M, NG, 1>();
      constexpr int MODE = (NS >= 4 && NS % 2 == 0) ? (M >= 28 ? 0 : (M >= 16 ? 2 : 1)) : 0;
      launch5_list<M, NG, NS, MODE, KPM>(a, s);
    }
    else launch5_list_ng<M, NG - 1, KPM>(a, ng, s);
  }
}

#ifdef FLUHIP_K5_OFFSIZE_TU
// the off-size forms' entry points (called from launch_nmf_update5 in the other translation unit); -1 / false: no such form,
// the caller runs the padded rank
int launch_nmf_update5_offsize(const UpdateArgs& a, int kc, int w, int ng, hipStream_t s)
{
  if (a.Kp == 32 && kc == 24) return launch5_off_ng<6, 9, 32>(a, w, ng, s);
  if (a.Kp == 64)
    switch (kc)
    {
    case 40: return launch5_off_ng<10, 4, 64>(a, w, ng, s);
    case 48: return launch5_off_ng<12, 4, 64>(a, w, ng, s);
    case 56: return launch5_off_ng<14, 4, 64>(a, w, ng, s);
    default: break;
    }
  if (a.Kp == 128)
    switch (kc)
    {
    case 72: return launch5_off_ng<18, 2, 128>(a, w, ng, s);
    case 80: return launch5_off_ng<20, 2, 128>(a, w, ng, s);
    case 88: return launch5_off_ng<22, 2, 128>(a, w, ng, s);
    case 96: return launch5_off_ng<24, 2, 128>(a, w, ng, s);
    case 104: return launch5_off_ng<26, 2, 128>(a, w, ng, s);
    case 112: return launch5_off_ng<28, 2, 128>(a, w, ng, s);
    default: break;
    }
  return -1;
}
bool launch_nmf_update5_offsize_list(const UpdateArgs& a, hipStream_t s)
{
  if (a.Kp == 32 && a.Kc == 24) { launch5_list_ng<6, 9, 32>(a, a.listNG, s); return true; }
  if (a.Kp == 64)
    switch (a.Kc)
    {
    case 40: launch5_list_ng<10, 4, 64>(a, a.listNG, s); return true;
    case 48: launch5_list_ng<12, 4, 64>(a, a.listNG, s); return true;
    case 56: launch5_list_ng<14, 4, 64>(a, a.listNG, s); return true;
    default: break;
    }
  if (a.Kp == 128)
    switch (a.Kc)
    {
    case 72: launch5_list_ng<18, 2, 128>(a, a.listNG, s); return true;
    case 80: launch5_list_ng<20, 2, 128>(a, a.listNG, s); return true;
    case 88: launch5_list_ng<22, 2, 128>(a, a.listNG, s); return true;
    case 96: launch5_list_ng<24, 2, 128>(a, a.listNG, s); return true;
    case 104: launch5_list_ng<26, 2, 128>(a, a.listNG, s); return true;
    case 112: launch5_list_ng<28, 2, 128>(a, a.listNG, s); return true;
    default: break;
    }
  return false;
}
#else
int launch_nmf_update5_offsize(const UpdateArgs& a, int kc, int w, int ng, hipStream_t s);
bool launch_nmf_update5_offsize_list(const UpdateArgs& a, hipStream_t s);

bool nmf_update5_supported(int Kp) { return Kp == 16 || Kp == 32 || Kp == 64 || Kp == 128; }
// compute rank of the off-size forms for a (true) rank K on arrays of rank Kp (the smallest form that holds K): 24 on arrays
// of rank 32; 40 / 48 / 56 on 64; 72 .. 112 in steps of 8 on 128; else Kp
int nmf_update5_compute_rank(int K, int Kp)
{
  if (Kp == 32) return K <= 24 ? 24 : 32;
  if (Kp == 64) return K <= 40 ? 40 : (K <= 48 ? 48 : (K <= 56 ? 56 : 64));
  if (Kp == 128) return K <= 72 ? 72 : (K <= 112 ? ((K + 7) / 8) * 8 : 128); // (30 of 32 products measured no faster than 32)
  return Kp;
}
static int k5_wps()
{
  // two wavefronts per SIMD measured no faster than one (profiles/r01/update_kernel_notes.md)
  static const int wps = [] { const char* e = fluhip::ab_getenv("FLUHIP_K5_WPS"); return e ? std::atoi(e) : 1; }();
  return wps;
}
int nmf_update5_max_groups(int Kp) { return Kp <= 32 ? 9 : (Kp <= 64 ? 4 : 2); }
int nmf_update5_strips(int C, int Kp, int B)
{
  const int G = (C + 15) / 16;
  const int w = nmf_update5_waves_per_buffer(C, Kp, B);
  const bool two = k5_wps() == 2 && (Kp == 16 || Kp == 32) && 2 * w <= G && (G + 2 * w - 1) / (2 * w) <= 4;
  return two ? 2 * w : w;
}

// strips per buffer for WPS wavefronts per SIMD: WPS x the one-wave plan, as long as every strip
// keeps at least one group
int launch_nmf_update5(const UpdateArgs& a, hipStream_t s)
{
  if (a.dryRun && (a.list || a.Kp > 64)) return 0; // (nothing but the plain un-split forms on arrays up to rank 64 take anything over)
  if (a.list)
  {
    // work-list mode: one wavefront per SIMD, the widest strip of the list picks the instantiation; the column sums
    // always ride in the kernel (no pre-pass: its per-buffer slots are dealt differently here)
    // (off-size ranks: fewer MFMAs per product on the arrays of the padded rank, as in the uniform forms)
    if (a.Kc > 0 && a.Kc != a.Kp && launch_nmf_update5_offsize_list(a, s)) return 0;
    switch (a.Kp / 4)
    {
    case 4: launch5_list_ng<4, 9>(a, a.listNG, s); break;
    case 8: launch5_list_ng<8, 9>(a, a.listNG, s); break;
    case 16: launch5_list_ng<16, 4>(a, a.listNG, s); break;
    case 32: launch5_list_ng<32, 2>(a, a.listNG, s); break;
    default: break;
    }
    return 0;
  }
  const int G = (a.C + 15) / 16;
  const int kc = a.Kc > 0 ? a.Kc : a.Kp; // (compute rank: planning is by the MFMAs per product)
  const int w = a.stripsOverride > 0 ? std::min(a.stripsOverride, G) : nmf_update5_strips(a.C, kc, a.B);
  const bool two = a.stripsOverride > 0 ? false : w != nmf_update5_waves_per_buffer(a.C, kc, a.B);
  const int ng = (G + w - 1) / w;
  if (kc != a.Kp)
  {
    const int r = launch_nmf_update5_offsize(a, kc, w, ng, s);
    if (r >= 0) return r;
  }
  if (two)
  {
    switch (a.Kp / 4)
    {
    case 4: return launch5_ng<4, 4, 2>(a, w, ng, s);
    case 8: return launch5_ng<8, 4, 2>(a, w, ng, s);
    default: break;
    }
  }
  else
  {
    switch (a.Kp / 4)
    {
    case 4: return launch5_ng<4, 9, 1>(a, w, ng, s);
    case 8: return launch5_ng<8, 9, 1>(a, w, ng, s);
    case 16: return launch5_ng<16, 4, 1>(a, w, ng, s);
    case 32: return launch5_ng<32, 2, 1>(a, w, ng, s);
    default: break;
    }
  }
  return 0;
}

#endif // FLUHIP_K5_OFFSIZE_TU

} // namespace fluhip
