// kernels_stft2.hip -- K1 in its block form: a workgroup of NW wavefronts transforms NW CONSECUTIVE frames of one
// buffer (one frame per wavefront, registers + a private LDS exchange buffer) and writes the magnitudes in BOTH
// layouts the factor updates stream -- frame-major rows straight from the registers, and the bin-major copy
// (`magT`, frame index contiguous) through an LDS staging step, NW frames of a bin at a time -- so no separate
// pass over V is needed after the STFT (the transpose kernel read and rewrote all of V: +1.8 GB on the bench shard).
//
// Same mathematics and DFT convention as kernels_stft.hip (see there for the reference lines):
//   algorithm::STFT::process / magnitude   include/flucoma/algorithms/public/STFT.hpp:90-108, 61-66
//   algorithm::FFT::process                include/flucoma/algorithms/util/FFT.hpp:92-108
// What changed against stft_wave_kernel (kernels_stft.hip), and what was measured (profiles/r02/stft_notes.md):
//   * the points cross the LDS as separate real and imaginary planes through ONE 8-byte-per-point buffer (the
//     wavefront's LDS instructions execute in order, so the imaginary plane may overwrite the real one as soon as
//     the reads of the real plane have issued): half the LDS per wavefront, and the same buffer then stages the
//     frame's magnitudes for the bin-major copy;
//   * the last pass is laid out so that the two bins of every real-FFT pair (k, N - k) end up in the SAME lane
//     (lane l owns butterflies l and Ns - l; lane 0 owns the two self-paired ones): the split needs no third
//     exchange (transforms with one last-pass butterfly per lane fetch the partner with ds_bpermute);
//   * the split's twiddle e^{-2 pi i k / fft}, k = j + Ns r, is (a per-lane constant) x (a compile-time constant),
//     so nothing is loaded for it inside the frame loop;
//   * everything the compiler would hoist out of the frame loop and keep in registers (per-point address offsets,
//     twiddle products) is either expressed as per-lane base + compile-time offset or kept opaque: 190 registers at
//     fft 2048 (two wavefronts per SIMD), 112 at fft 1024 (four).
// With its stores disabled the kernel transforms the bench shard's 110 336 frames in 430 us (the FP64 VALU ~70 % busy);
// writing V twice (1.86 GB, half of it as 64-byte pieces of 1025 different rows) brings it to 775-815 us -- against
// 973-1030 us for the wave kernel plus the transposing copy on the same boxes.  The stores, not the transform, set
// the pace: tools/hbm_write_probe.hip reproduces the figure with a plain FMA loop and the same two store patterns.
#include "fluhip_kernels.h"

#include <type_traits>

#include <algorithm>
#include <cstdio>
#include <vector>
#include <cstdlib>

namespace fluhip {

typedef double d2 __attribute__((ext_vector_type(2)));

// register arithmetic runs on (re, im) scalar pairs, not on 128-bit vectors: the two planes cross the LDS separately,
// and a vector register tuple cannot be half dead
struct cx
{
  double re, im;
};
__device__ __forceinline__ cx operator+(cx a, cx b) { return cx{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cx operator-(cx a, cx b) { return cx{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cx tocx(d2 v) { return cx{v[0], v[1]}; }

struct StftBArgs
{
  const float* audio;
  const double* audio64;
  int64_t n, audioStride;
  int win, fft, hop, T, F, B;
  const double* window;   // [max(win, fft)] zero past win
  const double* twiddle;  // [fft/2] e^{-2 pi i m / fft}
  double* mag;            // [B][*][ldMag] or nullptr
  int64_t magStride, ldMag;
  double* magT;           // [B][*][ldMagT] or nullptr
  int64_t magTStride, ldMagT;
  double* spec;
  int64_t specStride;
  int frameOffset;
  int blocksPerBuf;
  int64_t totalBlocks;
  const int64_t* nTab;    // ragged corpora: samples of every buffer (n is then the longest), or nullptr
  int prefetch;           // block kernel: request the next frame's samples ahead of this frame's stores
};

#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// The magnitudes leave once and are not read again before the whole corpus's transform is done (the factor updates stream them
// back much later, 1.9 GB on the bench shard): the two layouts' stores are NON-TEMPORAL (round 6) -- bench shard's STFT phase
// 0.560 - 0.586 -> 0.487 - 0.499 ms per launch with both layouts, 0.475 - 0.484 -> 0.444 - 0.451 with the frame-major one alone,
// alternating on one box (profiles/r06/stft_nt.txt).  -DFLUHIP_STFT_NT=0: plain stores.
#ifndef FLUHIP_STFT_NT
#define FLUHIP_STFT_NT 1
#endif
#ifndef FLUHIP_RESYNTH_NT
#define FLUHIP_RESYNTH_NT 0
#endif
__device__ __forceinline__ void store_mag16(double* p, double __attribute__((ext_vector_type(2))) v)
{
#if FLUHIP_STFT_NT
  __builtin_nontemporal_store(v, reinterpret_cast<double __attribute__((ext_vector_type(2)))*>(p));
#else
  *reinterpret_cast<double __attribute__((ext_vector_type(2)))*>(p) = v;
#endif
}
// -DFLUHIP_SPLIT_VIA_LDS=0: the ds_bpermute partner exchange of rounds 1 - 4 (FftCore::split)
#ifndef FLUHIP_SPLIT_VIA_LDS
#define FLUHIP_SPLIT_VIA_LDS 1
#endif
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the wavefront's global stores
// (s_waitcnt vmcnt(0)), which here would serialise every frame's stores with the next frame's arithmetic
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

namespace {

__device__ __forceinline__ cx cmul2(cx a, cx b) { return cx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cx mul_mi2(cx a) { return cx{a.im, -a.re}; } // a * (-i)

// sqrt(x), x >= DBL_MIN: v_rsq_f64 seed y = (1 + e) / sqrt(x), |e| ~ 2^-23; one coupled Goldschmidt step leaves g = sqrt(x) (1 - 1.5 e^2)
// (~2e-14) and h = 1 / (2 g) to the same order; the exact residual d = x - g^2 times h then squares that again: rounding
// error only, in 8 FP64 operations behind the seed (the wave kernel of round 1 ran a second Goldschmidt step first: 12).
// FAST (the fused feature kernel, whose bands leave as float32 behind a single-precision logarithm): the Goldschmidt step
// alone, 2e-14 relative -- nine orders below the float output's resolution -- in 5 operations.
// The callers keep x >= DBL_MIN by starting the sum of squares from DBL_MIN (mag_sumsq: adding it is exact-no-op for any
// power above 1e-290 and costs nothing, where fmax(x, DBL_MIN) was an instruction per bin).
constexpr double kDblMin = 2.2250738585072014e-308;
__device__ __forceinline__ double mag_sumsq(double re, double im) { return __builtin_fma(re, re, __builtin_fma(im, im, kDblMin)); }
template <bool FAST = false>
__device__ __forceinline__ double mag_sqrt2(double x)
{
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  if constexpr (FAST) return g;
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}

__device__ __forceinline__ void bf4(cx& a0, cx& a1, cx& a2, cx& a3)
{
  const cx t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = mul_mi2(a1 - a3);
  a0 = t0 + t2; a1 = t1 + t3; a2 = t0 - t2; a3 = t1 - t3;
}

constexpr double kC1 = 0.92387953251128673848; // cos(pi/8)
constexpr double kS1 = 0.38268343236508977173; // sin(pi/8)
constexpr double kC2 = 0.70710678118654752440; // sqrt(2)/2

template <int R>
__device__ __forceinline__ void bfr(cx (&v)[R]);

template <>
__device__ __forceinline__ void bfr<8>(cx (&v)[8])
{
  cx e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  cx o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  bf4(e0, e1, e2, e3);
  bf4(o0, o1, o2, o3);
  o1 = cx{kC2 * (o1.re + o1.im), kC2 * (o1.im - o1.re)};
  o2 = mul_mi2(o2);
  o3 = cx{kC2 * (o3.im - o3.re), -kC2 * (o3.re + o3.im)};
  v[0] = e0 + o0; v[4] = e0 - o0;
  v[1] = e1 + o1; v[5] = e1 - o1;
  v[2] = e2 + o2; v[6] = e2 - o2;
  v[3] = e3 + o3; v[7] = e3 - o3;
}

template <>
__device__ __forceinline__ void bfr<16>(cx (&v)[16])
{
  cx s[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
  {
    s[a][0] = v[a]; s[a][1] = v[a + 4]; s[a][2] = v[a + 8]; s[a][3] = v[a + 12];
    bf4(s[a][0], s[a][1], s[a][2], s[a][3]);
  }
  s[1][1] = cmul2(s[1][1], cx{kC1, -kS1});
  s[1][2] = cx{kC2 * (s[1][2].re + s[1][2].im), kC2 * (s[1][2].im - s[1][2].re)};
  s[1][3] = cmul2(s[1][3], cx{kS1, -kC1});
  s[2][1] = cx{kC2 * (s[2][1].re + s[2][1].im), kC2 * (s[2][1].im - s[2][1].re)};
  s[2][2] = mul_mi2(s[2][2]);
  s[2][3] = cx{kC2 * (s[2][3].im - s[2][3].re), -kC2 * (s[2][3].re + s[2][3].im)};
  s[3][1] = cmul2(s[3][1], cx{kS1, -kC1});
  s[3][2] = cx{kC2 * (s[3][2].im - s[3][2].re), -kC2 * (s[3][2].re + s[3][2].im)};
  s[3][3] = cmul2(s[3][3], cx{-kC1, kS1});
#pragma unroll
  for (int b = 0; b < 4; b++)
  {
    bf4(s[0][b], s[1][b], s[2][b], s[3][b]);
    v[b] = s[0][b]; v[b + 4] = s[1][b]; v[b + 8] = s[2][b]; v[b + 12] = s[3][b];
  }
}

// cos / sin of 2 pi r / (2 R) for the split's constant factor e^{-2 pi i r / (2 R)}
template <int R>
__device__ __forceinline__ cx split_const(int r)
{
  // e^{-2 pi i r / (2 R)}: R = 16: thirty-second roots of unity; R = 8: every second one; R = 4: every fourth
  constexpr double c32[16] = {1, 0.98078528040323043058, 0.92387953251128673848, 0.83146961230254523567, 0.70710678118654757274, 0.55557023301960228867, 0.38268343236508983729, 0.19509032201612833135, 0, -0.19509032201612819257, -0.38268343236508972627, -0.5555702330196019556, -0.70710678118654746172, -0.83146961230254534669, -0.92387953251128673848, -0.98078528040323043058};
  constexpr double s32[16] = {0, 0.19509032201612824808, 0.38268343236508978178, 0.55557023301960217765, 0.70710678118654746172, 0.83146961230254523567, 0.92387953251128673848, 0.98078528040323043058, 1, 0.98078528040323043058, 0.92387953251128673848, 0.83146961230254545772, 0.70710678118654757274, 0.55557023301960217765, 0.3826834323650898928, 0.19509032201612860891};
  static_assert(R == 16 || R == 8 || R == 4, "last-pass radix");
  const int q = r * (16 / R);
  return cx{c32[q], -s32[q]};
}

// ---- register <-> lane transposes without the LDS (round 6, gfx950) --------------------------------------------------
// One step swaps a bit of the register index with a bit of the lane index: for the register pair (A, B) that differs in the
// register bit, A keeps its lanes with the lane bit clear and takes B's partner lanes, B the reverse.  Lane bits 5 and 4 are
// ONE instruction per pair and dword (v_permlane32_swap / v_permlane16_swap: the upper half of A against the lower half of B;
// the odd rows of A against the even rows of B), lane bit 3 two DPP moves with a bank mask (row_ror:8 = lane ^ 8 inside a
// row of 16) and the copy that keeps A alive.
#ifndef FLUHIP_FFT_PERMLANE
#define FLUHIP_FFT_PERMLANE 0
#endif
template <int LANEBIT>
__device__ __forceinline__ void xpose_dword(int& a, int& b)
{
  static_assert(LANEBIT == 5 || LANEBIT == 4 || LANEBIT == 3, "lane bit of the transpose step");
  if constexpr (LANEBIT == 5)
  {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
  }
  else if constexpr (LANEBIT == 4)
  {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
  }
  else
  {
    const int na = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xc, false);   // lanes 8 .. 15 of a row <- b[lane - 8]
    const int nb = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false);   // lanes 0 .. 7           <- a[lane + 8]
    a = na; b = nb;
  }
}
template <int LANEBIT>
__device__ __forceinline__ void xpose_f64(double& a, double& b)
{
  const long long la = __double_as_longlong(a), lb = __double_as_longlong(b);
  int a0 = (int) (la & 0xffffffff), a1 = (int) (la >> 32), b0 = (int) (lb & 0xffffffff), b1 = (int) (lb >> 32);
  xpose_dword<LANEBIT>(a0, b0);
  xpose_dword<LANEBIT>(a1, b1);
  a = __longlong_as_double(((long long) a1 << 32) | (unsigned) a0);
  b = __longlong_as_double(((long long) b1 << 32) | (unsigned) b0);
}
// eight values per lane: register bits (2, 1, 0) <-> lane bits (5, 4, 3)
[[maybe_unused]] __device__ __forceinline__ void xpose8_hi(double (&v)[8])
{
#pragma unroll
  for (int r = 0; r < 4; r++) xpose_f64<5>(v[r], v[r + 4]);
#pragma unroll
  for (int r = 0; r < 8; r++) if (!(r & 2)) xpose_f64<4>(v[r], v[r + 2]);
#pragma unroll
  for (int r = 0; r < 8; r += 2) xpose_f64<3>(v[r], v[r + 1]);
}

} // namespace

// One frame's transform from the windowed points to the staged magnitudes, shared by the kernel forms below: pass 1,
// exchange, pass 2, exchange, pass 3, real-FFT split, |X|.  `pts` holds the windowed points m = lane + 64 bb + r N/R1
// (x[2m], x[2m+1]); on return the wavefront's buffer xb holds |X[k]| at xb[k], k = 0 .. N (plain index), and the
// complex bins have been written to specRow when that is not null.
// One ds_read_b64 that stays one: the compiler pairs neighbouring loads off one base into ds_read2(st64)_b64, which the LDS
// serves as two passes of 16-lane groups at 128 B per clock where two ds_read_b64 run at 256 (MI355X_MICROARCH.md, LDS
// table) -- and whose lane groups are not the ones the swizzled layouts below are laid out for.
__device__ __forceinline__ double lds_read1(const double* p)
{
  // (volatile keeps the load single; the explicit LDS address space keeps it a ds_read: address-space inference leaves
  //  volatile accesses through generic pointers as flat loads)
  return *(const volatile __attribute__((address_space(3))) double*) p;
}

template <int R1, int R2, int R3>
struct FftCore
{
  static constexpr int N = R1 * R2 * R3, PPL = N / 64;
  static constexpr int NB1 = N / (64 * R1), NB2 = N / (64 * R2), NB3 = N / (64 * R3);
  static_assert(NB1 >= 1 && NB2 >= 1 && (NB3 == 1 || NB3 == 2), "tiling");
  static constexpr int NS2 = R1, NS3 = R1 * R2;
  static constexpr bool LOCAL = NB3 == 2;        // both bins of a split pair in one lane
  // XL (round 5, the one-butterfly-per-lane transform of fft 1024): the real-FFT split fetches its partner bins through the
  // staging buffer instead of with ds_bpermute -- a plane of N + 64 slots, see split()
  static constexpr bool XL = FLUHIP_SPLIT_VIA_LDS != 0 && NB3 == 1;
  static constexpr int BUFD = N + N / 16 + (XL ? 34 : 2);    // doubles per wavefront (the + 2 staggers the buffers over the banks)
  // (the XL split writes slot N from every lane -- slots up to N + 63 -- so the padding must cover 64 slots; today's one
  //  XL form, N = 512, leaves 2 to spare, a smaller one would overrun into the next wavefront's buffer without this)
  static_assert(!XL || BUFD >= N + 64, "XL split: the staging buffer must hold slots N .. N + 63");
  static constexpr int T2 = (R2 - 1) * NS2, T3 = (R3 - 1) * NS3;
  // SWZ (round 4, the 8 x 8 x 8 transform of fft 1024): the exchange buffer is addressed through bit swizzles instead of the
  // one-in-sixteen padding.  The padding serves the first exchange's stores (ds_write_b64: groups of 16 lanes over 32 dword
  // banks) and costs every read behind both exchanges a third cycle (ds_read_b64: groups of 32 lanes over 64 banks -- 32
  // consecutive points span 33 slots) and the second exchange's stores a 2-way conflict: 19 % of the feature kernel's LDS
  // cycles (profiles/r04/c5_experiments.md).  Point i of the first exchange sits at slot s with
  //   s0 = i0 ^ i5, s1 = i1 ^ i6, s2 = i3, s3 = i4, s4 = i2, s5.. = i5..      (stores vary i3..i6, loads vary i0..i4)
  // and point i of the second at i ^ (i6 << 3) (stores vary i0..i2 and i6, loads i0..i4): every lane group of either
  // instruction meets every bank once.  Per lane that is four store bases + two load bases for the first exchange and two +
  // two for the second, all loop invariant; the offsets stay compile-time immediates.
  static constexpr bool SWZ = R1 == 8 && R2 == 8 && R3 == 8;
  // PLX (round 6): the first exchange of the 8 x 8 x 8 transform does not touch the LDS.  Pass 1 leaves point
  // i = 64 a + 8 b + r in register r of lane (a, b) (a = lane >> 3, b = lane & 7); pass 2 wants register a to hold it.  Swapping
  // the register index with the UPPER lane bits (xpose8_hi: 32 permlane swaps + 48 DPP moves per frame against 16
  // ds_write_b64 + 16 ds_read_b64, 128 cycles of the CU's one LDS pipe) puts it into lane (r, b): that lane runs butterfly
  // j = 8 b + r = 8 (lane & 7) + (lane >> 3) of pass 2 -- twiddle row k = j mod 8 = lane >> 3 -- and its outputs go to
  // i' = 64 (lane & 7) + (lane >> 3) + 8 r of the second exchange, whose slots are i' + 2 (i' >> 6): both the stores (16
  // consecutive lanes: 2 b + k over 16 bank pairs) and the loads (lane + 66 r) meet every bank once, one base each.
  // The data movement is exact: the results are the LDS form's bit for bit.
  static constexpr bool PLX = SWZ && FLUHIP_FFT_PERMLANE != 0;
  static_assert(!PLX || BUFD >= 7 * 66 + 64, "PLX: slots of the second exchange");
  static_assert(R1 == 16 || R1 == 8, "pass-1 radix");
  static_assert(NS2 % 8 == 0 && NS3 % 16 == 0 && (N / R2) % 16 == 0, "index arithmetic of the exchanges");
  int lane, jA, jB;
  cx wA, wB;
  double* xb;
  // LDS positions as (per-lane base) + (compile-time offset): slot i of the exchange buffer sits at i + (i >> 4)
  double* w1p;
  const double* r2p;
  double* w2p;
  const double *r3pA, *r3pB;
  const d2 *tw2p, *tw3pA, *tw3pB;
  // SWZ: store bases of the first exchange by r & 3, load bases by r & 1; second exchange: both by r & 1
  double* w1q[4];
  const double* r2q[2];
  double* w2q[2];
  const double* r3q[2];

  // the pass tables [R2-1][NS2] and [R3-1][NS3] from the natural table e^{-2 pi i m / fft} (m < fft/2)
  static __device__ __forceinline__ void fill_tables(d2* tw2, const d2* twg, int tid, int nthreads)
  {
    for (int i = tid; i < T2 + T3; i += nthreads)
    {
      int m;
      if (i < T2) { const int r = i / NS2 + 1, k = i % NS2; m = k * r * (2 * N / (NS2 * R2)); }
      else { const int ii = i - T2; const int r = ii / NS3 + 1, k = ii % NS3; m = k * r * (2 * N / (NS3 * R3)); }
      d2 w = twg[m >= N ? m - N : m];
      if (m >= N) w = d2{-w[0], -w[1]};
      tw2[i] = w;
    }
  }

  __device__ __forceinline__ void init(double* xb_, const d2* tw2, const d2* twg, int lane_)
  {
    const d2* tw3 = tw2 + T2;
    lane = lane_;
    xb = xb_;
    // last-pass butterflies of this lane and the split's per-lane twiddle factors e^{-2 pi i j / fft}
    jA = lane;
    jB = LOCAL ? (lane == 0 ? NS3 / 2 : NS3 - lane) : 0;
    wA = tocx(twg[jA]);
    wB = LOCAL ? tocx(twg[jB]) : cx{0.0, 0.0};
    wA = cx{0.5 * wA.re, 0.5 * wA.im};   // the split's 1/2 rides on its twiddle
    wB = cx{0.5 * wB.re, 0.5 * wB.im};
    //   pass-1 outputs  (lane + 64 bb) R1 + r    pass-2 inputs   lane + 64 bb + r N/R2
    //   pass-2 outputs  (j - k) R2 + k + r NS2   pass-3 inputs   j + r NS3
    w1p = xb + (R1 == 16 ? 17 * lane : 8 * lane + (lane >> 1));
    r2p = xb + lane + (lane >> 4);
    const int k2 = PLX ? (lane >> 3) : (lane & (NS2 - 1));
    const int hi2 = lane - k2;                                  // multiple of NS2
    w2p = xb + hi2 * R2 + ((hi2 * R2) >> 4) + k2;
    r3pA = xb + jA + (jA >> 4);
    r3pB = xb + jB + (jB >> 4);
    tw2p = tw2 + k2;
    tw3pA = tw3 + jA;
    tw3pB = tw3 + jB;
    if constexpr (SWZ)
    {
      const int l = lane;
      const int l0 = l & 1, l1 = (l >> 1) & 1, l2 = (l >> 2) & 1, l3 = (l >> 3) & 1, l4 = (l >> 4) & 1, l5 = (l >> 5) & 1;
      // first exchange, stores: i = 8 l + r -> (q ^ c) + 16 (r >> 2) + L, q = r & 3, c = (l >> 2) & 3
      const int c = (l >> 2) & 3, L = 4 * l0 + 8 * l1 + 32 * l2 + 64 * l3 + 128 * l4 + 256 * l5;
      for (int q = 0; q < 4; q++) w1q[q] = xb + L + (q ^ c);
      // first exchange, loads: i = l + 64 r -> (l0 ^ l5) + 2 (l1 ^ r0) + 4 l3 + 8 l4 + 16 l2 + 32 l5 + 64 r
      const int A = (l0 ^ l5) + 4 * l3 + 8 * l4 + 16 * l2 + 32 * l5;
      r2q[0] = xb + A + 2 * l1;
      r2q[1] = xb + A + 2 * (l1 ^ 1);
      // second exchange, stores: i = 64 a + k + 8 r (a = l >> 3, k = l & 7) -> i ^ (l3 << 3): r0 flips where l3 is set
      const int E = 64 * (l >> 3) + (l & 7);
      w2q[0] = xb + E + 8 * l3;
      w2q[1] = xb + E + 8 * (l3 ^ 1) - 8;                 // (+ 8 r with r odd)
      // second exchange, loads: i = l + 64 r -> i ^ ((r & 1) << 3)
      r3q[0] = xb + l;
      r3q[1] = xb + (l ^ 8);
      if constexpr (PLX)
      {
        w2q[0] = xb + 66 * (l & 7) + (l >> 3);
        r3q[0] = xb + l;
      }
    }
  }

  // the same lane positions in another staging buffer `delta` doubles further on (a wavefront that transforms several
  // frames per round keeps one buffer per frame)
  __device__ __forceinline__ void shift(int delta)
  {
    xb += delta; w1p += delta; r2p += delta; w2p += delta; r3pA += delta; r3pB += delta;
    if constexpr (SWZ)
    {
      for (int q = 0; q < 4; q++) w1q[q] += delta;
      for (int q = 0; q < 2; q++) { r2q[q] += delta; w2q[q] += delta; r3q[q] += delta; }
    }
  }

  // TAB: the split's twiddles e^{-2 pi i k / fft} / 2 come from a table in the LDS (splitTab[k], k < N) instead of being
  // formed per frame as products of the lane's factor with compile-time constants (28 FP64 instructions at R3 = 8)
  const d2* splitTab = nullptr;
  template <bool SPEC, bool TAB = false, bool FASTMAG = false>
  __device__ __forceinline__ void run(cx (&pts)[PPL], d2* specRow)
  {
    cx p3[PPL];
    passes(pts, p3);
    split<SPEC, TAB, FASTMAG>(p3, specRow);
  }

  // the three passes of the complex transform: points m = lane + 64 bb + r N/R1 in, bins j + r NS3 out (j = lane, and for
  // the two-butterfly forms jB = NS3 - lane as well: registers bb R3 + r)
  __device__ __forceinline__ void passes(cx (&pts)[PPL], cx (&p3)[PPL])
  {
  SCHED_FENCE();
  // ---- pass 1 (Ns = 1): butterfly j = lane + 64 bb, outputs at j R1 + r ------------------------------
#pragma unroll
  for (int bb = 0; bb < NB1; bb++)
  {
    cx v[R1];
#pragma unroll
    for (int r = 0; r < R1; r++) v[r] = pts[bb * R1 + r];
    bfr<R1>(v);
#pragma unroll
    for (int r = 0; r < R1; r++) pts[bb * R1 + r] = v[r];
  }
  SCHED_FENCE();
  cx p2[PPL];
  // exchange 1 -> distribution of pass 2 (butterfly j = lane + 64 bb reads j + r N/R2): real plane, then imaginary
  if constexpr (PLX)
  {
    double re[8], im[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { re[r] = pts[r].re; im[r] = pts[r].im; }
    xpose8_hi(re);
    xpose8_hi(im);
#pragma unroll
    for (int r = 0; r < 8; r++) p2[r] = cx{re[r], im[r]};
  }
  else if constexpr (SWZ)
  {
#pragma unroll
    for (int r = 0; r < R1; r++) w1q[r & 3][16 * (r >> 2)] = pts[r].re;
#pragma unroll
    for (int r = 0; r < R2; r++) p2[r].re = lds_read1(r2q[r & 1] + 64 * r);
#pragma unroll
    for (int r = 0; r < R1; r++) w1q[r & 3][16 * (r >> 2)] = pts[r].im;
#pragma unroll
    for (int r = 0; r < R2; r++) p2[r].im = lds_read1(r2q[r & 1] + 64 * r);
  }
  else
  {
#pragma unroll
  for (int bb = 0; bb < NB1; bb++)
#pragma unroll
    for (int r = 0; r < R1; r++) w1p[(64 * R1 + 4 * R1) * bb + ((R1 == 8 && r >= 8) ? 0 : r)] = pts[bb * R1 + r].re;
#pragma unroll
  for (int bb = 0; bb < NB2; bb++)
#pragma unroll
    for (int r = 0; r < R2; r++) p2[bb * R2 + r].re = r2p[68 * bb + (N / R2 + N / R2 / 16) * r];
#pragma unroll
  for (int bb = 0; bb < NB1; bb++)
#pragma unroll
    for (int r = 0; r < R1; r++) w1p[(64 * R1 + 4 * R1) * bb + ((R1 == 8 && r >= 8) ? 0 : r)] = pts[bb * R1 + r].im;
#pragma unroll
  for (int bb = 0; bb < NB2; bb++)
#pragma unroll
    for (int r = 0; r < R2; r++) p2[bb * R2 + r].im = r2p[68 * bb + (N / R2 + N / R2 / 16) * r];
  }
  SCHED_FENCE();
  // ---- pass 2 (Ns = R1): twiddle e^{-2 pi i k r / (R1 R2)}, k = j mod R1 = lane mod R1 -----------------
#pragma unroll
  for (int bb = 0; bb < NB2; bb++)
  {
    cx v[R2];
    v[0] = p2[bb * R2];
#pragma unroll
    for (int r = 1; r < R2; r++) v[r] = cmul2(p2[bb * R2 + r], tocx(tw2p[(r - 1) * NS2]));
    bfr<R2>(v);
#pragma unroll
    for (int r = 0; r < R2; r++) p2[bb * R2 + r] = v[r];
  }
  SCHED_FENCE();
  // exchange 2 -> distribution of pass 3: outputs of butterfly j at (j - k) R2 + k + r NS2
  auto in3 = [&](int bb, int r) -> const double* {
    return (LOCAL && bb == 1 ? r3pB : r3pA) + (NS3 + NS3 / 16) * r;
  };
  if constexpr (PLX)
  {
#pragma unroll
    for (int r = 0; r < R2; r++) w2q[0][8 * r] = p2[r].re;
#pragma unroll
    for (int r = 0; r < R3; r++) p3[r].re = lds_read1(r3q[0] + 66 * r);
#pragma unroll
    for (int r = 0; r < R2; r++) w2q[0][8 * r] = p2[r].im;
#pragma unroll
    for (int r = 0; r < R3; r++) p3[r].im = lds_read1(r3q[0] + 66 * r);
  }
  else if constexpr (SWZ)
  {
#pragma unroll
    for (int r = 0; r < R2; r++) w2q[r & 1][8 * r] = p2[r].re;
#pragma unroll
    for (int r = 0; r < R3; r++) p3[r].re = lds_read1(r3q[r & 1] + 64 * r);
#pragma unroll
    for (int r = 0; r < R2; r++) w2q[r & 1][8 * r] = p2[r].im;
#pragma unroll
    for (int r = 0; r < R3; r++) p3[r].im = lds_read1(r3q[r & 1] + 64 * r);
  }
  else
  {
#pragma unroll
  for (int bb = 0; bb < NB2; bb++)
#pragma unroll
    for (int r = 0; r < R2; r++)
      w2p[(64 * R2 + 4 * R2) * bb + NS2 * r + ((NS2 * r) >> 4)] = p2[bb * R2 + r].re;
#pragma unroll
  for (int bb = 0; bb < NB3; bb++)
#pragma unroll
    for (int r = 0; r < R3; r++) p3[bb * R3 + r].re = *in3(bb, r);
#pragma unroll
  for (int bb = 0; bb < NB2; bb++)
#pragma unroll
    for (int r = 0; r < R2; r++)
      w2p[(64 * R2 + 4 * R2) * bb + NS2 * r + ((NS2 * r) >> 4)] = p2[bb * R2 + r].im;
#pragma unroll
  for (int bb = 0; bb < NB3; bb++)
#pragma unroll
    for (int r = 0; r < R3; r++) p3[bb * R3 + r].im = *in3(bb, r);
  }
  SCHED_FENCE();
  // ---- pass 3 (Ns = R1 R2 = N / R3, so k = j): the outputs are the bins j + r NS3 and stay in registers -----
#pragma unroll
  for (int bb = 0; bb < NB3; bb++)
  {
    cx v[R3];
    v[0] = p3[bb * R3];
#pragma unroll
    for (int r = 1; r < R3; r++) v[r] = cmul2(p3[bb * R3 + r], tocx((LOCAL && bb == 1 ? tw3pB : tw3pA)[(r - 1) * NS3]));
    bfr<R3>(v);
#pragma unroll
    for (int r = 0; r < R3; r++) p3[bb * R3 + r] = v[r];
  }
  SCHED_FENCE();
  }

  template <bool SPEC, bool TAB = false, bool FASTMAG = false>
  __device__ __forceinline__ void split(cx (&p3)[PPL], d2* specRow)
  {
  // ---- real-FFT split (util/FFT.hpp:99-106) + magnitude (alg/STFT.hpp:61-66) ------------------------------
  // bin k = j + r NS3 pairs with N - k = (NS3 - j) + (R3 - 1 - r) NS3  (j > 0), or r -> R3 - r for j = 0
  // the products of the per-lane factors with the compile-time constants are loop invariant, and hoisted they
  // would sit in 56 registers for the whole frame loop: keep the factors opaque so they are formed where used
  asm volatile("" : "+v"(wA.re), "+v"(wA.im), "+v"(wB.re), "+v"(wB.im));
  int lq = lane;
  asm volatile("" : "+v"(lq));                    // (opaque: `lane == 0` is compared where it is used, not kept as a spilled exec mask)
  const bool l0 = lq == 0;
  const int srcAddr = ((64 - lane) & 63) * 4;
  (void) srcAddr;
  auto partner = [&](int bb, int r) -> cx {
    if constexpr (LOCAL)
    {
      // lane 0: butterfly 0 pairs within itself (r <-> R3 - r, bin 0 with itself), butterfly NS3/2 within itself
      if (bb == 0)
      {
        const cx x0 = p3[(R3 - r) & (R3 - 1)], x1 = p3[R3 + R3 - 1 - r];
        return cx{l0 ? x0.re : x1.re, l0 ? x0.im : x1.im};
      }
      const cx x1 = p3[R3 + R3 - 1 - r], y1 = p3[R3 - 1 - r];
      return cx{l0 ? x1.re : y1.re, l0 ? x1.im : y1.im};
    }
    else
    {
      // one butterfly per lane: the partner bin lives in lane 64 - l (register R3 - 1 - r); lane 0 pairs within itself
      const cx z = p3[R3 - 1 - r];
      const int lo0 = __builtin_amdgcn_ds_bpermute(srcAddr, (int) (__double_as_longlong(z.re) & 0xffffffff));
      const int hi0 = __builtin_amdgcn_ds_bpermute(srcAddr, (int) (__double_as_longlong(z.re) >> 32));
      const int lo1 = __builtin_amdgcn_ds_bpermute(srcAddr, (int) (__double_as_longlong(z.im) & 0xffffffff));
      const int hi1 = __builtin_amdgcn_ds_bpermute(srcAddr, (int) (__double_as_longlong(z.im) >> 32));
      const double re = __longlong_as_double(((long long) hi0 << 32) | (unsigned) lo0);
      const double im = __longlong_as_double(((long long) hi1 << 32) | (unsigned) lo1);
      const cx own = p3[(R3 - r) & (R3 - 1)];
      return cx{l0 ? own.re : re, l0 ? own.im : im};
    }
  };
  // XL: the partner of bin k = j + r NS3 is bin N - k = (NS3 - j) + (R3 - 1 - r) NS3 in lane NS3 - j -- and for j = 0 the same
  // expression names bin N - r NS3, which is lane 0's own register R3 - r once bin N is read as bin 0 (the transform is
  // periodic).  So the lanes lay one plane of Z (real parts, then imaginary parts) into the staging buffer, slot = bin, with
  // bin 0 once more at slot N, and every lane reads slot (NS3 - lane) + (R3 - 1 - r) NS3: no lane is special, where the
  // ds_bpermute form needed a select per dword for lane 0 (32 v_cndmask + 32 ds_bpermute per frame; now 18 + 16 LDS accesses of
  // 8 bytes and no VALU).  One plane at a time, and the magnitudes go to the same slots: the real parts of all R3 partners
  // are fetched before the imaginary plane overwrites them, and the magnitudes wait in registers for the last partner read
  // (lane 0's partners lie in its own column of slots).
  [[maybe_unused]] double pre[R3], mg[R3];
  [[maybe_unused]] const double* rpx = nullptr;
  if constexpr (XL)
  {
    double* wpx = xb + lq;
    rpx = xb + (NS3 - lq);
#pragma unroll
    for (int r = 0; r < R3; r++) wpx[NS3 * r] = p3[r].re;
    wpx[N] = p3[0].re;
#pragma unroll
    for (int r = 0; r < R3; r++) pre[r] = lds_read1(rpx + NS3 * (R3 - 1 - r));
#pragma unroll
    for (int r = 0; r < R3; r++) wpx[NS3 * r] = p3[r].im;
    wpx[N] = p3[0].im;
  }
#pragma unroll
  for (int bb = 0; bb < NB3; bb++)
  {
    const int j = LOCAL ? (bb == 0 ? jA : jB) : lane;
    const cx wj = (LOCAL && bb == 1) ? wB : wA;
#pragma unroll
    for (int r = 0; r < R3; r++)
    {
      const int i = bb * R3 + r;
      // X[k] = (Z[k] + conj Z[N-k]) / 2 - i / 2 e^{-2 pi i k / fft} (Z[k] - conj Z[N-k]); the halves ride on the twiddle
      // (wj = e^{...} / 2) and on one multiplier of the sum
      cx Bc;
      if constexpr (XL) Bc = cx{pre[r], lds_read1(rpx + NS3 * (R3 - 1 - r))};
      else Bc = partner(bb, r);
      const cx A = p3[i];
      const double er = A.re + Bc.re, ei = A.im - Bc.im;
      const double dr = A.re - Bc.re, di = A.im + Bc.im;
      cx w;
      if constexpr (TAB) w = tocx(splitTab[j + r * NS3]);
      else w = r == 0 ? wj : cmul2(wj, split_const<R3>(r));
      const double xr = __builtin_fma(0.5, er, __builtin_fma(w.re, di, w.im * dr));
      double xi = __builtin_fma(0.5, ei, __builtin_fma(w.im, di, -(w.re * dr)));
      const int k = j + r * NS3;
      if (k == 0) xi = 0.0;                     // DC is purely real (util/FFT.hpp:99-101)
      const double m = mag_sqrt2<FASTMAG>(mag_sumsq(xr, xi));
      if constexpr (XL) mg[r] = m;
      else xb[k] = m;                           // staged (the exchange buffer is idle now)
      if constexpr (SPEC) specRow[k] = d2{xr, xi};
      if (r & 1) SCHED_FENCE();                 // two bins' chains in flight, not sixteen
    }
  }
  if constexpr (XL)
  {
#pragma unroll
    for (int r = 0; r < R3; r++) xb[lq + NS3 * r] = mg[r];
  }
  if (lane == 0)
  {
    const cx z = p3[0];                         // Z[0]: Nyquist = Re - Im, purely real
    const double xr = z.re - z.im;
    xb[N] = fabs(xr);
    if constexpr (SPEC) specRow[N] = d2{xr, 0.0};
  }
  }

  // ---- NF frames at once (round 4, the fused feature kernel) ---------------------------------------------------------
  // The same passes with NF independent frames between the same fences: frame f stages through the buffer f * BUFD doubles
  // behind xb.  One frame's chain -- pass, exchange (write, read, wait), pass, ... -- leaves the wavefront waiting on the
  // LDS most of its life (a single wavefront at top priority needs 11.8 k cycles per frame for 2 760 cycles of VALU issue,
  // and four of them per SIMD fill the VALU to a half); two frames in one instruction stream give the scheduler something
  // to put into every one of those waits.
  template <int NF>
  __device__ __forceinline__ void passesN(cx (&pts)[NF][PPL], cx (&p3)[NF][PPL])
  {
  SCHED_FENCE();
#pragma unroll
  for (int f = 0; f < NF; f++)
#pragma unroll
    for (int bb = 0; bb < NB1; bb++)
    {
      cx v[R1];
#pragma unroll
      for (int r = 0; r < R1; r++) v[r] = pts[f][bb * R1 + r];
      bfr<R1>(v);
#pragma unroll
      for (int r = 0; r < R1; r++) pts[f][bb * R1 + r] = v[r];
    }
  SCHED_FENCE();
  cx p2[NF][PPL];
#pragma unroll
  for (int f = 0; f < NF; f++)
  {
#pragma unroll
    for (int bb = 0; bb < NB1; bb++)
#pragma unroll
      for (int r = 0; r < R1; r++) w1p[f * BUFD + (64 * R1 + 4 * R1) * bb + ((R1 == 8 && r >= 8) ? 0 : r)] = pts[f][bb * R1 + r].re;
#pragma unroll
    for (int bb = 0; bb < NB2; bb++)
#pragma unroll
      for (int r = 0; r < R2; r++) p2[f][bb * R2 + r].re = r2p[f * BUFD + 68 * bb + (N / R2 + N / R2 / 16) * r];
#pragma unroll
    for (int bb = 0; bb < NB1; bb++)
#pragma unroll
      for (int r = 0; r < R1; r++) w1p[f * BUFD + (64 * R1 + 4 * R1) * bb + ((R1 == 8 && r >= 8) ? 0 : r)] = pts[f][bb * R1 + r].im;
#pragma unroll
    for (int bb = 0; bb < NB2; bb++)
#pragma unroll
      for (int r = 0; r < R2; r++) p2[f][bb * R2 + r].im = r2p[f * BUFD + 68 * bb + (N / R2 + N / R2 / 16) * r];
  }
  SCHED_FENCE();
#pragma unroll
  for (int bb = 0; bb < NB2; bb++)
  {
    // (one read of a pass-2 twiddle serves every frame)
    cx tw[R2];
#pragma unroll
    for (int r = 1; r < R2; r++) tw[r] = tocx(tw2p[(r - 1) * NS2]);
#pragma unroll
    for (int f = 0; f < NF; f++)
    {
      cx v[R2];
      v[0] = p2[f][bb * R2];
#pragma unroll
      for (int r = 1; r < R2; r++) v[r] = cmul2(p2[f][bb * R2 + r], tw[r]);
      bfr<R2>(v);
#pragma unroll
      for (int r = 0; r < R2; r++) p2[f][bb * R2 + r] = v[r];
    }
  }
  SCHED_FENCE();
  auto in3 = [&](int f, int bb, int r) -> const double* {
    return (LOCAL && bb == 1 ? r3pB : r3pA) + f * BUFD + (NS3 + NS3 / 16) * r;
  };
#pragma unroll
  for (int f = 0; f < NF; f++)
  {
#pragma unroll
    for (int bb = 0; bb < NB2; bb++)
#pragma unroll
      for (int r = 0; r < R2; r++)
        w2p[f * BUFD + (64 * R2 + 4 * R2) * bb + NS2 * r + ((NS2 * r) >> 4)] = p2[f][bb * R2 + r].re;
#pragma unroll
    for (int bb = 0; bb < NB3; bb++)
#pragma unroll
      for (int r = 0; r < R3; r++) p3[f][bb * R3 + r].re = *in3(f, bb, r);
#pragma unroll
    for (int bb = 0; bb < NB2; bb++)
#pragma unroll
      for (int r = 0; r < R2; r++)
        w2p[f * BUFD + (64 * R2 + 4 * R2) * bb + NS2 * r + ((NS2 * r) >> 4)] = p2[f][bb * R2 + r].im;
#pragma unroll
    for (int bb = 0; bb < NB3; bb++)
#pragma unroll
      for (int r = 0; r < R3; r++) p3[f][bb * R3 + r].im = *in3(f, bb, r);
  }
  SCHED_FENCE();
#pragma unroll
  for (int bb = 0; bb < NB3; bb++)
  {
    cx tw[R3];
#pragma unroll
    for (int r = 1; r < R3; r++) tw[r] = tocx((LOCAL && bb == 1 ? tw3pB : tw3pA)[(r - 1) * NS3]);
#pragma unroll
    for (int f = 0; f < NF; f++)
    {
      cx v[R3];
      v[0] = p3[f][bb * R3];
#pragma unroll
      for (int r = 1; r < R3; r++) v[r] = cmul2(p3[f][bb * R3 + r], tw[r]);
      bfr<R3>(v);
#pragma unroll
      for (int r = 0; r < R3; r++) p3[f][bb * R3 + r] = v[r];
    }
  }
  SCHED_FENCE();
  }

  // the split + magnitude of frame f of NF (staged at xb + f * BUFD); one frame's worth of the single-frame split()
  template <int NF>
  __device__ __forceinline__ void splitN(cx (&p3)[NF][PPL])
  {
    double* const xb0 = xb;
#pragma unroll
    for (int f = 0; f < NF; f++)
    {
      xb = xb0 + f * BUFD;
      split<false>(p3[f], nullptr);
    }
    xb = xb0;
  }
};

// gather + window of frame t of buffer b: point m = x[2m] + i x[2m+1], m = lane + 64 bb + r N/R1 (alg/STFT.hpp:94-105;
// clients/nrt/NMFClient.hpp:240 float -> double); zero outside [0, n)
template <int R1, int N>
__device__ __forceinline__ void gather_points(const StftBArgs& a, int b, int t, int lane, const d2* wsrc, cx (&pts)[N / 64],
                                              int64_t nSamples)
{
  constexpr int NB1 = N / (64 * R1);
  const int64_t s0 = (int64_t) t * a.hop - a.win / 2 + a.frameOffset;
  // ---- gather + window: point m = x[2m] + i x[2m+1], m = lane + 64 bb + r N/R1 --------------------
  // sample positions are 32-bit offsets from the frame's first sample (a wave-uniform 64-bit base); only frames
  // that stick out of the buffer (or an odd base) take the clamped path
  const int lo = s0 < 0 ? (int) (-s0 < 2 * N ? -s0 : 2 * N) : 0;                 // first valid offset
  const int64_t room = nSamples - s0;
  const int hi = room < 2 * N ? (int) (room > 0 ? room : 0) : 2 * N;             // one past the last valid offset
  if (a.audio)
  {
    const float* fp = a.audio + (int64_t) b * a.audioStride + s0;
    const bool fast = lo == 0 && hi == 2 * N && ((reinterpret_cast<uintptr_t>(fp) & 7) == 0);
    if (fast)
    {
      const float2* lp = reinterpret_cast<const float2*>(fp) + lane;
#pragma unroll
      for (int bb = 0; bb < NB1; bb++)
#pragma unroll
        for (int r = 0; r < R1; r++)
        {
          const int mo = 64 * bb + r * (N / R1);
          const float2 x = lp[mo];
          const d2 w = wsrc[lane + mo];
          pts[bb * R1 + r] = cx{(double) x.x * w[0], (double) x.y * w[1]};
        }
    }
    else
    {
      int ln = lane;                       // (opaque: the per-point offsets of this rare path are not worth registers
      asm volatile("" : "+v"(ln));         //  across the whole frame loop, where the compiler would hoist them)
#pragma unroll
      for (int bb = 0; bb < NB1; bb++)
#pragma unroll
        for (int r = 0; r < R1; r++)
        {
          const int mo = 64 * bb + r * (N / R1);
          const int i0 = 2 * (ln + mo), i1 = i0 + 1;
          const bool ok0 = i0 >= lo && i0 < hi, ok1 = i1 >= lo && i1 < hi;
          const float v0 = fp[ok0 ? i0 : lo], v1 = fp[ok1 ? i1 : lo];   // lo is a valid offset whenever hi > lo
          const d2 w = wsrc[lane + mo];
          pts[bb * R1 + r] = cx{(ok0 ? (double) v0 : 0.0) * w[0], (ok1 ? (double) v1 : 0.0) * w[1]};
        }
    }
  }
  else
  {
    const double* dp = a.audio64 + (int64_t) b * a.audioStride + s0;
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int bb = 0; bb < NB1; bb++)
#pragma unroll
      for (int r = 0; r < R1; r++)
      {
        const int mo = 64 * bb + r * (N / R1);
        const int i0 = 2 * (ln + mo), i1 = i0 + 1;
        const bool ok0 = i0 >= lo && i0 < hi, ok1 = i1 >= lo && i1 < hi;
        const double v0 = dp[ok0 ? i0 : lo], v1 = dp[ok1 ? i1 : lo];
        const d2 w = wsrc[lane + mo];
        pts[bb * R1 + r] = cx{(ok0 ? v0 : 0.0) * w[0], (ok1 ? v1 : 0.0) * w[1]};
      }
  }
}

// The interior frames' samples as they lie in memory (the fast path of gather_points, split in two): requested early
// by the frame loop -- ahead of the stores of the frame before, see stft_block_kernel -- and windowed when their turn comes.
// false: an edge frame, an odd base or double-precision input (gather_points takes those).
template <int R1, int N>
__device__ __forceinline__ bool request_samples(const StftBArgs& a, int b, int t, int64_t nSamples, int lane, float2 (&raw)[N / 64])
{
  constexpr int NB1 = N / (64 * R1);
  if (!a.audio) return false;
  const int64_t s0 = (int64_t) t * a.hop - a.win / 2 + a.frameOffset;
  const float* fp = a.audio + (int64_t) b * a.audioStride + s0;
  if (s0 < 0 || s0 + 2 * N > nSamples || (reinterpret_cast<uintptr_t>(fp) & 7) != 0) return false;
  const float2* lp = reinterpret_cast<const float2*>(fp) + lane;
#pragma unroll
  for (int bb = 0; bb < NB1; bb++)
#pragma unroll
    for (int r = 0; r < R1; r++) raw[bb * R1 + r] = lp[64 * bb + r * (N / R1)];
  return true;
}
template <int R1, int N>
__device__ __forceinline__ void window_samples(const float2 (&raw)[N / 64], int lane, const d2* wsrc, cx (&pts)[N / 64])
{
  constexpr int NB1 = N / (64 * R1);
#pragma unroll
  for (int bb = 0; bb < NB1; bb++)
#pragma unroll
    for (int r = 0; r < R1; r++)
    {
      const float2 x = raw[bb * R1 + r];
      const d2 w = wsrc[lane + 64 * bb + r * (N / R1)];
      pts[bb * R1 + r] = cx{(double) x.x * w[0], (double) x.y * w[1]};
    }
}

// FPW frames per wavefront and round: a block stages NW * FPW consecutive frames before the bin-major flush, so a bin's
// piece of the transposed copy is NW * FPW * 8 bytes -- a full 128-byte line at fft 2048 with 8 wavefronts x 2 frames
// (64-byte pieces measured 3.3 TB/s against 4.4 - 5.9 for full lines, tools/hbm_write_probe.hip).
// DB (round 5): TWO sets of staging buffers, used in turn -- the flush of round r reads set r & 1 while the transforms of
// round r + 1 fill the other, so the barrier behind the flush goes (a wavefront reaches the next round's barrier only after its
// own flush reads, and nobody writes a set before everybody has passed that barrier).
template <int R1, int R2, int R3, int NW, int WINLDS, bool SPEC, int FPW = 1, bool DB = false>
__global__ __launch_bounds__(64 * NW) void stft_block_kernel(StftBArgs a)
{
  constexpr int N = R1 * R2 * R3;  // complex points per frame = fft / 2
  constexpr int PPL = N / 64;      // points per lane
  constexpr int NB1 = N / (64 * R1), NB2 = N / (64 * R2), NB3 = N / (64 * R3);
  static_assert(NB1 >= 1 && NB2 >= 1 && (NB3 == 1 || NB3 == 2), "tiling");
  constexpr int NS2 = R1, NS3 = R1 * R2;
  constexpr int BUFD = FftCore<R1, R2, R3>::BUFD; // doubles per wavefront
  constexpr int T2 = (R2 - 1) * NS2, T3 = (R3 - 1) * NS3;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  d2* tw2 = reinterpret_cast<d2*>(lds);       // [R2-1][NS2]
  d2* tw3 = tw2 + T2;                         // [R3-1][NS3]
  d2* wl = tw3 + T3;                          // [N] window pairs when WINLDS
  double* xall = reinterpret_cast<double*>(wl + (WINLDS ? N : 0));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* xb = xall + wave * FPW * BUFD;
  constexpr int FPB = NW * FPW; // frames per block and round

  const d2* twg = reinterpret_cast<const d2*>(a.twiddle);
  FftCore<R1, R2, R3>::fill_tables(tw2, twg, threadIdx.x, 64 * NW);
  const d2* wsrc = reinterpret_cast<const d2*>(a.window);
  if (WINLDS)
  {
    for (int m = threadIdx.x; m < N; m += 64 * NW) wl[m] = wsrc[m];
    wsrc = wl;
  }
  __syncthreads();

  FftCore<R1, R2, R3> core;
  core.init(xb, tw2, twg, lane);

  // blocks are dealt so that workgroups on one XCD (blockIdx & 7) take neighbouring blocks of a buffer: their
  // segments of a bin-major row fall into the same L2
  const int64_t chunk = (a.totalBlocks + 7) / 8;
  // block L of the dealing order -> (buffer, first frame); false past the end
  auto decode = [&](int64_t L, int& b, int& t0) -> bool {
    const int64_t slot = L >> 3;
    const int64_t blk = (L & 7) * chunk + slot;
    if (slot >= chunk || blk >= a.totalBlocks) return false;
    b = (int) (blk / a.blocksPerBuf);
    t0 = (int) (blk % a.blocksPerBuf) * FPB;
    return true;
  };
  // Prefetch (one frame per round, float audio): the samples of the NEXT round's frame are requested before this round's
  // stores are issued.  vmcnt counts loads and stores in one in-order counter, so samples requested AFTER the stores (as in
  // round 2's loop) can only be waited for by waiting for every store of the round before -- the store drain sat exposed
  // in front of every frame's arithmetic (the ISA had s_waitcnt vmcnt(7) behind eight fresh loads: "all stores done").
  // Requested BEFORE them, the samples are older than the stores and the wait leaves the stores in flight: 793 / 838 us
  // -> 697 / 691 us for the bench workload's STFT phase (profiles/r03/stft_prefetch.txt; FLUHIP_STFT_PREFETCH=0 is the
  // round-2 order).  Requesting them before the transform instead (35 more registers) measured the same, and two
  // 4-wavefront blocks per CU instead of one of 8 measured slower (754 / 763 us): neither is kept.
  constexpr bool PREFETCH = FPW == 1 && PPL == 16 && NW <= 12; // fft 2048 at two or three wavefronts per SIMD only: fft 1024 (4 wavefronts per SIMD) measures the same either way, fft 4096 has no registers to spare, and 16 wavefronts of fft 2048 need the 32 registers
  float2 raw[PPL];
  int rawB = -1, rawT = -1;
  auto prefetch = [&](int64_t Ln) {
    rawB = -1;
    if constexpr (PREFETCH)
    {
      int bn, t0n;
      if (!a.audio || !a.prefetch || (Ln >> 3) >= chunk || !decode(Ln, bn, t0n)) return;
      const int tn = t0n + wave;
      const int64_t nS = a.nTab ? a.nTab[bn] : a.n;
      const int Tn = a.nTab ? (int) ((nS + a.hop) / a.hop) : a.T;
      if (tn >= Tn) return;
      if (!request_samples<R1, N>(a, bn, tn, nS, lane, raw)) return;
      rawB = bn;
      rawT = tn;
    }
  };
  for (int64_t L = blockIdx.x;; L += gridDim.x)
  {
    int b, t0;
    if ((L >> 3) >= chunk) break;
    if (!decode(L, b, t0)) continue;
    [[maybe_unused]] double* const xcur = DB ? core.xb - wave * FPW * BUFD : xall;   // this round's set of staging buffers
    [[maybe_unused]] double* const xbw = DB ? core.xb : xb;
    // ragged corpora: the buffer's own length decides its frame count (alg/STFT.hpp:98-99); frames past it are padding
    const int64_t nSamples = a.nTab ? a.nTab[b] : a.n;
    const int Tb = a.nTab ? (int) ((nSamples + a.hop) / a.hop) : a.T;
#pragma unroll 1
    for (int fj = 0; fj < FPW; fj++)
    {
      const int t = t0 + wave * FPW + fj;
      if (t < Tb)
      {
        cx pts[PPL];
        if (PREFETCH && rawB == b && rawT == t)
          window_samples<R1, N>(raw, lane, wsrc, pts);
        else
          gather_points<R1, N>(a, b, t, lane, wsrc, pts, nSamples);
        SCHED_FENCE();
        core.template run<SPEC>(pts, SPEC ? reinterpret_cast<d2*>(a.spec + (int64_t) b * a.specStride + (int64_t) t * a.F * 2) : nullptr);
      }
      else if (a.magT)
      {
        // frames past the end of the buffer: zeros into the padding columns of the bin-major copy
#pragma unroll
        for (int i = 0; i < PPL; i++) core.xb[lane + 64 * i] = 0.0;
        if (lane == 0) core.xb[N] = 0.0;
      }
      if (FPW > 1) core.shift(fj + 1 < FPW ? BUFD : -(FPW - 1) * BUFD);
    }
    prefetch(L + gridDim.x);                          // ahead of this round's stores

    if (a.magT) LDS_BARRIER();                        // every wavefront of the block has staged its frames
    if (a.mag)
    {
#pragma unroll 1
      for (int fj = 0; fj < FPW; fj++)
      {
        const int t = t0 + wave * FPW + fj;
        if (t >= Tb) break;
        // frame-major row from the wavefront's own staging buffer, 16 bytes per lane
        double* magRow = a.mag + (int64_t) b * a.magStride + (int64_t) t * a.ldMag;
        const double* xs = xbw + fj * BUFD;
#pragma unroll
        for (int q = 0; q < N / 128; q++)
        {
          const int k = 2 * (lane + 64 * q);
          store_mag16(magRow + k, *reinterpret_cast<const d2*>(xs + k));
        }
        if (lane == 0) magRow[N] = xs[N];
      }
    }
    if (a.magT)
    {
      // ---- bin-major copy: the block's frames of a bin leave as one piece ------------------------------------------
      constexpr int HP = FPB / 2;                     // 16-byte pieces (two frames) per bin
      double* outT = a.magT + (int64_t) b * a.magTStride;
      constexpr int ITEMS = (N + 1) * HP, NTHR = 64 * NW, TRIPS = (ITEMS + NTHR - 1) / NTHR; // (a.F == N + 1)
      constexpr int UNR = PREFETCH ? TRIPS : 1;       // (a compile-time count of stores is what lets the wait ahead of the next transform skip them)
#pragma unroll UNR
      for (int k = 0; k < TRIPS; k++)
      {
        const int i = (int) threadIdx.x + k * NTHR;
        const int f = i / HP, p = i - f * HP;
        const int tc = t0 + 2 * p;
        if (i < ITEMS && tc < a.ldMagT)
        {
          const double v0 = xcur[(2 * p) * BUFD + f], v1 = xcur[(2 * p + 1) * BUFD + f];
          store_mag16(outT + (int64_t) f * a.ldMagT + tc, d2{v0, v1});
        }
      }
      if constexpr (!DB) LDS_BARRIER();
    }
    if constexpr (DB)
    {
      // the other set of staging buffers for the next round
      const int set = (int) ((core.xb - xall) >= NW * FPW * BUFD);
      core.shift(set ? -NW * FPW * BUFD : NW * FPW * BUFD);
    }
  }
}

// ---- wavefront scan of doubles with DPP moves (no LDS crossbar): lanes a DPP source does not reach read 0 ----------------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double x)
{
  const long long b = __double_as_longlong(x);
  if constexpr (ROWMASK == 0xf)
  {
    // every row enabled: bound_ctrl supplies the zeros of the lanes a shift does not reach, and the move has no `old`
    // operand to initialise (update_dpp(0, ...) cost a v_mov_b32 per half: 20 per frame of the feature kernel)
    const int lo = __builtin_amdgcn_mov_dpp((int) (b & 0xffffffff), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int) (b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
  }
  const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xffffffff), CTRL, ROWMASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, ROWMASK, 0xf, false);
  return __longlong_as_double(((long long) hi << 32) | (unsigned) lo);
}
// inclusive prefix sum over the 64 lanes, lane order
__device__ __forceinline__ double wave_scan(double x)
{
  x += dpp_f64<0x111, 0xf>(x);   // row_shr:1
  x += dpp_f64<0x112, 0xf>(x);   // row_shr:2
  x += dpp_f64<0x114, 0xf>(x);   // row_shr:4
  x += dpp_f64<0x118, 0xf>(x);   // row_shr:8
  x += dpp_f64<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
  x += dpp_f64<0x143, 0xc>(x);   // row_bcast:31 into rows 2 and 3
  return x;
}
// value of the lane below (0 for lane 0)
__device__ __forceinline__ double wave_shr1(double x) { return dpp_f64<0x138, 0xf>(x); } // wave_shr:1

// ---------------------------------------------------------------------------------------------------------------
// Fused feature form (BASELINE config 5: STFT -> MelBands -> MFCC): the frame's magnitudes never leave the
// wavefront's LDS buffer -- per frame the kernel reads hop 4 bytes of samples and writes nOut 4 bytes of features.
//   algorithm::MelBands::processFrame   include/flucoma/algorithms/public/MelBands.hpp:79-97
//   algorithm::DCT::processFrame        include/flucoma/algorithms/public/DCT.hpp:65-75
//   client glue                         include/flucoma/clients/rt/MFCCClient.hpp:122-130, rt/MelBandsClient.hpp:104-113
// A mel filter bank is a row of overlapping triangles: every bin lies on the rising edge of at most one band and on the
// falling edge of at most one (the one below).  With up[f] / dn[f] the two weights of bin f, band b is
//   sum_{f in interval b} up[f] m[f] + sum_{f in interval b+1} dn[f] m[f],   interval s = bins between centres s and s+1,
// two segment sums of two running sums over the bins: lane l forms the running sums of its CH consecutive bins, a
// wavefront scan gives each lane its offset, the lanes holding the last bin before an interval boundary publish the
// running sums there, and lane b takes band b as two differences.  (The host checks that the filter bank at hand has
// this shape -- api_features.hip features_common -- and falls back to the two-kernel path otherwise.)
// No workgroup barrier anywhere in the frame loop: wavefronts are independent.
// ---------------------------------------------------------------------------------------------------------------
struct FeatFusedArgs
{
  const double* up;    // [64 CH] rising-edge weight of bin f (0 where none)
  const double* dn;    // [64 CH] falling-edge weight of bin f
  const short* slot;   // [64 CH] s if bin f is the last bin before interval s starts (publishes the running sums), else -1
  const double* dct;   // [nDct][nBands] or nullptr
  int nBands, nDct, startCoeff, nOut;
  int magNorm, usePower, logOutput;
  float* out;          // [B][nOut][T]
  long long* dbg = nullptr; // FLUHIP_FEAT_CLOCK (A/B build): shader-cycle and 100 MHz stamps of workgroup 0's first wavefront
};

// DYN (round 4): the wavefronts of a workgroup take their frames from a counter in the LDS instead of a fixed share.  The issue
// arbiter of a SIMD serves its oldest wavefront first: with equal shares the first wavefront of every SIMD was done after
// 2.0 ms of a 3.86 ms launch (config 5, per-wavefront stamps) and the youngest ran the last third of the launch alone, at a
// quarter of the VALU's rate -- the counters show the VALU 87 % busy while four wavefronts are alive and 64 % over the launch.
// Handing out frames as wavefronts come free makes them finish together.
// SMALL (round 4): the layout for TWO workgroups per CU (20 wavefronts, five per SIMD, where one workgroup of 1024 threads stops
// at four): the window pairs come through the L1 instead of the LDS and the per-wavefront scratch of the band stage lies in
// the wavefront's staging buffer, which is idle by then -- 67 KB per workgroup of ten wavefronts at fft 1024.
// FASTMAG (round 5): the magnitudes behind one Goldschmidt step (mag_sqrt2<true>: 2e-14 relative, 24 VALU instructions per frame
// fewer at fft 1024) -- they feed band sums that leave as float32.  FLUHIP_FEAT_FASTMAG=0 (A/B build): the full square root.
template <int R1, int R2, int R3, int NW, bool DYN = true, bool SMALL = false, bool FASTMAG = true>
__global__ __launch_bounds__(64 * NW, SMALL ? 5 : 1) void stft_feat_kernel(StftBArgs a, FeatFusedArgs fa)
{
  using Core = FftCore<R1, R2, R3>;
  constexpr int N = Core::N, PPL = Core::PPL, BUFD = Core::BUFD;
  constexpr int CH = (N + 1 + 63) / 64;           // consecutive bins per lane in the band sums
  // per-wavefront scratch: boundary sums (rising, falling) -- 66 boundary slots, then one DUMP slot per lane: bins that end no
  // interval publish there, so that the publishing needs no compare, no exec mask and no select (36 v_cndmask + 9 v_cmp per
  // frame as the compiler had if-converted it) -- and the band values
  constexpr int WB = 66 + 64;
  constexpr int WS = SMALL ? 0 : 2 * WB + 64;
  static_assert(!SMALL || BUFD >= 2 * WB + 64, "the band stage's scratch inside the staging buffer");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  d2* tw2 = reinterpret_cast<d2*>(lds);
  d2* wl = tw2 + Core::T2 + Core::T3;             // [N] window pairs
  constexpr bool TAB = R1 == 8;                   // (fft 2048 has no LDS to spare for the table)
  d2* tws = wl + (SMALL ? 0 : N);                 // [N] the split's twiddles, halved
  double* xall = reinterpret_cast<double*>(tws + (TAB ? N : 0));
  double* scr = xall + NW * BUFD;                 // [NW][WS]
  double* upl = scr + NW * WS;                    // [64 CH]
  double* dnl = upl + 64 * CH;                    // [64 CH]
  double* dctl = dnl + 64 * CH;                   // [nDct * nBands]
  const int dq = (fa.nBands + 3) >> 2;            // bands per quarter row (four lanes share a coefficient)
  const int dld = 4 * dq + 1;                     // DCT rows: four quarters, zero-padded past nBands, one double apart in bank phase
                                                  // (rows of 40 doubles would put rows j and j + 4 on one bank)
  short* slotl = reinterpret_cast<short*>(dctl + (fa.dct ? fa.nDct * dld : 0));   // [64 CH]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* xb = xall + wave * BUFD;
  double* bu = SMALL ? xb : scr + wave * WS;
  double* bd = bu + WB;
  double* bands = bd + WB;

  const d2* twg = reinterpret_cast<const d2*>(a.twiddle);
  Core::fill_tables(tw2, twg, threadIdx.x, 64 * NW);
  if constexpr (!SMALL)
    for (int m = threadIdx.x; m < N; m += 64 * NW) wl[m] = reinterpret_cast<const d2*>(a.window)[m];
  if constexpr (TAB)
    for (int m = threadIdx.x; m < N; m += 64 * NW) { const d2 w = twg[m]; tws[m] = d2{0.5 * w[0], 0.5 * w[1]}; }
  // (per-lane tables lie [i][lane] in the LDS: a lane's CH consecutive bins are CH rows apart, a row is read without bank conflicts)
  for (int i = threadIdx.x; i < 64 * CH; i += 64 * NW)
  {
    const int q = (i % CH) * 64 + i / CH;
    upl[q] = fa.up[i]; dnl[q] = fa.dn[i];
    const short sl = fa.slot[i];
    slotl[q] = sl >= 0 ? sl : (short) (66 + i / CH);           // (the lane's dump slot)
  }
  if (fa.dct)
    for (int i = threadIdx.x; i < fa.nDct * dld; i += 64 * NW)
    {
      const int row = i / dld, col = i % dld;
      dctl[i] = col < fa.nBands ? fa.dct[row * fa.nBands + col] : 0.0;
    }
  if constexpr (!SMALL)
    for (int i = lane; i < WS; i += 64) bu[i] = 0.0;  // boundaries nobody publishes (before the first bin) stay 0
  __shared__ unsigned nextFrame;                      // DYN: frames of this workgroup handed out so far
  if (threadIdx.x == 0) nextFrame = 0;
  __syncthreads();

#ifdef FLUHIP_AB_SWITCHES
  if (fa.dbg && threadIdx.x == 0)
  {
    if (blockIdx.x == 0) fa.dbg[0] = (long long) __builtin_readcyclecounter();
    fa.dbg[4 + 2 * blockIdx.x] = (long long) wall_clock64();
  }
#endif
  Core core;
  core.init(xb, tw2, twg, lane);
  core.splitTab = tws;
  // FLUHIP_FEAT_PRIO (A/B): the wavefronts of a SIMD (wave, wave + 4, ...) at different issue priorities.  They run the same
  // instruction stream from the same start; under fair round-robin issue they stay in phase -- all in their LDS exchanges,
  // then all in their butterflies -- and the two pipes take turns instead of overlapping (PMC: VALU busy 46 % + LDS active
  // 42 % of the launch).  With strict priorities the first one runs ahead and the others fill its waits.
#ifdef FLUHIP_AB_SWITCHES
  if (a.prefetch & 2) { const int pr = (wave ^ (wave >> 2)) & 3; if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(1); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
#endif
  const double scale1 = 1.0 / ((double) a.win / 4.0);                          // alg/MelBands.hpp:49
  const double scale2 = 1.0 / (2.0 * (double) (2 * N) / (double) a.win);       // :52
  // Lane constants of the band stage kept in registers across the frame loop (round 5: the kernel sits at 101 of its 128
  // registers, and the LDS pipe co-limits it): the boundary slot of each of the lane's CH bins (was a ds_read_i16 and a shift
  // per bin and frame) and, for the MFCC default's ten products per lane, the lane's quarter row of the DCT table.
  constexpr bool HOIST = !SMALL;
  [[maybe_unused]] int slotReg[CH];
  if constexpr (HOIST)
  {
#pragma unroll
    for (int i = 0; i < CH; i++) slotReg[i] = slotl[i * 64 + lane];
  }
  // (the A/B build's kernel carries the experiment switches of rounds 3 - 4 and has no room for the ten coefficients: with them it
  //  spills 16 registers to scratch -- 3.4 ms on config 5)
  const bool dct10 = HOIST && !kAbSwitches && fa.dct && 4 * fa.nOut <= 64 && dq == 10;
  [[maybe_unused]] double drowReg[10];
  {
    const int j = lane >> 2, part = lane & 3;
    const bool live = j < fa.nOut && fa.startCoeff + j < fa.nDct;
    const double* drow = dctl + (live ? fa.startCoeff + j : 0) * dld + part * dq;
#pragma unroll
    for (int i = 0; i < 10; i++) drowReg[i] = dct10 ? drow[i] : 0.0;
  }

  const int64_t chunk = (a.totalBlocks + 7) / 8;
  // (Round 4: what these measurements were really showing is in the DYN comment at the head of the kernel.)
  // Round 3 measurements of this loop (config 5, 1.417 M frames, one box; profiles/r03/c5_breakdown.txt), parts switched off
  // one at a time: everything 3.93 ms; without the sample gather 3.38; without the transform 1.92; without the band sums /
  // DCT / stores 2.95 (the stores alone: 0.05); the loop with nothing in it 0.38.  Alone, the gather takes 0.82 ms (the HBM
  // time of 2.9 GB of samples), the transform 2.0, the band part 1.0 -- the parts add up: 93 KB of LDS traffic per frame
  // (transform exchanges 32, twiddles 14, window 8, staging 8 + 8, weights and DCT rows 23) keep the LDS pipe busy through
  // all three.  Requesting the next frame's samples ahead of the feature stores (as stft_block_kernel does) changes
  // nothing here (3.94 / 3.94 ms without, 3.92 / 3.92 with): four wavefronts per SIMD cover that wait.
  // Block index -> (buffer, frame block) WITHOUT a division per frame: the workgroup's blocks are blk0, blk0 + step, ...
  // (step = gridDim.x / 8 inside its XCD's chunk), so (buffer, frame block) advance by (step / blocksPerBuf, step %
  // blocksPerBuf) with a carry -- scalar adds.  The 64-bit blk / blocksPerBuf, blk % blocksPerBuf of rounds 2 - 3 ran on the
  // VALU (quarter-rate v_mul_hi / v_mul_lo sequences) once per frame and wavefront: most of the 0.38 ms the EMPTY frame
  // loop measured at config 5 (profiles/r03/c5_breakdown.txt).
  const int64_t blk0 = (int64_t) (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  const int step = (int) (gridDim.x >> 3);                      // (launch_feat_t: the grid is a multiple of 8)
  const int stepB = step / a.blocksPerBuf, stepF = step % a.blocksPerBuf;
  int bCur = __builtin_amdgcn_readfirstlane((int) (blk0 / a.blocksPerBuf));
  int fCur = __builtin_amdgcn_readfirstlane((int) (blk0 % a.blocksPerBuf));
  int64_t slot8 = blockIdx.x >> 3, blk = blk0;
  // DYN: unit u of the workgroup = wavefront slot u % NW of its block u / NW (block i of the workgroup is block blk0 + i step of
  // the launch); the division by blocksPerBuf is a multiplication by its reciprocal on the scalar unit
  const unsigned bpb = (unsigned) a.blocksPerBuf, bpbInv = (unsigned) (((unsigned long long) 1 << 32) / bpb);
  for (;; slot8 += step, blk += step, bCur += stepB, fCur += stepF)
  {
    int b, t;
    if constexpr (DYN)
    {
      unsigned u = 0;
      if (lane == 0) u = atomicAdd(&nextFrame, 1u);
      u = (unsigned) __builtin_amdgcn_readfirstlane((int) u);
      const unsigned i = u / NW, slot = u % NW;
      const int64_t bl = blk0 + (int64_t) i * step;
      if ((int64_t) (blockIdx.x >> 3) + (int64_t) i * step >= chunk || bl >= a.totalBlocks) break; // (blocks ascend: nothing behind this one either)
      const unsigned x = (unsigned) bl;               // (totalBlocks < 2^31: launch_feat_t)
      unsigned q = __umulhi(x, bpbInv), r = x - q * bpb;
      while (r >= bpb) { q++; r -= bpb; }
      b = (int) q;
      t = (int) (r * NW + slot);
      if (t >= a.T) continue;
    }
    else
    {
      if (fCur >= a.blocksPerBuf) { fCur -= a.blocksPerBuf; bCur += 1; }
      if (slot8 >= chunk) break;
      if (blk >= a.totalBlocks) continue;
      b = bCur;
      t = fCur * NW + wave;
      if (t >= a.T) continue;
    }
    {
      cx pts[PPL];
#ifdef FLUHIP_AB_SWITCHES
      // FLUHIP_FEAT_SAMEFRAME=1 (bit 2; a TIMING experiment, wrong output): every frame gathers frame 2 of buffer 0 -- cache hits
      // instead of the HBM's latency: the upper bound of what requesting the samples a frame ahead could buy
      gather_points<R1, N>(a, (a.prefetch & 4) ? 0 : b, (a.prefetch & 4) ? 2 : t, lane, (SMALL || (a.prefetch & 1)) ? reinterpret_cast<const d2*>(a.window) : wl, pts, a.n);
#else
      gather_points<R1, N>(a, b, t, lane, SMALL ? reinterpret_cast<const d2*>(a.window) : wl, pts, a.n);
#endif
      SCHED_FENCE();
      core.template run<false, TAB, FASTMAG>(pts, nullptr);
    }
    SCHED_FENCE();
    // ---- band sums ------------------------------------------------------------------------------------------
    int ln = lane;
    asm volatile("" : "+v"(ln));                    // (opaque: per-bin LDS positions are recomputed, not kept)
    double pu[CH], pd[CH];
    double su = 0.0, sd = 0.0, en = 0.0;
    // (the two options are launch-wide: a scalar branch around two copies of the loop -- if-converted they were two selects per
    //  bin, 36 v_cndmask + 18 multiplications per frame that the MFCC / mel-band defaults never use)
    auto band_sums = [&](auto plainTag) {
      constexpr bool PLAIN = decltype(plainTag)::value;
#pragma unroll
      for (int i = 0; i < CH; i++)
      {
        const int f = CH * ln + i;
        // (bins past N have zero weights: the clamped read costs a v_min where the select cost a compare and two v_cndmask)
        double m = xb[min(f, N)];
        if constexpr (!PLAIN)
        {
          if (fa.magNorm) { m = f <= N ? m * scale1 : 0.0; en += m; }     // :86-90
          if (fa.usePower) m = m * m;
        }
        su = __builtin_fma(lds_read1(upl + i * 64 + ln), m, su);    // (single ds_read_b64s: paired they run at half the rate)
        sd = __builtin_fma(lds_read1(dnl + i * 64 + ln), m, sd);
        pu[i] = su;
        pd[i] = sd;
      }
    };
    if (fa.magNorm | fa.usePower) band_sums(std::false_type{});
    else band_sums(std::true_type{});
    // inclusive scan of the lane totals over the wavefront (DPP: four shifts inside a row of 16, then the row totals
    // broadcast upwards), then shifted by one lane: what the lanes below contribute
    const double xu = wave_scan(su), xd = wave_scan(sd);
    const double eu = wave_shr1(xu), ed = wave_shr1(xd);
    if constexpr (SMALL)
    {
      // the boundary sums live in the staging buffer (the magnitudes have been read: LDS operations of a wavefront complete
      // in order): boundaries nobody publishes read 0
      bu[lane] = 0.0; bd[lane] = 0.0;
      if (lane < 2) { bu[64 + lane] = 0.0; bd[64 + lane] = 0.0; }
    }
#pragma unroll
    for (int i = 0; i < CH; i++)
    {
      int sl;
      if constexpr (HOIST) sl = slotReg[i];
      else sl = slotl[i * 64 + ln];
      bu[sl] = eu + pu[i];
      bd[sl] = ed + pd[i];
    }
    double v = 0.0;
    if (ln < fa.nBands) v = (bu[ln + 1] - bu[ln]) + (bd[ln + 2] - bd[ln + 1]);   // (predicates from the opaque lane: recomputed per frame, not kept as spilled exec masks)
    if (fa.magNorm)
    {
      // :93  bands * energy / max(eps, sum of the bands), energy = sum_f (m scale1) * scale2 (:86-87)
      double bs = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { bs += __shfl_xor(bs, off); en += __shfl_xor(en, off); }
      v = v * (en * scale2) / fmax(kEpsilon, bs);
    }
    // :95  20 log10(max(eps, band)).  The feature leaves as a float: the logarithm is taken in single precision (hardware
    // log2; the band energy rounds to float with 6e-8 relative error = 5e-7 dB, below the float output's own resolution)
    if (fa.logOutput) v = (double) (6.020599913279624f * __log2f((float) fmax(v, kEpsilon)));
    if (!fa.dct)
    {
      if (ln < fa.nBands) fa.out[((int64_t) b * fa.nOut + ln) * a.T + t] = (float) v;
      continue;
    }
    bands[ln] = ln < fa.nBands ? v : 0.0;                // (the DCT rows are zero there too: the quarters need no bounds)
    // ---- DCT-II rows startCoeff .. startCoeff + nOut - 1 (alg/DCT.hpp:73-75) -----------------------------------------
    if (4 * fa.nOut <= 64)
    {
      // four lanes per coefficient, a quarter of the bands each (ascending), then two exchange-adds.  Every lane walks dq
      // products -- a wave-uniform trip count, positions past nBands multiply zeros, lanes without a coefficient walk row 0
      // and store nothing -- instead of a loop whose bounds depend on the lane (exec-mask bookkeeping per product).
      const int j = ln >> 2, part = ln & 3;
      const bool live = j < fa.nOut && fa.startCoeff + j < fa.nDct;
      const double* drow = dctl + (live ? fa.startCoeff + j : 0) * dld + part * dq;
      const double* bq = bands + part * dq;
      double sacc = 0.0;
      if (dct10)
      {
        // 37 .. 40 bands (the MFCC default): the lane's quarter row of the table from registers, the bands at immediate offsets
#pragma unroll
        for (int i = 0; i < 10; i++) sacc = __builtin_fma(drowReg[i], lds_read1(bq + i), sacc);
      }
      else if (dq == 10)
      {
        // (the same unrolled: as a loop with a trip count from a register the compiler kept an address add per read -- 18
        //  v_add_u32 per eight products)
#pragma unroll
        for (int i = 0; i < 10; i++) sacc = __builtin_fma(lds_read1(drow + i), lds_read1(bq + i), sacc);
      }
      else
      {
#pragma unroll
        for (int i = 0; i < 16; i++)                            // (nBands <= 64: dq <= 16)
        {
          if (i >= dq) break;
          sacc = __builtin_fma(lds_read1(drow + i), lds_read1(bq + i), sacc);
        }
      }
      sacc += __shfl_xor(sacc, 1);
      sacc += __shfl_xor(sacc, 2);
      if (part == 0 && j < fa.nOut) fa.out[((int64_t) b * fa.nOut + j) * a.T + t] = (float) (live ? sacc : 0.0);
    }
    else
    {
      for (int j = lane; j < fa.nOut; j += 64)
      {
        double sacc = 0.0;
        if (fa.startCoeff + j < fa.nDct)
        {
          const double* drow = dctl + (fa.startCoeff + j) * dld;
          for (int band = 0; band < fa.nBands; band++) sacc = __builtin_fma(drow[band], bands[band], sacc);
        }
        fa.out[((int64_t) b * fa.nOut + j) * a.T + t] = (float) sacc;
      }
    }
  }
#ifdef FLUHIP_AB_SWITCHES
  if (fa.dbg && threadIdx.x == 0)
  {
    if (blockIdx.x == 0) fa.dbg[2] = (long long) __builtin_readcyclecounter();
    fa.dbg[5 + 2 * blockIdx.x] = (long long) wall_clock64();
  }
#endif
}

#ifdef FLUHIP_AB_SWITCHES
// The same kernel with FPW frames per wavefront in ONE instruction stream (FftCore::passesN): a block is NW * FPW
// consecutive frames, wavefront w takes frames w FPW .. w FPW + FPW - 1 of it.  Half the wavefronts per SIMD (twice the
// registers per wavefront), the same number of frames in flight -- but the frames of a wavefront are scheduled against each
// other by the compiler instead of against the luck of the round-robin.
template <int R1, int R2, int R3, int NW, int FPW>
__global__ __launch_bounds__(64 * NW) void stft_feat2_kernel(StftBArgs a, FeatFusedArgs fa)
{
  using Core = FftCore<R1, R2, R3>;
  constexpr int N = Core::N, PPL = Core::PPL, BUFD = Core::BUFD;
  constexpr int CH = (N + 1 + 63) / 64;
  constexpr int WS = 66 + 66 + 64;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  d2* tw2 = reinterpret_cast<d2*>(lds);
  d2* wl = tw2 + Core::T2 + Core::T3;
  double* xall = reinterpret_cast<double*>(wl + N);
  double* scr = xall + NW * FPW * BUFD;           // [NW][FPW][WS]
  double* upl = scr + NW * FPW * WS;
  double* dnl = upl + 64 * CH;
  double* dctl = dnl + 64 * CH;
  const int dld = fa.nBands + 1;
  short* slotl = reinterpret_cast<short*>(dctl + (fa.dct ? fa.nDct * dld : 0));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* xb = xall + wave * FPW * BUFD;
  double* bu0 = scr + wave * FPW * WS;

  const d2* twg = reinterpret_cast<const d2*>(a.twiddle);
  Core::fill_tables(tw2, twg, threadIdx.x, 64 * NW);
  for (int m = threadIdx.x; m < N; m += 64 * NW) wl[m] = reinterpret_cast<const d2*>(a.window)[m];
  for (int i = threadIdx.x; i < 64 * CH; i += 64 * NW) { const int q = (i % CH) * 64 + i / CH; upl[q] = fa.up[i]; dnl[q] = fa.dn[i]; slotl[q] = fa.slot[i]; }
  if (fa.dct)
    for (int i = threadIdx.x; i < fa.nDct * fa.nBands; i += 64 * NW) dctl[(i / fa.nBands) * dld + (i % fa.nBands)] = fa.dct[i];
  for (int i = lane; i < FPW * WS; i += 64) bu0[i] = 0.0;
  __syncthreads();

  Core core;
  core.init(xb, tw2, twg, lane);
  const double scale1 = 1.0 / ((double) a.win / 4.0);
  const double scale2 = 1.0 / (2.0 * (double) (2 * N) / (double) a.win);

  const int64_t chunk = (a.totalBlocks + 7) / 8;
  for (int64_t L = blockIdx.x;; L += gridDim.x)
  {
    const int64_t slot8 = L >> 3;
    if (slot8 >= chunk) break;
    const int64_t blk = (L & 7) * chunk + slot8;
    if (blk >= a.totalBlocks) continue;
    const int b = (int) (blk / a.blocksPerBuf);
    const int t0 = ((int) (blk % a.blocksPerBuf) * NW + wave) * FPW;
    if (t0 >= a.T) continue;
    {
      cx pts[FPW][PPL], p3[FPW][PPL];
#pragma unroll
      for (int f = 0; f < FPW; f++) gather_points<R1, N>(a, b, min(t0 + f, a.T - 1), lane, wl, pts[f], a.n); // (a frame past the last repeats it; not stored)
      SCHED_FENCE();
      core.template passesN<FPW>(pts, p3);
      core.template splitN<FPW>(p3);
    }
    SCHED_FENCE();
    // ---- band sums of the FPW frames, stage by stage ----------------------------------------------------------------
    int ln = lane;
    asm volatile("" : "+v"(ln));
    double v[FPW];
    {
      double pu[FPW][CH], pd[FPW][CH];
      double su[FPW], sd[FPW], en[FPW];
#pragma unroll
      for (int f = 0; f < FPW; f++) su[f] = sd[f] = en[f] = 0.0;
#pragma unroll
      for (int i = 0; i < CH; i++)
      {
        const int fb = CH * ln + i;
        const double wu = upl[i * 64 + ln], wd = dnl[i * 64 + ln];
#pragma unroll
        for (int f = 0; f < FPW; f++)
        {
          double m = fb <= N ? xb[f * BUFD + fb] : 0.0;
          if (fa.magNorm) { m *= scale1; en[f] += m; }
          if (fa.usePower) m = m * m;
          su[f] = __builtin_fma(wu, m, su[f]);
          sd[f] = __builtin_fma(wd, m, sd[f]);
          pu[f][i] = su[f];
          pd[f][i] = sd[f];
        }
      }
      double eu[FPW], ed[FPW];
#pragma unroll
      for (int f = 0; f < FPW; f++)
      {
        const double xu = wave_scan(su[f]), xd = wave_scan(sd[f]);
        eu[f] = wave_shr1(xu);
        ed[f] = wave_shr1(xd);
      }
#pragma unroll
      for (int i = 0; i < CH; i++)
      {
        const int sl = slotl[i * 64 + ln];
        if (sl >= 0)
        {
#pragma unroll
          for (int f = 0; f < FPW; f++) { bu0[f * WS + sl] = eu[f] + pu[f][i]; bu0[f * WS + 66 + sl] = ed[f] + pd[f][i]; }
        }
      }
#pragma unroll
      for (int f = 0; f < FPW; f++)
      {
        const double* bu = bu0 + f * WS;
        const double* bd = bu + 66;
        double vv = 0.0;
        if (lane < fa.nBands) vv = (bu[lane + 1] - bu[lane]) + (bd[lane + 2] - bd[lane + 1]);
        if (fa.magNorm)
        {
          double bs = vv, e = en[f];
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) { bs += __shfl_xor(bs, off); e += __shfl_xor(e, off); }
          vv = vv * (e * scale2) / fmax(kEpsilon, bs);
        }
        if (fa.logOutput) vv = (double) (6.020599913279624f * __log2f((float) fmax(vv, kEpsilon)));
        v[f] = vv;
      }
    }
    if (!fa.dct)
    {
#pragma unroll
      for (int f = 0; f < FPW; f++)
        if (lane < fa.nBands && t0 + f < a.T) fa.out[((int64_t) b * fa.nOut + lane) * a.T + t0 + f] = (float) v[f];
      continue;
    }
#pragma unroll
    for (int f = 0; f < FPW; f++) bu0[f * WS + 132 + lane] = v[f];
    if (4 * fa.nOut <= 64)
    {
      const int j = lane >> 2, part = lane & 3;
      const int q = (fa.nBands + 3) >> 2;
      double sacc[FPW];
#pragma unroll
      for (int f = 0; f < FPW; f++) sacc[f] = 0.0;
      if (j < fa.nOut && fa.startCoeff + j < fa.nDct)
      {
        const double* drow = dctl + (fa.startCoeff + j) * dld;
        const int b1 = min((part + 1) * q, fa.nBands);
        for (int band = part * q; band < b1; band++)
        {
          const double dv = drow[band];
#pragma unroll
          for (int f = 0; f < FPW; f++) sacc[f] = __builtin_fma(dv, bu0[f * WS + 132 + band], sacc[f]);
        }
      }
#pragma unroll
      for (int f = 0; f < FPW; f++)
      {
        sacc[f] += __shfl_xor(sacc[f], 1);
        sacc[f] += __shfl_xor(sacc[f], 2);
        if (part == 0 && j < fa.nOut && t0 + f < a.T) fa.out[((int64_t) b * fa.nOut + j) * a.T + t0 + f] = (float) sacc[f];
      }
    }
    else
    {
      for (int j = lane; j < fa.nOut; j += 64)
#pragma unroll
        for (int f = 0; f < FPW; f++)
        {
          double sacc = 0.0;
          if (fa.startCoeff + j < fa.nDct)
          {
            const double* drow = dctl + (fa.startCoeff + j) * dld;
            for (int band = 0; band < fa.nBands; band++) sacc = __builtin_fma(drow[band], bu0[f * WS + 132 + band], sacc);
          }
          if (t0 + f < a.T) fa.out[((int64_t) b * fa.nOut + j) * a.T + t0 + f] = (float) sacc;
        }
    }
  }
}

template <int R1, int R2, int R3, int NW, int FPW>
static bool launch_feat2_t(const StftBArgs& k0, const FeatFusedArgs& fa, hipStream_t s)
{
  using Core = FftCore<R1, R2, R3>;
  constexpr int N = Core::N, CH = (N + 1 + 63) / 64, WS = 66 + 66 + 64;
  const size_t shmem = ((size_t) Core::T2 + Core::T3 + N) * 16 + ((size_t) NW * FPW * (Core::BUFD + WS) + 2 * 64 * CH) * 8 +
                       (fa.dct ? (size_t) fa.nDct * (fa.nBands + 1) * 8 : 0) + (size_t) 64 * CH * 2 + 16;
  if (shmem > 160 * 1024) return false;
  StftBArgs k = k0;
  k.blocksPerBuf = (k.T + NW * FPW - 1) / (NW * FPW);
  k.totalBlocks = (int64_t) k.B * k.blocksPerBuf;
  if (k.totalBlocks < 1) return true;
  auto kern = stft_feat2_kernel<R1, R2, R3, NW, FPW>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) shmem) != hipSuccess)
    return false;
  const int64_t chunk = (k.totalBlocks + 7) / 8;
  int64_t grid = 8 * chunk;
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(kern, dim3((unsigned) grid), dim3(64 * NW), shmem, s, k, fa);
  return true;
}

#endif // FLUHIP_AB_SWITCHES

template <int R1, int R2, int R3, int NW, bool SMALL = false>
static bool launch_feat_t(const StftBArgs& k0, const FeatFusedArgs& fa, hipStream_t s)
{
  using Core = FftCore<R1, R2, R3>;
  constexpr int N = Core::N, CH = (N + 1 + 63) / 64, WS = SMALL ? 0 : 2 * (66 + 64) + 64;
  const size_t shmem = ((size_t) Core::T2 + Core::T3 + (SMALL ? 0 : N) + (R1 == 8 ? N : 0)) * 16 + ((size_t) NW * (Core::BUFD + WS) + 2 * 64 * CH) * 8 +
                       (fa.dct ? (size_t) fa.nDct * (4 * ((fa.nBands + 3) / 4) + 1) * 8 : 0) + (size_t) 64 * CH * 2 + 16;
  if (shmem > (SMALL ? 80 : 160) * 1024) return false;
  StftBArgs k = k0;
  k.blocksPerBuf = (k.T + NW - 1) / NW;
  k.totalBlocks = (int64_t) k.B * k.blocksPerBuf;
  if (k.totalBlocks < 1) return true;
  if (k.totalBlocks >= ((int64_t) 1 << 31)) return false; // (the kernel's block arithmetic is 32-bit; the two-kernel path takes it)
  auto kern = stft_feat_kernel<R1, R2, R3, NW, true, SMALL>;
#ifdef FLUHIP_AB_SWITCHES // FLUHIP_FEAT_DYN=0: a fixed share of the frames per wavefront (rounds 2 - 3)
  static const int dynOff = [] { const char* e = fluhip::ab_getenv("FLUHIP_FEAT_DYN"); return e && std::atoi(e) == 0 ? 1 : 0; }();
  if (dynOff && !SMALL) kern = stft_feat_kernel<R1, R2, R3, NW, false>;
  static const int slowMag = [] { const char* e = fluhip::ab_getenv("FLUHIP_FEAT_FASTMAG"); return e && std::atoi(e) == 0 ? 1 : 0; }();
  if (slowMag && !dynOff && !SMALL) kern = stft_feat_kernel<R1, R2, R3, NW, true, false, false>;
#endif
  request_dynamic_lds(kern, (size_t) (shmem));
  const int64_t chunk = (k.totalBlocks + 7) / 8;
  int64_t grid = 8 * chunk;
  if (grid > (SMALL ? 512 : 256)) grid = SMALL ? 512 : 256;
  if constexpr (kAbSwitches)
  {
    static const int clk = [] { const char* e = fluhip::ab_getenv("FLUHIP_FEAT_CLOCK"); return e ? std::atoi(e) : 0; }();
    if (clk)
    {
      FeatFusedArgs f2 = fa;
      long long* d = nullptr;
      if (hipMalloc(&d, (4 + 2 * 256) * 8) == hipSuccess)
      {
        f2.dbg = d;
        hipLaunchKernelGGL(kern, dim3((unsigned) grid), dim3(64 * NW), shmem, s, k, f2);
        std::vector<long long> h(4 + 2 * 256, 0);
        (void) hipStreamSynchronize(s);
        (void) hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        (void) hipFree(d);
        long long t0 = h[4], t1 = h[5];
        for (int g = 0; g < (int) grid; g++) { t0 = std::min(t0, h[4 + 2 * g]); t1 = std::max(t1, h[5 + 2 * g]); }
        const double us = (h[5] - h[4]) / 100.0;
        std::fprintf(stderr, "stft_feat_kernel: workgroup 0 alive %.1f us, %lld shader cycles -> %.0f MHz; all %d workgroups: %.1f us first start to last end\n",
                     us, h[2] - h[0], (h[2] - h[0]) / us, (int) grid, (t1 - t0) / 100.0);
        std::fprintf(stderr, "  start offsets / lifetimes (us) of workgroups 0, 1, 8, 9, 64, 128, 255:");
        for (int g : {0, 1, 8, 9, 64, 128, 255})
          if (g < (int) grid) std::fprintf(stderr, "  %.0f/%.0f", (h[4 + 2 * g] - t0) / 100.0, (h[5 + 2 * g] - h[4 + 2 * g]) / 100.0);
        int late = 0;
        for (int g = 0; g < (int) grid; g++) late += (h[4 + 2 * g] - t0) > 10000 ? 1 : 0;   // started more than 100 us after the first
        std::fprintf(stderr, "\n  workgroups that started more than 100 us after the first: %d\n", late);
        return true;
      }
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned) grid), dim3(64 * NW), shmem, s, k, fa);
  return true;
}

// STFT -> mel bands [-> DCT] in one kernel; false when the shape has no fused form
bool launch_stft_features(const StftArgs& a, const FeatArgs& f, const double* up, const double* dn, const short* slot,
                          hipStream_t s)
{
  if ((a.win % 2) != 0 || a.win > a.fft || f.nBands > 64 || f.nBands < 1) return false;
  StftBArgs k;
  k.audio = a.audio; k.audio64 = a.audio64; k.n = a.n; k.audioStride = a.audioStride;
  k.win = a.win; k.fft = a.fft; k.hop = a.hop; k.T = a.T; k.F = a.F; k.B = a.B;
  k.window = a.window; k.twiddle = a.twiddle;
  k.mag = nullptr; k.magStride = 0; k.ldMag = 0; k.magT = nullptr; k.magTStride = 0; k.ldMagT = 0;
  k.spec = nullptr; k.specStride = 0; k.frameOffset = a.frameOffset; k.blocksPerBuf = 0; k.totalBlocks = 0;
  k.nTab = nullptr; k.prefetch = 0;
  if constexpr (kAbSwitches) // FLUHIP_FEAT_WINDOW_GLOBAL=1: the window pairs through the L1 instead of the LDS (A/B; `prefetch` is unused by this kernel)
  {
    static const int wg = [] { const char* e = fluhip::ab_getenv("FLUHIP_FEAT_WINDOW_GLOBAL"); return e ? std::atoi(e) : 0; }();
    static const int pr = [] { const char* e = fluhip::ab_getenv("FLUHIP_FEAT_PRIO"); return e ? std::atoi(e) : 0; }();
    static const int sf = [] { const char* e = fluhip::ab_getenv("FLUHIP_FEAT_SAMEFRAME"); return e ? std::atoi(e) : 0; }();
    k.prefetch = (wg ? 1 : 0) | (pr ? 2 : 0) | (sf ? 4 : 0);
  }
  FeatFusedArgs fa;
  fa.up = up; fa.dn = dn; fa.slot = slot; fa.dct = f.dct;
  fa.nBands = f.nBands; fa.nDct = f.nDct; fa.startCoeff = f.startCoeff; fa.nOut = f.nOut;
  fa.magNorm = f.magNorm; fa.usePower = f.usePower; fa.logOutput = f.logOutput; fa.out = f.out;
  if (a.fft == 1024)
  {
#ifdef FLUHIP_AB_SWITCHES // (forms measured no faster than the production one, profiles/r04/c5_experiments.md: compiled into the A/B build only)
    // FLUHIP_FEAT_FPW=2: two frames per wavefront in one instruction stream, eight wavefronts per workgroup
    {
      static const int fpw = [] { const char* e = fluhip::ab_getenv("FLUHIP_FEAT_FPW"); return e ? std::atoi(e) : 1; }();
      if (fpw == 2) return launch_feat2_t<8, 8, 8, 8, 2>(k, fa, s);
      if (fpw == 22) return launch_feat2_t<8, 8, 8, 12, 2>(k, fa, s);
      if (fpw == 3) return launch_feat2_t<8, 8, 8, 4, 3>(k, fa, s);
    }
    // FLUHIP_FEAT_NW=8|12: wavefronts per workgroup of the fused feature kernel (production 16)
    {
      static const int nw = [] { const char* e = fluhip::ab_getenv("FLUHIP_FEAT_NW"); return e ? std::atoi(e) : 16; }();
      if (nw == 8) return launch_feat_t<8, 8, 8, 8>(k, fa, s);
      if (nw == 12) return launch_feat_t<8, 8, 8, 12>(k, fa, s);
      if (nw == 10) return launch_feat_t<8, 8, 8, 10, true>(k, fa, s);   // two workgroups per CU
      if (nw == 9) return launch_feat_t<8, 8, 8, 9, true>(k, fa, s);
    }
#endif
    return launch_feat_t<8, 8, 8, 16>(k, fa, s);
  }
  if (a.fft == 2048) return launch_feat_t<16, 8, 8, 8>(k, fa, s);
  return false;
}
int stft_features_bins_per_lane(int fft) { return (fft / 2 + 1 + 63) / 64; }

template <int R1, int R2, int R3, int NW, int WINLDS, int FPW = 1, bool DB = false>
static bool launch_block_t(const StftBArgs& k0, hipStream_t s)
{
  constexpr int N = R1 * R2 * R3;
  constexpr int BUFD = FftCore<R1, R2, R3>::BUFD;
  constexpr int TW = (R2 - 1) * R1 + (R3 - 1) * R1 * R2;
  constexpr size_t shmem = ((size_t) TW + (WINLDS ? N : 0)) * 16 + (size_t) (DB ? 2 : 1) * NW * FPW * BUFD * 8;
  static_assert(shmem <= 160 * 1024, "LDS");
  StftBArgs k = k0;
  static const int pf = [] { const char* e = fluhip::ab_getenv("FLUHIP_STFT_PREFETCH"); return e ? std::atoi(e) : 1; }();
  k.prefetch = pf;
  k.blocksPerBuf = (k.T + NW * FPW - 1) / (NW * FPW);
  k.totalBlocks = (int64_t) k.B * k.blocksPerBuf;
  if (k.totalBlocks < 1) return true;
  auto kern = k.spec ? stft_block_kernel<R1, R2, R3, NW, WINLDS, true, FPW, DB> : stft_block_kernel<R1, R2, R3, NW, WINLDS, false, FPW, DB>;   // the complex spectrum is kept for resynthesis / BufSTFT only
  request_dynamic_lds(kern, (size_t) (shmem));
  const int64_t chunk = (k.totalBlocks + 7) / 8;
  int64_t grid = 8 * chunk;
  const int perCu = (int) ((160 * 1024) / shmem);
  const int64_t cap = 256 * (int64_t) (perCu < 1 ? 1 : perCu);
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(kern, dim3((unsigned) grid), dim3(64 * NW), shmem, s, k);
  return true;
}

// both magnitude layouts from one kernel; false when the shape has no block form (the caller then falls back to
// launch_stft + launch_transpose)
bool launch_stft_block(const StftArgs& a, double* magT, int64_t magTStride, int64_t ldMagT, hipStream_t s)
{
  if ((a.win % 2) != 0 || a.win > a.fft) return false;
  static const bool off = [] { const char* e = fluhip::ab_getenv("FLUHIP_STFT_BLOCK"); return e && std::atoi(e) == 0; }();
  if (off) return false;   // A/B: the round-1 wave kernel + transposing copy
  StftBArgs k;
  k.audio = a.audio; k.audio64 = a.audio64; k.n = a.n; k.audioStride = a.audioStride;
  k.win = a.win; k.fft = a.fft; k.hop = a.hop; k.T = a.T; k.F = a.F; k.B = a.B;
  k.window = a.window; k.twiddle = a.twiddle;
  k.mag = a.mag; k.magStride = a.magStride; k.ldMag = a.ldMag;
  k.magT = magT; k.magTStride = magTStride; k.ldMagT = ldMagT;
  k.spec = a.spec; k.specStride = a.specStride;
  k.frameOffset = a.frameOffset;
  k.blocksPerBuf = 0; k.totalBlocks = 0;
  k.nTab = a.nTab;
  if (magT && (ldMagT % 2) != 0) return false;
  if (a.fft == 2048)
  {
    // 8 frames per block, two wavefronts per SIMD (143 registers without the spectrum output; 12 per workgroup fit too and
    // measured the same).  FLUHIP_STFT_FPW=2: TWO frames per wavefront and round, so that 16 frames of a bin leave as one
    // 128-byte line of the bin-major copy instead of 64-byte pieces (the 16 staging buffers leave no room for the window in
    // the LDS: it is read through the L1).  Built on the round-2 review's suggestion and measured on one box, back to back:
    // 803 / 820 us against 765 / 783 us for the one-frame form (profiles/r03/stft_fpw.txt) -- full lines are not what the
    // phase waits for (fft 1024 already writes 128-byte pieces and runs at the same bytes per second), so it stays off.
    static const int fpw = [] { const char* e = fluhip::ab_getenv("FLUHIP_STFT_FPW"); return e ? std::atoi(e) : 1; }();
    if (fpw == 2) return launch_block_t<16, 8, 8, 8, 0, 2>(k, s);
    // Two sets of staging buffers used in turn (round 5): the bin-major flush of round r runs under the transforms of round
    // r + 1 and the barrier behind the flush goes -- 565 - 580 -> 513 - 542 us per launch of the bench corpus's STFT, same box,
    // alternating (profiles/r05/stft_double_buffer.txt).  The second set takes the room of the window table: the window is
    // read through the L1.  Without a bin-major copy there is no barrier to save; FLUHIP_STFT_DB=0 (A/B build): one set.
    static const int db = [] { const char* e = fluhip::ab_getenv("FLUHIP_STFT_DB"); return e ? std::atoi(e) : 1; }();
#ifdef FLUHIP_AB_SWITCHES
    // FLUHIP_STFT_NW=16 (A/B build, round 5): sixteen frames per block, four wavefronts per SIMD, one set of staging buffers, no
    // sample prefetch -- the form fft 1024 runs in; a bin's piece of the bin-major copy is a full 128-byte line.  At 128
    // registers the 16-point-per-lane transform spills 41 of them: 878 - 887 us per STFT phase of the bench corpus against
    // 651 - 652 for the production form (tools/stft_timing.py, profiles/r05/stft_nw16.txt; twelve wavefronts at 168 registers
    // spill 59: 1 240 us).  Not adopted.
    {
      static const int nw = [] { const char* e = fluhip::ab_getenv("FLUHIP_STFT_NW"); return e ? std::atoi(e) : 8; }();
      if (nw == 16) return launch_block_t<16, 8, 8, 16, 0>(k, s);
    }
#endif
    if (db == 1 && magT) return launch_block_t<16, 8, 8, 8, 0, 1, true>(k, s);
    return launch_block_t<16, 8, 8, 8, 1>(k, s);
  }
  if (a.fft == 4096)
  {
    // BASELINE config 3: 32 points per lane (one wavefront per SIMD), 4 frames per block
    return launch_block_t<16, 8, 16, 4, 1>(k, s);
  }
  if (a.fft == 1024)
  {
    // 16 frames per block (128 registers, four wavefronts per SIMD): 128-byte pieces of the bin-major rows
    // (two sets of staging buffers, as at fft 2048, measured slower here -- 681 - 690 against 659 - 663 us for 128 x 10 s at hop
    //  256: four wavefronts per SIMD already overlap the flush, and the window through the L1 costs more than the barrier)
    return launch_block_t<8, 8, 8, 16, 1>(k, s);
  }
  return false;
}


// ---------------------------------------------------------------------------------------------------------------
// Batched resynthesis (fluhip_kernels.h ResynthBatchArgs):
//   algorithm::NMF::estimate      include/flucoma/algorithms/public/NMF.hpp:33-42       est[t][f] = H1[t][k] W1[k][f]
//   algorithm::RatioMask::process include/flucoma/algorithms/public/RatioMask.hpp:33-57  Y = X min(est (1/max(Vhat,eps)), 1)
//   algorithm::ISTFT::process     include/flucoma/algorithms/public/STFT.hpp:178-199     inverse real FFT, 1/fft, window,
//                                                                                       overlap-add, / max(sum w^2, eps), trim
//   driver                        include/flucoma/clients/nrt/NMFClient.hpp:302-334
// Arithmetic as in resynth_frames_kernel / resynth_ola_kernel (kernels_istft.hip): Z[k] = conj(E[k] + i O[k]) from the
// masked bins k and N - k, forward transform of N = fft/2 points, x[2m] = Re z / N, x[2m+1] = -Im z / N, times the window.
// The input point k = lane + 64 r pairs with N - k, the point of lane 64 - l in register N/64 - 1 - r (lane 0 pairs
// within itself: register N/64 - r, and bin 0 with the Nyquist bin): one ds_bpermute round, no LDS buffer.
// The transform's outputs m = j + r NS3 (j = lane, and NS3 - lane for the two-butterfly forms) are 2 NS3 samples apart
// from register to register: a hop of S 2 NS3 samples shifts the overlap-add state by S registers, all in the lane.
// ---------------------------------------------------------------------------------------------------------------
namespace {

template <int R1, int R2, int R3, int NW, int S, bool SHARED = false>
__global__ __launch_bounds__(64 * NW) void resynth_seq_kernel(ResynthBatchArgs a, int runSlots, int runsPerBuf, int kGroups)
{
  using Core = FftCore<R1, R2, R3>;
  constexpr int N = Core::N, PPL = Core::PPL, BUFD = Core::BUFD, NS3 = Core::NS3, NB3 = Core::NB3;
  static_assert(Core::NB1 == 1, "one pass-1 butterfly per lane: point k = lane + 64 r");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  d2* tw2 = reinterpret_cast<d2*>(lds);
  d2* wl = tw2 + Core::T2 + Core::T3;             // [N] window pairs
  double* nrml = reinterpret_cast<double*>(wl + N); // [hop]
  double* xall = nrml + a.hop;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* xb = xall + wave * BUFD;
  // SHARED: the eight wavefronts of the workgroup are eight components of one (buffer, run) and walk the same frames: the
  // frame's spectrum row and reciprocal V-hat row are brought into the LDS once per workgroup, by LDS-DMA, a frame ahead
  // (two row buffers), instead of every wavefront loading them through its registers at the head of its frame
  constexpr int XROW = (N + 2) * 2, MROW = N + 2;  // doubles per row buffer (spectrum complex, reciprocal V-hat), 16-byte multiples
  double* xrow = xall + NW * BUFD;                // [2][XROW]
  double* mrowS = xrow + 2 * XROW;                // [2][MROW]
  const d2* twg = reinterpret_cast<const d2*>(a.twiddle);
  Core::fill_tables(tw2, twg, threadIdx.x, 64 * NW);
  for (int m = threadIdx.x; m < N; m += 64 * NW) wl[m] = reinterpret_cast<const d2*>(a.window)[m];
  for (int m = threadIdx.x; m < a.hop; m += 64 * NW) nrml[m] = a.nrmTab[m];
  __syncthreads();
  Core core;
  core.init(xb, tw2, twg, lane);

  // task: the workgroups of one (buffer, run) sit on one XCD (workgroup i runs on XCD i & 7), NW components each; at
  // ranks below NW a workgroup takes NW / K (buffer, run) pairs instead
  const int64_t wg = blockIdx.x;
  const int64_t slot8 = wg >> 3;
  int64_t pair = (slot8 / kGroups) * 8 + (wg & 7);
  int comp = (int) (slot8 % kGroups) * NW + wave;
  if (a.K < NW)
  {
    const int ppw = NW / a.K;
    if (wave >= ppw * a.K) return;
    pair = pair * ppw + wave / a.K;
    comp = wave % a.K;
  }
  const int b = (int) (pair / runsPerBuf), run = (int) (pair % runsPerBuf);
  if (b >= a.B) return;                           // (the whole workgroup)
  const bool active = comp < a.K;                 // SHARED: idle wavefronts still fetch and meet the barriers
  if (!SHARED && !active) return;                 // (no workgroup barrier below)
  if (!active) comp = 0;
  const int64_t nSamples = a.nTab ? a.nTab[b] : a.n;
  const int Tb = a.nTab ? (int) ((nSamples + a.hop) / a.hop) : a.T;
  const int cover = a.win / a.hop;                // frames over a position (hop | win)
  static_assert(S >= 1 && S <= R3 / 2, "registers the state shifts per frame: hop = S 2 NS3 samples");
  const int sFirst = (int) (a.trim / a.hop);
  const int sLast = (int) ((nSamples - 1 + a.trim) / a.hop);
  const int sa = sFirst + run * runSlots;
  const int sb = min(sa + runSlots, sLast + 1);
  if (sa > sLast) return;

  const double* spec = a.spec + (int64_t) b * a.specStride;
  const double* mult = a.mult + (int64_t) b * a.multStride;
  const double* wrow = a.Wt + (int64_t) b * a.wtStride + (int64_t) comp * a.F;
  const double* hcol = a.H1 + (int64_t) b * a.hStride + comp;
  float* out = a.out32 + ((int64_t) b * a.K + comp) * a.outStride;
  const bool pairStores = (reinterpret_cast<uintptr_t>(out) & 7) == 0 && (a.trim & 1) == 0;
  const double inv = 1.0 / (double) N;
  const int srcAddr = ((64 - lane) & 63) * 4;
  const bool l0 = lane == 0;
  // output samples of this lane in a frame: 2 m, 2 m + 1 with m = jA + r NS3 (and jB + r NS3)
  const int jA = lane, jB = NB3 == 2 ? (lane == 0 ? NS3 / 2 : NS3 - lane) : 0;

  cx acc[PPL];
#pragma unroll
  for (int i = 0; i < PPL; i++) acc[i] = cx{0.0, 0.0};

  // SHARED: rows of frame t into row buffer `which` (every wavefront of the workgroup takes its 1 KB pieces)
  auto fetch_rows = [&](int t, int which) {
    if constexpr (SHARED)
    {
      if (t < 0 || t >= Tb) return;
      const char* xs = reinterpret_cast<const char*>(spec + (int64_t) t * a.F * 2);
      const char* ms = reinterpret_cast<const char*>(mult + (int64_t) t * a.F);
      char* xd = reinterpret_cast<char*>(xrow + which * XROW);
      char* md = reinterpret_cast<char*>(mrowS + which * MROW);
      constexpr int XI = N / (64 * NW) > 0 ? N / (64 * NW) : 1;   // spectrum: N 16-byte bins, 1 KB per instruction
#pragma unroll
      for (int j = 0; j < XI; j++)
      {
        const int cb = (XI * wave + j) * 64;
        if (cb < N)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (xs + (int64_t) (cb + lane) * 16),
                                           (__attribute__((address_space(3))) void*) (xd + cb * 16), 16, 0, 0);
      }
      const int cbm = wave * 64;                                  // reciprocal V-hat: N / 2 16-byte pairs
      if (cbm < N / 2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (ms + (int64_t) (cbm + lane) * 16),
                                         (__attribute__((address_space(3))) void*) (md + cbm * 16), 16, 0, 0);
    }
  };
  const int tFirst = sa - (cover - 1);
  if constexpr (SHARED)
  {
    fetch_rows(tFirst, 0);
    __syncthreads();                                 // (drains the DMA: s_waitcnt vmcnt(0) is part of the barrier)
  }

  for (int t = tFirst; t < sb; t++)
  {
    const int which = (t - tFirst) & 1;
    fetch_rows(t + 1 < sb ? t + 1 : -1, which ^ 1);   // the next frame's rows, a frame ahead
    if (active && t >= 0 && t < Tb)
    {
      // ---- masked bins of this lane: k = lane + 64 r ----------------------------------------------------------------
      int ln = lane;
      asm volatile("" : "+v"(ln));                  // (opaque: nothing of the frame-invariant loads is kept across frames)
      const double hk = hcol[(int64_t) t * a.Kp];
      const d2* srow = reinterpret_cast<const d2*>(spec + (int64_t) t * a.F * 2);
      const double* mrow = mult + (int64_t) t * a.F;
      const d2* srowL = reinterpret_cast<const d2*>(xrow + which * XROW);
      const double* mrowL = mrowS + which * MROW;
      cx Y[PPL];
#pragma unroll
      for (int r = 0; r < PPL; r++)
      {
        const int f = ln + 64 * r;
        const d2 x = SHARED ? srowL[f] : srow[f];
        const double mu = SHARED ? mrowL[f] : mrow[f];
        const double m = fmin((hk * wrow[f]) * mu, 1.0);        // NMF.hpp:41, RatioMask.hpp:52-56 (exponent 1)
        Y[r] = cx{x[0] * m, x[1] * m};
      }
      if (l0) Y[0].im = 0.0;                         // packed DC (util/FFT.hpp:155-160)
      // the Nyquist bin, partner of bin 0 (lane 0)
      double yn = 0.0;
      {
        const d2 x = srow[N];
        const double m = fmin((hk * wrow[N]) * mrow[N], 1.0);
        yn = x[0] * m;
      }
      cx pts[PPL];
#pragma unroll
      for (int ii = 0; ii < PPL; ii++)
      {
        // (order 0, PPL-1, 1, PPL-2, ...: point r needs the bins of registers r, PPL-1-r and -- lane 0 -- PPL-r, so the
        //  masked bins die as the points are formed instead of all sixteen living beside all sixteen points)
        const int r = (ii & 1) ? PPL - 1 - (ii >> 1) : (ii >> 1);
        // partner N - k: lane 64 - l, register PPL - 1 - r; lane 0: its own register PPL - r (r > 0), the Nyquist bin (r = 0)
        const cx z = Y[PPL - 1 - r];
        const int lo0 = __builtin_amdgcn_ds_bpermute(srcAddr, (int) (__double_as_longlong(z.re) & 0xffffffff));
        const int hi0 = __builtin_amdgcn_ds_bpermute(srcAddr, (int) (__double_as_longlong(z.re) >> 32));
        const int lo1 = __builtin_amdgcn_ds_bpermute(srcAddr, (int) (__double_as_longlong(z.im) & 0xffffffff));
        const int hi1 = __builtin_amdgcn_ds_bpermute(srcAddr, (int) (__double_as_longlong(z.im) >> 32));
        cx Xn = cx{__longlong_as_double(((long long) hi0 << 32) | (unsigned) lo0),
                   __longlong_as_double(((long long) hi1 << 32) | (unsigned) lo1)};
        const cx own = r == 0 ? cx{yn, 0.0} : Y[PPL - r];
        if (l0) Xn = own;
        const cx X = Y[r];
        // E = (X + conj Xn) / 2, D = (X - conj Xn) / 2, O = D conj(w), Z = E + i O, point = conj Z
        const double er = 0.5 * (X.re + Xn.re), ei = 0.5 * (X.im - Xn.im);
        const double dr = 0.5 * (X.re - Xn.re), di = 0.5 * (X.im + Xn.im);
        const d2 w = twg[ln + 64 * r];               // e^{-2 pi i k / fft}
        const double orr = dr * w[0] + di * w[1], oi = di * w[0] - dr * w[1];
        pts[r] = cx{er - oi, -(ei + orr)};
      }
      cx p3[PPL];
      core.passes(pts, p3);
      // ---- x[2m] = Re / N, x[2m+1] = -Im / N, times the window, onto the running state -----------------------------------
#pragma unroll
      for (int bb = 0; bb < NB3; bb++)
#pragma unroll
        for (int r = 0; r < R3; r++)
        {
          const int i = bb * R3 + r;
          const int m = (bb == 0 ? jA : jB) + r * NS3;
          const d2 w = wl[m];
          acc[i].re += (p3[i].re * inv) * w[0];
          acc[i].im += (-p3[i].im * inv) * w[1];
        }
    }
    // ---- the oldest hop samples are final: slot t --------------------------------------------------------------------
    if (active && t >= sa)
    {
      const int64_t p0 = (int64_t) t * a.hop;
      const bool interior = t - (cover - 1) >= 0 && t <= Tb - 1;
#pragma unroll
      for (int bb = 0; bb < NB3; bb++)
#pragma unroll
        for (int r = 0; r < R3; r++)
        {
          if (r >= S) continue;
          const int i = bb * R3 + r;
          const int m = (bb == 0 ? jA : jB) + r * NS3;
          const int q = 2 * m;
          double n0, n1;
          if (interior) { n0 = nrml[q]; n1 = nrml[q + 1]; }
          else
          {
            // frames t' with t' hop <= p < t' hop + win, 0 <= t' < T, in increasing order (resynth_ola_kernel)
            n0 = 0.0; n1 = 0.0;
            for (int tt = max(0, t - (cover - 1)); tt <= min(t, Tb - 1); tt++)
            {
              const int off = (t - tt) * a.hop + q;
              const d2 w = wl[off >> 1];
              n0 += w[0] * w[0];
              n1 += w[1] * w[1];
            }
          }
          const double y0 = acc[i].re / fmax(n0, kEpsilon), y1 = acc[i].im / fmax(n1, kEpsilon);
          const int64_t i0 = p0 + q - a.trim;
          if (pairStores && i0 >= 0 && i0 + 1 < nSamples)
          {
            // (K x the input's samples leave here and nobody on the device reads them again: non-temporal, FLUHIP_RESYNTH_NT)
            typedef float f2v __attribute__((ext_vector_type(2)));
#if FLUHIP_RESYNTH_NT
            __builtin_nontemporal_store(f2v{(float) y0, (float) y1}, reinterpret_cast<f2v*>(out + i0));
#else
            *reinterpret_cast<f2v*>(out + i0) = f2v{(float) y0, (float) y1};
#endif
          }
          else
          {
            if (i0 >= 0 && i0 < nSamples) out[i0] = (float) y0;
            if (i0 + 1 >= 0 && i0 + 1 < nSamples) out[i0 + 1] = (float) y1;
          }
        }
    }
    // ---- shift the state by one hop ---------------------------------------------------------------------------------
#pragma unroll
    for (int bb = 0; bb < NB3; bb++)
#pragma unroll
      for (int r = 0; r < R3; r++)
      {
        const int i = bb * R3 + r;
        if constexpr (true) acc[i] = (r + S < R3) ? acc[bb * R3 + ((r + S < R3) ? r + S : 0)] : cx{0.0, 0.0};
      }
    if constexpr (SHARED) __syncthreads();          // the next frame's rows have landed; everyone is done with this frame's
  }
}

__global__ void resynth_mult_kernel(const double* Wt, int64_t strideWt, const double* H1, int64_t strideH, double* mult, int T,
                                    int F, int K, int Kp)
{
  // block = (8 frames, buffer): thread f sums k ascending for the 8 frames, W read component-major (Wt[k][f]: consecutive
  // threads, consecutive addresses -- with W as it lies, [f][Kp], every thread walked its own 256-byte row: 1.4 ms on
  // the bench shard)
  extern __shared__ double hs[]; // [8][Kp]
  const int b = blockIdx.y, t0 = blockIdx.x * 8;
  const double* W = Wt + (int64_t) b * strideWt;
  const double* H = H1 + (int64_t) b * strideH;
  for (int i = threadIdx.x; i < 8 * Kp; i += blockDim.x)
  {
    const int tt = i / Kp, k = i % Kp;
    hs[i] = t0 + tt < T ? H[(int64_t) (t0 + tt) * Kp + k] : 0.0;
  }
  __syncthreads();
  for (int f = threadIdx.x; f < F; f += blockDim.x)
  {
    double s[8];
#pragma unroll
    for (int tt = 0; tt < 8; tt++) s[tt] = 0.0;
    for (int k = 0; k < K; k++)
    {
      const double wk = W[(int64_t) k * F + f];
#pragma unroll
      for (int tt = 0; tt < 8; tt++) s[tt] = __builtin_fma(wk, hs[tt * Kp + k], s[tt]);
    }
#pragma unroll
    for (int tt = 0; tt < 8; tt++)
      if (t0 + tt < T) mult[((int64_t) b * T + t0 + tt) * F + f] = 1.0 / fmax(s[tt], kEpsilon); // RatioMask.hpp:39-41
  }
}

} // namespace

__global__ void resynth_nrm_kernel(const double* window, int win, int hop, double* tab)
{
  // normaliser of a padded position covered by all win / hop frames: sum of window^2 over the covering frames in
  // increasing frame order, i.e. decreasing offset (resynth_ola_kernel)
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= hop) return;
  double acc = 0.0;
  for (int j = win / hop - 1; j >= 0; j--)
  {
    const double w = window[q + j * hop];
    acc += w * w;
  }
  tab[q] = acc;
}
void launch_resynth_normaliser(const double* window, int win, int hop, double* tab, hipStream_t s)
{
  hipLaunchKernelGGL(resynth_nrm_kernel, dim3((unsigned) ((hop + 255) / 256)), dim3(256), 0, s, window, win, hop, tab);
}

void launch_resynth_mult(const double* Wf, int64_t strideW, const double* H1, int64_t strideH, double* Wt, double* mult,
                         int T, int F, int K, int Kp, int B, hipStream_t s)
{
  // Wt[b][k][f] = W[b][f][k] (Kp rows of F per buffer), then the reciprocal V-hat from it
  launch_transpose(Wf, Kp, strideW, Wt, F, (int64_t) Kp * F, F, Kp, B, s);
  hipLaunchKernelGGL(resynth_mult_kernel, dim3((unsigned) ((T + 7) / 8), (unsigned) B), dim3(256), (size_t) 8 * Kp * sizeof(double), s,
                     Wt, (int64_t) Kp * F, H1, strideH, mult, T, F, K, Kp);
}

bool resynth_batch_supported(int win, int fft, int hop)
{
  // the transform's outputs lie 2 NS3 samples apart from register to register (256 at fft 2048, 128 at fft 1024): the hop
  // 1, 2 or 4 of those (and so a divisor of the window)
  // (a window shorter than the transform: the samples past it meet a zero window and the state's upper registers stay zero)
  if (win > fft || win < hop || win % hop != 0 || (win & 1)) return false;
  if (fft == 2048) return hop == 256 || hop == 512 || hop == 1024;
  if (fft == 1024) return hop == 128 || hop == 256 || hop == 512;
  return false;
}

template <int R1, int R2, int R3, int NW>
static bool launch_resynth_batch_t(const ResynthBatchArgs& a, hipStream_t s)
{
  using Core = FftCore<R1, R2, R3>;
  const size_t shmem = ((size_t) Core::T2 + Core::T3 + Core::N) * 16 + (size_t) a.hop * 8 + (size_t) NW * Core::BUFD * 8;
  const int64_t sFirst = a.trim / a.hop, sLast = (a.n - 1 + a.trim) / a.hop;
  const int64_t slots = sLast - sFirst + 1;
  static const int runEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_RESYNTH_RUN"); return e ? std::atoi(e) : 0; }();
  // A run's first win / hop - 1 frames only fill the state (2 % at 128 slots and hop = win / 4; measured on the bench shard:
  // 32 slots 26.1 ms, 64: 26.1, 128: 25.3, 256: 28.8 -- too few workgroups per buffer there).  Few buffers / components:
  // shorter runs, down to 8 slots, until there are some 1024 workgroups (a single buffer's frames are otherwise walked by
  // a handful of wavefronts one after the other).
  const int kGroups = (a.K + NW - 1) / NW;
  const int ppw = a.K < NW ? NW / a.K : 1;
  int runSlots = 128;
  while (runSlots > 8 && (((slots + runSlots - 1) / runSlots) * a.B + ppw - 1) / ppw * kGroups < 1024) runSlots /= 2;
  if (runEnv > 0) runSlots = runEnv;
  const int runsPerBuf = (int) ((slots + runSlots - 1) / runSlots);
  const int64_t pairs = ((int64_t) a.B * runsPerBuf + ppw - 1) / ppw;
  const int64_t wgs = ((pairs + 7) / 8) * 8 * kGroups;
  // the workgroup's spectrum / V-hat rows through the LDS (see the kernel) from rank NW on: 25.3 -> 19.2 ms on the bench shard,
  // alternating on one box; FLUHIP_RESYNTH_SHARED=0: every wavefront loads its own
  static const int sharedEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_RESYNTH_SHARED"); return e ? std::atoi(e) : 1; }();
  const bool shared = sharedEnv != 0 && a.K >= NW;
  const size_t shmemS = shmem + (size_t) (2 * (Core::N + 2) * 2 + 2 * (Core::N + 2)) * 8;
  auto go = [&](auto kern, size_t sh) {
    request_dynamic_lds(kern, (size_t) (sh));
    hipLaunchKernelGGL(kern, dim3((unsigned) wgs), dim3(64 * NW), sh, s, a, runSlots, runsPerBuf, kGroups);
  };
  if (shared && shmemS <= 160 * 1024)
  {
    switch (a.hop / (2 * Core::NS3))
    {
    case 1: go(resynth_seq_kernel<R1, R2, R3, NW, 1, true>, shmemS); break;
    case 2: go(resynth_seq_kernel<R1, R2, R3, NW, 2, true>, shmemS); break;
    case 4: go(resynth_seq_kernel<R1, R2, R3, NW, 4, true>, shmemS); break;
    default: return false;
    }
    return true;
  }
  switch (a.hop / (2 * Core::NS3))
  {
  case 1: go(resynth_seq_kernel<R1, R2, R3, NW, 1>, shmem); break;
  case 2: go(resynth_seq_kernel<R1, R2, R3, NW, 2>, shmem); break;
  case 4: go(resynth_seq_kernel<R1, R2, R3, NW, 4>, shmem); break;
  default: return false;
  }
  return true;
}

bool launch_resynth_batch(const ResynthBatchArgs& a, hipStream_t s)
{
  if (!resynth_batch_supported(a.win, a.fft, a.hop) || a.F != a.fft / 2 + 1) return false;
  if (a.fft == 2048) return launch_resynth_batch_t<16, 8, 8, 8>(a, s);
  return launch_resynth_batch_t<8, 8, 8, 8>(a, s);
}

} // namespace fluhip
