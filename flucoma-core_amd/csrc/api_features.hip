// api_features.hip -- BufSTFT (clients/nrt/BufSTFTClient.hpp:81-276) and the feature pipeline BufMelBands / BufMFCC
// (algorithms/public/MelBands.hpp, DCT.hpp; clients/rt/MFCCClient.hpp, MelBandsClient.hpp behind StreamingControl) of the C ABI.
#include "api_internal.h"

extern "C" {

// ---- BufSTFT ------------------------------------------------------------------
static int64_t bufstft_padding(int64_t win, int64_t hop, int mode)
{
  return mode == 0 ? 0 : (mode == 1 ? win >> 1 : win - hop); // cc/ParameterTypes.hpp:315-323
}

int fluhip_bufstft_forward_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride, int64_t win,
                               int64_t fft, int64_t hop, int padding_mode, float* mag, float* phase,
                               int64_t* hops_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!audio) return fail(ctx, "No input buffer supplied");
  if (!mag && !phase) return fail(ctx, "Neither magnitude nor phase buffer supplied");
  if (padding_mode < 0 || padding_mode > 2) return fail(ctx, "padding mode must be 0, 1 or 2");
  if (stride < 1) return fail(ctx, "stride must be >= 1");
  int rc = check_shape(ctx, n, win, fft, hop, 1);
  if (rc) return rc;
  if (fft / 2 + 1 >= 65536) // nrt/BufSTFTClient.hpp:135-138
    return fail(ctx, "Can produce up to 65536 channels. Split your data up and try again");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int64_t F = fft / 2 + 1, pad = bufstft_padding(win, hop, padding_mode);
  int64_t padded = n + 2 * pad;                                      // :121-124
  if (padding_mode == 2) padded = ((padded + hop - 1) / hop) * hop;   // :125-127
  if (padded < win) return fail(ctx, "not enough frames");
  const int64_t T = 1 + (padded - win) / hop;                         // :129-130
  if (hops_out) *hops_out = T;
  const double *wtab = nullptr, *ttab = nullptr;
  rc = get_window(ctx, win, fft, FLUHIP_WINDOW_HANN, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, fft, &ttab);
  if (rc) return rc;
  DevBuf in, spec, dm, dp;
  HIPCHK(ctx, in.alloc((size_t) n * sizeof(float), false, s));
  HIPCHK(ctx, upload_strided(in.p, audio, (size_t) n, (size_t) stride, sizeof(float), s));
  HIPCHK(ctx, spec.alloc((size_t) T * F * 2 * sizeof(double), false, s));
  if (mag) HIPCHK(ctx, dm.alloc((size_t) T * F * sizeof(float), false, s));
  if (phase) HIPCHK(ctx, dp.alloc((size_t) T * F * sizeof(float), false, s));
  StftArgs sa;
  sa.audio = in.as<float>(); sa.audio64 = nullptr; sa.n = n; sa.audioStride = n;
  sa.win = (int) win; sa.fft = (int) fft; sa.hop = (int) hop; sa.T = (int) T; sa.F = (int) F; sa.B = 1;
  sa.window = wtab; sa.twiddle = ttab; sa.mag = nullptr; sa.magStride = 0; sa.ldMag = 0;
  sa.spec = spec.as<double>(); sa.specStride = 0;
  sa.frameOffset = (int) (win / 2 - pad); // frame i starts at sample i*hop - padding (:151-162)
  sa.bigScratch = big_fft_scratch(ctx, win, fft, T);
  if (stft_needs_scratch(win, fft) && !sa.bigScratch) return FLUHIP_ERROR;
  launch_stft(sa, s);
  launch_spec_to_magphase(spec.as<double>(), (int) T, (int) F, mag ? dm.as<float>() : nullptr,
                          phase ? dp.as<float>() : nullptr, s);
  HIPCHK(ctx, hipGetLastError());
  if (mag) HIPCHK(ctx, hipMemcpyAsync(mag, dm.p, (size_t) T * F * sizeof(float), hipMemcpyDeviceToHost, s));
  if (phase) HIPCHK(ctx, hipMemcpyAsync(phase, dp.p, (size_t) T * F * sizeof(float), hipMemcpyDeviceToHost, s));
  HIPCHK(ctx, hipStreamSynchronize(s));
  return FLUHIP_OK;
}

int fluhip_bufstft_inverse_f32(fluhip_ctx* ctx, const float* mag, const float* phase, int64_t hops, int64_t win,
                               int64_t fft, int64_t hop, int padding_mode, float* out, int64_t* n_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!mag || !phase) return fail(ctx, "Need both magnutude and phase buffers for inverse transform");
  if (padding_mode < 0 || padding_mode > 2) return fail(ctx, "padding mode must be 0, 1 or 2");
  if (hops < 1) return fail(ctx, "not enough frames");
  int rc = check_shape(ctx, 1, win, fft, hop, 1);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int64_t F = fft / 2 + 1, T = hops, pad = bufstft_padding(win, hop, padding_mode);
  const int64_t paddedOut = (T - 1) * hop + win; // nrt/BufSTFTClient.hpp:233
  const int64_t finalOut = paddedOut - pad;      // :234
  if (n_out) *n_out = finalOut;
  if (!out) return FLUHIP_OK;                    // size query
  const double *wtab = nullptr, *ttab = nullptr;
  rc = get_window(ctx, win, fft, FLUHIP_WINDOW_HANN, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, fft, &ttab);
  if (rc) return rc;
  DevBuf dm, dp, spec, frames, dout;
  HIPCHK(ctx, dm.alloc((size_t) T * F * sizeof(float), false, s));
  HIPCHK(ctx, dp.alloc((size_t) T * F * sizeof(float), false, s));
  HIPCHK(ctx, spec.alloc((size_t) T * F * 2 * sizeof(double), false, s));
  HIPCHK(ctx, frames.alloc((size_t) T * win * sizeof(double), false, s));
  HIPCHK(ctx, dout.alloc((size_t) finalOut * sizeof(float), false, s));
  HIPCHK(ctx, hipMemcpyAsync(dm.p, mag, (size_t) T * F * sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(dp.p, phase, (size_t) T * F * sizeof(float), hipMemcpyHostToDevice, s));
  launch_polar_to_spec(dm.as<float>(), dp.as<float>(), (int) T, (int) F, spec.as<double>(), s);
  ResynthArgs ra;
  ra.spec = spec.as<double>(); ra.Wf = nullptr; ra.H1 = nullptr; ra.Vhat = nullptr; ra.ldV = 0; ra.Kp = 0; ra.k = 0;
  ra.win = (int) win; ra.fft = (int) fft; ra.hop = (int) hop; ra.T = (int) T; ra.F = (int) F;
  ra.window = wtab; ra.twiddle = ttab; ra.frames = frames.as<double>(); ra.out = nullptr;
  ra.out32 = dout.as<float>(); ra.n = finalOut; ra.trim = pad;
  ra.bigScratch = big_fft_scratch(ctx, ra.win, ra.fft, ra.T);
  if (stft_needs_scratch(ra.win, ra.fft) && !ra.bigScratch) return FLUHIP_ERROR;
  launch_resynth(ra, s);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, dout.p, (size_t) finalOut * sizeof(float), hipMemcpyDeviceToHost, s));
  HIPCHK(ctx, hipStreamSynchronize(s));
  return FLUHIP_OK;
}

// ---- feature pipeline -------------------------------------------------------
static int features_common(fluhip_ctx* ctx, bool mfcc, const float* audio, int64_t count, int64_t n, int64_t win,
                           int64_t fft, int64_t hop, int64_t nBands, int64_t nCoefs, int64_t startCoeff,
                           double minFreq, double maxFreq, double sampleRate, int normalize, int scaleDb,
                           int paddingMode, float* out, int64_t* frames_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!audio || !out) return fail(ctx, "null buffer");
  if (paddingMode < 0 || paddingMode > 2) return fail(ctx, "padding mode must be 0 (None), 1 (Default) or 2 (Full)");
  if (count < 1) return fail(ctx, "need at least one buffer");
  int rc = check_shape(ctx, n, win, fft, hop, 1);
  if (rc) return rc;
  if (nBands < 2 || nBands > fft / 2 + 1) return fail(ctx, "numBands must be in [2, fft/2 + 1]");
  if (!(maxFreq > minFreq)) return fail(ctx, "maxFreq must be above minFreq");
  if (mfcc && (nCoefs < 2 || nCoefs > nBands || startCoeff < 0 || startCoeff > 1))
    return fail(ctx, "numCoeffs must be in [2, numBands] and startCoeff in [0, 1]");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int64_t F = fft / 2 + 1;
  // StreamingControl bookkeeping (cc/FluidNRTClientWrapper.hpp:564-579, 642-644)
  // userPadding.first = FFTParams::padding (cc/ParameterTypes.hpp:315-323): 0 / win/2 / win - hop; the input sits that
  // far into the padded signal, the client's latency (= win) is added in front of the analysis, the padded length is
  // rounded up to whole hops in Full mode (:572-574), and the first latency / hop output frames are dropped (:643-656)
  const int64_t latencyHops = win / hop;
  const int64_t userPad = paddingMode == 0 ? 0 : paddingMode == 1 ? win / 2 : win - hop;
  int64_t paddedLength = n + win + 2 * userPad;
  if (paddingMode == 2) paddedLength = ((paddedLength + hop - 1) / hop) * hop;
  const int64_t T = 1 + (paddedLength - win) / hop - latencyHops;
  // kept frame k starts at sample latencyHops hop - win - userPad + k hop; the kernels place frame t at
  // t hop - win/2 + frameOffset
  const int64_t frameOffset = latencyHops * hop - win + win / 2 - userPad;
  if (T < 1) return fail(ctx, "not enough frames");
  if (frames_out) *frames_out = T;
  const int64_t Tp = round_up(T, 32), Fp = round_up(F, 32);
  const int64_t bandsPad = round_up(nBands, 64);
  // mel filter bank (alg/MelBands.hpp:53-73), bin-major and zero padded; f64 on the host like the reference
  std::vector<double> filtT((size_t) F * bandsPad, 0.0);
  {
    auto hz2mel = [](double x) { return 1127.01048 * std::log(x / 700.0 + 1.0); };
    const int64_t nc = nBands + 2;
    std::vector<double> centres((size_t) nc);
    const double mlo = hz2mel(minFreq), mhi = hz2mel(maxFreq);
    for (int64_t i = 0; i < nc; i++)
      centres[(size_t) i] = 700.0 * (std::exp((mlo + (double) i * (mhi - mlo) / (double) (nc - 1)) / 1127.01048) - 1.0);
    for (int64_t b = 0; b < nBands; b++)
    {
      const double d0 = std::fabs(centres[(size_t) b] - centres[(size_t) b + 1]);
      const double d1 = std::fabs(centres[(size_t) b + 1] - centres[(size_t) b + 2]);
      for (int64_t f = 0; f < F; f++)
      {
        const double hz = (double) f * (sampleRate / 2.0) / (double) (F - 1);
        const double lower = -(centres[(size_t) b] - hz) / d0, upper = (centres[(size_t) b + 2] - hz) / d1;
        filtT[(size_t) (f * bandsPad + b)] = std::max(0.0, std::min(lower, upper));
      }
    }
  }
  // the filter bank over each band's support only (kernels_feat.hip): first non-zero bin and packed weights
  std::vector<int> bandLo((size_t) bandsPad, 0);
  int64_t maxLen = 1;
  {
    std::vector<int64_t> hi((size_t) bandsPad, -1);
    for (int64_t b = 0; b < nBands; b++)
    {
      int64_t lo = -1;
      for (int64_t f = 0; f < F; f++)
        if (filtT[(size_t) (f * bandsPad + b)] != 0.0) { if (lo < 0) lo = f; hi[(size_t) b] = f; }
      bandLo[(size_t) b] = (int) std::max<int64_t>(lo, 0);
      if (lo >= 0) maxLen = std::max(maxLen, hi[(size_t) b] - lo + 1);
    }
  }
  std::vector<double> wpack((size_t) maxLen * bandsPad, 0.0);
  for (int64_t b = 0; b < nBands; b++)
    for (int64_t j = 0; j < maxLen; j++)
    {
      const int64_t f = bandLo[(size_t) b] + j;
      if (f < F) wpack[(size_t) (j * bandsPad + b)] = filtT[(size_t) (f * bandsPad + b)];
    }
  const int64_t nDct = mfcc ? std::min(nCoefs + startCoeff, nBands) : 0; // rt/MFCCClient.hpp:104-105
  std::vector<double> dct((size_t) std::max<int64_t>(1, nDct * nBands));
  for (int64_t i = 0; i < nDct; i++) // alg/DCT.hpp:53-61
  {
    const double scale = i == 0 ? 1.0 / std::sqrt((double) nBands) : std::sqrt(2.0 / (double) nBands);
    for (int64_t j = 0; j < nBands; j++)
      dct[(size_t) (i * nBands + j)] = std::cos((M_PI / (double) nBands) * (double) i * (0.5 + (double) j)) * scale;
  }
  const int64_t nOut = mfcc ? nCoefs : nBands;
  const double *wtab = nullptr, *ttab = nullptr;
  rc = get_window(ctx, win, fft, FLUHIP_WINDOW_HANN, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, fft, &ttab);
  if (rc) return rc;
  DevBuf dFilt, dDct, dAudio, dMag, dOut, dLo, dPack;
  HIPCHK(ctx, dLo.alloc(bandLo.size() * sizeof(int), false, s));
  HIPCHK(ctx, dPack.alloc(wpack.size() * sizeof(double), false, s));
  HIPCHK(ctx, hipMemcpyAsync(dLo.p, bandLo.data(), bandLo.size() * sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(dPack.p, wpack.data(), wpack.size() * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, dFilt.alloc(filtT.size() * sizeof(double), false, s));
  HIPCHK(ctx, dDct.alloc(dct.size() * sizeof(double), false, s));
  HIPCHK(ctx, hipMemcpyAsync(dFilt.p, filtT.data(), filtT.size() * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(dDct.p, dct.data(), dct.size() * sizeof(double), hipMemcpyHostToDevice, s));
  // ---- fused form (kernels_stft2.hip stft_feat_kernel): the magnitudes never leave the chip --------------------------
  // Every bin must lie on the rising edge of at most one band and the falling edge of the band below it, with the
  // bins of each edge contiguous: then band b = (sum of up[f] m[f] over interval b) + (sum of dn[f] m[f] over interval
  // b + 1), interval s = the bins between centres s and s + 1.  True of any filter bank whose triangles are wider than
  // a bin; checked here against the dense matrix, coefficient by coefficient, and anything else takes the two-kernel path.
  {
    const int CH = stft_features_bins_per_lane((int) fft);
    std::vector<double> up((size_t) 64 * CH, 0.0), dn((size_t) 64 * CH, 0.0);
    std::vector<short> slot((size_t) 64 * CH, (short) -1);
    std::vector<int64_t> interval((size_t) F, -1); // interval of bin f, -1: no band touches it
    bool ok = nBands <= 64 && (!mfcc || nDct * nBands <= 4096) && !stft_needs_scratch(win, fft) && (fft == 1024 || fft == 2048) &&
              (win % 2) == 0;
    if (const char* e = fluhip::ab_getenv("FLUHIP_FEAT_FUSED")) // A/B and tests: 0 forces the two-kernel form
      if (std::atoi(e) == 0) ok = false;
    std::vector<int64_t> peak((size_t) nBands, 0);
    for (int64_t b = 0; ok && b < nBands; b++)
    {
      double best = -1.0;
      for (int64_t f = 0; f < F; f++)
        if (filtT[(size_t) (f * bandsPad + b)] > best) { best = filtT[(size_t) (f * bandsPad + b)]; peak[(size_t) b] = f; }
      if (best <= 0.0) ok = false; // a band no bin falls into
    }
    for (int64_t f = 0; ok && f < F; f++)
    {
      int64_t b1 = -1, b2 = -1, cnt = 0;
      for (int64_t b = 0; b < nBands; b++)
        if (filtT[(size_t) (f * bandsPad + b)] != 0.0) { if (cnt == 0) b1 = b; else b2 = b; cnt++; }
      if (cnt == 0) continue;
      if (cnt > 2 || (cnt == 2 && b2 != b1 + 1)) { ok = false; break; }
      if (cnt == 2)
      {
        interval[(size_t) f] = b2;
        up[(size_t) f] = filtT[(size_t) (f * bandsPad + b2)];
        dn[(size_t) f] = filtT[(size_t) (f * bandsPad + b1)];
      }
      else if (f <= peak[(size_t) b1]) { interval[(size_t) f] = b1; up[(size_t) f] = filtT[(size_t) (f * bandsPad + b1)]; }
      else { interval[(size_t) f] = b1 + 1; dn[(size_t) f] = filtT[(size_t) (f * bandsPad + b1)]; }
    }
    // interval s starts at bin g[s].  The touched bins must be one contiguous run whose intervals ascend one at a time
    // from some i0 up to nBands (the falling edge of the last band); intervals below i0 are empty and the running
    // sums are still 0 at their boundaries, which therefore publish nothing.
    std::vector<int64_t> g((size_t) nBands + 2, -1);
    if (ok)
    {
      int64_t prev = -1, first = -1, last = -1;
      bool ended = false;
      for (int64_t f = 0; f < F && ok; f++)
      {
        const int64_t iv = interval[(size_t) f];
        if (iv < 0) { if (first >= 0) ended = true; continue; }
        if (ended) { ok = false; break; }              // touched bins are not one contiguous run
        if (first < 0) first = f;
        else if (iv != prev && iv != prev + 1) { ok = false; break; }
        if (iv != prev) g[(size_t) iv] = f;
        prev = iv;
        last = f;
      }
      if (first < 0 || prev != nBands) ok = false;
      if (ok)
      {
        int64_t i0 = 0;
        while (g[(size_t) i0] < 0) i0++;
        for (int64_t sI = 0; sI < i0; sI++) g[(size_t) sI] = first;
        g[(size_t) nBands + 1] = last + 1;
        for (int64_t sI = i0 + 1; sI <= nBands + 1; sI++) slot[(size_t) (g[(size_t) sI] - 1)] = (short) sI;
      }
      // reconstruction: the segment sums must give back the dense matrix exactly
      for (int64_t b = 0; ok && b < nBands; b++)
        for (int64_t f = 0; f < F; f++)
        {
          double w = 0.0;
          if (f >= g[(size_t) b] && f < g[(size_t) b + 1]) w += up[(size_t) f];
          if (f >= g[(size_t) b + 1] && f < g[(size_t) b + 2]) w += dn[(size_t) f];
          if (w != filtT[(size_t) (f * bandsPad + b)]) { ok = false; break; }
        }
    }
    if (ok)
    {
      DevBuf dUp, dDn, dSlot, dDct2, dAud, dOutF;
      HIPCHK(ctx, dUp.alloc(up.size() * sizeof(double), false, s));
      HIPCHK(ctx, dDn.alloc(dn.size() * sizeof(double), false, s));
      HIPCHK(ctx, dSlot.alloc(slot.size() * sizeof(short), false, s));
      HIPCHK(ctx, dDct2.alloc(dct.size() * sizeof(double), false, s));
      HIPCHK(ctx, hipMemcpyAsync(dUp.p, up.data(), up.size() * sizeof(double), hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(dDn.p, dn.data(), dn.size() * sizeof(double), hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(dSlot.p, slot.data(), slot.size() * sizeof(short), hipMemcpyHostToDevice, s));
      HIPCHK(ctx, hipMemcpyAsync(dDct2.p, dct.data(), dct.size() * sizeof(double), hipMemcpyHostToDevice, s));
      // device-resident audio / output are used in place; host buffers go through staging chunks of bounded size
      hipPointerAttribute_t pa;
      const bool audDev = hipPointerGetAttributes(&pa, audio) == hipSuccess && pa.type == hipMemoryTypeDevice;
      const bool outDev = hipPointerGetAttributes(&pa, out) == hipSuccess && pa.type == hipMemoryTypeDevice;
      (void) hipGetLastError();
      int64_t chunkBytes = (int64_t) 1 << 31;
      if (const char* e = fluhip::ab_getenv("FLUHIP_FEAT_CHUNK_BYTES")) chunkBytes = std::max<int64_t>(1, std::atoll(e)); // tests: force several chunks
      const int64_t chunkB = (audDev && outDev) ? count
                                                : std::max<int64_t>(1, std::min<int64_t>(count, chunkBytes / (n * (int64_t) sizeof(float))));
      if (!audDev) HIPCHK(ctx, dAud.alloc((size_t) chunkB * n * sizeof(float), false, s));
      if (!outDev) HIPCHK(ctx, dOutF.alloc((size_t) chunkB * nOut * T * sizeof(float), false, s));
      for (int64_t b0 = 0; b0 < count; b0 += chunkB)
      {
        const int64_t nb = std::min(chunkB, count - b0);
        const float* aPtr = audio + b0 * n;
        if (!audDev)
        {
          HIPCHK(ctx, hipMemcpyAsync(dAud.p, aPtr, (size_t) nb * n * sizeof(float), hipMemcpyDefault, s));
          aPtr = dAud.as<float>();
        }
        float* oPtr = outDev ? out + b0 * nOut * T : dOutF.as<float>();
        StftArgs sa;
        sa.audio = aPtr; sa.audio64 = nullptr; sa.n = n; sa.audioStride = n;
        sa.win = (int) win; sa.fft = (int) fft; sa.hop = (int) hop; sa.T = (int) T; sa.F = (int) F; sa.B = (int) nb;
        sa.window = wtab; sa.twiddle = ttab;
        sa.mag = nullptr; sa.magStride = 0; sa.ldMag = 0; sa.spec = nullptr; sa.specStride = 0;
        sa.frameOffset = (int) frameOffset; sa.bigScratch = nullptr;
        FeatArgs fa;
        fa.mag = nullptr; fa.magStride = 0; fa.ldMag = 0;
        fa.T = (int) T; fa.F = (int) F; fa.B = (int) nb; fa.win = (int) win;
        fa.filtT = nullptr; fa.nBands = (int) nBands; fa.bandsPad = (int) bandsPad;
        fa.bandLo = nullptr; fa.wpack = nullptr; fa.maxLen = 0;
        fa.magNorm = mfcc ? 0 : (normalize ? 1 : 0); fa.usePower = 0; fa.logOutput = mfcc ? 1 : (scaleDb ? 1 : 0);
        fa.dct = mfcc ? dDct2.as<double>() : nullptr; fa.nDct = (int) nDct; fa.startCoeff = (int) startCoeff;
        fa.nOut = (int) nOut; fa.out = oPtr;
        bool launched;
        {
          ProfScope p(ctx, 2);
          launched = launch_stft_features(sa, fa, dUp.as<double>(), dDn.as<double>(), dSlot.as<short>(), s);
        }
        if (!launched) { ok = false; break; }
        HIPCHK(ctx, hipGetLastError());
        if (!outDev)
          HIPCHK(ctx, hipMemcpyAsync(out + b0 * nOut * T, oPtr, (size_t) nb * nOut * T * sizeof(float), hipMemcpyDefault, s));
        if (!audDev || !outDev) HIPCHK(ctx, hipStreamSynchronize(s));   // the staging buffers are reused by the next chunk
      }
      if (ok)
      {
        HIPCHK(ctx, hipStreamSynchronize(s));
        return FLUHIP_OK;
      }
    }
  }
  // ---- two-kernel form: magnitudes through HBM, any filter bank / fft size -------------------------------------------
  // buffers are processed in chunks that keep the magnitude scratch around 2 GiB
  const int64_t perBuf = Tp * Fp * (int64_t) sizeof(double);
  int64_t scratchBytes = 2LL << 30;
  if (const char* e = fluhip::ab_getenv("FLUHIP_FEAT_CHUNK_BYTES")) scratchBytes = std::max<int64_t>(1, std::atoll(e)); // tests: force several chunks
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(count, 65535), scratchBytes / perBuf));
  HIPCHK(ctx, dAudio.alloc((size_t) chunk * n * sizeof(float), false, s));
  HIPCHK(ctx, dMag.alloc((size_t) chunk * perBuf, true, s));
  HIPCHK(ctx, dOut.alloc((size_t) chunk * nOut * T * sizeof(float), false, s));
  for (int64_t b0 = 0; b0 < count; b0 += chunk)
  {
    const int64_t nb = std::min(chunk, count - b0);
    // hipMemcpyDefault: `audio` and `out` may be host or device pointers (a corpus already resident in HBM skips PCIe)
    HIPCHK(ctx, hipMemcpyAsync(dAudio.p, audio + b0 * n, (size_t) nb * n * sizeof(float), hipMemcpyDefault, s));
    StftArgs sa;
    sa.audio = dAudio.as<float>(); sa.audio64 = nullptr; sa.n = n; sa.audioStride = n;
    sa.win = (int) win; sa.fft = (int) fft; sa.hop = (int) hop; sa.T = (int) T; sa.F = (int) F; sa.B = (int) nb;
    sa.window = wtab; sa.twiddle = ttab;
    sa.mag = dMag.as<double>(); sa.magStride = Tp * Fp; sa.ldMag = Fp;
    sa.spec = nullptr; sa.specStride = 0; sa.frameOffset = (int) frameOffset;
    sa.bigScratch = big_fft_scratch(ctx, win, fft, nb * T);
    if (stft_needs_scratch(win, fft) && !sa.bigScratch) return FLUHIP_ERROR;
    {
      ProfScope p(ctx, 0);
      launch_stft(sa, s);
    }
    FeatArgs fa;
    fa.mag = dMag.as<double>(); fa.magStride = Tp * Fp; fa.ldMag = Fp;
    fa.T = (int) T; fa.F = (int) F; fa.B = (int) nb; fa.win = (int) win;
    fa.filtT = dFilt.as<double>(); fa.nBands = (int) nBands; fa.bandsPad = (int) bandsPad;
    fa.bandLo = dLo.as<int>(); fa.wpack = dPack.as<double>(); fa.maxLen = (int) maxLen;
    // rt/MFCCClient.hpp:123-124 (false, false, true); rt/MelBandsClient.hpp:106-108 (normalize, false, scale == dB)
    fa.magNorm = mfcc ? 0 : (normalize ? 1 : 0); fa.usePower = 0; fa.logOutput = mfcc ? 1 : (scaleDb ? 1 : 0);
    fa.dct = mfcc ? dDct.as<double>() : nullptr; fa.nDct = (int) nDct; fa.startCoeff = (int) startCoeff;
    fa.nOut = (int) nOut; fa.out = dOut.as<float>();
    {
      ProfScope p(ctx, 2);
      launch_features(fa, s);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out + b0 * nOut * T, dOut.p, (size_t) nb * nOut * T * sizeof(float),
                               hipMemcpyDefault, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
  }
  return FLUHIP_OK;
}

int fluhip_bufmelbands_padded_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win,
                                  int64_t fft, int64_t hop, int64_t n_bands, double min_freq, double max_freq,
                                  double sample_rate, int normalize, int scale_db, int padding_mode, float* out,
                                  int64_t* frames_out)
{
  return features_common(ctx, false, audio, count, n, win, fft, hop, n_bands, 0, 0, min_freq, max_freq,
                         sample_rate, normalize, scale_db, padding_mode, out, frames_out);
}
int fluhip_bufmelbands_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win,
                           int64_t fft, int64_t hop, int64_t n_bands, double min_freq, double max_freq,
                           double sample_rate, int normalize, int scale_db, float* out, int64_t* frames_out)
{
  return fluhip_bufmelbands_padded_f32(ctx, audio, count, n, win, fft, hop, n_bands, min_freq, max_freq, sample_rate,
                                       normalize, scale_db, 1, out, frames_out);
}

int fluhip_bufmfcc_padded_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                              int64_t hop, int64_t n_bands, int64_t n_coefs, int64_t start_coeff, double min_freq,
                              double max_freq, double sample_rate, int padding_mode, float* out, int64_t* frames_out)
{
  return features_common(ctx, true, audio, count, n, win, fft, hop, n_bands, n_coefs, start_coeff, min_freq,
                         max_freq, sample_rate, 0, 0, padding_mode, out, frames_out);
}
int fluhip_bufmfcc_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                       int64_t hop, int64_t n_bands, int64_t n_coefs, int64_t start_coeff, double min_freq,
                       double max_freq, double sample_rate, float* out, int64_t* frames_out)
{
  return fluhip_bufmfcc_padded_f32(ctx, audio, count, n, win, fft, hop, n_bands, n_coefs, start_coeff, min_freq,
                                   max_freq, sample_rate, 1, out, frames_out);
}

} // extern "C"
