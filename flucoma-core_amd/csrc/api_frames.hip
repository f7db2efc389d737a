// api_frames.hip -- the users of NMF::processFrame behind the C ABI: the real-time clients NMFMatch and NMFFilter
// (include/flucoma/clients/rt/NMFMatchClient.hpp:76-118, NMFFilterClient.hpp:69-118) as the reference's own offline
// wrapper templates drive a real-time client over a buffer (clients/common/FluidNRTClientWrapper.hpp: StreamingControl
// :551-660 for the control-rate output of NMFMatch, Streaming :466-547 for the audio outputs of NMFFilter).
//
// The reference pushes host vectors through the client and solves one frame at a time; frames are independent (a fresh
// generator of the seed per processFrame call, the dictionary re-read from its buffer per call), so here every frame of
// every channel is ONE batch of the H update with the dictionary fixed (process_frames_on_device, api_algorithms.hip),
// in front of it the STFT kernels at the ring buffers' frame positions, behind it -- for NMFFilter -- the ratio-mask /
// inverse-transform / overlap-add kernels of the BufNMF resynthesis at those positions.  The frame positions are the
// closed form of the FluidSource / FluidSink bookkeeping; the test suite holds both a literal model of the clients
// and the closed form (numpy, under oracle/) and holds them against each other.
#include "api_internal.h"

extern "C" {

int fluhip_nmfmatch_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft, int64_t hop,
                        const float* bases, int64_t K, int64_t seed, int padding_mode, float* out, int64_t* frames_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!audio) return fail(ctx, "null buffer");
  if (padding_mode < 0 || padding_mode > 2) return fail(ctx, "padding mode must be 0 (None), 1 (Default) or 2 (Full)");
  if (count < 1) return fail(ctx, "need at least one channel");
  int rc = check_shape(ctx, n, win, fft, hop, K);
  if (rc) return rc;
  // StreamingControl::process, cc/FluidNRTClientWrapper.hpp:557-579, 643-647
  const int64_t pad = padding_mode == 0 ? 0 : padding_mode == 1 ? win >> 1 : win - hop; // FFTParams::padding
  int64_t padded = n + win + 2 * pad;
  if (padding_mode == 2) padded = ((padded + hop - 1) / hop) * hop;
  const int64_t nAnalysis = 1 + (padded - win) / hop;
  const int64_t latencyHops = win / hop;
  const int64_t T = nAnalysis - latencyHops;
  if (frames_out) *frames_out = T;
  if (T < 1) return fail(ctx, "not enough frames");
  if (!out) return FLUHIP_OK; // size query
  if (!bases) return fail(ctx, "null buffer");
  const int64_t F = fft / 2 + 1, Ttot = count * T;
  if (int rcr = check_rank(ctx, Ttot, F, K)) return rcr;
  if (Ttot > 2000000000LL / 16) return fail(ctx, "too many frames");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const double *wtab = nullptr, *ttab = nullptr;
  rc = get_window(ctx, win, fft, FLUHIP_WINDOW_HANN, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, fft, &ttab);
  if (rc) return rc;

  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.K = K;
  c.hop = 1; c.n = Ttot - 1; c.fft = fft; c.win = fft;
  c.T = Ttot; c.F = F;
  c.Tp = round_up(Ttot, 32); c.Fp = round_up(F, 32); c.Kp = padded_rank(K);
  DevBuf in;
  HIPCHK(ctx, in.alloc((size_t) count * n * sizeof(float), false, s));
  HIPCHK(ctx, hipMemcpyAsync(in.p, audio, (size_t) count * n * sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, c.mag.alloc((size_t) c.Tp * c.Fp * sizeof(double), true, s));
  // Kept column k is what the client returned in call k + latencyHops: its activations BEFORE that call's frame, i.e. those
  // of the frame of the call before -- FluidSource hands call j the window that ends where block j begins, padded samples
  // [j hop - win, j hop) -- so column k belongs to the frame at audio sample (k + latencyHops - 1) hop - win - pad
  // (rt/NMFMatchClient.hpp:104 writes the output, :108-117 then process the call's frame).
  StftArgs sa;
  sa.audio = in.as<float>(); sa.audio64 = nullptr; sa.n = n; sa.audioStride = n;
  sa.win = (int) win; sa.fft = (int) fft; sa.hop = (int) hop; sa.T = (int) T; sa.F = (int) F; sa.B = (int) count;
  sa.window = wtab; sa.twiddle = ttab;
  sa.mag = c.mag.as<double>(); sa.magStride = T * c.Fp; sa.ldMag = c.Fp; // the channels' frames one after the other: one matrix
  sa.spec = nullptr; sa.specStride = 0;
  sa.frameOffset = (int) ((latencyHops - 1) * hop - win - pad + win / 2);
  sa.bigScratch = big_fft_scratch(ctx, win, fft, Ttot);
  if (stft_needs_scratch(win, fft) && !sa.bigScratch) return FLUHIP_ERROR;
  launch_stft(sa, s);
  HIPCHK(ctx, hipGetLastError());
  // the filter buffer's channels as the client copies them (:100-101, float -> double)
  std::vector<double> W0((size_t) K * F);
  for (size_t i = 0; i < W0.size(); i++) W0[i] = (double) bases[i];
  rc = process_frames_on_device(ctx, c, W0.data(), 10, seed); // :113-116: ten iterations, whatever `iterations` says
  if (rc) return rc;
  DevBuf dh;
  HIPCHK(ctx, dh.alloc((size_t) Ttot * K * sizeof(double), false, s));
  launch_gather_h_f64(c.H1.as<double>(), 0, dh.as<double>(), 0, (int) Ttot, (int) K, (int) c.Kp, 1, s);
  std::vector<double> h((size_t) Ttot * K);
  HIPCHK(ctx, hipMemcpyAsync(h.data(), dh.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(ctx, hipStreamSynchronize(s));
  // out[channel][component][column], the layout the features of the other control clients leave (fluhip_bufmfcc_f32)
  for (int64_t b = 0; b < count; b++)
    for (int64_t k = 0; k < K; k++)
    {
      float* o = out + (b * K + k) * T;
      for (int64_t t = 0; t < T; t++) o[t] = (float) h[(size_t) ((b * T + t) * K + k)];
      if (latencyHops == 0) o[0] = 0.0f; // no call before the first one: mActivations as constructed (:63)
    }
  return FLUHIP_OK;
}

int fluhip_nmffilter_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft, int64_t hop,
                         const float* bases, int64_t K, int64_t iters, int64_t seed, float* out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!audio || !bases || !out) return fail(ctx, "null buffer");
  if (count < 1) return fail(ctx, "need at least one channel");
  if (iters < 0) return fail(ctx, "negative iteration count");
  int rc = check_shape(ctx, n, win, fft, hop, K);
  if (rc) return rc;
  if (hop > win) return fail(ctx, "hop sizes above the window size are not supported by the filter");
  // Streaming::process (cc/FluidNRTClientWrapper.hpp:466-547): the ring buffers delay the signal by one window (latency(),
  // rt/NMFFilterClient.hpp:64), the wrapper drops that many output samples -- frame m = 1, 2, ... covers the audio samples
  // [m hop - win, m hop) and is overlap-added where it came from.  Frames that start at or behind the end add nothing.
  const int64_t T = (n + win + hop - 1) / hop - 1;
  if (T < 1) return fail(ctx, "not enough frames");
  const int64_t F = fft / 2 + 1, Ttot = count * T;
  if (int rcr = check_rank(ctx, Ttot, F, K)) return rcr;
  if (Ttot > 2000000000LL / 16) return fail(ctx, "too many frames");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const double *wtab = nullptr, *ttab = nullptr;
  rc = get_window(ctx, win, fft, FLUHIP_WINDOW_HANN, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, fft, &ttab);
  if (rc) return rc;

  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.K = K;
  c.hop = 1; c.n = Ttot - 1; c.fft = fft; c.win = fft;
  c.T = Ttot; c.F = F;
  c.Tp = round_up(Ttot, 32); c.Fp = round_up(F, 32); c.Kp = padded_rank(K);
  DevBuf in, spec;
  HIPCHK(ctx, in.alloc((size_t) count * n * sizeof(float), false, s));
  HIPCHK(ctx, hipMemcpyAsync(in.p, audio, (size_t) count * n * sizeof(float), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, c.mag.alloc((size_t) c.Tp * c.Fp * sizeof(double), true, s));
  HIPCHK(ctx, spec.alloc((size_t) Ttot * F * 2 * sizeof(double), false, s));
  StftArgs sa;
  sa.audio = in.as<float>(); sa.audio64 = nullptr; sa.n = n; sa.audioStride = n;
  sa.win = (int) win; sa.fft = (int) fft; sa.hop = (int) hop; sa.T = (int) T; sa.F = (int) F; sa.B = (int) count;
  sa.window = wtab; sa.twiddle = ttab;
  sa.mag = c.mag.as<double>(); sa.magStride = T * c.Fp; sa.ldMag = c.Fp;
  sa.spec = spec.as<double>(); sa.specStride = T * F * 2;
  sa.frameOffset = (int) (hop - win + win / 2); // frame t = m - 1 starts at (t + 1) hop - win
  sa.bigScratch = big_fft_scratch(ctx, win, fft, Ttot);
  if (stft_needs_scratch(win, fft) && !sa.bigScratch) return FLUHIP_ERROR;
  launch_stft(sa, s);
  HIPCHK(ctx, hipGetLastError());
  std::vector<double> W0((size_t) K * F);
  for (size_t i = 0; i < W0.size(); i++) W0[i] = (double) bases[i];
  rc = process_frames_on_device(ctx, c, W0.data(), iters, seed); // rt/NMFFilterClient.hpp:102-105
  if (rc) return rc;
  // :106-113  mask.init(estimate); per component NMF::estimate -> RatioMask::process -> (BufferedProcess) ISTFT::processFrame,
  // overlap-add, division by the overlap-added window^2.  c.Wf holds the dictionary as processFrame left tmpFilt: clamped, normalised.
  DevBuf vhat, frames, dout;
  HIPCHK(ctx, vhat.alloc((size_t) T * F * sizeof(double), false, s));
  const int64_t compsPerLaunch = std::max<int64_t>(1, std::min<int64_t>(K, ((int64_t) 1 << 30) / (T * win * 8)));
  HIPCHK(ctx, frames.alloc((size_t) compsPerLaunch * T * win * sizeof(double), false, s));
  HIPCHK(ctx, dout.alloc((size_t) count * K * n * sizeof(float), false, s));
  for (int64_t b = 0; b < count; b++)
  {
    const double* Hb = c.H1.as<double>() + b * T * c.Kp;
    launch_vhat(c.Wf.as<double>(), 0, Hb, 0, vhat.as<double>(), F, 0, (int) T, (int) F, (int) c.Kp, 1, s); // NMF.hpp:87 v = W^T h
    ResynthArgs ra;
    ra.spec = spec.as<double>() + b * T * F * 2; ra.Wf = c.Wf.as<double>(); ra.H1 = Hb;
    ra.Vhat = vhat.as<double>(); ra.ldV = F; ra.Kp = (int) c.Kp;
    ra.win = (int) win; ra.fft = (int) fft; ra.hop = (int) hop; ra.T = (int) T; ra.F = (int) F;
    ra.window = wtab; ra.twiddle = ttab; ra.frames = frames.as<double>(); ra.out = nullptr; ra.n = n; ra.outStride = n;
    ra.trim = win - hop; // frame t lies at [t hop - trim, t hop - trim + win) of the output
    for (int64_t k = 0; k < K; k += compsPerLaunch)
    {
      ra.k = (int) k;
      ra.nComp = (int) std::min(compsPerLaunch, K - k);
      ra.out32 = dout.as<float>() + (b * K + k) * n;
      ra.bigScratch = big_fft_scratch(ctx, ra.win, ra.fft, ra.T);
      if (stft_needs_scratch(ra.win, ra.fft) && !ra.bigScratch) return FLUHIP_ERROR;
      launch_resynth(ra, s);
    }
  }
  HIPCHK(ctx, hipGetLastError());
  const size_t nbytes = (size_t) count * K * n * sizeof(float);
  return copy_to_host(ctx, out, nbytes, dout.p, nbytes, nbytes, 1, s);
}

} // extern "C"
