// api_algorithms.hip -- the algorithm- and client-level single-buffer entry points of the C ABI:
//   algorithm::STFT::process / magnitude   include/flucoma/algorithms/public/STFT.hpp:90-108,61-66   fluhip_stft_f64 / _f32
//   algorithm::NMF::process                include/flucoma/algorithms/public/NMF.hpp:91-134          fluhip_nmf_process_*
//   algorithm::NMF::processFrame           NMF.hpp:45-89                                             fluhip_nmf_process_frames_f64
//   algorithm::NNDSVD / BufNMFSeed         NNDSVD.hpp:30-132, clients/nrt/NMFSeedClient.hpp:73-131   fluhip_nndsvd_f64, fluhip_bufnmfseed_f32
//   the channel-loop body of BufNMF        clients/nrt/NMFClient.hpp:240-334                         fluhip_bufnmf_channel_f32
#include "api_internal.h"

extern "C" {

static int stft_common(fluhip_ctx* ctx, const float* a32, const double* a64, int64_t n, int64_t stride,
                       int64_t win, int64_t fft, int64_t hop, int window_type, double* spec,
                       double* mag, int64_t* frames_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!a32 && !a64) return fail(ctx, "null audio");
  if (stride < 1) return fail(ctx, "stride must be >= 1");
  int rc = check_shape(ctx, n, win, fft, hop, 1);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.n = n; c.win = win; c.fft = fft; c.hop = hop; c.K = 1;
  c.windowType = window_type;
  c.keepSpec = spec != nullptr;
  c.stftOnly = true;   // STFT::process + magnitude: the frame-major magnitudes (and the spectrum), nothing for factor updates
  rc = corpus_alloc(ctx, &c);
  if (rc) return rc;
  // strided host view -> contiguous device copy (clients/nrt/NMFClient.hpp:240 `tmp <<= samps(...)`)
  DevBuf in;
  const size_t esz = a32 ? sizeof(float) : sizeof(double);
  HIPCHK(ctx, in.alloc((size_t) n * esz, false, ctx->stream));
  HIPCHK(ctx, upload_strided(in.p, a32 ? (const void*) a32 : (const void*) a64, (size_t) n, (size_t) stride, esz,
                             ctx->stream));
  rc = corpus_stft(&c, a32 ? in.as<float>() : nullptr, a32 ? nullptr : in.as<double>(), n, true);
  if (rc) return rc;
  if (frames_out) *frames_out = c.T;
  // (long buffers: through the pinned staging blocks -- a minute of audio at fft 2048 is 42 + 85 MB)
  if (mag && (rc = copy_to_host(ctx, mag, (size_t) c.F * sizeof(double), c.mag.p, (size_t) c.Fp * sizeof(double),
                                (size_t) c.F * sizeof(double), (size_t) c.T, ctx->stream)))
    return rc;
  if (spec)
  {
    const size_t nb = (size_t) c.T * c.F * 2 * sizeof(double);
    if ((rc = copy_to_host(ctx, spec, nb, c.spec.p, nb, nb, 1, ctx->stream))) return rc;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLUHIP_OK;
}

int fluhip_stft_f64(fluhip_ctx* ctx, const double* audio, int64_t n, int64_t stride, int64_t win,
                    int64_t fft, int64_t hop, int window_type, double* spec, double* mag,
                    int64_t* frames_out)
{
  return stft_common(ctx, nullptr, audio, n, stride, win, fft, hop, window_type, spec, mag, frames_out);
}

int fluhip_stft_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride, int64_t win,
                    int64_t fft, int64_t hop, int window_type, double* spec, double* mag,
                    int64_t* frames_out)
{
  return stft_common(ctx, audio, nullptr, n, stride, win, fft, hop, window_type, spec, mag, frames_out);
}

// ---- two-stride views at the algorithm boundary (data/FluidTensor_Support.hpp:260-420, util/FluidEigenMappings.hpp:35-225)
} // extern "C" (the helpers below are C++)
namespace {
inline bool view_empty(const fluhip_matrix_view* v) { return !v || !v->data || v->rows == 0 || v->cols == 0; }
// contiguous row-major host copy of a view (small matrices: seeds)
std::vector<double> view_gather(const fluhip_matrix_view& v)
{
  std::vector<double> out((size_t) (v.rows * v.cols));
  for (int64_t r = 0; r < v.rows; r++)
    for (int64_t c = 0; c < v.cols; c++) out[(size_t) (r * v.cols + c)] = v.data[r * v.row_stride + c * v.col_stride];
  return out;
}
void view_scatter(const fluhip_matrix_view& v, const double* src)
{
  for (int64_t r = 0; r < v.rows; r++)
    for (int64_t c = 0; c < v.cols; c++) v.data[r * v.row_stride + c * v.col_stride] = src[r * v.cols + c];
}
} // namespace
extern "C" {

int fluhip_nmf_process_views_f64(fluhip_ctx* ctx, const fluhip_matrix_view* Xv, int64_t K, int64_t iters, int update_w,
                                 int update_h, int64_t seed, const fluhip_matrix_view* W0v, const fluhip_matrix_view* H0v,
                                 const fluhip_matrix_view* W1v, const fluhip_matrix_view* H1v, const fluhip_matrix_view* V1v,
                                 fluhip_progress_fn progress, void* user)
{
  if (!ctx) return FLUHIP_ERROR;
  if (view_empty(Xv)) return fail(ctx, "bad input matrix");
  const int64_t T = Xv->rows, F = Xv->cols;
  if (K < 1) return fail(ctx, "rank must be >= 1");
  if (iters < 0) return fail(ctx, "negative iteration count");
  if (int rcr = check_rank(ctx, T, F, K)) return rcr;
  if ((Xv->col_stride == 1 && Xv->row_stride < F) || (Xv->row_stride == 1 && Xv->col_stride < T && Xv->col_stride != 1))
    return fail(ctx, "bad input matrix");
  // alg/NMF.hpp:109-110, 121-122 assert these shapes
  if (!view_empty(W0v) && (W0v->rows != K || W0v->cols != F)) return fail(ctx, "W0 must be rank x bins");
  if (!view_empty(H0v) && (H0v->rows != T || H0v->cols != K)) return fail(ctx, "H0 must be frames x rank");
  if (!view_empty(W1v) && (W1v->rows != K || W1v->cols != F)) return fail(ctx, "W1 must be rank x bins");
  if (!view_empty(H1v) && (H1v->rows != T || H1v->cols != K)) return fail(ctx, "H1 must be frames x rank");
  if (!view_empty(V1v) && (V1v->rows != T || V1v->cols != F)) return fail(ctx, "V1 must be frames x bins");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.K = K;
  // shape the corpus directly from the matrix extents (no audio behind it)
  c.hop = 1; c.n = T - 1; c.fft = (F - 1) * 2; c.win = c.fft;
  c.T = T; c.F = F;
  c.Tp = round_up(T, 32); c.Fp = round_up(F, 32); c.Kp = padded_rank(K);
  {
    HIPCHK(ctx, c.mag.alloc((size_t) c.Tp * c.Fp * sizeof(double), true, s));
    HIPCHK(ctx, c.magT.alloc((size_t) c.Fp * c.Tp * sizeof(double), true, s));
    HIPCHK(ctx, c.Wf.alloc((size_t) c.Fp * c.Kp * sizeof(double), true, s));
    HIPCHK(ctx, c.H1.alloc((size_t) c.Tp * c.Kp * sizeof(double), true, s));
    HIPCHK(ctx, c.hmax.alloc(sizeof(double), true, s));
    if (int rc2 = plan_updates(ctx, &c)) return rc2;
  }
  // alg/NMF.hpp:125  V = X^T.  A view with unit column stride is the T x F row-major image the frame-major copy wants;
  // a view with unit ROW stride (FluidTensorView::transpose() of an F x T matrix) is byte for byte the bin-major copy:
  // either goes up as one strided 2-D copy and the other layout is made on the device.  Anything else (both strides
  // non-unit) is gathered on the host first.
  // (a one-row view with a non-unit column stride is NOT a contiguous row: the frame-major path needs unit column stride
  //  or a single column; such a view is byte for byte a bin-major image of one frame and takes the second path)
  std::vector<double> xtmp;
  if (Xv->col_stride == 1 || F == 1)
  {
    HIPCHK(ctx, hipMemcpy2DAsync(c.mag.p, (size_t) c.Fp * sizeof(double), Xv->data, (size_t) Xv->row_stride * sizeof(double),
                                 (size_t) F * sizeof(double), (size_t) T, hipMemcpyHostToDevice, s));
    launch_transpose(c.mag.as<double>(), c.Fp, c.Tp * c.Fp, c.magT.as<double>(), c.Tp, c.Fp * c.Tp, (int) T, (int) F, 1, s);
  }
  else if (Xv->row_stride == 1 || T == 1)
  {
    HIPCHK(ctx, hipMemcpy2DAsync(c.magT.p, (size_t) c.Tp * sizeof(double), Xv->data, (size_t) Xv->col_stride * sizeof(double),
                                 (size_t) T * sizeof(double), (size_t) F, hipMemcpyHostToDevice, s));
    launch_transpose(c.magT.as<double>(), c.Tp, c.Fp * c.Tp, c.mag.as<double>(), c.Fp, c.Tp * c.Fp, (int) F, (int) T, 1, s);
  }
  else
  {
    xtmp = view_gather(*Xv);
    HIPCHK(ctx, hipMemcpy2DAsync(c.mag.p, (size_t) c.Fp * sizeof(double), xtmp.data(), (size_t) F * sizeof(double),
                                 (size_t) F * sizeof(double), (size_t) T, hipMemcpyHostToDevice, s));
    launch_transpose(c.mag.as<double>(), c.Fp, c.Tp * c.Fp, c.magT.as<double>(), c.Tp, c.Fp * c.Tp, (int) T, (int) F, 1, s);
  }
  c.haveMag = true;
  // seeds are small (K x F, T x K): contiguous host images whatever their strides
  std::vector<double> w0tmp, h0tmp;
  FactorInit fi;
  fi.sharedW = fi.sharedH = true;
  if (!view_empty(W0v))
  {
    if (W0v->col_stride == 1 && W0v->row_stride == F) fi.W0host = W0v->data;
    else { w0tmp = view_gather(*W0v); fi.W0host = w0tmp.data(); }
  }
  if (!view_empty(H0v))
  {
    if (H0v->col_stride == 1 && H0v->row_stride == K) fi.H0host = H0v->data;
    else { h0tmp = view_gather(*H0v); fi.H0host = h0tmp.data(); }
  }
  int rc = corpus_init_factors(&c, seed, nullptr, fi);
  if (rc) return rc;
  rc = corpus_iterate(&c, iters, update_w != 0, update_h != 0, progress, user);
  if (rc != FLUHIP_OK && rc != FLUHIP_CANCELLED) return rc;
  const bool cancelled = rc == FLUHIP_CANCELLED;
  // alg/NMF.hpp:127-133 outputs; :182 V = W*H only when the loop ran to completion
  DevBuf dw, dh, dv, dvt;
  std::vector<double> w1tmp, h1tmp, v1tmp;
  if (!view_empty(W1v))
  {
    const bool direct = W1v->col_stride == 1 && W1v->row_stride == F;
    if (!direct) w1tmp.resize((size_t) (K * F));
    HIPCHK(ctx, dw.alloc((size_t) K * F * sizeof(double), false, s));
    launch_gather_w_f64(c.Wf.as<double>(), 0, dw.as<double>(), 0, (int) F, (int) K, (int) c.Kp, 1, s);
    HIPCHK(ctx, hipMemcpyAsync(direct ? W1v->data : w1tmp.data(), dw.p, (size_t) K * F * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (!view_empty(H1v))
  {
    const bool direct = H1v->col_stride == 1 && H1v->row_stride == K;
    if (!direct) h1tmp.resize((size_t) (T * K));
    HIPCHK(ctx, dh.alloc((size_t) T * K * sizeof(double), false, s));
    launch_gather_h_f64(c.H1.as<double>(), 0, dh.as<double>(), 0, (int) T, (int) K, (int) c.Kp, 1, s);
    HIPCHK(ctx, hipMemcpyAsync(direct ? H1v->data : h1tmp.data(), dh.p, (size_t) T * K * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (!view_empty(V1v) && !cancelled)
  {
    HIPCHK(ctx, dv.alloc((size_t) T * F * sizeof(double), false, s));
    launch_vhat(c.Wf.as<double>(), 0, c.H1.as<double>(), 0, dv.as<double>(), F, 0, (int) T, (int) F,
                (int) c.Kp, 1, s);
    if (V1v->col_stride == 1 || F == 1)
      HIPCHK(ctx, hipMemcpy2DAsync(V1v->data, (size_t) V1v->row_stride * sizeof(double), dv.p, (size_t) F * sizeof(double),
                                   (size_t) F * sizeof(double), (size_t) T, hipMemcpyDeviceToHost, s));
    else if (V1v->row_stride == 1 || T == 1)
    {
      // a transposed view: the F x T image, made on the device
      HIPCHK(ctx, dvt.alloc((size_t) F * T * sizeof(double), false, s));
      launch_transpose(dv.as<double>(), F, 0, dvt.as<double>(), T, 0, (int) T, (int) F, 1, s);
      HIPCHK(ctx, hipMemcpy2DAsync(V1v->data, (size_t) V1v->col_stride * sizeof(double), dvt.p, (size_t) T * sizeof(double),
                                   (size_t) T * sizeof(double), (size_t) F, hipMemcpyDeviceToHost, s));
    }
    else
    {
      v1tmp.resize((size_t) (T * F));
      HIPCHK(ctx, hipMemcpyAsync(v1tmp.data(), dv.p, (size_t) T * F * sizeof(double), hipMemcpyDeviceToHost, s));
    }
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(s));
  if (!w1tmp.empty()) view_scatter(*W1v, w1tmp.data());
  if (!h1tmp.empty()) view_scatter(*H1v, h1tmp.data());
  if (!v1tmp.empty()) view_scatter(*V1v, v1tmp.data());
  return cancelled ? FLUHIP_CANCELLED : FLUHIP_OK;
}

int fluhip_nmf_process_f64(fluhip_ctx* ctx, const double* X, int64_t T, int64_t F, int64_t ldx,
                           int64_t K, int64_t iters, int update_w, int update_h, int64_t seed,
                           const double* W0, const double* H0, double* W1, double* H1,
                           double* V1, fluhip_progress_fn progress, void* user)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!X || T < 1 || F < 1 || ldx < F) return fail(ctx, "bad input matrix");
  const fluhip_matrix_view xv{const_cast<double*>(X), T, F, ldx, 1};
  const fluhip_matrix_view w0{const_cast<double*>(W0), K, F, F, 1}, h0{const_cast<double*>(H0), T, K, K, 1};
  const fluhip_matrix_view w1{W1, K, F, F, 1}, h1{H1, T, K, K, 1}, v1{V1, T, F, F, 1};
  return fluhip_nmf_process_views_f64(ctx, &xv, K, iters, update_w, update_h, seed, W0 ? &w0 : nullptr, H0 ? &h0 : nullptr,
                                      W1 ? &w1 : nullptr, H1 ? &h1 : nullptr, V1 ? &v1 : nullptr, progress, user);
}

} // extern "C"

// NMF::processFrame (alg/NMF.hpp:45-89) over the rows of a magnitude matrix that is ALREADY ON THE DEVICE: c->mag holds
// T rows of F magnitudes (row stride c->Fp, allocated with the padding zero), c->T / F / K / Tp / Fp / Kp are set.  Leaves
// the clamped, row-normalised dictionary in c->Wf and the activations of every frame in c->H1 ([Tp][Kp]).
int process_frames_on_device(fluhip_ctx* ctx, fluhip_corpus& c, const double* W0host, int64_t iters, int64_t seed)
{
  hipStream_t s = ctx->stream;
  const int64_t T = c.T, F = c.F, K = c.K;
  HIPCHK(ctx, c.magT.alloc((size_t) c.Fp * c.Tp * sizeof(double), true, s));
  HIPCHK(ctx, c.Wf.alloc((size_t) c.Fp * c.Kp * sizeof(double), true, s));
  HIPCHK(ctx, c.H1.alloc((size_t) c.Tp * c.Kp * sizeof(double), true, s));
  if (int rc2 = plan_updates(ctx, &c)) return rc2;
  // :57-58, 61  v0 = max(x, eps)
  launch_clamp_eps(c.mag.as<double>(), c.Fp, 0, (int) T, (int) F, 1, s);
  launch_transpose(c.mag.as<double>(), c.Fp, c.Tp * c.Fp, c.magT.as<double>(), c.Tp, c.Fp * c.Tp, (int) T,
                   (int) F, 1, s);
  // :59, 64-65  W = max(W, eps), every component divided by its L2 norm over the bins
  HIPCHK(ctx, c.stage.alloc((size_t) K * F * sizeof(double), false, s));
  HIPCHK(ctx, hipMemcpyAsync(c.stage.p, W0host, (size_t) K * F * sizeof(double), hipMemcpyHostToDevice, s));
  launch_scatter_factor(c.stage.as<double>(), 0, c.Wf.as<double>(), c.Fp * c.Kp, (int) F, (int) K, (int) c.Kp, 1,
                        true, s);
  HIPCHK(ctx, c.normScratch.alloc((size_t) colnorm_scratch_doubles((int) F, (int) c.Kp, 1) * sizeof(double), false, s));
  launch_colnorm(c.Wf.as<double>(), c.Fp * c.Kp, (int) F, (int) K, (int) c.Kp, 1, true, false,
                 c.normScratch.as<double>(), s);
  // :55-56, 60  h = max(uniform(0,1)^K, eps): the same K draws for every frame when seeded
  std::vector<double> h0, rows((size_t) T * K);
  if (seed >= 0)
  {
    draw_uniform(seed, (size_t) K, h0);
    for (int64_t t = 0; t < T; t++) std::memcpy(&rows[(size_t) t * K], h0.data(), (size_t) K * sizeof(double));
  }
  else
    draw_uniform(seed, (size_t) T * K, rows);
  DevBuf hs;
  HIPCHK(ctx, hs.alloc((size_t) T * K * sizeof(double), false, s));
  HIPCHK(ctx, hipMemcpyAsync(hs.p, rows.data(), (size_t) T * K * sizeof(double), hipMemcpyHostToDevice, s));
  launch_scatter_factor(hs.as<double>(), 0, c.H1.as<double>(), c.Tp * c.Kp, (int) T, (int) K, (int) c.Kp, 1, false, s);
  launch_clamp_eps(c.H1.as<double>(), c.Kp, 0, (int) T, (int) K, 1, s);
  HIPCHK(ctx, hipStreamSynchronize(s)); // host staging vectors go out of use
  c.haveMag = c.haveFactors = true;
  // :71-79  nIterations of the H update
  return corpus_iterate(&c, iters, false, true, nullptr, nullptr);
}

extern "C" {

int fluhip_nmf_process_frames_f64(fluhip_ctx* ctx, const double* X, int64_t T, int64_t F, int64_t ldx,
                                  const double* W0, int64_t K, int64_t iters, int64_t seed, double* H, double* V)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!X || T < 1 || F < 1 || ldx < F) return fail(ctx, "bad input matrix");
  if (!W0 || K < 1) return fail(ctx, "bad dictionary");
  if (iters < 0) return fail(ctx, "negative iteration count");
  if (int rcr = check_rank(ctx, T, F, K)) return rcr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.K = K;
  c.hop = 1; c.n = T - 1; c.fft = (F - 1) * 2; c.win = c.fft;
  c.T = T; c.F = F;
  c.Tp = round_up(T, 32); c.Fp = round_up(F, 32); c.Kp = padded_rank(K);
  HIPCHK(ctx, c.mag.alloc((size_t) c.Tp * c.Fp * sizeof(double), true, s));
  HIPCHK(ctx, hipMemcpy2DAsync(c.mag.p, (size_t) c.Fp * sizeof(double), X, (size_t) ldx * sizeof(double),
                               (size_t) F * sizeof(double), (size_t) T, hipMemcpyHostToDevice, s));
  int rc = process_frames_on_device(ctx, c, W0, iters, seed);
  if (rc != FLUHIP_OK) return rc;
  DevBuf dh, dv;
  if (H)
  {
    HIPCHK(ctx, dh.alloc((size_t) T * K * sizeof(double), false, s));
    launch_gather_h_f64(c.H1.as<double>(), 0, dh.as<double>(), 0, (int) T, (int) K, (int) c.Kp, 1, s);
    HIPCHK(ctx, hipMemcpyAsync(H, dh.p, (size_t) T * K * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (V) // :87  v = W^T h
  {
    HIPCHK(ctx, dv.alloc((size_t) T * F * sizeof(double), false, s));
    launch_vhat(c.Wf.as<double>(), 0, c.H1.as<double>(), 0, dv.as<double>(), F, 0, (int) T, (int) F, (int) c.Kp, 1, s);
    HIPCHK(ctx, hipMemcpyAsync(V, dv.p, (size_t) T * F * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(s));
  return FLUHIP_OK;
}


// ---------------------------------------------------------------------------------------
// NNDSVD (alg/NNDSVD.hpp) -- the SVD by one-sided Jacobi on the device (kernels_svd.hip), the O(k (F + T))
// construction on the host
// ---------------------------------------------------------------------------------------
// G: device [F][ldg], row f = bin f over the T frames (the transposed magnitude copy; destroyed).  Top factors
// to the host: s [min(F,T)] descending, U [k][F] (row j = u_j), VT [k][T]; k from the coverage rule.
static int nndsvd_device(fluhip_ctx* ctx, double* G, int64_t F, int64_t T, int64_t ldg, int64_t minRank, int64_t maxRank,
                         double amount, std::vector<double>& s, std::vector<double>& U, std::vector<double>& VT,
                         int64_t* kOut)
{
  hipStream_t st = ctx->stream;
  DevBuf dJ, dN, dFlag;
  HIPCHK(ctx, dJ.alloc((size_t) F * F * sizeof(double), false, st));
  HIPCHK(ctx, dN.alloc((size_t) F * sizeof(double), false, st));
  HIPCHK(ctx, dFlag.alloc(sizeof(unsigned), true, st));
  const int sweeps = launch_jacobi_svd(G, ldg, (int) F, (int) T, dJ.as<double>(), dN.as<double>(), dFlag.as<unsigned>(),
                                       40, st);
  HIPCHK(ctx, hipGetLastError());
  if (sweeps < 0) return fail(ctx, "the SVD did not converge");
  std::vector<double> norms((size_t) F);
  HIPCHK(ctx, hipMemcpyAsync(norms.data(), dN.p, (size_t) F * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  std::vector<int64_t> order((size_t) F);
  for (int64_t i = 0; i < F; i++) order[(size_t) i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return norms[(size_t) a] > norms[(size_t) b]; });
  const int64_t r = std::min(F, T);
  s.resize((size_t) r);
  for (int64_t i = 0; i < r; i++) s[(size_t) i] = norms[(size_t) order[(size_t) i]];
  // alg/NNDSVD.hpp:47-58
  int64_t k = 0;
  if (amount == 0) k = minRank;
  else
  {
    double current = 0, total = 0;
    for (double v : s) total += v;
    while ((current / total) < amount && k < r) current += s[(size_t) k++];
  }
  if (k < minRank) k = minRank;
  if (k > maxRank) k = maxRank;
  if (k > r) return fail(ctx, "rank above min(bins, frames)");
  *kOut = k;
  U.assign((size_t) std::max<int64_t>(k, 1) * F, 0.0);
  VT.assign((size_t) std::max<int64_t>(k, 1) * T, 0.0);
  for (int64_t j = 0; j < k; j++)
  {
    const int64_t row = order[(size_t) j];
    HIPCHK(ctx, hipMemcpyAsync(&U[(size_t) j * F], dJ.as<double>() + row * F, (size_t) F * sizeof(double),
                               hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(&VT[(size_t) j * T], G + row * ldg, (size_t) T * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(ctx, hipStreamSynchronize(st));
  for (int64_t j = 0; j < k; j++)
  {
    const double sj = s[(size_t) j];
    if (sj > 0)
      for (int64_t t = 0; t < T; t++) VT[(size_t) j * T + t] /= sj;
  }
  return FLUHIP_OK;
}

// alg/NNDSVD.hpp:60-129 from (U, s, V^T).  W: wRows x F row-major, H: T x wRows row-major.
static void nndsvd_construct(const std::vector<double>& s, const std::vector<double>& U, const std::vector<double>& VT,
                             int64_t F, int64_t T, int64_t k, int64_t wRows, int method, int64_t seed, double mean,
                             double* W, double* H)
{
  const double eps = kEpsilon;
  std::fill(W, W + wRows * F, 0.0);
  std::fill(H, H + T * wRows, 0.0);
  auto u = [&](int64_t j) { return &U[(size_t) j * F]; };
  auto v = [&](int64_t j) { return &VT[(size_t) j * T]; };
  if (method == 0)
  {
    for (int64_t j = 0; j < k; j++)
    {
      for (int64_t f = 0; f < F; f++) W[j * F + f] = std::fabs(u(j)[f]);
      for (int64_t t = 0; t < T; t++) H[t * wRows + j] = std::fabs(s[(size_t) j] * v(j)[t]);
    }
    return;
  }
  if (k > 0)
  {
    for (int64_t f = 0; f < F; f++) W[f] = std::fabs(u(0)[f]);                                  // :68
    const double sq = std::sqrt(s[0]);
    for (int64_t t = 0; t < T; t++) H[t * wRows] = sq * std::fabs(v(0)[t]);                       // :69
  }
  for (int64_t j = 1; j < k; j++)
  {
    double xP = 0, yP = 0, xN = 0;
    for (int64_t f = 0; f < F; f++) { const double x = u(j)[f]; if (x > 0) xP += x * x; else xN += x * x; }
    for (int64_t t = 0; t < T; t++) { const double y = v(j)[t]; if (y > 0) yP += y * y; }
    const double xPn = std::sqrt(xP), yPn = std::sqrt(yP), xNn = std::sqrt(xN);
    const double yNn = xNn;                                                                       // :85 as written
    const double mP = xPn * yPn, mN = xNn * yNn;
    const bool pos = mP > mN;
    const double sigma = pos ? mP : mN;
    const double lbd = std::sqrt(s[(size_t) j] * sigma);
    const double xn = pos ? xPn : xNn, yn = pos ? yPn : yNn; // :90-100 (yNn is ||xN||, see above)
    for (int64_t f = 0; f < F; f++)
    {
      const double x = u(j)[f];
      W[j * F + f] = (pos ? std::max(x, 0.0) : std::fabs(std::min(x, 0.0))) / xn;
    }
    for (int64_t t = 0; t < T; t++)
    {
      const double y = v(j)[t];
      H[t * wRows + j] = lbd * ((pos ? std::max(y, 0.0) : std::fabs(std::min(y, 0.0))) / yn);
    }
  }
  if (method == 1)
  {
    // :107-116: the lazily evaluated random matrix is only sampled where the condition holds, in the assignment's
    // column-major traversal (WT is F x wRows, HT is wRows x T); a fresh generator of the same seed for each
    std::random_device rd;
    const double lo = eps, hi = mean * 0.001;
    {
      std::mt19937_64 g{seed >= 0 ? (size_t) seed : (size_t) rd()};
      std::uniform_real_distribution<double> d{lo, hi};
      for (int64_t j = 0; j < wRows; j++)
        for (int64_t f = 0; f < F; f++)
          if (W[j * F + f] < eps) W[j * F + f] = d(g);
    }
    {
      std::mt19937_64 g{seed >= 0 ? (size_t) seed : (size_t) rd()};
      std::uniform_real_distribution<double> d{lo, hi};
      for (int64_t t = 0; t < T; t++)
        for (int64_t j = 0; j < wRows; j++)
          if (H[t * wRows + j] < eps) H[t * wRows + j] = d(g);
    }
  }
  else if (method == 2)
  {
    for (int64_t i = 0; i < wRows * F; i++) if (W[i] < eps) W[i] = mean;
    for (int64_t i = 0; i < T * wRows; i++) if (H[i] < eps) H[i] = mean;
  }
}

int fluhip_nndsvd_f64(fluhip_ctx* ctx, const double* X, int64_t T, int64_t F, int64_t ldx, int64_t w_rows,
                      int64_t min_rank, int64_t max_rank, double amount, int method, int64_t seed, double* W,
                      double* H, int64_t* rank_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!X || !W || !H || T < 1 || F < 1 || ldx < F) return fail(ctx, "bad matrix arguments");
  if (method < 0 || method > 3) return fail(ctx, "method must be 0..3");
  if (!(amount > 0 || min_rank > 0)) return fail(ctx, "coverage or minimum rank must be positive"); // :40 assert
  if (amount > 1) return fail(ctx, "coverage must be <= 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf A, G;
  HIPCHK(ctx, A.alloc((size_t) T * F * sizeof(double), false, st));
  HIPCHK(ctx, G.alloc((size_t) F * T * sizeof(double), false, st));
  HIPCHK(ctx, hipMemcpy2DAsync(A.p, (size_t) F * sizeof(double), X, (size_t) ldx * sizeof(double),
                               (size_t) F * sizeof(double), (size_t) T, hipMemcpyHostToDevice, st));
  launch_transpose(A.as<double>(), F, 0, G.as<double>(), T, 0, (int) T, (int) F, 1, st); // one bin per row
  std::vector<double> s, U, VT;
  int64_t k = 0;
  if (int rc = nndsvd_device(ctx, G.as<double>(), F, T, T, min_rank, max_rank, amount, s, U, VT, &k)) return rc;
  if (k > w_rows) return fail(ctx, "rank exceeds the rows of W");
  double mean = 0;
  for (int64_t t = 0; t < T; t++)
    for (int64_t f = 0; f < F; f++) mean += X[t * ldx + f];
  mean /= (double) (T * F);
  nndsvd_construct(s, U, VT, F, T, k, w_rows, method, seed, mean, W, H);
  if (rank_out) *rank_out = k;
  return FLUHIP_OK;
}

int fluhip_bufnmfseed_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride, int64_t win,
                          int64_t fft, int64_t hop, int64_t min_rank, int64_t max_rank, double coverage,
                          int method, int64_t seed, float* bases_out, float* acts_out, int64_t* rank_out)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!audio) return fail(ctx, "null audio");
  if (stride < 1) return fail(ctx, "stride must be >= 1");
  if (max_rank < 1) return fail(ctx, "maximum rank must be >= 1");
  int rc = check_shape(ctx, n, win, fft, hop, 1);
  if (rc) return rc;
  if (method < 0 || method > 3) return fail(ctx, "method must be 0..3");
  if (!(coverage > 0 || min_rank > 0)) return fail(ctx, "coverage or minimum rank must be positive");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.n = n; c.win = win; c.fft = fft; c.hop = hop; c.K = 1;
  rc = corpus_alloc(ctx, &c);
  if (rc) return rc;
  DevBuf in;
  HIPCHK(ctx, in.alloc((size_t) n * sizeof(float), false, st));
  HIPCHK(ctx, upload_strided(in.p, audio, (size_t) n, (size_t) stride, sizeof(float), st));
  rc = corpus_stft(&c, in.as<float>(), nullptr, n); // NMFSeedClient.hpp:97-98
  if (rc) return rc;
  const int64_t T = c.T, F = c.F;
  // mean of the magnitudes for methods 1 and 2 (alg/NNDSVD.hpp:105): on the host from a copy of the
  // spectrogram (it is small next to the SVD)
  std::vector<double> mag((size_t) T * F);
  HIPCHK(ctx, hipMemcpy2DAsync(mag.data(), (size_t) F * sizeof(double), c.mag.p, (size_t) c.Fp * sizeof(double),
                               (size_t) F * sizeof(double), (size_t) T, hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  double mean = 0;
  for (double v : mag) mean += v;
  mean /= (double) (T * F);
  std::vector<double> s, U, VT;
  int64_t k = 0;
  // the transposed copy holds one bin per row, as the Jacobi kernels want it; they work in place
  rc = nndsvd_device(ctx, c.magT.as<double>(), F, T, c.Tp, min_rank, max_rank, coverage, s, U, VT, &k);
  if (rc) return rc;
  std::vector<double> W((size_t) max_rank * F), H((size_t) T * max_rank);
  nndsvd_construct(s, U, VT, F, T, k, max_rank, method, seed, mean, W.data(), H.data());
  // NMFSeedClient.hpp:108-128
  if (bases_out)
    for (int64_t i = 0; i < max_rank * F; i++) bases_out[i] = i < k * F ? (float) W[(size_t) i] : 0.f;
  if (acts_out)
  {
    double maxH = H[0];
    for (double v : H) maxH = std::max(maxH, v);
    const float scale = (float) (1.0 / maxH);
    for (int64_t j = 0; j < max_rank; j++)
      for (int64_t t = 0; t < T; t++)
        acts_out[j * T + t] = j < k ? (float) H[(size_t) t * max_rank + j] * scale : 0.f;
  }
  if (rank_out) *rank_out = k;
  return FLUHIP_OK;
}

int fluhip_bufnmf_channel_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride,
                              int64_t win, int64_t fft, int64_t hop, int64_t K, int64_t iters,
                              int update_w, int update_h, int64_t seed, const float* bases_seed,
                              const float* acts_seed, float* bases_out, float* acts_out,
                              float* resynth_out, fluhip_progress_fn progress, void* user)
{
  if (!ctx) return FLUHIP_ERROR;
  if (!audio) return fail(ctx, "null audio");
  if (stride < 1) return fail(ctx, "stride must be >= 1");
  int rc = check_shape(ctx, n, win, fft, hop, K);
  if (rc) return rc;
  if (iters < 0) return fail(ctx, "negative iteration count");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  fluhip_corpus c;
  c.ctx = ctx; c.B = 1; c.n = n; c.win = win; c.fft = fft; c.hop = hop; c.K = K;
  c.keepSpec = resynth_out != nullptr; // the complex spectrogram is only needed for resynthesis
  rc = corpus_alloc(ctx, &c);
  if (rc) return rc;
  DevBuf in;
  HIPCHK(ctx, in.alloc((size_t) n * sizeof(float), false, s));
  HIPCHK(ctx, upload_strided(in.p, audio, (size_t) n, (size_t) stride, sizeof(float), s));
  rc = corpus_stft(&c, in.as<float>(), nullptr, n); // nrt/NMFClient.hpp:240-242
  if (rc) return rc;
  FactorInit fi;
  fi.W0f32 = bases_seed; // :246-258 seeds gathered channel by channel
  fi.H0f32 = acts_seed;
  rc = corpus_init_factors(&c, seed, nullptr, fi);
  if (rc) return rc;
  rc = corpus_iterate(&c, iters, update_w != 0, update_h != 0, progress, user); // :268-271
  if (rc) return rc;                                                             // :273-274
  rc = fluhip_corpus_writeback_host(&c, bases_out, acts_out);                    // :277-300
  if (rc) return rc;
  if (resynth_out) // :302-334  estimate -> ratio mask -> ISTFT per component (the corpus form, one buffer)
  {
    DevBuf out32;
    HIPCHK(ctx, out32.alloc((size_t) K * n * sizeof(float), false, s));
    c.haveFactors = true;
    rc = fluhip_corpus_resynth_dev(&c, out32.as<float>());
    if (rc) return rc;
    const size_t nbytes = (size_t) K * n * sizeof(float);
    rc = copy_to_host(ctx, resynth_out, nbytes, out32.p, nbytes, nbytes, 1, s);
    if (rc) return rc;
  }
  return FLUHIP_OK;
}

} // extern "C"
