/*
 * flucoma_hip.h -- C ABI of libflucoma_hip.so: the MI355X (gfx950) implementation of
 * flucoma-core's buffered spectral-decomposition hot path (STFT -> magnitude -> KL-NMF).
 *
 * This is the drop-in boundary.  Every entry point replaces a reference interface that is
 * cited next to it (paths relative to <flucoma-core>/include/flucoma/).  Plain pointers and
 * sizes only: no C++ types, no torch types, no exceptions cross this boundary.
 *
 * Conventions
 *   - status codes map 1:1 onto client::Result::Status (clients/common/Result.hpp:24);
 *     the message of the last non-OK result is read with fluhip_last_error().
 *   - "host" pointers are ordinary CPU memory; "dev" pointers are HIP device memory on the
 *     context's device.  Nothing is retained after a call returns except inside handles.
 *   - all matrices are dense row-major with the layouts of the reference's FluidTensorViews:
 *       spectrogram / magnitude  T x F       (F = fft/2 + 1, T = (N + hop) / hop)
 *       W1 (bases)               K x F
 *       H1 (activations)         T x K
 *   - the library is re-entrant; a context may be used by one thread at a time (the reference
 *     runs one job per std::thread: clients/common/FluidNRTClientWrapper.hpp:1048).
 *   - there is NO CPU fallback: without a usable gfx950 device every compute entry point
 *     returns FLUHIP_ERROR.
 */
#ifndef FLUCOMA_HIP_H
#define FLUCOMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 6) = 4 + fluhip_last_error_is_out_of_memory, fluhip_clear_error (the host clients' batched -> channel-by-channel
 * fallback classifies by code: include/flucoma_hip/NMFClient.hpp needs them), fluhip_corpus_stft_mag_only,
 * fluhip_corpus_last_loop_ms, fluhip_debug_plan_shape, fluhip_debug_wnorm_form.  Nothing of version 4 changed meaning. */
#define FLUHIP_ABI_VERSION 5

/* clients/common/Result.hpp:24  enum class Status { kOk, kWarning, kError, kCancelled } */
enum fluhip_status
{
  FLUHIP_OK = 0,
  FLUHIP_WARNING = 1,
  FLUHIP_ERROR = 2,
  FLUHIP_CANCELLED = 3
};

/* algorithms/public/WindowFuncs.hpp:26-32  enum class WindowTypes */
enum fluhip_window
{
  FLUHIP_WINDOW_HANN = 0,
  FLUHIP_WINDOW_HANND = 1,
  FLUHIP_WINDOW_HAMMING = 2,
  FLUHIP_WINDOW_BLACKMANHARRIS = 3,
  FLUHIP_WINDOW_GAUSSIAN = 4
};

typedef struct fluhip_ctx    fluhip_ctx;    /* one device + one HIP stream + scratch arena */
typedef struct fluhip_corpus fluhip_corpus; /* a device-resident batch of equal-shape buffers */

/* algorithms/public/NMF.hpp:31 ProgressCallback = std::function<bool(index)>; invoked on the
 * calling thread with iteration = 1..iters in order; returning 0 cancels (NMF.hpp:175-176). */
typedef int (*fluhip_progress_fn)(int64_t iteration, void* user);

/* ---- library / context ------------------------------------------------------------- */
int         fluhip_abi_version(void);
int         fluhip_device_count(void);                    /* 0 when no HIP device is visible */
int         fluhip_ctx_create(int device, fluhip_ctx** out);
/* Teardown order: destroy every corpus created on a context BEFORE the context (a corpus keeps using its context's
 * stream until fluhip_corpus_destroy returns). */
void        fluhip_ctx_destroy(fluhip_ctx* ctx);
const char* fluhip_last_error(const fluhip_ctx* ctx);     /* never NULL */
/* 1 when the last non-OK result of this context was an allocation the device (hipErrorOutOfMemory, the FFT workspace) or
 * the host (std::bad_alloc inside the library) could not serve -- the one failure a caller may answer by retrying with a
 * smaller job (the clients' batched -> channel-by-channel fallback, NMFClient.hpp).  Classified by error CODE at the point
 * of failure, never by message text.  fluhip_clear_error empties message and flag (call it in front of a block of calls
 * whose failure you are going to classify, so that nothing stale is read). */
int         fluhip_last_error_is_out_of_memory(const fluhip_ctx* ctx);
void        fluhip_clear_error(fluhip_ctx* ctx);
/* device name / gcnArchName into caller buffers (for bench reports) */
int         fluhip_ctx_device_info(const fluhip_ctx* ctx, char* name, int name_len, char* arch,
                                   int arch_len, int* compute_units);
/* raw HIP stream (hipStream_t) the context launches on -- for event timing by the caller */
void*       fluhip_ctx_stream(const fluhip_ctx* ctx);
/* Device buffers are recycled through a per-device cache of freed blocks (at most 8 GiB per device; a context's
 * destruction empties its device's cache).  fluhip_ctx_trim hands the cached blocks of the context's device back to
 * the driver now -- for hosts that share the GPU with other allocators (torch, RCCL). */
int         fluhip_ctx_trim(fluhip_ctx* ctx);
int         fluhip_ctx_synchronize(fluhip_ctx* ctx);
/* How far the device may run ahead of the last iteration reported to a progress callback (default 8: at most 7 iterations
 * are enqueued beyond the one a callback refuses).  lag = 1 is the reference's exact behaviour -- NMF.hpp:175-176 stops AT the
 * iteration whose callback returns false, the factors are those of that iteration -- at the price of one host round trip per
 * iteration (~10 - 20 us: nothing for a corpus, a third of a single small buffer's iteration). */
int         fluhip_ctx_set_progress_lag(fluhip_ctx* ctx, int lag);

/* ---- parameter arithmetic (integer, bit-exact) --------------------------------------- */
/* clients/common/ParameterTypes.hpp:295-312 FFTParams::fftSize/hopSize/frameSize.
 * in: win, hop (<=0: win/2), fft (<0: nextPow2(win)).  Returns FLUHIP_ERROR when fft is not a
 * power of two >= win or win < 4 (the reference clamps these at set time: :371-393). */
int fluhip_fft_params(int64_t win, int64_t hop, int64_t fft, int64_t* win_out, int64_t* hop_out,
                      int64_t* fft_out, int64_t* bins_out);
/* algorithms/public/STFT.hpp:98-99, clients/nrt/NMFClient.hpp:111-112: (n + hop) / hop */
int64_t fluhip_stft_num_frames(int64_t n, int64_t win, int64_t hop);

/* ---- algorithm::STFT ------------------------------------------------------------------ */
/* Replaces STFT::STFT(win, fft, hop, windowType) + STFT::process(audio, spectrogram) +
 * STFT::magnitude(spectrogram, magnitude)  (algorithms/public/STFT.hpp:36-47, 90-108, 61-66).
 * audio: n host doubles with element stride `stride`.  spec (may be NULL): T*F interleaved
 * (re,im) doubles == std::complex<double>[T][F].  mag (may be NULL): T*F doubles. */
int fluhip_stft_f64(fluhip_ctx* ctx, const double* audio, int64_t n, int64_t stride, int64_t win,
                    int64_t fft, int64_t hop, int window_type, double* spec, double* mag,
                    int64_t* frames_out);
/* same with the float->double conversion of clients/nrt/NMFClient.hpp:240 folded in */
int fluhip_stft_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride, int64_t win,
                    int64_t fft, int64_t hop, int window_type, double* spec, double* mag,
                    int64_t* frames_out);

/* ---- algorithm::NMF ------------------------------------------------------------------- */
/* Replaces NMF::process(X, W1, H1, V1, rank, nIterations, updateW, updateH, randomSeed, W0, H0)
 * + NMF::addProgressCallback  (algorithms/public/NMF.hpp:91-139, 144-183).
 * X: T x F with row stride ldx (doubles).  W0: K x F or NULL ("0x0 view").  H0: T x K or NULL.
 * W1: K x F, H1: T x K, V1: T x F or NULL.  seed < 0 => std::random_device like the reference.
 * progress is called once per completed iteration, in order; the device is never more than 8 iterations ahead of the last
 * one reported.  On cancellation returns FLUHIP_CANCELLED; W1/H1 then hold the factors as they stand when the device has
 * stopped -- at most 7 iterations after the one the callback refused (the reference stops at it exactly, NMF.hpp:175-176;
 * its client discards the factors of a cancelled job, clients/nrt/NMFClient.hpp:273-274) -- and V1 is left unwritten.
 * fluhip_ctx_set_progress_lag(ctx, 1) makes the stop exact.
 * Numerics: FP64 throughout.  The quotients V / max(W H, eps) of the update loop are formed as V * (1 / d) with a Newton-
 * refined reciprocal, relative error <= 2^-46 (1.4e-14) per quotient instead of a correctly rounded division; measured,
 * the factors stay within 1e-13 of the restatement after 200 iterations (the tests assert 1e-9; north_star asks 1e-5).
 * Two quotients share one reciprocal 1 / (d d'): magnitudes whose products W H reach sqrt(DBL_MAX) ~ 1e154 overflow it
 * (the ratio becomes 0 there) -- far outside anything a spectrogram holds; build with -DFLUHIP_SHARED_RECIPROCAL=0
 * -DFLUHIP_QUOTIENT_CORRECTION=1 for division to the last bit over the whole double range. */
int fluhip_nmf_process_f64(fluhip_ctx* ctx, const double* X, int64_t T, int64_t F, int64_t ldx,
                           int64_t K, int64_t iters, int update_w, int update_h, int64_t seed,
                           const double* W0, const double* H0, double* W1, double* H1,
                           double* V1, fluhip_progress_fn progress, void* user);

/* The same with every matrix as a two-stride view, the shape FluidTensorView<double, 2> hands to asEigen
 * (data/FluidTensor_Support.hpp:260-420 strides, :386-393 transpose(); util/FluidEigenMappings.hpp:35-225 maps them as
 * Stride<Dynamic, Dynamic>): element (r, c) at data[r * row_stride + c * col_stride].  NULL, data == NULL or a zero
 * extent is the reference's "0x0 view" (no seed / output not wanted).  Shapes as above (X T x F, W0 / W1 K x F,
 * H0 / H1 T x K, V1 T x F).  Unit-column-stride views and their transposes (unit row stride) move as strided copies
 * with no host-side repacking of X or V1; views with two non-unit strides are packed on the host. */
typedef struct fluhip_matrix_view
{
  double* data;
  int64_t rows, cols, row_stride, col_stride;
} fluhip_matrix_view;
int fluhip_nmf_process_views_f64(fluhip_ctx* ctx, const fluhip_matrix_view* X, int64_t K, int64_t iters, int update_w,
                                 int update_h, int64_t seed, const fluhip_matrix_view* W0, const fluhip_matrix_view* H0,
                                 const fluhip_matrix_view* W1, const fluhip_matrix_view* H1, const fluhip_matrix_view* V1,
                                 fluhip_progress_fn progress, void* user);

/* Replaces NMF::processFrame(x, W0, out, nIterations, v, randomSeed, alloc) (algorithms/public/NMF.hpp:45-89)
 * applied to every row of X at once -- the per-frame activation solve that
 * clients/rt/NMFMatchClient.hpp:113-118 (10 iterations, no estimate) and clients/rt/NMFFilterClient.hpp:102-116
 * (kIterations, estimate kept for the ratio mask) run on each spectral frame.  Frames are independent, so the
 * whole matrix is one batch of the H update with the dictionary fixed.
 * X: T x F magnitudes with row stride ldx.  W0: K x F dictionary (read only here; the reference clamps and
 * row-normalises its argument in place -- pass a copy there, the result is the same).
 * H (may be NULL): T x K activations.  V (may be NULL): T x F estimates W^T h.
 * seed >= 0: every frame starts from the same K draws, like the reference's fresh generator per call;
 * seed < 0: std::random_device. */
int fluhip_nmf_process_frames_f64(fluhip_ctx* ctx, const double* X, int64_t T, int64_t F, int64_t ldx,
                                  const double* W0, int64_t K, int64_t iters, int64_t seed, double* H, double* V);

/* ---- algorithm::NNDSVD / client::nndsvd::NMFSeedClient ------------------------------------------ */
/* Replaces NNDSVD::process(X, W, H, minRank, maxRank, amount, method, seed) (algorithms/public/NNDSVD.hpp:30-132):
 * thin SVD of X^T (bins x frames), rank k = the smallest number of leading singular values covering `amount`
 * of their sum (clamped to [min_rank, max_rank]; amount == 0 => min_rank), then the non-negative factors
 * method 0 (NMF-SVD): W = |U_k|, H = |S_k V_k^T|;  1 (NNDSVDar) / 2 (NNDSVDa) / 3 (NNDSVD): positive/negative
 * split of every singular pair, zeros filled with small random values / the mean / left at zero.
 * X: T x F with row stride ldx.  W: w_rows x F, H: T x w_rows (the reference's W.rows(); rows / columns >= k are
 * zero, or filled like the rest for methods 1 and 2, exactly as the reference treats its zero-initialised
 * outputs); k is returned in *rank_out and must not exceed w_rows.
 * The SVD is a one-sided Jacobi iteration on the device (kernels_svd.hip).  A singular pair is only defined up to a
 * common sign:
 * method 0 does not depend on it, methods 1..3 do (in the reference as well -- they follow whatever Eigen's
 * BDCSVD returns), so for those only the construction from a given SVD is pinned by the tests. */
int fluhip_nndsvd_f64(fluhip_ctx* ctx, const double* X, int64_t T, int64_t F, int64_t ldx, int64_t w_rows,
                      int64_t min_rank, int64_t max_rank, double amount, int method, int64_t seed, double* W,
                      double* H, int64_t* rank_out);
/* Replaces NMFSeedClient::process (clients/nrt/NMFSeedClient.hpp:73-131; BufNMFSeed): STFT -> magnitude ->
 * NNDSVD -> bases (max_rank x F floats, rows >= rank untouched by the reference's resize-to-rank: here zero) and
 * activations (max_rank x T floats, scaled by 1 / max over the whole T x max_rank envelope matrix, :120-128). */
int fluhip_bufnmfseed_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride, int64_t win,
                          int64_t fft, int64_t hop, int64_t min_rank, int64_t max_rank, double coverage,
                          int method, int64_t seed, float* bases_out, float* acts_out, int64_t* rank_out);

/* ---- client::bufnmf::NMFClient::process, one channel ---------------------------------- */
/* Replaces the body of the channel loop, clients/nrt/NMFClient.hpp:240-300 (STFT -> magnitude
 * -> NMF -> float write-back with H/max(H)), without a host round trip in between.
 * audio: n host floats, element stride `stride`.
 * bases_seed: K x F floats or NULL (basesMode Seed/Fixed);  acts_seed: K x T floats or NULL
 * (channel-major like BufferAdaptor::samps(channel)).
 * bases_out: K x F floats or NULL;  acts_out: K x T floats or NULL (already scaled by
 * float(1/max H) in float arithmetic, :297-298);  resynth_out: K x n floats or NULL
 * (resynthMode 1, :302-334). */
int fluhip_bufnmf_channel_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride,
                              int64_t win, int64_t fft, int64_t hop, int64_t K, int64_t iters,
                              int update_w, int update_h, int64_t seed, const float* bases_seed,
                              const float* acts_seed, float* bases_out, float* acts_out,
                              float* resynth_out, fluhip_progress_fn progress, void* user);

/* ---- client::bufstft::BufferSTFTClient ----------------------------------------------------------- */
/* Replaces processFwd, clients/nrt/BufSTFTClient.hpp:81-184: padding_mode None/Default/Full = 0/1/2
 * (padding = 0, win/2, win-hop: clients/common/ParameterTypes.hpp:315-323), hops = 1 + (padded - win)/hop,
 * frame i = padded[i*hop, i*hop + win).  mag / phase (either may be NULL): bins x hops floats, one bin per
 * buffer channel like mags.allFrames().transpose() <<= tmpMags (:168-178). */
int fluhip_bufstft_forward_f32(fluhip_ctx* ctx, const float* audio, int64_t n, int64_t stride, int64_t win,
                               int64_t fft, int64_t hop, int padding_mode, float* mag, float* phase,
                               int64_t* hops_out);
/* Replaces processInverse, :186-276: std::polar(mag, phase) -> ISTFT::processFrame -> overlap-add /
 * window^2 normaliser -> drop `padding` leading samples.  out: (hops-1)*hop + win - padding floats
 * (call with out == NULL to query *n_out). */
int fluhip_bufstft_inverse_f32(fluhip_ctx* ctx, const float* mag, const float* phase, int64_t hops, int64_t win,
                               int64_t fft, int64_t hop, int padding_mode, float* out, int64_t* n_out);

/* ---- feature pipeline: BufMelBands / BufMFCC (BASELINE config 5) ------------------------------ */
/* Replaces, for `count` equal-length mono buffers at once, the offline-wrapped real-time clients
 *   NRTThreadedMelBandsClient  clients/rt/MelBandsClient.hpp:77-119  (MelBands::processFrame, alg/MelBands.hpp:79-97)
 *   NRTThreadedMFCCClient      clients/rt/MFCCClient.hpp:86-131      (+ DCT::processFrame, alg/DCT.hpp:65-75)
 * as driven by StreamingControl with the default padding (clients/common/FluidNRTClientWrapper.hpp:551-660):
 * T = 1 + (n + 2 (win/2))/hop - win/hop frames, frame k starting at sample (win/hop)*hop - win - win/2 + k*hop
 * (for hop | win: [k*hop - win/2, k*hop + win/2), T = n/hop + 1).
 * audio: count x n floats.  out: count x nFeatures x T floats, feature-major per buffer like
 * BufferAdaptor::samps(feature).  Either may be a host or a device pointer (a corpus that already sits in HBM
 * skips the PCIe copies, which otherwise dominate: config 5 moves 2.9 GB in).  window: Hann (the clients pass no
 * window type). */
int fluhip_bufmelbands_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win,
                           int64_t fft, int64_t hop, int64_t n_bands, double min_freq, double max_freq,
                           double sample_rate, int normalize, int scale_db, float* out, int64_t* frames_out);
int fluhip_bufmfcc_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                       int64_t hop, int64_t n_bands, int64_t n_coefs, int64_t start_coeff, double min_freq,
                       double max_freq, double sample_rate, float* out, int64_t* frames_out);
/* The same with the wrapper's "padding" parameter (None / Default / Full = 0 / 1 / 2; the two calls above are mode 1):
 * userPadding = FFTParams::padding = 0 / win/2 / win - hop (clients/common/ParameterTypes.hpp:315-323,
 * FluidNRTClientWrapper.hpp:357-364), paddedLength = n + win + 2 userPadding, rounded up to whole hops in Full mode
 * (:572-574), T = 1 + (paddedLength - win)/hop - win/hop, kept frame k starting at sample
 * (win/hop) hop - win - userPadding + k hop. */
int fluhip_bufmelbands_padded_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win,
                                  int64_t fft, int64_t hop, int64_t n_bands, double min_freq, double max_freq,
                                  double sample_rate, int normalize, int scale_db, int padding_mode, float* out,
                                  int64_t* frames_out);
int fluhip_bufmfcc_padded_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                              int64_t hop, int64_t n_bands, int64_t n_coefs, int64_t start_coeff, double min_freq,
                              double max_freq, double sample_rate, int padding_mode, float* out, int64_t* frames_out);

/* ---- the users of NMF::processFrame: NMFMatch and NMFFilter ------------------------------------------------------------
 * clients/rt/NMFMatchClient.hpp:76-118 and clients/rt/NMFFilterClient.hpp:69-118 are real-time clients: a host vector in, a
 * host vector out, one NMF::processFrame per hop against the dictionary in the `bases` buffer.  These two calls are those
 * clients over a whole buffer, as the reference's offline wrapper templates drive a real-time client
 * (clients/common/FluidNRTClientWrapper.hpp: StreamingControl :551-660, Streaming :466-547) -- every frame of every channel
 * in one batch.  audio: count x n host floats (channels of one job; the client is reset per channel, :602 / :515).
 * bases: K x F host floats, F = fft/2 + 1 (the filter buffer's channels, K = min(channels, maxComponents)).
 *
 * fluhip_nmfmatch_f32 -- the control output of NMFMatch per hop: out[channel][component][T], T as for the feature clients
 * (fluhip_bufmfcc_padded_f32; *frames_out, also with out == NULL as a size query).  The client writes its output BEFORE
 * it processes the call's frame (:104 against :108-117), so column k holds the activations of the frame at audio sample
 * (k + win/hop - 1) hop - win - userPadding: one hop behind the frame BufMFCC analyses for that column (zeros where no
 * frame precedes it).  processFrame runs TEN iterations there whatever the client's `iterations` parameter says (:113-116);
 * seed >= 0: every frame starts from the same K draws (a fresh generator of the seed per call), seed < 0: random_device.
 *
 * fluhip_nmffilter_f32 -- the audio outputs of NMFFilter: out[channel][component][n].  Per frame: processFrame (`iters`
 * iterations), the estimate W^T h as the ratio mask's denominator, component i's rank-one estimate through the mask
 * (exponent 1), inverse transform, window, overlap-add, division by the overlap-added squared window; the ring buffers'
 * delay of one window is removed as the wrapper removes it (frame m = 1, 2, ... covers samples [m hop - win, m hop)).
 * hop <= win. */
int fluhip_nmfmatch_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft, int64_t hop,
                        const float* bases, int64_t K, int64_t seed, int padding_mode, float* out, int64_t* frames_out);
int fluhip_nmffilter_f32(fluhip_ctx* ctx, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft, int64_t hop,
                         const float* bases, int64_t K, int64_t iters, int64_t seed, float* out);

/* ---- corpus: many independent equal-shape buffers, resident in HBM --------------------- */
/* The data-parallel form of the same path (BASELINE config 4): `count` mono buffers of n
 * samples each; every buffer is an independent BufNMF job (clients/nrt/NMFClient.hpp:233 loop
 * body; no cross-buffer state).  Lifetime: create -> set_audio -> stft -> nmf -> read-back. */
int  fluhip_corpus_create(fluhip_ctx* ctx, int64_t count, int64_t n, int64_t win, int64_t fft,
                          int64_t hop, int64_t K, fluhip_corpus** out);
void fluhip_corpus_destroy(fluhip_corpus* c);
int64_t fluhip_corpus_frames(const fluhip_corpus* c);   /* T */
int64_t fluhip_corpus_bins(const fluhip_corpus* c);     /* F */
int64_t fluhip_corpus_device_bytes(const fluhip_corpus* c);
/* audio: count x n floats, host (synchronous copy) or device pointer */
int fluhip_corpus_set_audio_host(fluhip_corpus* c, const float* audio);
int fluhip_corpus_set_audio_dev(fluhip_corpus* c, const float* audio_dev);
/* K1: batched window + real FFT + magnitude for every frame of every buffer */
int fluhip_corpus_stft(fluhip_corpus* c);
/* The same transform with the frame-major magnitudes ALONE -- STFT::process + STFT::magnitude of every buffer (alg/STFT.hpp:
 * 90-108, 61-66), no bin-major copy for the H update: what the spectrogram-only callers (fluhip_stft_*, BufSTFT) run, as a
 * corpus-sized batch.  fluhip_corpus_nmf then needs fluhip_corpus_stft again. */
int fluhip_corpus_stft_mag_only(fluhip_corpus* c);
/* NMF on every buffer; seeds: `count` seeds or NULL (then `seed` for all, like one ParameterSet
 * shared by all jobs).  Asynchronous on the context stream unless a progress callback is
 * given (progress is then reported per iteration across the whole batch). */
int fluhip_corpus_nmf(fluhip_corpus* c, int64_t iters, int update_w, int update_h, int64_t seed,
                      const int64_t* seeds, fluhip_progress_fn progress, void* user);
/* Seed / Fixed factors of the batched form: basesMode / actMode of clients/nrt/NMFClient.hpp:246-258 as they reach
 * NMF::process (alg/NMF.hpp:102-124: a given W0 / H0 replaces the random draw; clamping and normalisation :150-153 run
 * either way).  bases_seed: count x K x F floats or NULL; acts_seed: count x K x T floats or NULL -- channel-major per
 * buffer like BufferAdaptor::samps(component), the layout of fluhip_bufnmf_channel_f32's seeds.  The arrays are copied;
 * they apply to every later fluhip_corpus_nmf of this corpus until replaced (NULL: back to random draws from the seed).
 * Seed mode = seeds + update flag 1, Fixed mode = seeds + update flag 0 (fluhip_corpus_nmf's update_w / update_h). */
int fluhip_corpus_set_factors(fluhip_corpus* c, const float* bases_seed, const float* acts_seed);
/* RAGGED corpus: `count` mono buffers of DIFFERENT lengths n[i] (a folder of sound files) as one device-resident batch --
 * one STFT launch and one set of factor-update launches per iteration over all of them, the work dealt per wavefront
 * by each buffer's own frame count T_i = (n[i] + hop) / hop (long contractions are split, the strips of the H update
 * follow the buffer's frames).  Every buffer is the same independent BufNMF job as in an equal-length corpus
 * (clients/nrt/NMFClient.hpp:233 loop body).  Supported: ranks up to 128, fft 1024 / 2048 / 4096 with an even window
 * (FLUHIP_ERROR with a message otherwise: run such buffers as equal-length groups).  The other corpus entry points work
 * on a ragged corpus with T (n) = the longest buffer's frame (sample) count as the stride of every per-buffer array:
 * fluhip_corpus_stft, fluhip_corpus_nmf (seed / seeds), fluhip_corpus_set_factors (acts_seed: count x K x T, the entries
 * past a buffer's own frames are ignored), fluhip_corpus_read_f64 (frames past a buffer's own are zero),
 * fluhip_corpus_keep_spectrum + fluhip_corpus_resynth_dev (count x K x n, samples past a buffer's own untouched),
 * fluhip_corpus_plan. */
int fluhip_corpus_create_ragged(fluhip_ctx* ctx, int64_t count, const int64_t* n, int64_t win, int64_t fft, int64_t hop,
                                int64_t K, fluhip_corpus** out);
int64_t fluhip_corpus_frames_of(const fluhip_corpus* c, int64_t i);   /* T_i (T for an equal-length corpus) */
/* audio[i]: n[i] host floats */
int fluhip_corpus_set_audio_ragged_host(fluhip_corpus* c, const float* const* audio);
/* bases[i]: K x F floats, acts[i]: K x T_i floats (either array, or single entries, may be NULL) */
int fluhip_corpus_writeback_ragged_host(fluhip_corpus* c, float* const* bases, float* const* acts);
/* out[i]: K x n[i] floats (entries may be NULL): the resynthesised components of every buffer (see fluhip_corpus_resynth_dev) */
int fluhip_corpus_resynth_ragged_host(fluhip_corpus* c, float* const* out);
/* write-back (clients/nrt/NMFClient.hpp:277-300) into device or host float arrays:
 * bases: count x K x F, acts: count x K x T.  Either may be NULL. */
int fluhip_corpus_writeback_dev(fluhip_corpus* c, float* bases_dev, float* acts_dev);
int fluhip_corpus_writeback_host(fluhip_corpus* c, float* bases, float* acts);
/* Resynthesis of every component of every buffer (clients/nrt/NMFClient.hpp:302-334: NMF::estimate -> RatioMask ->
 * ISTFT): out = count x K x n floats.  The complex spectrogram has to be kept for it: switch that on before
 * fluhip_corpus_stft (it costs count x T x F x 16 bytes of HBM). */
int fluhip_corpus_keep_spectrum(fluhip_corpus* c, int on);
int fluhip_corpus_resynth_dev(fluhip_corpus* c, float* out_dev);
int fluhip_corpus_resynth_host(fluhip_corpus* c, float* out);
/* The resynthesis written the way an interleaved host buffer holds it (frames x channels, like MemoryBufferAdaptor,
 * clients/common/MemoryBufferAdaptor.hpp:96-100): out[t * frame_stride + b * K + k] = component k of buffer b at sample t,
 * i.e. what resynth.samps(b * rank + k) <<= ... leaves there (clients/nrt/NMFClient.hpp:321-326), transposed on the device
 * and streamed to the host through pinned staging blocks.  frame_stride >= count x K floats; equal-length corpora only. */
int fluhip_corpus_resynth_interleaved_host(fluhip_corpus* c, float* out, int64_t frame_stride);
/* raw f64 results for parity tests: mag count x T x F, W1 count x K x F, H1 count x T x K */
int fluhip_corpus_read_f64(fluhip_corpus* c, double* mag, double* W1, double* H1);
/* How the factor updates of this corpus are scheduled on the device (introspection for tests, benchmarks and
 * bug reports; no effect on results beyond summation order): out8 = { kernel form (5 = 4x4x4 MFMA + LDS-DMA, ranks up to
 * 128; 0 = the un-fused any-rank path above that), contraction splits of the W update, of the H update (low 16 bits;
 * bits 16.. = the pieces of the tail launch when the H update goes out as two launches, 0 otherwise),
 * deferred column normalisation of W (alg/NMF.hpp:162 applied on load) 0/1, Nyquist bin as a side column 0/1,
 * wavefronts per buffer of the W update, padded rank (low 16 bits: the rank the arrays are laid out for -- 16, 32, 64, 128;
 * bits 16..: the rank the factor updates COMPUTE -- 24 for ranks 17 .. 24; 40 / 48 / 56 for 33 .. 40 / 48 / 56; 72, 80 .. 112 for 65 .. 72, .. 80, .. 112; else the padded rank),
 * frame-strip schedule 0/1 (a single buffer of rank <= 16: the H
 * update local to a strip of frames, the W update's numerator as per-workgroup partials + a reduce launch; the split
 * counts before it then describe the schedule it replaces) }. */
int fluhip_corpus_plan(const fluhip_corpus* c, int64_t* out8);

/* Introspection for the tests (pure host code, no device needed): the work lists the planner builds for `count` buffers of
 * frames[i] frames and `bins` bins at rank K -- which = 0 the W update's, 1 the H update's.  desc (may be NULL) receives up to
 * `cap` descriptors of 12 ints {buffer, first column group, groups, first step, end step, partial slot, statistics slot,
 * denominator slot, group word (leader | rank << 4 | size << 8 | barrier << 16), 0, 0, 0}, four per workgroup; info8 =
 * {workgroups, widest strip, partials in memory 0/1, most partials per buffer, most pieces per contraction, partial slots,
 * Nyquist side column 0/1, statistics parts per buffer}.  Returns the number of descriptors (-1: bad arguments). */
int64_t fluhip_debug_plan_lists(int64_t count, const int64_t* frames, int64_t bins, int64_t K, int which, int32_t* desc,
                                int64_t cap, int32_t* info8);

/* Likewise for the two-launch form of the H update (api_corpus.hip plan_tail): `count` equal-length buffers of `frames` frames and
 * `bins` bins at rank K.  out4 = {pieces of the tail launch's contraction (0: the update stays one launch), strips per buffer
 * of the first launch, strips per buffer of the tail launch, frames per buffer in the first launch}. */
int fluhip_debug_plan_tail(int64_t count, int64_t frames, int64_t bins, int64_t K, int64_t* out4);
/* Which schedule family an equal-length corpus of that shape gets when nothing else decides first (a single buffer of rank <= 16
 * that fits the frame-strip schedule never asks): 1 = the work lists, 0 = the uniform schedule; -1: bad arguments.  The rules are
 * measured ones (api_corpus.hip list_plan_pays, profiles/r03/plan_regimes.txt); the CPU tests pin them for the BASELINE shapes. */
int fluhip_debug_plan_kind(int64_t count, int64_t frames, int64_t bins, int64_t K);
/* What the H update of an equal-length corpus of that shape takes over from the launches that used to run between the two
 * factor updates (kernels_nmf5.hip SIDEQ forms; no device needed): bit 0 = the next W update's side column (its last bin)
 * comes out of this launch's epilogue, bit 1 = the norm combine of the W update in front is done in its prologue; 0 = neither
 * (rank above 64, work lists, two-launch H update, no side column at this bin count); -1: bad arguments. */
int fluhip_debug_plan_h_update(int64_t count, int64_t frames, int64_t bins, int64_t K);
/* The WHOLE schedule of an equal-length corpus of that shape as data, without a device (api_corpus.hip decide_update_plan: the
 * one function fluhip_corpus_create plans from).  out32 =
 *   0 kernel family (5 / 0)   1 pieces of the W update's contraction   2 of the H update's   3 deferred normalisation
 *   4 Nyquist bin as a side column   5 statistics records per buffer of a W update   6 padded rank   7 computed rank
 *   8 frame-strip schedule (0 no, 1 fused, 2 / 3 the A/B forms)   9 work lists   10 pieces of the tail launch of a two-launch H
 *   update (0: one launch)   11 strips of its first launch   12 of its tail launch   13 frames in the first launch
 *   14 wavefronts per buffer of a uniform H update
 *   15 .. 21 workspaces in doubles, as allocated: split partials, denominators, column-sum pre-pass, norm / side-column scratch,
 *            column partials of the W update, frame-strip partials, any-rank scratch
 *   22 what the H update takes over in the steady state (bit 0 side column, bit 1 norm combine, bit 2 column sums from the W update)
 *   23 form of the norm-combine launch between the updates (fluhip_debug_wnorm_form's codes; -1: no such launch)
 *   24 slices of the side-column launch (0: none)   25 side-partial slots per buffer and generation   26 doubles of wscratch the
 *      statistics records occupy.
 * tests/test_plan_table.py holds this to its invariants over a grid of shapes and to the pinned plans of the BASELINE shapes. */
int fluhip_debug_plan_shape(int64_t count, int64_t frames, int64_t bins, int64_t K, int64_t* out32);
/* the form the norm combine of a W update takes (kernels_nmf.hip kWnormForms, the one table of its thresholds): 0 side column +
 * combine in one launch, 1 side-column launch only, 2 pre-reduction + combine, 3 one workgroup of 1024 threads, 4 of 256 */
int fluhip_debug_wnorm_form(int Kp, int count, int parts, int slices, int side_rows, int side_phase, int want_colsum);

/* ---- device pool: one host process, several GPUs ------------------------------------------------------------ */
/* The reference runs one std::thread per job (clients/common/FluidNRTClientWrapper.hpp:1042-1048) and the buffers of
 * a corpus are independent jobs (clients/nrt/NMFClient.hpp:233 loop body): a pool holds one context per listed device
 * and, per call, one host thread per device; buffers are dealt in contiguous blocks (fluhip_shard_range, remainder
 * over the first members) and every device writes its share of the result straight into the caller's arrays.
 * devices == NULL: every visible device.  A device may be listed more than once (two contexts on one GPU).
 * Multi-PROCESS jobs (one rank per GPU) use the same dealing and gather with RCCL: flucoma-core_amd/sharding.py. */
typedef struct fluhip_pool fluhip_pool;
int         fluhip_pool_create(const int* devices, int n_devices, fluhip_pool** out);
void        fluhip_pool_destroy(fluhip_pool* pool);
int         fluhip_pool_size(const fluhip_pool* pool);
int         fluhip_pool_device(const fluhip_pool* pool, int member);
const char* fluhip_pool_last_error(const fluhip_pool* pool);
/* BufNMF over `count` equal-length mono buffers (BASELINE config 4): audio count x n host floats; bases count x K x F,
 * acts count x K x T host floats (either may be NULL); seeds: `count` seeds or NULL.  progress (may be NULL) is called
 * on the calling thread with iteration = 1..iters in order, an iteration counting once every device has passed it;
 * returning 0 cancels every device's share (FLUHIP_CANCELLED). */
int fluhip_pool_bufnmf_f32(fluhip_pool* pool, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                           int64_t hop, int64_t K, int64_t iters, int update_w, int update_h, int64_t seed,
                           const int64_t* seeds, float* bases, float* acts, fluhip_progress_fn progress, void* user);
/* The whole BufNMF parameter set of the batched form in one description: Seed / Fixed factors (basesMode / actMode,
 * clients/nrt/NMFClient.hpp:63-67, 246-258) and the resynthesis output (resynthMode 1, :302-334) on top of what
 * fluhip_pool_bufnmf_f32 takes.  Zero-initialise, then fill in what the job uses. */
typedef struct fluhip_bufnmf_job
{
  int64_t count, n;                  /* equal-length mono buffers, samples each */
  int64_t win, fft, hop, K, iters;
  int     update_w, update_h;        /* 0 together with the matching seed array = Fixed mode */
  int64_t seed;                      /* randomSeed for every buffer ... */
  const int64_t* seeds;              /* ... or `count` seeds (NULL: `seed`) */
  const float* audio;                /* count x n */
  const float* bases_seed;           /* count x K x F or NULL */
  const float* acts_seed;            /* count x K x T or NULL */
  float* bases;                      /* count x K x F or NULL */
  float* acts;                       /* count x K x T or NULL */
  float* resynth;                    /* count x K x n or NULL */
} fluhip_bufnmf_job;
int fluhip_pool_bufnmf_job_f32(fluhip_pool* pool, const fluhip_bufnmf_job* job, fluhip_progress_fn progress, void* user);
/* The same over buffers of DIFFERENT lengths (a folder of sound files): audio[i] = n[i] host floats; bases[i] receives
 * K x F, acts[i] K x T_i floats with T_i = fluhip_stft_num_frames(n[i], win, hop) (either array, or single entries, may be
 * NULL).  Buffers are dealt by fluhip_balanced_assignment over their frame counts; a device's share runs as ONE ragged
 * corpus (all its buffers advance together, iteration by iteration); shapes the ragged form does not cover are
 * processed run by run of equal length.  progress (may be NULL) is called on the calling thread with the number of
 * buffers finished so far, ascending up to count -- in the ragged form a device's whole share finishes at once, so the
 * count moves in steps of a share, not buffer by buffer; returning 0 stops every device at its next iteration
 * (FLUHIP_CANCELLED; the outputs of unfinished buffers are not written).  Without a callback the iterations are
 * enqueued without the per-iteration host round trip a cancellable job needs. */
int fluhip_pool_bufnmf_ragged_f32(fluhip_pool* pool, const float* const* audio, const int64_t* n, int64_t count, int64_t win,
                                  int64_t fft, int64_t hop, int64_t K, int64_t iters, int update_w, int update_h, int64_t seed,
                                  const int64_t* seeds, float* const* bases, float* const* acts, fluhip_progress_fn progress,
                                  void* user);
/* The feature pipeline over the pool (BASELINE config 5, "1 -> 8 GPU scaling"): `count` equal-length slices, dealt in
 * contiguous blocks; audio: count x n host floats, out: count x nFeatures x T host floats (see fluhip_bufmfcc_padded_f32 /
 * fluhip_bufmelbands_padded_f32 for the arguments).  The slices are independent analyses: no exchange between devices. */
int fluhip_pool_bufmfcc_f32(fluhip_pool* pool, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                            int64_t hop, int64_t n_bands, int64_t n_coefs, int64_t start_coeff, double min_freq,
                            double max_freq, double sample_rate, int padding_mode, float* out, int64_t* frames_out);
int fluhip_pool_bufmelbands_f32(fluhip_pool* pool, const float* audio, int64_t count, int64_t n, int64_t win, int64_t fft,
                                int64_t hop, int64_t n_bands, double min_freq, double max_freq, double sample_rate,
                                int normalize, int scale_db, int padding_mode, float* out, int64_t* frames_out);

/* the dealing itself (integer arithmetic, also used by the multi-process launcher): contiguous blocks [begin, end) of
 * n_items over `world` ranks; and the greedy longest-processing-time deal for ragged corpora (cost ~ T F K per buffer,
 * ties to the lower index / rank): rank_of_item[i] = rank of item i */
void fluhip_shard_range(int64_t n_items, int world, int rank, int64_t* begin, int64_t* end);
int  fluhip_balanced_assignment(const double* costs, int64_t n, int world, int32_t* rank_of_item);

/* ---- live kernel timing (HIP events on the context stream) ----------------------------- */
/* When enabled, every launch of the two dominant kernel classes is bracketed by hipEvents on
 * the stream it is launched on.  Classes: 0 = the STFT kernel (both magnitude layouts), 1 = nmf_update (both factor
 * updates share one kernel; the split-contraction finalize counts with it), 2 = feature kernels, 3 = the small kernels
 * between the factor updates, 4 = the transposing copy of shapes the block STFT kernel does not cover.  fluhip_prof_read synchronises and returns the launch count and the summed duration
 * since the last reset. */
int fluhip_prof_enable(fluhip_ctx* ctx, int on);
int fluhip_prof_reset(fluhip_ctx* ctx);
int fluhip_prof_read(fluhip_ctx* ctx, int kernel_class, int64_t* launches, double* total_ms);
/* Box-invariant cost of the factor-update launches of a corpus (kernels_nmf5.hip): one wavefront per launch -- it lives as
 * long as the launch does -- adds its shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) to a device record.
 * out8 = { W update: launches, shader cycles, 100 MHz ticks, 0;  H update: the same four }.  cycles / launches is what a
 * kernel change moves whatever clock the box sustains; cycles / ticks * 100 MHz is that sustained clock.  Zero counts when
 * the corpus runs a schedule without the stamps (frame-strip schedule, A/B kernel forms).  reset != 0 clears the record. */
int fluhip_corpus_update_clocks(fluhip_corpus* c, int64_t* out8, int reset);
/* Device time of the last fluhip_corpus_nmf iteration loop of this corpus: two HIP events on the context stream, the first
 * recorded BEHIND the host-side initialisation (random draws, uploads), the second behind the loop's last launch (the
 * deferred-normalisation apply excluded).  Synchronises on the second event.  What tools/perf_matrix.py divides by the
 * iteration count: no host scheduling, no initialisation inside the figure. */
int fluhip_corpus_last_loop_ms(fluhip_corpus* c, double* ms);
/* Kernel-developer diagnostic: with FLUHIP_K5_INSTR=1 in the environment the factor-update kernel of the
 * c4-shaped schedules runs an instrumented build that leaves cycle counters and a timeline of one wavefront in the
 * corpus' scratch; this copies the first 32 words out (tools/phase_breakdown.py decodes them).  Without the
 * environment variable the words are whatever the scratch holds. */
int fluhip_corpus_debug_words(fluhip_corpus* c, int64_t* out32);

#ifdef __cplusplus
}
#endif
#endif /* FLUCOMA_HIP_H */
