/*
 * fluid_oracle.h -- CPU restatement ("oracle") of flucoma-core's BufNMF hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call it, and
 * there only as the checker / the timed CPU baseline, never as the thing shipped.
 *
 * PARITY, two halves.
 * PINNED to outputs of the reference itself: the STFT -> magnitude -> mel -> DCT chain (fo_stft's framing, window and
 * transform as the buffered clients drive them, fo_bufmfcc_channel).  flucoma-core ships the analysis of its demo corpus by
 * a FluCoMa build (Resources/Data/flucoma_corpus_mfcc.json: mean and deviation of BufMFCC's coefficients 1..13 per slice);
 * the 299 slices whose audio is in the checkout come out of this file to the float32 the JSON stores
 * (tests/test_oracle.py::test_oracles_reproduce_the_references_pre_analysed_corpus; four of them are the fixture
 * tests/golden/reference_corpus_mfcc.npz).
 * UNPINNED: the NMF arithmetic.  The reference (flucoma-core, C++17 header-only) cannot be compiled in
 * this image -- every header on the path needs Eigen 3.4.0, HISSTools_Library@f3292ad and
 * foonathan/memory, all network FetchContent dependencies (reference CMakeLists.txt:54-123)
 * that are absent -- and its own test-suite holds no known-answer vectors for STFT spectra
 * or NMF factors (tests/algorithms/public/TestNMF.cpp:11-46 only checks same-seed
 * repeatability).  What *is* pinned: the RNG stream (libstdc++ <random>, checked in
 * tests/test_oracle.py against std::mt19937_64 + uniform_real_distribution compiled here),
 * the Hann formula (tests/clients/common/TestBufferedProcess.cpp:42-44), frame counts
 * (include/flucoma/clients/nrt/NMFClient.hpp:111-112), and agreement to <=1e-12 with an
 * independent numpy restatement (oracle/oracle_np.py) whose outputs are committed as
 * tests/golden fixtures.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/include/flucoma/).
 */
#ifndef FLUID_ORACLE_H
#define FLUID_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* util/AlgorithmUtils.hpp:19  epsilon = std::numeric_limits<double>::epsilon() */
#define FO_EPSILON 2.220446049250313e-16

/* alg/WindowFuncs.hpp:41-45  (kHann, periodic): w[i] = 0.5 - 0.5*cos(2*pi*i/size) */
void fo_window_hann(int64_t win, double* w);

/* alg/STFT.hpp:98-99 and nrt/NMFClient.hpp:111-112: T = (N + hop) / hop (integer division) */
int64_t fo_stft_num_frames(int64_t n, int64_t win, int64_t hop);

/* alg/STFT.hpp:90-108 + util/FFT.hpp:92-108 + alg/STFT.hpp:61-66.
 * audio: n doubles.  spec (may be NULL): T*F interleaved (re,im).  mag (may be NULL): T*F.
 * Row-major T x F, F = fft/2+1.  Returns T. */
int64_t fo_stft(const double* audio, int64_t n, int64_t win, int64_t fft, int64_t hop,
                double* spec, double* mag);

/* float -> double converting copy then fo_stft (nrt/NMFClient.hpp:240-242); stride in floats */
int64_t fo_stft_f32(const float* audio, int64_t n, int64_t stride, int64_t win, int64_t fft,
                    int64_t hop, double* spec, double* mag);

/* util/EigenRandom.hpp:73-101 with libstdc++'s uniform_real_distribution<double>(0,1) over
 * std::mt19937_64{seed}: out[i] = double(g())/2^64 (clamped below 1). */
void fo_rng_uniform01(uint64_t seed, int64_t count, double* out);

/* `reps` rounds of independent register FMAs (*flop_per_rep flop each): the caller times it -> the core's measured FMA peak */
double fo_fma_burst(int64_t reps, int64_t* flop_per_rep);

typedef int (*fo_progress_fn)(int64_t iteration, void* user); /* return 0 => cancel */

/* alg/NMF.hpp:91-134 (process) + :144-183 (multiplicativeUpdates).
 * X: T x F row-major (ldx = F).  W0: K x F or NULL.  H0: T x K or NULL.
 * Outputs W1: K x F, H1: T x K, V1: T x F (may be NULL).
 * faithful != 0 executes all seven GEMMs of alg/NMF.hpp:158-173 per iteration (incl. the two
 * "ones" GEMMs and the dead R = W*H) -- this is what the CPU baseline times; faithful == 0
 * replaces the "ones" GEMMs by sums and drops R (same maths, summation order in den differs).
 * Returns 0, or 1 if a progress callback cancelled (outputs are then still written from the
 * current W,H like the reference's early return leaves V untouched: V1 = X^T copy). */
int fo_nmf_process(const double* X, int64_t T, int64_t F, int64_t K, int64_t iters,
                   int updateW, int updateH, int64_t seed, const double* W0,
                   const double* H0, double* W1, double* H1, double* V1, int faithful,
                   fo_progress_fn progress, void* user);

/* nrt/NMFClient.hpp:277-300 write-back of one channel.
 * bases_out: K x F floats (channel-major: row k = bases channel k).
 * acts_out : K x T floats (row k = activations channel k): float(H1[t][k]) * float(1/max(H1)). */
void fo_bufnmf_writeback(const double* W1, const double* H1, int64_t T, int64_t F, int64_t K,
                         float* bases_out, float* acts_out);

/* One channel of nrt/NMFClient.hpp:233-300 end to end (random init, both factors updated).
 * Returns T.  mag_out (T*F) may be NULL. */
int64_t fo_bufnmf_channel(const float* audio, int64_t n, int64_t win, int64_t fft, int64_t hop,
                          int64_t K, int64_t iters, int64_t seed, int faithful,
                          float* bases_out, float* acts_out, double* mag_out);

/* ---- "next" rows: per-frame activations of a fixed dictionary ------------------------ */
/* alg/NMF.hpp:45-89 processFrame (what rt/NMFMatchClient.hpp:113-118 and rt/NMFFilterClient.hpp:102-116 run
 * on every spectral frame): h = uniform(0,1)^K from a fresh generator of `seed`, W = max(W0, eps) with every
 * row (component) divided by its L2 norm, v0 = max(x, eps); nIterations of
 *   v1 = max(W^T h, eps); h = h * (W (v0 / v1)) / max(W 1, eps).
 * x: F.  W0: K x F row-major (not modified here; the reference normalises its argument in place).
 * h_out: K.  v_out (may be NULL): F = W^T h. */
void fo_nmf_process_frame(const double* x, const double* W0, int64_t K, int64_t F, int64_t iters, int64_t seed,
                          double* h_out, double* v_out);

/* ---- "next" rows: resynthesis ------------------------------------------ */
/* alg/NMF.hpp:33-42 + alg/RatioMask.hpp:33-57 + alg/STFT.hpp:178-199 for component k.
 * spec: T*F interleaved complex; W1 KxF; H1 TxK; V1 TxF (= W*H estimate);
 * out: n doubles (already trimmed by win/2). */
void fo_resynth_component(const double* spec, const double* W1, const double* H1,
                          const double* V1, int64_t T, int64_t F, int64_t K, int64_t k,
                          int64_t win, int64_t fft, int64_t hop, int64_t n, double* out);

/* ---- "next" rows: MelBands / MFCC feature pipeline --------------------------- */
/* alg/MelBands.hpp:41-77 init(): triangular filters on linear-Hz FFT bins between mel-spaced
 * centres.  filt: nBands x nBins row-major. */
void fo_mel_filters(double lo, double hi, int64_t nBands, int64_t nBins, double sampleRate, double* filt);
/* alg/DCT.hpp:36-63 init(): orthonormal DCT-II table, nOut x nIn row-major. */
void fo_dct_table(int64_t nIn, int64_t nOut, double* table);
/* alg/MelBands.hpp:79-97 processFrame() over every row of mag (T x F, modified like the reference
 * modifies its input when magNorm is set): out T x nBands. */
void fo_melbands(const double* mag, int64_t T, int64_t F, const double* filt, int64_t nBands, int64_t win,
                 int magNorm, int usePower, int logOutput, double* out);
/* BufMFCC on one channel with the default padding mode (rt/MFCCClient.hpp:86-131 driven by
 * StreamingControl, cc/FluidNRTClientWrapper.hpp:551-660): kept frame k starts at audio sample
 * (win/hop)*hop - win - win/2 + k*hop; T = 1 + (n + 2 (win/2))/hop - win/hop.  For hop | win that is the
 * framing of fo_stft: [k*hop - win/2, k*hop + win/2), T = n/hop + 1.
 * out: nCoefs x T floats (channel-major like BufferAdaptor::samps(i)).  Returns T. */
int64_t fo_bufmfcc_channel(const float* audio, int64_t n, int64_t win, int64_t fft, int64_t hop, int64_t nBands,
                           int64_t nCoefs, int64_t startCoeff, double minFreq, double maxFreq,
                           double sampleRate, float* out);
/* BufMelBands on one channel (rt/MelBandsClient.hpp:77-119): out nBands x T floats. */
int64_t fo_bufmelbands_channel(const float* audio, int64_t n, int64_t win, int64_t fft, int64_t hop,
                               int64_t nBands, double minFreq, double maxFreq, double sampleRate,
                               int normalize, int scaleDb, float* out);

/* ---- "next" rows: BufSTFT ----------------------------------------------------- */
/* nrt/BufSTFTClient.hpp:81-184 processFwd: padding = {0, win/2, win-hop}[mode]
 * (cc/ParameterTypes.hpp:315-323), numHops = 1 + (paddedLength - win)/hop, frame i = padded[i*hop, +win);
 * magnitude (alg/STFT.hpp:61-66) and phase (alg/STFT.hpp:75-79: arg) as float, bin-major [F][numHops].
 * Returns numHops. */
int64_t fo_bufstft_forward(const float* audio, int64_t n, int64_t win, int64_t fft, int64_t hop, int padding_mode,
                           float* mag, float* phase);
int64_t fo_bufstft_num_hops(int64_t n, int64_t win, int64_t hop, int padding_mode);
/* nrt/BufSTFTClient.hpp:186-276 processInverse: std::polar(mag, phase) -> ISTFT::processFrame ->
 * overlap-add with window^2 normaliser -> drop `padding` leading samples.
 * mag/phase: [F][T] floats.  out: (T-1)*hop + win - padding doubles.  Returns that length. */
int64_t fo_bufstft_inverse(const float* mag, const float* phase, int64_t T, int64_t win, int64_t fft, int64_t hop,
                           int padding_mode, double* out);

#ifdef __cplusplus
}
#endif
#endif
