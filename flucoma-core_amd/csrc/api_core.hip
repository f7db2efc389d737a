// api_core.hip -- the context side of the C ABI of libflucoma_hip.so (include/flucoma_hip.h): contexts and their tables
// (windows, twiddles), the caching device-block pool, staged device -> host copies, the profiling aid.
//
// There is no CPU fallback anywhere behind the ABI: every compute path launches HIP kernels and fails with FLUHIP_ERROR
// when the device is unusable.
#include "api_internal.h"

// workspace for transforms whose frame does not fit the LDS; null (with the error set) when it cannot be had
double* big_fft_scratch(fluhip_ctx* ctx, int64_t win, int64_t fft, int64_t frames)
{
  if (!stft_needs_scratch(win, fft)) return nullptr;
  const size_t need = (size_t) big_fft_scratch_bytes(fft, frames, nullptr);
  if (need > ctx->bigFftBytes)
  {
    (void) hipStreamSynchronize(ctx->stream);
    if (ctx->bigFft) (void) hipFree(ctx->bigFft);
    ctx->bigFft = nullptr;
    ctx->bigFftBytes = 0;
    if (hipMalloc(&ctx->bigFft, need) != hipSuccess) { (void) hipGetLastError(); fail_oom(ctx, "out of device memory for the FFT workspace"); return nullptr; }
    ctx->bigFftBytes = need;
  }
  return static_cast<double*>(ctx->bigFft);
}

int fail(fluhip_ctx* ctx, const std::string& msg, int status)
{
  if (ctx) { ctx->err = msg; ctx->errOom = false; }
  return status;
}

int fail_oom(fluhip_ctx* ctx, const std::string& msg)
{
  if (ctx) { ctx->err = msg; ctx->errOom = true; }
  return FLUHIP_ERROR;
}

int fail_hip(fluhip_ctx* ctx, hipError_t e, const char* what)
{
  const std::string msg = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
  if (e == hipErrorOutOfMemory) { (void) hipGetLastError(); return fail_oom(ctx, msg); } // (the sticky-less error is consumed: a retry with less memory starts clean)
  return fail(ctx, msg);
}

// Large results to pageable host memory: a plain hipMemcpy stages them through the runtime's small pinned buffers (measured
// 4.7 - 5.3 GB/s: 85 - 95 ms for the 451 MB of an 8-channel x 32-component resynthesis).  Here: two pinned blocks of 8 MiB,
// the DMA of block i + 1 running while the host copies block i to its place.  `rows` rows of `width` bytes, source rows
// spitch and destination rows dpitch bytes apart (a contiguous copy: rows = 1).  Work queued on `s` before the call is
// complete when it returns.  Small copies take the plain path.
int copy_to_host(fluhip_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows,
                        hipStream_t s)
{
  constexpr size_t kStage = (size_t) 8 << 20;
  const size_t total = width * rows;
  static const int off = [] { const char* e = fluhip::ab_getenv("FLUHIP_PINNED_D2H"); return e && std::atoi(e) == 0 ? 1 : 0; }();
  if (off || total < 2 * kStage || width > kStage)
  {
    HIPCHK(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return FLUHIP_OK;
  }
  for (int i = 0; i < 2; i++)
  {
    if (!ctx->stage[i]) HIPCHK(ctx, hipHostMalloc(&ctx->stage[i], kStage, hipHostMallocDefault));
    if (!ctx->stageEv[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->stageEv[i], hipEventDisableTiming));
  }
  // units: whole rows per block when there are several rows, byte ranges of the one row otherwise
  const bool byRows = rows > 1;
  const size_t unit = byRows ? width : 1;
  const size_t unitsPerBlock = kStage / unit;
  const size_t units = byRows ? rows : width;
  auto issue = [&](size_t u0, int slot) -> hipError_t {
    const size_t nu = std::min(unitsPerBlock, units - u0);
    hipError_t e = byRows ? hipMemcpy2DAsync(ctx->stage[slot], width, static_cast<const char*>(src) + u0 * spitch, spitch, width, nu,
                                             hipMemcpyDeviceToHost, s)
                          : hipMemcpyAsync(ctx->stage[slot], static_cast<const char*>(src) + u0, nu, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    return hipEventRecord(ctx->stageEv[slot], s);
  };
  size_t u = 0;
  int slot = 0;
  HIPCHK(ctx, issue(0, 0));
  while (u < units)
  {
    const size_t nu = std::min(unitsPerBlock, units - u);
    const size_t next = u + nu;
    if (next < units) HIPCHK(ctx, issue(next, slot ^ 1));
    HIPCHK(ctx, hipEventSynchronize(ctx->stageEv[slot]));
    if (byRows)
      for (size_t r = 0; r < nu; r++)
        std::memcpy(static_cast<char*>(dst) + (u + r) * dpitch, static_cast<const char*>(ctx->stage[slot]) + r * width, width);
    else
      std::memcpy(static_cast<char*>(dst) + u, ctx->stage[slot], nu);
    u = next;
    slot ^= 1;
  }
  return FLUHIP_OK;
}


BlockPool g_pool;

hipEvent_t take_event(fluhip_ctx* ctx)
{
  if (!ctx->eventPool.empty())
  {
    hipEvent_t e = ctx->eventPool.back();
    ctx->eventPool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void) hipEventCreate(&e);
  return e;
}

// Strided host view -> contiguous device copy (clients/nrt/NMFClient.hpp:240 `tmp <<= samps(...)`).
// A contiguous source is one plain copy; a strided one (a channel of a frame-interleaved host buffer) is gathered
// on the host first -- a 2-D copy with element-sized rows would be issued row by row.
hipError_t upload_strided(void* dst, const void* src, size_t n, size_t stride, size_t esz, hipStream_t s)
{
  if (stride == 1)
  {
    hipError_t e = hipMemcpyAsync(dst, src, n * esz, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);
  }
  std::vector<char> tmp(n * esz);
  const char* p = static_cast<const char*>(src);
  if (esz == 4)
    for (size_t i = 0; i < n; i++) reinterpret_cast<float*>(tmp.data())[i] = reinterpret_cast<const float*>(p)[i * stride];
  else
    for (size_t i = 0; i < n; i++) reinterpret_cast<double*>(tmp.data())[i] = reinterpret_cast<const double*>(p)[i * stride];
  hipError_t e = hipMemcpyAsync(dst, tmp.data(), n * esz, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s); // tmp goes out of scope
}

// ---------------------------------------------------------------------------------------
// tables: window (alg/WindowFuncs.hpp:38-72) and FFT twiddles, computed on the host in f64
// ---------------------------------------------------------------------------------------
bool make_window(int type, int64_t size, std::vector<double>& out)
{
  const double pi = M_PI; // util/AlgorithmUtils.hpp:21
  out.resize((size_t) size);
  switch (type)
  {
  case FLUHIP_WINDOW_HANN: // alg/WindowFuncs.hpp:41-45
    for (int64_t i = 0; i < size; i++) out[(size_t) i] = 0.5 - 0.5 * std::cos((pi * 2 * i) / size);
    return true;
  case FLUHIP_WINDOW_HANND: // :46-51
  {
    double norm = pi / size;
    for (int64_t i = 0; i < size; i++) out[(size_t) i] = norm * std::sin((2 * pi * i) / size);
    return true;
  }
  case FLUHIP_WINDOW_HAMMING: // :52-56
    for (int64_t i = 0; i < size; i++) out[(size_t) i] = 0.54 - 0.46 * std::cos((pi * 2 * i) / size);
    return true;
  case FLUHIP_WINDOW_BLACKMANHARRIS: // :57-65 (all three cosines share one argument, as written there)
    for (int64_t i = 0; i < size; i++)
      out[(size_t) i] = 0.35875 - 0.48829 * std::cos((pi * 2 * i) / size) +
                        0.14128 * std::cos((pi * 2 * i) / size) +
                        0.01168 * std::cos((pi * 2 * i) / size);
    return true;
  case FLUHIP_WINDOW_GAUSSIAN: // :66-72 (requires odd size; sigma = size / 3 in integer arithmetic)
  {
    if (size % 2 == 0) return false;
    double sigma = (double) (size / 3);
    int64_t h = (size - 1) / 2;
    for (int64_t i = -h; i <= h; i++) out[(size_t) (i + h)] = std::exp(-i * i / (2 * sigma * sigma));
    return true;
  }
  default: return false;
  }
}

// device table of `fft` doubles: the window followed by zeros (a frame shorter than the transform
// is zero-padded at its tail, util/FFT.hpp:97-98)
int get_window(fluhip_ctx* ctx, int64_t win, int64_t fft, int type, const double** out)
{
  auto key = std::make_tuple(win, fft, type);
  auto it = ctx->windows.find(key);
  if (it == ctx->windows.end())
  {
    std::vector<double> w;
    if (!make_window(type, win, w)) return fail(ctx, "unsupported window type / size");
    w.resize((size_t) std::max(win, fft), 0.0);
    double* d = nullptr;
    HIPCHK(ctx, hipMalloc(&d, w.size() * sizeof(double)));
    HIPCHK(ctx, hipMemcpy(d, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
    it = ctx->windows.emplace(key, d).first;
  }
  *out = it->second;
  return FLUHIP_OK;
}

int get_twiddle(fluhip_ctx* ctx, int64_t fft, const double** out)
{
  auto it = ctx->twiddles.find((int) fft);
  if (it == ctx->twiddles.end())
  {
    const size_t nc = (size_t) fft / 2;
    std::vector<double> t(2 * nc);
    for (size_t j = 0; j < nc; j++)
    {
      const double ang = -2.0 * M_PI * (double) j / (double) fft;
      t[2 * j] = std::cos(ang);
      t[2 * j + 1] = std::sin(ang);
    }
    double* d = nullptr;
    HIPCHK(ctx, hipMalloc(&d, std::max<size_t>(16, t.size() * sizeof(double))));
    HIPCHK(ctx, hipMemcpy(d, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    it = ctx->twiddles.emplace((int) fft, d).first;
  }
  *out = it->second;
  return FLUHIP_OK;
}

// ---------------------------------------------------------------------------------------
// C ABI: context, parameter arithmetic, profiling aid
// ---------------------------------------------------------------------------------------
extern "C" {

int fluhip_abi_version(void) { return FLUHIP_ABI_VERSION; }

int fluhip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int fluhip_ctx_create(int device, fluhip_ctx** out)
{
  if (!out) return FLUHIP_ERROR;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return FLUHIP_ERROR;
  std::unique_ptr<fluhip_ctx> ctx(new fluhip_ctx);
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) return FLUHIP_ERROR;
  if (hipGetDeviceProperties(&ctx->props, device) != hipSuccess) return FLUHIP_ERROR;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return FLUHIP_ERROR;
  *out = ctx.release();
  return FLUHIP_OK;
}

void fluhip_ctx_destroy(fluhip_ctx* ctx)
{
  if (!ctx) return;
  (void) hipSetDevice(ctx->device);
  if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
  if (ctx->bigFft) (void) hipFree(ctx->bigFft);
  for (int i = 0; i < 2; i++)
  {
    if (ctx->stage[i]) (void) hipHostFree(ctx->stage[i]);
    if (ctx->stageEv[i]) (void) hipEventDestroy(ctx->stageEv[i]);
  }
  for (auto& kv : ctx->windows) (void) hipFree(kv.second);
  for (auto& kv : ctx->twiddles) (void) hipFree(kv.second);
  for (auto& r : ctx->profRecs) { (void) hipEventDestroy(r.start); (void) hipEventDestroy(r.stop); }
  for (auto e : ctx->eventPool) (void) hipEventDestroy(e);
  if (ctx->sideStream) (void) hipStreamSynchronize(ctx->sideStream);
  for (auto e : ctx->sideEv) if (e) (void) hipEventDestroy(e);
  if (ctx->sideStream) (void) hipStreamDestroy(ctx->sideStream);
  if (ctx->copyStream) (void) hipStreamDestroy(ctx->copyStream);
  if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
  g_pool.trim(ctx->device); // cached device blocks go with the context
  delete ctx;
}

const char* fluhip_last_error(const fluhip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int fluhip_last_error_is_out_of_memory(const fluhip_ctx* ctx) { return ctx && ctx->errOom ? 1 : 0; }
void fluhip_clear_error(fluhip_ctx* ctx) { if (ctx) { ctx->err.clear(); ctx->errOom = false; } }

int fluhip_ctx_device_info(const fluhip_ctx* ctx, char* name, int name_len, char* arch,
                           int arch_len, int* compute_units)
{
  if (!ctx) return FLUHIP_ERROR;
  if (name && name_len > 0) { std::snprintf(name, (size_t) name_len, "%s", ctx->props.name); }
  if (arch && arch_len > 0) { std::snprintf(arch, (size_t) arch_len, "%s", ctx->props.gcnArchName); }
  if (compute_units) *compute_units = ctx->props.multiProcessorCount;
  return FLUHIP_OK;
}

void* fluhip_ctx_stream(const fluhip_ctx* ctx) { return ctx ? (void*) ctx->stream : nullptr; }

int fluhip_ctx_trim(fluhip_ctx* ctx)
{
  if (!ctx) return FLUHIP_ERROR;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  g_pool.trim(ctx->device);
  return FLUHIP_OK;
}

int fluhip_ctx_set_progress_lag(fluhip_ctx* ctx, int lag)
{
  if (!ctx) return FLUHIP_ERROR;
  if (lag < 1 || lag > 64) return fail(ctx, "progress lag must be 1 .. 64");
  ctx->progressLag = lag;
  return FLUHIP_OK;
}

int fluhip_ctx_synchronize(fluhip_ctx* ctx)
{
  if (!ctx) return FLUHIP_ERROR;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLUHIP_OK;
}

int fluhip_fft_params(int64_t win, int64_t hop, int64_t fft, int64_t* win_out, int64_t* hop_out,
                      int64_t* fft_out, int64_t* bins_out)
{
  // clients/common/ParameterTypes.hpp:295-312
  if (win < 4) return FLUHIP_ERROR;
  int64_t h = hop > 0 ? hop : win >> 1;
  int64_t f = fft;
  if (f < 0)
  {
    f = 1;
    while (f < win) f <<= 1; // nextPow2(win, up)
  }
  if ((f & (f - 1)) || f < win) return FLUHIP_ERROR;
  if (win_out) *win_out = win;
  if (hop_out) *hop_out = h;
  if (fft_out) *fft_out = f;
  if (bins_out) *bins_out = (f >> 1) + 1;
  return FLUHIP_OK;
}

int64_t fluhip_stft_num_frames(int64_t n, int64_t win, int64_t hop)
{
  (void) win;
  return hop > 0 ? (n + hop) / hop : 0;
}

// ---- corpus ---------------------------------------------------------------------------
int fluhip_prof_enable(fluhip_ctx* ctx, int on)
{
  if (!ctx) return FLUHIP_ERROR;
  ctx->prof = on != 0;
  return FLUHIP_OK;
}

int fluhip_prof_reset(fluhip_ctx* ctx)
{
  if (!ctx) return FLUHIP_ERROR;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& r : ctx->profRecs)
  {
    ctx->eventPool.push_back(r.start);
    ctx->eventPool.push_back(r.stop);
  }
  ctx->profRecs.clear();
  return FLUHIP_OK;
}

int fluhip_prof_read(fluhip_ctx* ctx, int kernel_class, int64_t* launches, double* total_ms)
{
  if (!ctx) return FLUHIP_ERROR;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  int64_t n = 0;
  double tot = 0.0;
  for (auto& r : ctx->profRecs)
  {
    if (r.cls != kernel_class) continue;
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, r.start, r.stop));
    tot += ms;
    n++;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = tot;
  return FLUHIP_OK;
}

} // extern "C"
