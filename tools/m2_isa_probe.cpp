// ISA-level probe for nmf_update5_kernel instantiations (round 5, the MODE-2 wrong result): loads a device code object
// assembled from (possibly hand-patched) compiler output of kernels_nmf5.hip, launches ONE factor update of a full chip
// (128 buffers x 8 strips of 8 column groups, rank 32) through a test instantiation and through a reference instantiation
// of the same code object, and reports where the results differ.  Nothing of the library is involved: the kernel is fed a
// hand-filled Upd5Args (a copy of the struct in kernels_nmf5.hip).
//
//   hipcc -S --cuda-device-only ... kernels_nmf5.hip -o k.s ; <patch k.s> ;
//   clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c k.s -o k.o ; ld.lld -shared k.o -o k.co
//   g++ -O2 tools/m2_isa_probe.cpp -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lamdhip64 -o tools/bin/m2_isa_probe
//   tools/bin/m2_isa_probe k.co <test kernel symbol> <reference kernel symbol> [steps] [repeats] [shmem bytes]
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <set>
#include <vector>

struct WaveDesc;
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
  long long* clk;
  const WaveDesc* list;
  double* sidePart;
  double* sideWold;
  const double* cmbStat;
  const double* cmbSide;
  const double* cmbWold;
  double* cmbNrmOut;
  double* cmbRowOut;
  int cmbParts, cmbSlices, cmbK;
};
static_assert(sizeof(Upd5Args) == 240, "kernarg size of the compiled kernel");

#define CHK(x)                                                                                         \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } \
  } while (0)

int main(int argc, char** argv)
{
  if (argc < 4) { std::fprintf(stderr, "usage: %s code-object test-symbol reference-symbol [steps] [repeats] [shmem]\n", argv[0]); return 2; }
  const int steps = argc > 4 ? std::atoi(argv[4]) : 1, repeats = argc > 5 ? std::atoi(argv[5]) : 3;
  const unsigned shmem = argc > 6 ? (unsigned) std::atoi(argv[6]) : 122880u;
  const int B = 128, Kp = 32, C = 1024, Cp = 1056, R = 4 * steps, Rp = ((R + 31) / 32) * 32, D = 4;
  hipModule_t mod;
  CHK(hipModuleLoad(&mod, argv[1]));
  hipFunction_t ftest, fref;
  CHK(hipModuleGetFunction(&ftest, mod, argv[2]));
  CHK(hipModuleGetFunction(&fref, mod, argv[3]));
  std::vector<double> V((size_t) B * Rp * Cp, 0.0), Mv((size_t) B * Rp * Kp, 0.0), S0((size_t) B * Cp * Kp, 0.0);
  std::mt19937_64 g(12345);
  std::uniform_real_distribution<double> u(0.05, 1.0);
  for (int b = 0; b < D; b++)
  {
    for (int r = 0; r < R; r++)
      for (int c = 0; c < C; c++) V[((size_t) b * Rp + r) * Cp + c] = u(g);
    for (int r = 0; r < R; r++)
      for (int k = 0; k < Kp; k++) Mv[((size_t) b * Rp + r) * Kp + k] = u(g);
    for (int c = 0; c < C; c++)
      for (int k = 0; k < Kp; k++) S0[((size_t) b * Cp + c) * Kp + k] = u(g);
  }
  for (int b = D; b < B; b++)
  {
    std::memcpy(&V[(size_t) b * Rp * Cp], &V[(size_t) (b % D) * Rp * Cp], sizeof(double) * Rp * Cp);
    std::memcpy(&Mv[(size_t) b * Rp * Kp], &Mv[(size_t) (b % D) * Rp * Kp], sizeof(double) * Rp * Kp);
    std::memcpy(&S0[(size_t) b * Cp * Kp], &S0[(size_t) (b % D) * Cp * Kp], sizeof(double) * Cp * Kp);
  }
  double *dV, *dM, *dS;
  CHK(hipMalloc(&dV, V.size() * 8)); CHK(hipMalloc(&dM, Mv.size() * 8)); CHK(hipMalloc(&dS, S0.size() * 8));
  CHK(hipMemcpy(dV, V.data(), V.size() * 8, hipMemcpyHostToDevice));
  CHK(hipMemcpy(dM, Mv.data(), Mv.size() * 8, hipMemcpyHostToDevice));
  Upd5Args a;
  std::memset(&a, 0, sizeof a);
  a.V = dV; a.ldv = Cp; a.strideV = (int64_t) Rp * Cp;
  a.Mv = dM; a.strideM = (int64_t) Rp * Kp;
  a.S = dS; a.strideS = (int64_t) Cp * Kp;
  a.R = R; a.C = C; a.B = B;
  a.nGroups = C / 16; a.wavesPerBuf = 8; a.wgPerBuf = 2; a.nSteps = steps; a.nsplit = 1; a.stepsPerSplit = steps;
  a.Cp = Cp; a.xcdMap = 1;
  auto run = [&](hipFunction_t f, std::vector<double>& out) {
    CHK(hipMemcpy(dS, S0.data(), S0.size() * 8, hipMemcpyHostToDevice));
    void* params[] = {&a};
    CHK(hipModuleLaunchKernel(f, (unsigned) (B * 2), 1, 1, 256, 1, 1, shmem, nullptr, params, nullptr));
    CHK(hipDeviceSynchronize());
    out.resize(S0.size());
    CHK(hipMemcpy(out.data(), dS, out.size() * 8, hipMemcpyDeviceToHost));
  };
  std::vector<double> ref, got;
  run(fref, ref);
  // the reference against plain arithmetic (buffer 0, a few columns): the probe's arguments mean what the kernel means
  {
    double worst = 0.0;
    for (int c : {0, 3, 129, 1023})
    {
      double q[64];
      for (int r = 0; r < R; r++)
      {
        double t = 0.0;
        for (int k = 0; k < Kp; k++) t += Mv[(size_t) r * Kp + k] * S0[(size_t) c * Kp + k];
        q[r] = V[(size_t) r * Cp + c] / std::max(t, 2.220446049250313e-16);
      }
      for (int k = 0; k < Kp; k++)
      {
        double num = 0.0, den = 0.0;
        for (int r = 0; r < R; r++) { num += q[r] * Mv[(size_t) r * Kp + k]; den += Mv[(size_t) r * Kp + k]; }
        const double want = S0[(size_t) c * Kp + k] * num / den;
        worst = std::max(worst, std::fabs(ref[(size_t) c * Kp + k] - want) / want);
      }
    }
    std::printf("reference kernel against plain arithmetic: %.3g\n", worst);
  }
  int totalBad = 0;
  for (int rep = 0; rep < repeats; rep++)
  {
    run(ftest, got);
    int bad = 0, badBufs = 0, zeros = 0;
    std::map<int, int> byComp, byColInStrip;
    for (int b = 0; b < B; b++)
    {
      int nb = 0;
      for (int c = 0; c < C; c++)
        for (int k = 0; k < Kp; k++)
        {
          const size_t i = ((size_t) b * Cp + c) * Kp + k;
          if (std::fabs(got[i] - ref[i]) > 1e-9 * std::fabs(ref[i]))
          {
            nb++; byComp[k]++; byColInStrip[c % 128]++;
            if (got[i] == 0.0) zeros++;
          }
        }
      bad += nb; badBufs += nb > 0;
    }
    std::printf("repeat %d: %d entries off the reference in %d buffers (%d of them exactly 0)", rep, bad, badBufs, zeros);
    if (bad)
    {
      std::printf("; components:");
      for (auto& kv : byComp) std::printf(" %d", kv.first);
      std::printf("; columns of a strip:");
      int n = 0;
      for (auto& kv : byColInStrip) if (n++ < 12) std::printf(" %d", kv.first);
    }
    std::printf("\n");
    if (bad && rep == 0)
      for (int c : {0, 1, 128})
      {
        std::printf("  buffer 0 column %d, components 22..31: old / reference / got / got:reference\n", c);
        for (int k = 22; k < 32; k++)
        {
          const size_t i = (size_t) c * Kp + k;
          std::printf("    %2d  %.9g  %.9g  %.9g  %.6g\n", k, S0[i], ref[i], got[i], got[i] / ref[i]);
        }
      }
    totalBad += bad;
  }
  std::printf("SUMMARY %s: %d\n", argv[2], totalBad);
  return 0;
}
