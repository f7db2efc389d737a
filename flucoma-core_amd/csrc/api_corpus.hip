// api_corpus.hip -- the corpus level of the C ABI: B equal-shape (or ragged) buffers resident in HBM, the planner that picks
// the schedule of the factor updates for a shape, the iteration loop, and the fluhip_corpus_* entry points.
//   NMF::multiplicativeUpdates             include/flucoma/algorithms/public/NMF.hpp:144-183
//   bufnmf::NMFClient::process write-back  include/flucoma/clients/nrt/NMFClient.hpp:277-300
#include "api_internal.h"

// which factor-update path runs:
//   5 = v_mfma_f64_4x4x4_4b + LDS-DMA operand streaming (kernels_nmf5.hip): every rank up to 128, padded to 16 / 32 / 64 / 128
//   0 = un-fused, over a materialised ratio matrix (kernels_nmf_wide.hip): any rank, used above 128
//       (FLUHIP_NMF_KERNEL=-1 forces it: an independent second implementation for the tests)
int update_variant(int Kp)
{
  if (Kp > 128) return 0;
  static const int forced = [] {
    const char* e = fluhip::ab_getenv("FLUHIP_NMF_KERNEL");
    return e ? std::atoi(e) : 0;
  }();
  if (forced == -1) return 0;
  return 5;
}
// padded rank: the 4x4x4 kernel is built for 16 / 32 / 64 / 128 (components up to the padded rank are zero and stay zero);
// above 128 the any-rank path takes multiples of 16
int64_t padded_rank(int64_t K)
{
  if (K <= 16) return 16;
  if (K <= 32) return 32;
  if (K <= 64) return 64;
  if (K <= 128) return 128;
  return round_up(K, 16);
}

static int choose_split4(int64_t B, int C, int R, int Kp)
{
  static const int forceS = [] { const char* e = fluhip::ab_getenv("FLUHIP_PLAN_SPLIT"); return e ? std::atoi(e) : 0; }();
  const int64_t nSteps = (R + 3) / 4;
  const int64_t smax = std::max<int64_t>(1, std::min<int64_t>(64, nSteps / 12)); // >= 12 steps per wavefront
  if (forceS > 0) return (int) std::min<int64_t>(std::min<int64_t>(forceS, 64), std::max<int64_t>(1, nSteps / 2)); // the finalize kernel sums at most 64 splits
  const int64_t waves = B * nmf_update5_waves_per_buffer(C, Kp, (int) B);
  if (waves >= 768) return 1;
  // one wavefront per SIMD (the kernel's register footprint allows no more): never exceed 1024 in
  // total, a 1025th wavefront would wait for a whole pass of the others
  const int64_t s = 1024 / waves;
  return (int) std::max<int64_t>(1, std::min(s, smax));
}

// A whole-contraction update whose wavefronts need a last, poorly filled round of the 1024 SIMDs (config 3's H update: 2 x 808
// strips = 1.58 rounds, paid as 2) goes out as TWO launches: the strips that fill whole rounds as before, then the remaining
// strips with their contraction cut into `split` pieces (uniform split schedule: partials + finalize), which deals the tail
// over the chip in short rounds -- 513 + 2 x 171 steps instead of 2 x 513.  Costs in 4-row steps of this strip width:
// ~20 k cycles of prologue + epilogue per wavefront, partials written and read back at ~3 TB/s plus the finalize launch.
// Returns the split (0: one launch pays) and the strips per buffer of the first launch.  FLUHIP_TAIL_SPLIT: 0 off, n forces n
// pieces; FLUHIP_TAIL_SLOTS: the wavefronts of a round (tests: small corpora take the path).
static int plan_tail(int64_t B, int C, int R, int Kp, int* stripsA)
{
  static const int forceS = [] { const char* e = fluhip::ab_getenv("FLUHIP_TAIL_SPLIT"); return e ? std::atoi(e) : -1; }();
  static const int slotsEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_TAIL_SLOTS"); return e ? std::atoi(e) : 0; }();
  *stripsA = 0;
  if (forceS == 0) return 0;
  const int64_t slots = slotsEnv > 0 ? slotsEnv : 1024;
  const int G = (C + 15) / 16;
  const int w = nmf_update5_strips(C, Kp, (int) B);
  const int ng = (G + w - 1) / w;
  const int64_t total = B * w;
  const int64_t roundsA = total / slots;
  if (roundsA < 1 || roundsA > 3 || total % slots == 0) return 0;
  // (a workgroup is four wavefronts of ONE buffer and one workgroup fills a CU: strips are dealt in fours, and a launch
  //  occupies round_up(strips, 4) slots per buffer -- 200 buffers x 5 strips are 400 workgroups, two rounds, not 1000 slots)
  int wA = (int) std::min<int64_t>(w - 1, roundsA * slots / B);
  wA -= wA % 4;
  if (wA < 4) return 0;
  const int64_t nRest = B * round_up(w - wA, 4);
  const int64_t nSteps = (R + 3) / 4;
  const double cycles = 400.0 + 270.0 * ng * (Kp / 32.0); // per step (measured: rank 32 400 + 270 NG; rank 128, 2 groups 2 376)
  const double ovh = 20000.0 / cycles, stepUs = cycles / 2300.0;
  const double cur = (double) ((B * round_up(w, 4) + slots - 1) / slots) * (nSteps + ovh);
  const int64_t restCols = C - (int64_t) wA * ng * 16;
  int best = 0;
  double bestCost = 0.93 * cur; // a clear win only
  const int smax = (int) std::min<int64_t>(8, nSteps / 12);
  for (int sp = 2; sp <= smax; sp++)
  {
    if (forceS > 0 && sp != std::min(forceS, smax)) continue;
    const double finUs = 2.0 * sp * B * restCols * Kp * 8.0 / 3.0e6 + 8.0;
    const double cost = (double) ((B * wA + slots - 1) / slots) * (nSteps + ovh) +
                        (double) ((nRest * sp + slots - 1) / slots) * ((nSteps + sp - 1) / sp + ovh) + finUs / stepUs;
    // (more pieces only for a clear gain: they cost memory, and config 3 measures the same with 3 and 5)
    if ((best == 0 ? cost < bestCost : cost < 0.97 * bestCost) || (forceS > 0 && best == 0)) { best = sp; bestCost = cost; }
  }
  if (best) *stripsA = wA;
  return best;
}

static bool offsize_enabled()
{
  static const int offEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_OFFSIZE"); return e ? std::atoi(e) : 1; }();
  return offEnv != 0;
}
// where the work-list form beats the uniform split schedule for equal-length corpora (tools/batch_timing.py with
// FLUHIP_LIST_PLAN=0|1, profiles/r03/small_batches_*.jsonl)
struct PlanShape { int64_t B, T, F, Kp; }; // what the choice of schedule depends on (also reachable without a corpus: fluhip_debug_plan_kind)
static bool list_plan_pays(const PlanShape* c)
{
  // measured at rank 32, 10 s buffers (us per iteration, uniform schedule -> lists): 1 buffer 47 -> 51, 2: 53 -> 57, 4: 73 -> 65,
  // 8: 92 -> 73, 16: 135 -> 109, 24: 196 -> 168, 32: 250 -> 202, 48: 346 -> 289, 64: 370 -> 345, 96: 758 -> 511, 112: 878 -> 564,
  // 128: 619 = 621 (the same schedule either way); ranks 16 / 64 / 128 at 4 and 16 buffers likewise (49 -> 49, 89 -> 76;
  // 130 -> 98, 227 -> 193; 211 -> 160, 486 -> 476).  Between one and two rounds (profiles/r03/midsize_ab.txt, two rounds of
  // the uniform schedule = ~1 250 us): 144 buffers 729, 176: 854, 200: 1 007, 232: 1 102, 250: 1 153.  So: from three buffers on,
  // while whole contractions at the widest strips do not fill the chip in whole rounds.
  const int maxNG = nmf_update5_max_groups((int) c->Kp);
  const int G = ((int) c->F - 1 + 15) / 16;
  const int64_t w0 = c->B * ((G + maxNG - 1) / maxNG);
  // Beyond one round the lists win as well, whole rounds or not (profiles/r03/midsize_ab.txt parts 3 - 4, uniform schedule in
  // round-major windows -> lists: 288 buffers 1 723 -> 1 475 us, 400: 2 300 -> 2 015, 520: 2 864 -> 2 589, 900: 4 568 -> 4 320,
  // 1000: 4 586 -> 4 424; exact multiples of a round 256: 1 222 -> 1 206, 512: 2 346 -> 2 285, 1024: 4 611 -> 4 485, and with a
  // progress callback -- iteration-major launches of several rounds -- 512: 2 427 -> 2 234, 1024: 5 157 -> 4 452): several
  // thousand workgroups handed out as CUs free up keep the chip busy across what the uniform launch runs as lock-step rounds.
  // Exactly one round (the bench shard: 128 buffers) is the same schedule either way and stays on the uniform kernel.
  // (Rank 128 apart: the list kernel's instantiation there -- grouped refill, column sums in the kernel -- runs 3 580 cycles per
  //  step against the uniform one's 2 376.)
  // Rank 128 (profiles/r03/plan_regimes.txt): 4 buffers 211 -> 160 on the lists, but 16: 452 -> 500, 24: 581 -> 820, 32: 639 -> 818,
  // 40: 902 (uniform, two-launch H update) -> 1 330 -- lists only while whole contractions fill less than half a round.
  // (few buffers, 10 s each: 3: 177 -> 167, 4: 168 -> 163, 8: 240 -> 273, 12: 516 -> 405; long ones lose: 4 x 60 s 558 -> 753)
  if (c->Kp > 64) return c->B >= 3 && w0 < 512 && c->T <= 2048;
  // Rank 64 follows rank 32 (100 x 10 s 1 333 -> 1 043, 300: 2 999 -> 2 771, 128 x 2 s 330 -> 317, 1 x 60 s 131 -> 106, 1 x 300 s
  // 417 -> 263, 2 x 300 s 800 -> 614; 40 x 10 s 456 -> 473 the one loss).
  // One or two buffers (rank <= 64; a single buffer of rank <= 16 that fits the frame-strip schedule never gets here): the
  // uniform split schedule up to ~45 s of frames in all (1 x 30 s 62.5 us per iteration against 66.7, 2 x 10 s 53 against 57),
  // lists beyond (1 x 60 s 80.5 -> 72.8, 1 x 300 s 253 -> 163, 2 x 30 s 79.3 -> 71.3, 2 x 300 s 454 -> 340).
  if (c->B <= 2) return c->B * c->T * c->F >= 4200000;
  return w0 != 1024;
}

// ---------------------------------------------------------------------------------------
// ragged corpora: work lists of the two factor updates
// ---------------------------------------------------------------------------------------
namespace {
// up to four wavefronts that share a strip and split its contraction (WaveDesc::grp); they stay together in one workgroup
struct WaveGroup
{
  std::vector<WaveDesc> waves; // waves[0] is the leader; grp holds (rank << 4 | size << 8) until the packing
  int64_t work;                // what the group takes: its longest member
  int buf;
};

// groups -> workgroups of 4 wavefronts: longest first (the hardware hands the next workgroup to whichever CU frees up),
// first fit (a later, shorter group fills the slots an earlier workgroup has left).  Workgroup i runs on XCD i & 7 and a
// launch of up to 256 workgroups is ONE round only if every XCD gets at most 32 of them, so the workgroups are numbered
// in list order -- 8 consecutive ones on 8 different XCDs -- rather than by buffer.
void pack_groups(std::vector<WaveGroup>& groups, std::vector<WaveDesc>& out, int* wgs)
{
  std::stable_sort(groups.begin(), groups.end(), [](const WaveGroup& a, const WaveGroup& b) { return a.work > b.work; });
  std::vector<std::vector<WaveDesc>> q; // workgroups of up to 4 descriptors
  size_t firstOpen = 0;
  for (const auto& g : groups)
  {
    bool placed = false;
    while (firstOpen < q.size() && q[firstOpen].size() >= 4) firstOpen++;
    for (size_t i = std::max(firstOpen, q.size() > 256 ? q.size() - 256 : 0); i < q.size(); i++) // (a bounded look-back: linear time)
      if (q[i].size() + g.waves.size() <= 4) { q[i].insert(q[i].end(), g.waves.begin(), g.waves.end()); placed = true; break; }
    if (!placed) q.push_back(g.waves);
  }
  out.assign(q.size() * 4, WaveDesc{0, 0, 0, 0, 0, -1, 0, -1, 0, 0, 0, 0});
  for (size_t i = 0; i < q.size(); i++)
  {
    auto& wgp = q[i];
    bool barrier = false;
    for (const auto& d : wgp) barrier = barrier || ((d.grp >> 8) & 15) > 1;
    size_t leaderAt = 0; // the leader's wavefront index is known only now
    for (size_t w = 0; w < wgp.size(); w++)
    {
      WaveDesc d = wgp[w];
      const int rank = (d.grp >> 4) & 15, size = (d.grp >> 8) & 15;
      if (rank == 0) leaderAt = w;
      d.grp = (int) leaderAt | (rank << 4) | (size << 8) | (barrier ? (1 << 16) : 0);
      out[i * 4 + w] = d;
    }
  }
  *wgs = (int) q.size();
}

int upload_list(fluhip_ctx* ctx, DevBuf& dst, const void* src, size_t bytes)
{
  HIPCHK(ctx, dst.alloc(bytes, false, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // the host image may go out of scope
  return FLUHIP_OK;
}
} // namespace

// Work lists of the two factor updates over buffers with their own frame counts tOf[b]: ragged corpora, and equal-length
// corpora too small to fill the chip with whole contractions (the intra-workgroup reduction is what this form has over the
// uniform split schedule: fewer or no partials in memory, no finalize launch).
namespace {
struct ListSide
{
  std::vector<WaveDesc> list;
  std::vector<int> splitTab;
  int wgs = 0, ng = 0, partial = 0, maxSplit = 1, pieces = 1, statParts = 0;
  int64_t nPartials = 0;
};
struct ListPlanHost
{
  ListSide W, H;
  bool sideW = false;
};
} // namespace
// pure host code (no device, no context): also reachable through fluhip_debug_plan_lists for the CPU tests
static void build_list_plan(const std::vector<int>& tOf, int Tmax, int F, int Kp, ListPlanHost& out)
{
  const int B = (int) tOf.size();
  const int maxNG = nmf_update5_max_groups(Kp);
  // ---- W update: strips over the bins (the same for every buffer), the contraction over a buffer's own frames ------------
  {
    auto envInt = [](const char* name, int dflt) { const char* e = fluhip::ab_getenv(name); return e ? std::atoi(e) : dflt; };
    std::vector<int> steps((size_t) B);
    for (int b = 0; b < B; b++) steps[(size_t) b] = (tOf[(size_t) b] + 3) / 4;
    // the Nyquist bin as a side column (fluhip_kernels.h SideColumn): every power-of-two transform has 16 m + 1 bins, and a
    // wavefront runs the loop of the widest strip of the launch -- 65 column groups never deal evenly
    static const int sideOff = [] { const char* e = fluhip::ab_getenv("FLUHIP_NO_SIDE"); return e ? std::atoi(e) : 0; }();
    out.sideW = !sideOff && nmf_side_column_supported(Tmax, F, Kp);
    const int C = F - (out.sideW ? 1 : 0);
    const int G = (C + 15) / 16;
    // One candidate schedule per strip width.  Contractions are cut into pieces when one round of wavefronts (1024 SIMDs, one
    // workgroup of four per CU) would stay part empty: the wavefront budget goes to whichever buffer has the longest pieces
    // (down to 12 steps); the pieces of one strip go to up to four wavefronts of one workgroup (added up through the LDS:
    // WaveDesc::grp) and only what is left beyond four becomes partials in memory for the finalize launch.  Cost model as
    // for the H update below: ~400 + 270 NG cycles per step of a strip of NG column groups, ~40 k per round, ~60 k per
    // finalize launch -- narrow strips cost little per group, so few buffers take many narrow strips and few partials.
    struct Cand
    {
      std::vector<WaveDesc> list;
      std::vector<int> splitTab;
      bool anyPartial = false;
      int64_t pbase = 0;
      int maxSplit = 1, maxPieces = 1, wgs = 0, wW = 0, ngW = 0;
      double cost = 1e300;
    };
    auto build = [&](int wW, int forcedBudget) -> Cand {
      Cand r;
      r.wW = wW;
      r.ngW = (G + wW - 1) / wW;
      r.splitTab.assign((size_t) B * 2, 0);
      const int gmax = std::min(envInt("FLUHIP_RG_GMAX", 4), nmf_update5_groups_fit(Kp, r.ngW) ? 4 : 1);
      const int64_t waves0 = (int64_t) B * wW;
      const int64_t wgs0 = (waves0 + 3) / 4;
      const int64_t roundsWanted = wgs0 <= 384 ? std::max<int64_t>(1, (wgs0 + 255) / 256) : 0; // 0: many rounds, no cutting
      std::vector<int> nsOf((size_t) B, 1);
      for (int64_t budgetWaves = forcedBudget > 0 ? forcedBudget : 1024 * std::max<int64_t>(roundsWanted, 1);; budgetWaves -= 32)
      {
        std::fill(nsOf.begin(), nsOf.end(), 1);
        if (roundsWanted > 0 && budgetWaves > waves0)
        {
          const int64_t budget = budgetWaves / wW;
          auto piece = [&](int b) { return (steps[(size_t) b] + nsOf[(size_t) b] - 1) / nsOf[(size_t) b]; };
          std::vector<std::pair<int, int>> heap; // (piece, buffer)
          for (int b = 0; b < B; b++) heap.emplace_back(piece(b), b);
          std::make_heap(heap.begin(), heap.end());
          for (int64_t units = B; units < budget; units++)
          {
            std::pop_heap(heap.begin(), heap.end());
            const int b = heap.back().second;
            if (heap.back().first <= 12 || nsOf[(size_t) b] >= 64) break; // the longest piece cannot get shorter
            nsOf[(size_t) b]++;
            heap.back() = std::make_pair(piece(b), b);
            std::push_heap(heap.begin(), heap.end());
          }
        }
        r.anyPartial = false;
        for (int b = 0; b < B; b++) r.anyPartial = r.anyPartial || nsOf[(size_t) b] > gmax;
        std::vector<WaveGroup> groups;
        r.pbase = 0;
        r.maxSplit = 1;
        r.maxPieces = 1;
        int64_t longest = 0;
        for (int b = 0; b < B; b++)
        {
          const int ns = nsOf[(size_t) b];
          const int per = (steps[(size_t) b] + ns - 1) / ns;
          const int np = (ns + gmax - 1) / gmax; // partials in memory
          r.splitTab[(size_t) b * 2] = (int) r.pbase;
          r.splitTab[(size_t) b * 2 + 1] = np;
          r.maxSplit = std::max(r.maxSplit, np);
          r.maxPieces = std::max(r.maxPieces, ns);
          longest = std::max<int64_t>(longest, per);
          int piece = 0;
          for (int q = 0; q < np; q++)
          {
            const int size = ns / np + (q < ns % np ? 1 : 0);
            const int base = G / wW, rem = G % wW;
            int g0 = 0;
            for (int st = 0; st < wW; st++)
            {
              const int ng = base + (st < rem ? 1 : 0);
              WaveGroup grp;
              grp.buf = b; grp.work = 0;
              for (int rk = 0; rk < size; rk++)
              {
                const int s0 = std::min(steps[(size_t) b], (piece + rk) * per), s1 = std::min(steps[(size_t) b], s0 + per); // (an empty tail piece: no steps, zeros)
                WaveDesc d{};
                d.buf = b; d.g0 = g0; d.ng = ng; d.s0 = s0; d.s1 = std::max(s0, s1);
                d.partIdx = (r.anyPartial && rk == 0) ? (int) r.pbase + q : -1;
                d.statIdx = b * wW + st;
                d.dIdx = (r.anyPartial && st == 0 && rk == 0) ? d.partIdx : -1;
                d.grp = (rk << 4) | (size << 8);
                grp.work = std::max<int64_t>(grp.work, d.s1 - d.s0);
                grp.waves.push_back(d);
              }
              if (ng > 0) groups.push_back(std::move(grp));
              g0 += ng;
            }
            piece += size;
          }
          r.pbase += np;
        }
        pack_groups(groups, r.list, &r.wgs);
        const double rounds = roundsWanted == 0 ? (double) r.wgs / 256.0 : (double) ((r.wgs + 255) / 256);
        r.cost = rounds * ((double) longest * (400.0 + 270.0 * r.ngW) + 40000.0) + (r.maxPieces > 1 ? 6000.0 : 0.0) +
                 (r.anyPartial ? 60000.0 + 3000.0 * r.maxSplit : 0.0);
        if (roundsWanted == 0 || r.wgs <= 256 * roundsWanted || budgetWaves <= waves0 || forcedBudget > 0) break;
      }
      return r;
    };
    Cand best;
    const int forcedNG = envInt("FLUHIP_RG_WNG", 0), forcedBudget = envInt("FLUHIP_RG_WBUDGET", 0);
    int lastW = -1;
    for (int ngc = maxNG; ngc >= 1; ngc--)
    {
      if (forcedNG > 0 && ngc != std::min(forcedNG, maxNG)) continue;
      const int wW = (G + ngc - 1) / ngc;
      if (wW == lastW) continue; // the same strips as the wider candidate
      lastW = wW;
      // wide strips re-use the moving factor's rows over more columns: with enough buffers to fill the chip without cutting
      // anything the widest form is what the batched kernel was tuned for -- narrower candidates only while the chip is not full
      if (forcedNG == 0 && (int64_t) B * ((G + maxNG - 1) / maxNG) >= 1536 && ngc < maxNG) break;
      Cand cnd = build(wW, forcedBudget);
      if (cnd.cost < best.cost - 1.0) best = std::move(cnd);
    }
    out.W.wgs = best.wgs;
    out.W.ng = best.ngW; out.W.partial = best.anyPartial ? 1 : 0; out.W.maxSplit = best.maxSplit;
    out.W.nPartials = best.anyPartial ? best.pbase : 0;
    out.W.statParts = best.anyPartial ? update_finalize_parts(C, Kp) : best.wW;
    out.W.pieces = best.maxPieces;
    out.W.list = std::move(best.list);
    out.W.splitTab = std::move(best.splitTab);
  }
  // ---- H update: strips over a buffer's own frames, the contraction over the bins (the same for every buffer) ------------
  {
    std::vector<int> groupsOf((size_t) B);
    for (int b = 0; b < B; b++) groupsOf[(size_t) b] = (tOf[(size_t) b] + 15) / 16;
    // Strip width NG, pieces of the bin contraction that share a workgroup (gsz: added up through the LDS) and partials in
    // memory (ns: a finalize launch) by a small cost model in shader cycles, fitted to tools/ragged_sweep.sh
    // and tools/batch_timing.py: a wavefront runs the loop of the launch's widest strip at ~400 + 270 NG cycles per 4-row step
    // (measured: 1 group 695, 4: 1 370, 6: 1 780, 7: 2 250, 8: 2 620, 9: 2 775), a round of up to 256 workgroups pays ~40 k
    // cycles of prologue, epilogue and launch, a finalize launch ~60 k.
    const int steps = (F + 3) / 4;
    int NG = maxNG, gsz = 1, ns = 1;
    double bestCost = 1e300;
    for (int cand = maxNG; cand >= 1; cand--)
    {
      int64_t strips = 0;
      int widestStrip = 1;
      for (int b = 0; b < B; b++)
      {
        const int w = (groupsOf[(size_t) b] + cand - 1) / cand;
        strips += w;
        widestStrip = std::max(widestStrip, (groupsOf[(size_t) b] + w - 1) / std::max(w, 1));
      }
      const bool fit = nmf_update5_groups_fit(Kp, widestStrip);
      for (int g : {1, 2, 4})
      {
        if (g > 1 && !fit) continue;
        for (int m : {1, 2, 3, 4, 6, 8, 12, 16})
        {
          const int pieces = g * m;
          if (pieces > 1 && steps / pieces < 12) continue;
          const int64_t wgs = (strips * m + (4 / g) - 1) / (4 / g);
          const double rounds = (double) ((wgs + 255) / 256);
          const double perWave = (double) ((steps + pieces - 1) / pieces) * (400.0 + 270.0 * widestStrip) + 40000.0;
          const double cost = rounds * perWave + (g > 1 ? 6000.0 : 0.0) + (m > 1 ? 60000.0 + 3000.0 * m : 0.0);
          if (cost < bestCost - 1.0) { bestCost = cost; NG = cand; gsz = g; ns = m; }
        }
      }
    }
    {
      auto envInt = [](const char* name, int dflt) { const char* e = fluhip::ab_getenv(name); return e ? std::atoi(e) : dflt; };
      NG = std::max(1, std::min(maxNG, envInt("FLUHIP_RG_HNG", NG)));
      gsz = envInt("FLUHIP_RG_HG", gsz);
      ns = envInt("FLUHIP_RG_HM", ns);
    }
    int ngH = 1;
    for (int b = 0; b < B; b++)
    {
      const int w = (groupsOf[(size_t) b] + NG - 1) / NG;
      ngH = std::max(ngH, (groupsOf[(size_t) b] + w - 1) / std::max(w, 1));
    }
    if (!nmf_update5_groups_fit(Kp, ngH)) gsz = 1;
    const int pieces = gsz * ns;
    const int per = (steps + pieces - 1) / pieces;
    std::vector<WaveGroup> groups;
    std::vector<int> splitTab((size_t) B * 2);
    for (int b = 0; b < B; b++) { splitTab[(size_t) b * 2] = b * ns; splitTab[(size_t) b * 2 + 1] = ns; }
    for (int b = 0; b < B; b++)
    {
      const int G = groupsOf[(size_t) b];
      const int w = (G + NG - 1) / NG;
      for (int j = 0; j < ns; j++)
      {
        const int base = G / w, rem = G % w;
        int g0 = 0;
        for (int st = 0; st < w; st++)
        {
          const int ng = base + (st < rem ? 1 : 0);
          WaveGroup grp;
          grp.buf = b; grp.work = 0;
          for (int r = 0; r < gsz; r++)
          {
            WaveDesc d{};
            d.buf = b; d.g0 = g0; d.ng = ng;
            d.s0 = std::min(steps, (j * gsz + r) * per); d.s1 = std::max(d.s0, std::min(steps, d.s0 + per));
            d.partIdx = (ns > 1 && r == 0) ? b * ns + j : -1;
            d.statIdx = 0;
            d.dIdx = (ns > 1 && st == 0 && r == 0) ? d.partIdx : -1;
            d.grp = (r << 4) | (gsz << 8);
            grp.work = std::max<int64_t>(grp.work, (int64_t) (d.s1 - d.s0) * ng);
            grp.waves.push_back(d);
          }
          if (ng > 0) groups.push_back(std::move(grp));
          g0 += ng;
        }
      }
    }
    pack_groups(groups, out.H.list, &out.H.wgs);
    out.H.ng = ngH; out.H.partial = ns > 1 ? 1 : 0; out.H.maxSplit = ns; out.H.nPartials = ns > 1 ? (int64_t) B * ns : 0;
    out.H.pieces = pieces;
    out.H.splitTab = std::move(splitTab);
  }
}

// ---------------------------------------------------------------------------------------
// The schedule of the factor updates of a shape as DATA (round 6).  decide_update_plan() is a pure function
// of (B, T, F, K): no device, no allocation, nothing read from a corpus -- every choice between the uniform / split / work-list /
// frame-strip / two-launch / any-rank forms, the side column, the off-size compute rank, and the size in doubles of every
// workspace the launches of that plan index.  plan_updates() copies it onto the corpus and allocates exactly those sizes;
// fluhip_debug_plan_shape() reports it without a device; tests/test_plan_table.py holds it to its invariants over a grid of
// shapes and to the pinned plans of the BASELINE and perf-matrix shapes.
void decide_update_plan(int64_t B, int64_t T, int64_t F, int64_t K, UpdatePlan& p)
{
  p = UpdatePlan{};
  const int64_t Tp = round_up(T, 32), Fp = round_up(F, 32);
  p.Kp = padded_rank(K);
  p.variant = update_variant((int) p.Kp);
  p.Kc = (int) p.Kp;
  p.nsplitW = p.nsplitH = 1;
  if (p.variant == 0)
  {
    p.wideDoubles = std::max(nmf_update_wide_scratch_doubles((int) T, (int) F, (int) p.Kp, (int) B),
                             nmf_update_wide_scratch_doubles((int) F, (int) T, (int) p.Kp, (int) B));
  }
  else
  {
    const int Kp = (int) p.Kp;
    p.nsplitW = choose_split4(B, (int) F, (int) T, Kp);
    p.nsplitH = choose_split4(B, (int) T, (int) F, Kp);
    // Fast path: W stays un-normalised in memory during the loop (UpdateArgs::nrm), the column statistics
    // come out of the update kernel's epilogue and the Nyquist bin is a side column when that shortens the
    // widest strip of the MFMA kernel (fluhip_kernels.h SideColumn).
    static const int lazyOff = [] { const char* e = fluhip::ab_getenv("FLUHIP_NO_LAZY"); return e ? std::atoi(e) : 0; }();
    static const int sideOff = [] { const char* e = fluhip::ab_getenv("FLUHIP_NO_SIDE"); return e ? std::atoi(e) : 0; }();
    p.lazy = !lazyOff;
    if (p.lazy && !sideOff && p.nsplitW == 1 && nmf_side_column_supported((int) T, (int) F, Kp) &&
        choose_split4(B, (int) F - 1, (int) T, Kp) == 1)
    {
      const int G = ((int) F + 15) / 16, G1 = G - 1;
      const int w = nmf_update5_strips((int) F, Kp, (int) B);
      const int w1 = nmf_update5_strips((int) F - 1, Kp, (int) B);
      // worth it when the widest strip gets shorter, or when the launch needs fewer passes over the 1024 SIMDs
      const int64_t passes = (B * w + 1023) / 1024, passes1 = (B * w1 + 1023) / 1024;
      p.sideW = w1 <= w && ((G1 + w1 - 1) / w1 < (G + w - 1) / w || passes1 < passes);
    }
    else if (p.lazy && !sideOff && p.nsplitW > 1 && nmf_side_column_supported((int) T, (int) F, Kp))
    {
      // Split contraction (few buffers): without the 16 m + 1-th bin the strips deal evenly and the pieces get shorter --
      // config 3 (2 x 2049 bins, rank 128): 130 strips x 7 pieces of 923 steps (910 wavefronts) -> 128 strips x 8 pieces
      // of 808 steps (1024 wavefronts).  Taken when the longest piece shrinks.
      const int s1 = choose_split4(B, (int) F - 1, (int) T, Kp);
      const int64_t nSteps = (T + 3) / 4;
      const int64_t w = B * nmf_update5_waves_per_buffer((int) F, Kp, (int) B);
      const int64_t w1 = B * nmf_update5_waves_per_buffer((int) F - 1, Kp, (int) B);
      if (s1 > 1 && w1 * s1 <= 1024 && w * p.nsplitW <= 1024 && (nSteps + s1 - 1) / s1 < (nSteps + p.nsplitW - 1) / p.nsplitW)
      {
        p.sideW = true;
        p.nsplitW = s1;
      }
    }
    // A single buffer of rank <= 16 runs the frame-strip schedule while one round of workgroups covers it (at most 6 frame
    // quads per CU: 71 s at hop 512): two launches per iteration instead of five and V read once.  Longer buffers and
    // batches stay with the split / batched kernels, which win there (tools/strip_vs_split.py).  FLUHIP_STRIP=0 off,
    // =1 wherever the kernel supports the shape.
    static const int stripEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_STRIP"); return e ? std::atoi(e) : -1; }();
    p.strip = p.lazy && stripEnv != 0 && nmf_strip_supported((int) F, (int) T, Kp) &&
              (stripEnv == 1 || (B == 1 && nmf_strip_workgroups((int) T) <= 512));
    if (p.strip)
    {
      p.sideW = false;
      p.stripPartDoubles = nmf_strip_part_doubles((int) F, (int) T, (int) B);
      // FLUHIP_STRIP_BIN=1 (A/B build only): the W update as its own launch over bin strips instead of the fused form (W
      // partials behind the H phase + the reduce launch).  Built and measured in round 4 (profiles/r04/c2_forms.md): 45.1 us
      // per iteration against 40.3 at config 2 -- a tenth of the partial bytes, but the last arriver's chain of cross-XCD
      // round trips (ticket, partials in two rounds, update) costs what the reduce launch cost.  Not adopted.
      static const int binEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_STRIP_BIN"); return e ? std::atoi(e) : 0; }();
      p.stripBin = kAbSwitches && binEnv != 0;
      // Round 5: the W update as the BIN-TILED launch (kernels_nmf_bintile.hip: four bins and ALL frames per workgroup, no
      // numerator partials in memory, no reduce launch, no ticket), the strip kernel doing the H update and the Nyquist bin's
      // partials.  Built to the review's specification, parity-green at full size, and measured at config 2
      // (profiles/r05/c2_bintile.md): 20.3 us for the launch in three layouts against the review's 13 us kill line.  Not
      // adopted: A/B build only (the kernel is not part of the production library), FLUHIP_STRIP_TILE=1.
#ifdef FLUHIP_AB_SWITCHES
      static const int tileEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_STRIP_TILE"); return e ? std::atoi(e) : 0; }();
      p.stripTile = !p.stripBin && tileEnv == 1 && nmf_bintile_supported((int) F, (int) T, Kp) &&
                    nmf_strip_tile_supported((int) F, (int) T, Kp);
#endif
    }
    // Equal-length corpora too small to fill the chip with whole contractions: the work-list form (plan_lists) instead of
    // the uniform split schedule -- narrow strips, the pieces of a contraction added up inside a workgroup, few or no
    // partials in memory.  FLUHIP_LIST_PLAN=0 keeps the uniform split schedule, =1 takes the lists whenever something is split.
    static const int listEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_LIST_PLAN"); return e ? std::atoi(e) : -1; }();
    const PlanShape ps{B, T, F, Kp};
    p.useLists = !p.strip && p.lazy && listEnv != 0 && (listEnv == 1 || list_plan_pays(&ps));
    // Off-size ranks (round 5): a rank between two array ranks keeps the ARRAYS of the padded rank (32 / 64 / 128) -- every
    // helper kernel runs its form of that rank on zero columns -- and the factor updates compute fewer MFMAs per product
    // (kernels_nmf5.hip KPM, nmf_update5_compute_rank): 6 of 8 for ranks 17 .. 24; 10 / 12 / 14 of 16 for 33 .. 40 / 48 / 56; 18, 20 .. 28
    // of 32 for 65 .. 72, .. 80, .. 112.  The plain and the split-contraction schedules and the work lists; the strip schedule
    // (rank <= 16) has no off-size rank.  FLUHIP_OFFSIZE=0 (A/B build): the padded forms, for the comparison.
    p.Kc = (offsize_enabled() && p.lazy && !p.strip) ? nmf_update5_compute_rank((int) K, Kp) : Kp;
    // (the launcher deals its strips by the COMPUTE rank, everything planned here -- statistics records, side
    //  slices, the tail launch, the scratch sizes -- by the padded rank.  The two agree because the group limits are the same
    //  within an off-size class; should a future form break that, the shape keeps the padded rank rather than a layout mismatch)
    if (p.Kc != Kp && (nmf_update5_strips((int) F, p.Kc, (int) B) != nmf_update5_strips((int) F, Kp, (int) B) ||
                       nmf_update5_strips((int) F - 1, p.Kc, (int) B) != nmf_update5_strips((int) F - 1, Kp, (int) B) ||
                       nmf_update5_strips((int) T, p.Kc, (int) B) != nmf_update5_strips((int) T, Kp, (int) B)))
      p.Kc = Kp;
    if (p.useLists)
    {
      // the lists themselves are built (and checked descriptor by descriptor: tests/test_list_plan.py) by build_list_plan
      ListPlanHost lp;
      build_list_plan(std::vector<int>((size_t) B, (int) T), (int) T, (int) F, Kp, lp);
      p.sideW = lp.sideW;
      p.stripsW = lp.W.statParts;
      p.nsplitW = lp.W.pieces;
      p.nsplitH = lp.H.pieces;
      const int64_t nPart = std::max(lp.W.nPartials, lp.H.nPartials);
      p.partDoubles = nPart * std::max(Fp, Tp) * Kp;
      p.dpartDoubles = std::max<int64_t>(32, std::max<int64_t>(nPart, B) * Kp);
      p.wscratchDoubles = wnorm_scratch_doubles(Kp, (int) B, p.stripsW);
      return;
    }
    // statistics partials of the W update: one per wavefront of a buffer, or one per 64-row chunk from the
    // finalize kernel when the contraction is split
    p.stripsW = p.nsplitW > 1 ? update_finalize_parts((int) F - (p.sideW ? 1 : 0), Kp)
                              : nmf_update5_strips((int) F - (p.sideW ? 1 : 0), Kp, (int) B);
    if (!p.strip && p.lazy && p.nsplitH == 1)
    {
      int wA = 0;
      const int sp = plan_tail(B, (int) T, (int) F, Kp, &wA);
      if (sp > 1)
      {
        const int G = ((int) T + 15) / 16, w = nmf_update5_strips((int) T, Kp, (int) B);
        p.tailSplitH = sp;
        p.tailStripsH = wA;
        p.tailRestH = w - wA;
        p.tailColsH = wA * ((G + w - 1) / w) * 16;
      }
    }
    if (!p.strip) p.stripsH = nmf_update5_strips((int) T, p.Kc, (int) B);
    // workspaces of the factor updates: split-contraction partials, denominators, column-sum pre-pass
    const int ns = std::max(std::max(p.nsplitW, p.nsplitH), p.tailSplitH);
    if (ns > 1 && !p.strip)
    {
      // partial numerators [B][pieces][rows][Kp]: the W update's rows are the bins, the H update's the frames (of the tail launch:
      // the frames behind the whole-contraction ones); the launches set UpdateArgs::Cp to their own row count
      const int64_t rowsW = p.nsplitW > 1 ? (int64_t) p.nsplitW * Fp : 0;
      const int64_t rowsH = p.nsplitH > 1 ? (int64_t) p.nsplitH * Tp : 0;
      const int64_t rowsT = p.tailSplitH > 1 ? (int64_t) p.tailSplitH * round_up(T - p.tailColsH, 32) : 0;
      p.partDoubles = B * std::max(std::max(rowsW, rowsH), rowsT) * Kp;
    }
    // (the tail launch keeps its denominator slots behind the first launch's B x Kp)
    p.dpartDoubles = std::max<int64_t>(32, B * std::max(ns, 1 + p.tailSplitH) * Kp);
    if (Kp > 64) p.csumDoubles = colsum_scratch_doubles((int) std::max(T, F), Kp, (int) B);
    if (p.lazy)
    {
      p.wscratchDoubles = wnorm_scratch_doubles(Kp, (int) B, p.stripsW);
      // column sums of the rows of W' every wavefront of a W update writes (UpdateArgs::colOut; rank 32, the two-launch iteration)
      if (Kp == 32 && !p.strip && p.stripsW > 0) p.colPartDoubles = B * p.stripsW * Kp;
    }
  }
}

static int plan_lists(fluhip_ctx* ctx, fluhip_corpus* c);
// the plan of the corpus's shape onto the corpus + its workspaces; needs B, T, F, Tp, Fp, K, Kp
int plan_updates(fluhip_ctx* ctx, fluhip_corpus* c)
{
  hipStream_t s = ctx->stream;
  UpdatePlan p;
  decide_update_plan(c->B, c->T, c->F, c->K, p);
  if (p.Kp != c->Kp) return fail(ctx, "internal error: the plan's padded rank is not the corpus's");
  c->nsplitW = p.nsplitW; c->nsplitH = p.nsplitH;
  c->lazy = p.lazy; c->sideW = p.sideW;
  c->strip = p.strip; c->stripBin = p.stripBin; c->stripTile = p.stripTile;
  c->stripsW = p.stripsW;
  c->tailSplitH = p.tailSplitH; c->tailStripsH = p.tailStripsH; c->tailRestH = p.tailRestH; c->tailColsH = p.tailColsH;
  c->Kc = p.Kc;
  if (p.useLists)
  {
    c->tOf.assign((size_t) c->B, (int) c->T);
    c->useLists = true;
    return plan_lists(ctx, c);
  }
  auto bytes = [](int64_t doubles) { return (size_t) doubles * sizeof(double); };
  if (p.wideDoubles) HIPCHK(ctx, c->wideScratch.alloc(bytes(p.wideDoubles), false, s));
  // (stripPart zeroed once: with the Nyquist bin as a side column only 16 values of its partial blocks are ever written, and the
  //  reduce launch adds the whole blocks before it masks the bins that do not exist)
  if (p.stripPartDoubles) HIPCHK(ctx, c->stripPart.alloc(bytes(p.stripPartDoubles), true, s));
#ifdef FLUHIP_AB_SWITCHES
  if (c->stripBin)
    HIPCHK(ctx, c->binWork.alloc((size_t) nmf_binstrip_doubles((int) c->F, (int) c->T, (int) c->B) * sizeof(double), true, s));
  if (c->stripTile)
    HIPCHK(ctx, c->tileWork.alloc((size_t) nmf_bintile_doubles((int) c->F, (int) c->B, nmf_strip_workgroups((int) c->T)) * sizeof(double), true, s));
#endif
  if (p.variant != 0)
  {
    if (p.partDoubles) HIPCHK(ctx, c->part.alloc(bytes(p.partDoubles), true, s));
    HIPCHK(ctx, c->dpart.alloc(bytes(p.dpartDoubles), true, s));
    if (p.csumDoubles) HIPCHK(ctx, c->csumScratch.alloc(bytes(p.csumDoubles), false, s));
  }
  HIPCHK(ctx, c->clk.alloc(8 * sizeof(long long), true, s));
  if (c->lazy && p.variant != 0)
  {
    HIPCHK(ctx, c->wnorm.alloc((size_t) c->B * c->Kp * sizeof(double), false, s));
    launch_fill_ones(c->wnorm.as<double>(), (int64_t) (c->B * c->Kp), s);
    HIPCHK(ctx, c->wscratch.alloc(bytes(p.wscratchDoubles), true, s));
    if (p.colPartDoubles) HIPCHK(ctx, c->colPart.alloc(bytes(p.colPartDoubles), true, s));
  }
  return FLUHIP_OK;
}

static int plan_lists(fluhip_ctx* ctx, fluhip_corpus* c)
{
  hipStream_t s = ctx->stream;
  const int B = (int) c->B, Kp = (int) c->Kp;
  c->lazy = true; c->strip = false;
  c->Kc = offsize_enabled() ? nmf_update5_compute_rank((int) c->K, Kp) : Kp; // (off-size ranks: see plan_updates)
  ListPlanHost plan;
  build_list_plan(c->tOf, (int) c->T, (int) c->F, Kp, plan);
  c->sideW = plan.sideW;
  auto take = [&](fluhip_corpus::WorkList& dst, const ListSide& src) -> int {
    dst.wgs = src.wgs; dst.ng = src.ng; dst.partial = src.partial; dst.maxSplit = src.maxSplit; dst.nPartials = src.nPartials;
    dst.statParts = src.statParts;
    if (int rc = upload_list(ctx, dst.list, src.list.data(), src.list.size() * sizeof(WaveDesc))) return rc;
    return upload_list(ctx, dst.splitTab, src.splitTab.data(), src.splitTab.size() * sizeof(int));
  };
  if (int rc = take(c->listW, plan.W)) return rc;
  if (int rc = take(c->listH, plan.H)) return rc;
  c->stripsW = plan.W.statParts;
  c->nsplitW = plan.W.pieces;
  c->nsplitH = plan.H.pieces;

  // workspaces
  const int64_t nPart = std::max(c->listW.nPartials, c->listH.nPartials);
  const size_t Cp = (size_t) std::max(c->Fp, c->Tp);
  if (nPart > 0) HIPCHK(ctx, c->part.alloc((size_t) nPart * Cp * Kp * sizeof(double), true, s));
  HIPCHK(ctx, c->dpart.alloc(std::max<size_t>(256, (size_t) std::max<int64_t>(nPart, B) * Kp * sizeof(double)), true, s));
  HIPCHK(ctx, c->clk.alloc(8 * sizeof(long long), true, s));
  HIPCHK(ctx, c->wnorm.alloc((size_t) B * Kp * sizeof(double), false, s));
  launch_fill_ones(c->wnorm.as<double>(), (int64_t) B * Kp, s);
  HIPCHK(ctx, c->wscratch.alloc((size_t) wnorm_scratch_doubles(Kp, B, c->stripsW) * sizeof(double), true, s));
  return FLUHIP_OK;
}

int corpus_alloc(fluhip_ctx* ctx, fluhip_corpus* c)
{
  hipStream_t s = ctx->stream;
  c->T = (c->n + c->hop) / c->hop; // alg/STFT.hpp:98-99; nrt/NMFClient.hpp:111-112
  c->F = c->fft / 2 + 1;
  c->Tp = round_up(c->T, 32);
  c->Fp = round_up(c->F, 32);
  c->Kp = padded_rank(c->K);
  const size_t B = (size_t) c->B;
  HIPCHK(ctx, c->mag.alloc(B * c->Tp * c->Fp * sizeof(double), true, s));
  // (a corpus that only ever transforms -- algorithm::STFT::process / magnitude at the algorithm level, fluhip_stft_* -- has one
  //  layout of the magnitudes and no factor workspaces: the bin-major copy exists for the H update alone)
  if (c->stftOnly) return FLUHIP_OK;
  HIPCHK(ctx, c->magT.alloc(B * c->Fp * c->Tp * sizeof(double), true, s));
  HIPCHK(ctx, c->Wf.alloc(B * c->Fp * c->Kp * sizeof(double), true, s));
  HIPCHK(ctx, c->H1.alloc(B * c->Tp * c->Kp * sizeof(double), true, s));
  HIPCHK(ctx, c->hmax.alloc(B * sizeof(double), true, s));
  if (c->ragged)
  {
    HIPCHK(ctx, c->nTab.alloc(B * sizeof(int64_t), false, s));
    HIPCHK(ctx, c->tTab.alloc(B * sizeof(int), false, s));
    HIPCHK(ctx, hipMemcpyAsync(c->nTab.p, c->nOf.data(), B * sizeof(int64_t), hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(c->tTab.p, c->tOf.data(), B * sizeof(int), hipMemcpyHostToDevice, s));
    c->useLists = true;
    return plan_lists(ctx, c);
  }
  if (int rc = plan_updates(ctx, c)) return rc;
  return FLUHIP_OK;
}

// launch-geometry limits of the factor-update paths, refused up front with a message instead of surfacing as an
// "invalid configuration" launch error: the normalisation kernels put the padded rank in one workgroup (<= 1024
// threads), and the any-rank path (rank above 128, kernels_nmf_wide.hip) puts frames / bins in gridDim.y (<= 65535)
int check_rank(fluhip_ctx* ctx, int64_t T, int64_t F, int64_t K)
{
  if (padded_rank(K) > 1024) return fail(ctx, "ranks above 1024 are not supported");
  if (padded_rank(K) > 128 && std::max(T, F) > 65535)
    return fail(ctx, "ranks above 128 are limited to 65535 frames and bins");
  return FLUHIP_OK;
}

int check_shape(fluhip_ctx* ctx, int64_t n, int64_t win, int64_t fft, int64_t hop, int64_t K)
{
  if (n <= 0) return fail(ctx, "not enough frames");
  if (win < 1 || hop < 1) return fail(ctx, "window and hop sizes must be positive");
  if (fft < 4 || (fft & (fft - 1)) || fft < win)
    return fail(ctx, "fft size must be a power of two >= window size");
  if (!stft_supported(win, fft))
    return fail(ctx, "fft sizes above 65536 are not supported");
  if (K < 1) return fail(ctx, "rank must be >= 1");
  if ((n + hop) / hop > 2000000000LL / 16) return fail(ctx, "too many frames");
  return check_rank(ctx, (n + hop) / hop, fft / 2 + 1, K);
}

int corpus_stft(fluhip_corpus* c, const float* a32, const double* a64, int64_t audioStride, bool magOnly)
{
  // magOnly: the frame-major magnitudes alone (what STFT::process + STFT::magnitude compute, alg/STFT.hpp:90-108, 61-66);
  // the bin-major copy is written for the factor updates only -- half the kernel's store traffic
  fluhip_ctx* ctx = c->ctx;
  const double *wtab = nullptr, *ttab = nullptr;
  int rc = get_window(ctx, c->win, c->fft, c->windowType, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, c->fft, &ttab);
  if (rc) return rc;
  if (c->keepSpec && !c->spec.p)
    HIPCHK(ctx, c->spec.alloc((size_t) c->B * c->T * c->F * 2 * sizeof(double), false, ctx->stream));
  StftArgs a;
  a.audio = a32; a.audio64 = a64; a.n = c->n; a.audioStride = audioStride;
  a.win = (int) c->win; a.fft = (int) c->fft; a.hop = (int) c->hop;
  a.T = (int) c->T; a.F = (int) c->F; a.B = (int) c->B;
  a.window = wtab; a.twiddle = ttab;
  a.mag = c->mag.as<double>(); a.magStride = c->Tp * c->Fp; a.ldMag = c->Fp;
  a.spec = c->keepSpec ? c->spec.as<double>() : nullptr; a.specStride = c->T * c->F * 2;
  a.frameOffset = 0;
  a.nTab = c->ragged ? c->nTab.as<int64_t>() : nullptr;
  a.bigScratch = big_fft_scratch(ctx, c->win, c->fft, c->B * c->T);
  if (stft_needs_scratch(c->win, c->fft) && !a.bigScratch) return FLUHIP_ERROR;
  // V is kept in both layouts (frame-major for the W update, bin-major for the H update).  The block form of K1
  // writes both in one pass; shapes it does not cover take the wave / generic kernel and a transposing copy.
  bool both = false;
  {
    ProfScope p(ctx, 0);
    if (!stft_needs_scratch(c->win, c->fft))
      both = launch_stft_block(a, magOnly ? nullptr : c->magT.as<double>(), c->Fp * c->Tp, c->Tp, ctx->stream);
    if (!both && c->ragged) return fail(ctx, "ragged corpora need an STFT shape with a block form (fft 1024 / 2048 / 4096, even window)");
    if (!both) launch_stft(a, ctx->stream);
  }
  if (!both && !magOnly)
  {
    ProfScope p(ctx, 4);
    launch_transpose(c->mag.as<double>(), c->Fp, c->Tp * c->Fp, c->magT.as<double>(), c->Tp,
                     c->Fp * c->Tp, (int) c->T, (int) c->F, (int) c->B, ctx->stream);
  }
  HIPCHK(ctx, hipGetLastError());
  c->haveMag = !magOnly;   // (the factor updates need both layouts)
  c->touched = true;
  return FLUHIP_OK;
}

// util/EigenRandom.hpp:73-101: std::mt19937_64 g{seed ? *seed : rd()} +
// std::uniform_real_distribution<double>{0, 1}; one draw per coefficient in Eigen's column-major
// linear order.  libstdc++'s <random> is used verbatim, exactly as the reference does.
void draw_uniform(int64_t seed, size_t count, std::vector<double>& out)
{
  std::random_device rd;
  std::mt19937_64 g{seed >= 0 ? (size_t) seed : (size_t) rd()};
  std::uniform_real_distribution<double> d{0.0, 1.0};
  out.resize(count);
  for (size_t i = 0; i < count; i++) out[i] = d(g);
}


int corpus_init_factors(fluhip_corpus* c, int64_t seed, const int64_t* seeds,
                               const FactorInit& fi)
{
  fluhip_ctx* ctx = c->ctx;
  hipStream_t s = ctx->stream;
  const size_t FK = (size_t) c->F * c->K, TK = (size_t) c->T * c->K;
  const int B = (int) c->B;
  // The draws run on the calling thread while the device is still busy with whatever was enqueued before (the STFT
  // of this job); the host images stay alive to the single synchronisation at the end, so the two uploads and the
  // scatter kernels queue up behind each other without a host round trip in between.
  std::vector<double> hostW, hostH;
  DevBuf stageH;
  // --- W ---
  if (fi.W0f32)
  {
    HIPCHK(ctx, c->stage.alloc((size_t) B * FK * sizeof(float), false, s));
    HIPCHK(ctx, hipMemcpyAsync(c->stage.p, fi.W0f32, (size_t) B * FK * sizeof(float), hipMemcpyHostToDevice, s));
    launch_scatter_factor_f32(c->stage.as<float>(), (int64_t) FK, c->Wf.as<double>(), c->Fp * c->Kp,
                              (int) c->F, (int) c->K, (int) c->Kp, B, s);
  }
  else
  {
    std::vector<double>& host = hostW;
    const double* src = fi.W0host;
    int nsrc = fi.W0host ? (fi.sharedW ? 1 : B) : 1;
    if (!src)
    {
      if (seeds)
      {
        nsrc = B;
        host.resize((size_t) B * FK);
        std::vector<double> tmp;
        std::map<int64_t, int> seen;
        for (int b = 0; b < B; b++)
        {
          auto it = seeds[b] >= 0 ? seen.find(seeds[b]) : seen.end();
          if (it != seen.end())
            std::memcpy(&host[(size_t) b * FK], &host[(size_t) it->second * FK], FK * sizeof(double));
          else
          {
            draw_uniform(seeds[b], FK, tmp);
            std::memcpy(&host[(size_t) b * FK], tmp.data(), FK * sizeof(double));
            if (seeds[b] >= 0) seen[seeds[b]] = b;
          }
        }
      }
      else
        draw_uniform(seed, FK, host); // alg/NMF.hpp:104-105
      src = host.data();
    }
    HIPCHK(ctx, c->stage.alloc((size_t) nsrc * FK * sizeof(double), false, s));
    HIPCHK(ctx, hipMemcpyAsync(c->stage.p, src, (size_t) nsrc * FK * sizeof(double), hipMemcpyHostToDevice, s));
    // K x F row-major source (random: column-major F x K fill; seeded: W0 transposed, :102-112)
    launch_scatter_factor(c->stage.as<double>(), nsrc == 1 ? 0 : (int64_t) FK, c->Wf.as<double>(),
                          c->Fp * c->Kp, (int) c->F, (int) c->K, (int) c->Kp, B, true, s);
  }
  // --- H ---
  if (fi.H0f32)
  {
    HIPCHK(ctx, stageH.alloc((size_t) B * TK * sizeof(float), false, s));
    HIPCHK(ctx, hipMemcpyAsync(stageH.p, fi.H0f32, (size_t) B * TK * sizeof(float), hipMemcpyHostToDevice, s));
    launch_scatter_factor_f32(stageH.as<float>(), (int64_t) TK, c->H1.as<double>(), c->Tp * c->Kp,
                              (int) c->T, (int) c->K, (int) c->Kp, B, s, c->ragged ? c->tTab.as<int>() : nullptr);
  }
  else
  {
    std::vector<double>& host = hostH;
    const double* src = fi.H0host;
    int nsrc = fi.H0host ? (fi.sharedH ? 1 : B) : 1;
    if (!src)
    {
      if (seeds)
      {
        nsrc = B;
        host.resize((size_t) B * TK);
        std::vector<double> tmp;
        std::map<int64_t, int> seen;
        for (int b = 0; b < B; b++)
        {
          auto it = seeds[b] >= 0 ? seen.find(seeds[b]) : seen.end();
          if (it != seen.end())
            std::memcpy(&host[(size_t) b * TK], &host[(size_t) it->second * TK], TK * sizeof(double));
          else
          {
            draw_uniform(seeds[b], TK, tmp);
            std::memcpy(&host[(size_t) b * TK], tmp.data(), TK * sizeof(double));
            if (seeds[b] >= 0) seen[seeds[b]] = b;
          }
        }
      }
      else
        draw_uniform(seed, TK, host); // alg/NMF.hpp:116-117 (a fresh generator from the same seed)
      src = host.data();
    }
    HIPCHK(ctx, stageH.alloc((size_t) nsrc * TK * sizeof(double), false, s));
    HIPCHK(ctx, hipMemcpyAsync(stageH.p, src, (size_t) nsrc * TK * sizeof(double), hipMemcpyHostToDevice, s));
    // T x K row-major source (random: column-major K x T fill; seeded: H0 transposed, :113-124)
    launch_scatter_factor(stageH.as<double>(), nsrc == 1 ? 0 : (int64_t) TK, c->H1.as<double>(),
                          c->Tp * c->Kp, (int) c->T, (int) c->K, (int) c->Kp, B, false, s,
                          c->ragged ? c->tTab.as<int>() : nullptr); // (K x T_b column-major = the first T_b K draws)
  }
  // alg/NMF.hpp:150-153: clamp both to eps, normalise columns of W and rows of H (= columns of H1)
  if (!c->normScratch.p)
  {
    const size_t nd = (size_t) colnorm_scratch_doubles((int) std::max(c->T, c->F), (int) c->Kp, B);
    HIPCHK(ctx, c->normScratch.alloc(nd * sizeof(double), false, s));
  }
  launch_colnorm(c->H1.as<double>(), c->Tp * c->Kp, (int) c->T, (int) c->K, (int) c->Kp, B, true, false,
                 c->normScratch.as<double>(), s, c->ragged ? c->tTab.as<int>() : nullptr);
  launch_colnorm(c->Wf.as<double>(), c->Fp * c->Kp, (int) c->F, (int) c->K, (int) c->Kp, B, true, false,
                 c->normScratch.as<double>(), s);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(s)); // the host images and the second staging buffer go out of scope
  c->haveFactors = true;
  c->stripStatsValid = false;
  return FLUHIP_OK;
}

// the second stream and the fork / join events of the W update's side column (created on first use; false: stay on one stream)
constexpr bool kSideBesideDefault = false;
constexpr bool kSideFirstCorpora = false;
static bool side_stream_ready(fluhip_ctx* ctx)
{
  if (ctx->sideStream) return ctx->sideEv[7] != nullptr;
  if (hipStreamCreateWithFlags(&ctx->sideStream, hipStreamNonBlocking) != hipSuccess) { ctx->sideStream = nullptr; return false; }
  for (auto& e : ctx->sideEv)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
  return true;
}

static void enqueue_iteration(fluhip_corpus* c, bool updateW, bool updateH, bool last)
{
  fluhip_ctx* ctx = c->ctx;
  hipStream_t s = ctx->stream;
  const int B = (int) c->B;
  if (c->strip)
  {
    StripArgs a;
    a.V = c->mag.as<double>(); a.strideV = c->Tp * c->Fp; a.ldv = c->Fp;
    a.W = c->Wf.as<double>(); a.strideW = c->Fp * c->Kp;
    a.H = c->H1.as<double>(); a.strideH = c->Tp * c->Kp;
    a.part = c->stripPart.as<double>(); a.nrm = c->wnorm.as<double>();
    a.F = (int) c->F; a.T = (int) c->T; a.K = (int) c->K; a.B = B;
    a.doH = a.doW = 0; a.wPend = c->wPending ? 1 : 0;
#ifdef FLUHIP_AB_SWITCHES
    BinTileArgs bt{};
    if (c->stripTile)
    {
      bt.VT = c->magT.as<double>(); bt.strideVT = c->Fp * c->Tp; bt.ldT = c->Tp;
      bt.W = a.W; bt.strideW = a.strideW; bt.H = a.H; bt.strideH = a.strideH;
      bt.work = c->tileWork.as<double>(); bt.nSideWG = nmf_strip_workgroups((int) c->T);
      bt.F = a.F; bt.K = a.K; bt.B = B;
      a.nRec = nmf_bintile_records(a.F); a.sideOut = nmf_bintile_side_area(bt);
    }
    // (the strip launches of the tiled form read the tile records of the generation current at their launch)
    auto tile_gen = [&]() { if (c->stripTile) a.tileStat = nmf_bintile_records_ptr(bt, c->stripGen); };
#else
    auto tile_gen = [] {};
#endif
    if (!c->stripStatsValid)
    {
      // W was written by something else (initialisation, the normalisation at the end of the last call)
      a.statGen = c->stripGen;
#ifdef FLUHIP_AB_SWITCHES
      if (c->stripTile) { bt.statGen = c->stripGen; bt.wPend = 0; launch_nmf_bintile_wstats(bt, s); }
      else
#endif
      launch_nmf_strip_wstats(a, s);
      c->stripStatsValid = true;
    }
#ifdef FLUHIP_AB_SWITCHES
    if (updateW && c->stripTile)
    {
      // alg/NMF.hpp:158-161 as ONE launch over tiles of four bins; :162 stays deferred (tile records of the other generation)
      if (!c->stripSideReady)
      {
        // the Nyquist bin's numerator partials from the H in memory (first iteration of a call, W-only iterations)
        a.doH = 0; a.doW = 2; a.wPend = c->wPending ? 1 : 0; a.statGen = c->stripGen; tile_gen();
        ProfScope p(ctx, 3);
        launch_nmf_strip(a, s);
      }
      bt.sidePart = a.sideOut; bt.statGen = c->stripGen; bt.wPend = c->wPending ? 1 : 0;
      {
        ProfScope p(ctx, 1);
        launch_nmf_bintile(bt, s);
      }
      c->stripGen ^= 1;
      c->wPending = true;
      c->stripSideReady = false;
      c->stripReady = false;
      c->stripNormFresh = false;
    }
    else
#endif
    if (updateW && c->stripBin)
    {
      // alg/NMF.hpp:158-161 as one launch over bin strips: reads W', its records and the H in memory, leaves the new W' and
      // its records (other generation); :162 stays deferred as in the fused form
      a.wPend = c->wPending ? 1 : 0; a.statGen = c->stripGen;
      {
        ProfScope p(ctx, 1);
#ifdef FLUHIP_AB_SWITCHES
        launch_nmf_binstrip(a, c->binWork.as<double>(), s);
#endif
      }
      c->stripGen ^= 1;
      c->wPending = true;
      c->stripReady = false;
      c->stripNormFresh = false;
    }
    else if (updateW)
    {
      // alg/NMF.hpp:158-161; :162 is implicit in the next staging of W'
      if (!c->stripReady)
      {
        a.doH = 0; a.doW = 1; a.wPend = c->wPending ? 1 : 0; a.statGen = c->stripGen;
        ProfScope p(ctx, 1);
        launch_nmf_strip(a, s);
      }
      {
        a.wPend = c->wPending ? 1 : 0; a.statGen = c->stripGen;
        ProfScope p(ctx, 3);
        launch_nmf_strip_reduce(a, s);
        c->stripGen ^= 1;
      }
      c->wPending = true;
      c->stripReady = false;
      c->stripNormFresh = false;
    }
    if (updateH)
    {
      // :165-170, and behind it the numerator of the next iteration's W update while the new H is at hand
      a.doH = 1; a.doW = (updateW && !last && !c->stripBin) ? (c->stripTile ? 2 : 1) : 0; a.wPend = c->wPending ? 1 : 0; a.statGen = c->stripGen;
      tile_gen();
      ProfScope p(ctx, 1);
      launch_nmf_strip(a, s);
      c->stripReady = a.doW == 1;
      c->stripSideReady = a.doW == 2;
      c->stripNormFresh = true;
    }
    return;
  }
  // (a window of the corpus: every per-buffer array starts b0 buffers in; the scratch of the small kernels is reused)
  const int64_t b0 = c->winB ? c->winB0 : 0;
  const int Bw = c->winB ? (int) c->winB : B;
  double* const magW = c->mag.as<double>() + b0 * c->Tp * c->Fp;
  double* const magTW = c->magT.as<double>() + b0 * c->Fp * c->Tp;
  double* const WfW = c->Wf.as<double>() + b0 * c->Fp * c->Kp;
  double* const H1W = c->H1.as<double>() + b0 * c->Tp * c->Kp;
  double* const wnormW = c->wnorm.p ? c->wnorm.as<double>() + b0 * c->Kp : nullptr;
  // the H update's arguments (alg/NMF.hpp:165-170; V2 is formed from the already updated W) -- also wanted by the W update's
  // step, which asks what the H launch behind it will take over (dryRun)
  auto h_args = [&]() {
    UpdateArgs a;
    a.V = magTW; a.ldv = c->Tp; a.strideV = c->Fp * c->Tp;
    a.Mv = WfW; a.strideM = c->Fp * c->Kp;
    a.S = H1W; a.strideS = c->Tp * c->Kp;
    a.R = (int) c->F; a.C = (int) c->T; a.B = Bw; a.Kp = (int) c->Kp; a.Kc = c->Kc;
    if (c->winB) a.stripsOverride = c->winStripsH;
    a.nsplit = c->nsplitH; a.part = c->part.as<double>(); a.dpart = c->dpart.as<double>();
    a.Cp = c->useLists ? std::max(c->Fp, c->Tp) : c->Tp; a.colsumScratch = c->csumScratch.as<double>();
    a.clk = c->clk.as<long long>() + 4;
    return a;
  };
  const int uvH = update_variant((int) c->Kp);
  // (the plain one-launch form of the H update: no work lists, no second launch for a poorly filled last round)
  const bool hPlain = !c->useLists && uvH == 5 && !(c->tailSplitH > 1 && !c->winB);
  // FLUHIP_SIDE_FROM_H=0 (A/B build): side-column launch and norm-combine launch between the updates, as before round 4
  static const bool fromH = [] { const char* e = fluhip::ab_getenv("FLUHIP_SIDE_FROM_H"); return e ? std::atoi(e) != 0 : true; }();
  static const bool normInH = [] { const char* e = fluhip::ab_getenv("FLUHIP_NORM_IN_H"); return e ? std::atoi(e) != 0 : true; }();
  auto side_io = [&](UpdateArgs& a, bool wantNorm) {
    double* scr = c->wscratch.as<double>();
    const int outGen = c->sideGen ^ 1;
    a.sideOut = wnorm_side_part(scr, (int) c->Kp, Bw, c->stripsW, outGen);
    a.sideWold = wnorm_side_wold(scr, (int) c->Kp, Bw, c->stripsW, outGen);
    if (wantNorm)
    {
      a.cmbStat = scr; a.cmbParts = c->stripsW;
      a.cmbSide = wnorm_side_part(scr, (int) c->Kp, Bw, c->stripsW, c->sideGen);
      a.cmbWold = wnorm_side_wold(scr, (int) c->Kp, Bw, c->stripsW, c->sideGen);
      a.cmbSlices = c->sideFromHSlices; a.cmbK = (int) c->K;
      a.cmbNrmOut = wnormW; a.cmbRowOut = WfW;
    }
  };
  if (updateW)
  {
    // alg/NMF.hpp:158-161
    UpdateArgs a;
    a.V = magW; a.ldv = c->Fp; a.strideV = c->Tp * c->Fp;
    a.Mv = H1W; a.strideM = c->Tp * c->Kp;
    a.S = WfW; a.strideS = c->Fp * c->Kp;
    a.R = (int) c->T; a.C = (int) c->F - (c->sideW ? 1 : 0); a.B = Bw; a.Kp = (int) c->Kp; a.Kc = c->Kc;
    if (c->winB) a.stripsOverride = c->winStripsW;
    a.nsplit = c->nsplitW; a.part = c->part.as<double>(); a.dpart = c->dpart.as<double>();
    a.Cp = c->useLists ? std::max(c->Fp, c->Tp) : c->Fp; a.colsumScratch = c->csumScratch.as<double>();
    a.clk = c->clk.as<long long>();
    if (c->useLists)
    {
      a.list = c->listW.list.as<WaveDesc>(); a.listWGs = c->listW.wgs; a.listNG = c->listW.ng; a.listPartial = c->listW.partial;
    }
    if (c->lazy)
    {
      // W' = W diag(wnorm) in memory: the kernel divides its stationary rows by wnorm, writes the new W'
      // and its per-wavefront column statistics; [side column ->] new wnorm.  alg/NMF.hpp:162 is then
      // implicit in every later use of (W', wnorm).
      a.nrm = wnormW; a.nrmMode = 1; a.statPart = c->wscratch.as<double>();
      SideColumn sc{magTW + (c->F - 1) * c->Tp, c->Fp * c->Tp, H1W, c->Tp * c->Kp, (int) c->T};
      // The side column (bin F-1 of W, scalar FMAs) reads H, V, the old W row and the old norms -- nothing the update launch
      // writes -- so it runs BESIDE that launch on a second stream: forked behind the H update in front, joined before the norm
      // combine.  The update launch is enqueued first; its wavefronts are the older ones on every SIMD and are served first.
      static const bool sideBeside = [] { const char* e = fluhip::ab_getenv("FLUHIP_SIDE_STREAM"); return e ? std::atoi(e) != 0 : kSideBesideDefault; }();
      const bool sideReady = c->sideFromH;
      c->sideFromH = false;
      hipEvent_t join = nullptr;
      if (c->sideW && !sideReady && sideBeside && side_stream_ready(ctx))
      {
        hipEvent_t fork = ctx->sideEv[(ctx->sideTurn * 2) % 8];
        join = ctx->sideEv[(ctx->sideTurn * 2 + 1) % 8];
        ctx->sideTurn++;
        if (hipEventRecord(fork, s) != hipSuccess || hipStreamWaitEvent(ctx->sideStream, fork, 0) != hipSuccess) join = nullptr;
      }
      // Arrays of rank 128 at compute ranks above 104 take the update's column sums of H from a pre-pass -- and the side
      // column's denominator is that same sum.  There the side-column launch goes IN FRONT of the update (it reads H, V, the old
      // side row and the old norms: nothing the update writes) and its slices' denominators are added into the update's
      // denominator slots: one sweep over H instead of two (config 3: colsum_part_kernel's 11 - 16 us per iteration gone).
      // FLUHIP_COLSUM_FROM_SIDE=0 (A/B build): the pre-pass as before.
      static const bool csFromSide = [] { const char* e = fluhip::ab_getenv("FLUHIP_COLSUM_FROM_SIDE"); return e ? std::atoi(e) != 0 : true; }();
      static const bool sideFused = [] { const char* e = fluhip::ab_getenv("FLUHIP_SIDE_FUSED"); return e && std::atoi(e) == 1; }();   // (A/B: the combine rides in the side-column launch, which must then follow the update)
      // FLUHIP_SIDE_FIRST_CORPORA=0|1 (A/B build): whether corpora whose side column and norm combine are ONE launch behind the
      // update (side_norm_kernel) give that up for the side-first order
      static const bool sideFirstCorpora = [] { const char* e = fluhip::ab_getenv("FLUHIP_SIDE_FIRST_CORPORA"); return e ? std::atoi(e) != 0 : kSideFirstCorpora; }();
      const bool sideFirst = csFromSide && !sideFused && c->sideW && !sideReady && !join && !c->useLists && c->Kp == 128 &&
                             (c->Kc <= 0 || c->Kc > 104) && a.colsumScratch && update_variant(a.Kp) == 5 &&
                             (sideFirstCorpora || !wnorm_side_norm_shape(Bw, c->stripsW, sc.R, (int) c->Kp));
      if (sideFirst)
      {
        ProfScope p(ctx, 3);
        launch_wnorm_combine(WfW, c->Fp * c->Kp, (int) c->F, (int) c->K, (int) c->Kp, Bw, c->stripsW,
                             c->wscratch.as<double>(), wnormW, &sc, s, 1);
        const int ns = a.nsplit < 1 ? 1 : a.nsplit;
        launch_colsum_from_side(c->wscratch.as<double>(), (int) c->Kp, Bw, c->stripsW, wnorm_side_slices(sc.R, (int) c->Kp),
                                a.dpart, (int64_t) ns * c->Kp, ns - 1, s);
        a.colsumInPlace = true;
      }
      // rank 32, steady state of the two-launch iteration: the H update in front left the side column's partials, whose
      // denominators are the column sums of H this update divides by -- no accumulators in its loop; it leaves the column sums
      // of its rows for the H update behind (the launcher decides whether the form exists for these arguments: bit 2)
      if (sideReady && c->sideW && !c->useLists && c->colPart.p && c->sideFromHSlices > 0 && update_variant(a.Kp) == 5)
      {
        a.colIn = wnorm_side_part(c->wscratch.as<double>(), (int) c->Kp, Bw, c->stripsW, c->sideGen);
        a.colInN = c->sideFromHSlices;
        a.colOut = c->colPart.as<double>();
      }
      {
        ProfScope p(ctx, 1);
        c->colPartValid = (launch_nmf_update5(a, s) & 4) != 0;
        if (c->useLists && c->listW.partial)
          launch_update_finalize(a.S, a.strideS, a.part, a.dpart, a.C, a.Kp, a.Cp, c->listW.maxSplit, a.B, s, a.nrm, a.nrmMode,
                                 a.statPart, c->listW.splitTab.as<int>());
      }
      if (join)
      {
        launch_wnorm_combine(WfW, c->Fp * c->Kp, (int) c->F, (int) c->K, (int) c->Kp, Bw, c->stripsW,
                             c->wscratch.as<double>(), wnormW, &sc, ctx->sideStream, 1);
        (void) hipEventRecord(join, ctx->sideStream);
        (void) hipStreamWaitEvent(s, join, 0);
      }
      c->normDue = false;
      if (c->sideW && sideReady && updateH && hPlain && normInH && c->stripsW <= 16 && c->sideFromHSlices <= 16)
      {
        // ... and the norm combine too can wait for the H update behind this launch, whose wavefronts then do it in their
        // prologue -- if the form that launch will take does that (dryRun: nothing is launched)
        UpdateArgs ah = h_args();
        ah.nrm = wnormW; ah.nrmMode = 2; ah.dryRun = true;
        side_io(ah, true);
        c->normDue = (launch_nmf_update5(ah, s) & 2) != 0;
      }
      if (!c->normDue)   // (nothing between the updates otherwise: no event pair of the profiling aid there either)
      {
        ProfScope p(ctx, 3);
        if (c->sideW && sideReady)   // the H update in front left the side column's partials: the norm combine is all that is due
          launch_wnorm_combine(WfW, c->Fp * c->Kp, (int) c->F, (int) c->K, (int) c->Kp, Bw, c->stripsW,
                               c->wscratch.as<double>(), wnormW, &sc, s, 2, c->sideFromHSlices, c->sideGen);
        else
        {
          // ... and where the H update behind takes ITS column sums (of the new W') from a pre-pass, the combine launch leaves
          // them in that update's denominator slots (long factors: launch_wnorm_combine says whether it did)
          WnormColsum wc;
          if (csFromSide && updateH && !c->useLists && uvH == 5 && c->Kp == 128 && (c->Kc <= 0 || c->Kc > 104) && c->csumScratch.p)
          {
            double* dp = c->dpart.as<double>();
            if (c->tailSplitH > 1 && !c->winB)
            {
              wc.out1 = dp; wc.stride1 = c->Kp; wc.zero1 = 0;
              wc.out2 = dp + (int64_t) Bw * c->Kp; wc.stride2 = (int64_t) c->tailSplitH * c->Kp; wc.zero2 = c->tailSplitH - 1;
            }
            else
            {
              const int nsH = c->nsplitH < 1 ? 1 : c->nsplitH;
              wc.out1 = dp; wc.stride1 = (int64_t) nsH * c->Kp; wc.zero1 = nsH - 1;
            }
          }
          c->colsumWInPlace = launch_wnorm_combine(WfW, c->Fp * c->Kp, (int) c->F, (int) c->K, (int) c->Kp, Bw, c->stripsW,
                                                   c->wscratch.as<double>(), wnormW, c->sideW ? &sc : nullptr, s,
                                                   (join || sideFirst) ? 2 : 0, 0, -1, wc.out1 ? &wc : nullptr);
        }
      }
      c->wPending = true;
    }
    else
    {
      {
        ProfScope p(ctx, 1);
        if (update_variant(a.Kp) == 5) launch_nmf_update5(a, s);
        else launch_nmf_update_wide(a, c->wideScratch.as<double>(), s);
      }
      // :162  if (W.maxCoeff() > epsilon) W.colwise().normalize()
      launch_colnorm(WfW, c->Fp * c->Kp, (int) c->F, (int) c->K, (int) c->Kp, Bw, false, true,
                     c->normScratch.as<double>(), s);
    }
  }
  if (updateH)
  {
    UpdateArgs a = h_args();
    const bool colValid = c->colPartValid;
    c->colPartValid = false;
    if (c->wPending) { a.nrm = wnormW; a.nrmMode = 2; }
    const bool csIn = c->colsumWInPlace;   // (the column sums of W' are in this update's denominator slots already)
    c->colsumWInPlace = false;
    ProfScope p(ctx, 1);
    const int uv = uvH;
    if (c->useLists)
    {
      a.list = c->listH.list.as<WaveDesc>(); a.listWGs = c->listH.wgs; a.listNG = c->listH.ng; a.listPartial = c->listH.partial;
      launch_nmf_update5(a, s);
      if (c->listH.partial)
        launch_update_finalize(a.S, a.strideS, a.part, a.dpart, a.C, a.Kp, a.Cp, c->listH.maxSplit, a.B, s, a.nrm, a.nrmMode,
                               nullptr, c->listH.splitTab.as<int>());
    }
    else if (uv == 5 && c->tailSplitH > 1 && !c->winB)
    {
      // two launches (plan_tail): whole contractions for the frames that fill whole rounds, split ones for the rest
      UpdateArgs a1 = a;
      a1.C = c->tailColsH; a1.stripsOverride = c->tailStripsH; a1.nsplit = 1;
      a1.colsumInPlace = csIn;
      launch_nmf_update5(a1, s);
      UpdateArgs a2 = a;
      a2.V = a.V + c->tailColsH; a2.S = a.S + (int64_t) c->tailColsH * c->Kp; a2.C = a.C - c->tailColsH;
      a2.stripsOverride = c->tailRestH; a2.nsplit = c->tailSplitH; a2.Cp = round_up(a2.C, 32);
      a2.dpart = a.dpart + (int64_t) Bw * c->Kp;
      if (csIn) a2.colsumInPlace = true;
      else if (c->Kp > 64) a2.colsumGiven = a.dpart; // (the first launch's pre-pass left the column sums of W there)
      a2.clk = nullptr;                         // the clock stamps stay those of the whole-contraction wavefront
      launch_nmf_update5(a2, s);
    }
    else if (uv == 5)
    {
      // a W update follows (same call, same window) and keeps bin F - 1 as a side column: this launch, which holds the new H
      // in registers at its end, leaves that side column's contraction as per-wavefront partials (UpdateArgs::sideOut) and the
      // side-column launch -- a second pass over H, 10 us per iteration of the bench shard -- is not needed.
      // FLUHIP_SIDE_FROM_H=0 (A/B build): the side-column launch as before.
      // (the norm combine of the W update in front, if that was left to this launch: c->normDue)
      const bool wantNorm = c->normDue;
      a.colsumInPlace = csIn;
      if (wantNorm && updateW && colValid) { a.colIn = c->colPart.as<double>(); a.colInN = c->stripsW; }
      if (wantNorm || (fromH && updateW && !last && c->lazy && c->sideW && c->wPending && c->nsplitH == 1)) side_io(a, wantNorm);
      const int did = launch_nmf_update5(a, s);
      if (wantNorm && !(did & 2)) c->planError = true; // (cannot happen: the dry run in the W update's step took the same arguments)
      c->normDue = false;
      c->sideFromH = (did & 1) != 0 && updateW && !last;
      if (did & 1)
      {
        const int G = ((int) c->T + 15) / 16;
        c->sideGen ^= 1;
        c->sideFromHSlices = a.stripsOverride > 0 ? std::min(a.stripsOverride, G) : nmf_update5_strips((int) c->T, (int) c->Kp, Bw);
      }
    }
    else launch_nmf_update_wide(a, c->wideScratch.as<double>(), s);
  }
}

// alg/NMF.hpp:154-181 loop + :175-176 callbacks.  Iterations are enqueued back to back; with a progress callback
// every iteration is followed by an event and the callback is invoked once per completed iteration, in order, on
// the calling thread, with a bounded run-ahead.
static int corpus_iterate_loop(fluhip_corpus* c, int64_t iters, bool updateW, bool updateH,
                               fluhip_progress_fn progress, void* user)
{
  fluhip_ctx* ctx = c->ctx;
  c->sideFromH = false; // (side-column partials an H update leaves are only ever used by the W update enqueued right behind it)
  c->colPartValid = false;
  c->colsumWInPlace = false;
  c->normDue = false;
  if (!progress)
  {
    // Corpora of several rounds of wavefronts (more than 1024 / strips buffers) run ROUND-MAJOR: all iterations of the
    // first 128 buffers (at 8 strips), then of the next 128, ...  Buffers are independent jobs (nrt/NMFClient.hpp:233),
    // so the order is free, and a round's factor matrices (63 MB at the bench shape) stay in the 256 MB last-level cache
    // from launch to launch, which the whole corpus's (500 MB at 1024 buffers) do not: 8-round launches measured 0.304 -
    // 0.319 ms per round against 0.271 - 0.278 for one-round launches (profiles/r03/bench_v*_1024_buffers_one_gpu.json).
    // With a progress callback the iterations stay outermost: "iteration i" means every buffer has passed it.
    // FLUHIP_ROUND_MAJOR=0: iteration-major as before.
    static const int roundEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_ROUND_MAJOR"); return e ? std::atoi(e) : 1; }();
    int64_t chunk = 0;
    if (roundEnv && !c->strip && !c->useLists && c->lazy && c->nsplitW == 1 && c->nsplitH == 1 && update_variant((int) c->Kp) == 5)
    {
      const int CW = (int) c->F - (c->sideW ? 1 : 0);
      const int wW = nmf_update5_strips(CW, (int) c->Kp, (int) c->B), wH = nmf_update5_strips((int) c->T, (int) c->Kp, (int) c->B);
      const int64_t per = 1024 / std::max(wW, wH);
      if (per >= 8 && c->B >= 2 * per)
      {
        // the schedule of ONE round of `per` buffers, kept for every window (the update kernel would otherwise deal its
        // strips from the number of buffers it is launched on); the W update's strips also fix the layout of its column
        // statistics, which was planned for the whole corpus
        const int wWc = nmf_update5_strips(CW, (int) c->Kp, (int) per), wHc = nmf_update5_strips((int) c->T, (int) c->Kp, (int) per);
        // (both updates must fill most of a round at that window: short buffers at a wide rank -- 128 x 1.6 s, rank 128: 32
        //  strips over the bins, 5 over the frames -- ran H updates of 160 wavefronts per window: 1 510 us per iteration against
        //  608 iteration-major, tools/wide_list_ab.sh)
        if (wWc == wW && per * std::max(wWc, wHc) <= 1024 && per * std::min(wWc, wHc) >= 768)
        {
          chunk = per; c->winStripsW = wWc; c->winStripsH = wHc;
        }
      }
    }
    if (chunk > 0)
    {
      for (int64_t b0 = 0; b0 < c->B; b0 += chunk)
      {
        c->winB0 = b0;
        c->winB = std::min(chunk, c->B - b0);
        c->sideFromH = false;
        for (int64_t i = 0; i < iters; i++) enqueue_iteration(c, updateW, updateH, i + 1 == iters);
      }
      c->winB0 = c->winB = 0;
    }
    else
    {
      // FLUHIP_GRAPH_ITERS=n (experiment, default off -- DESIGN.md "hipGraph"): after the first iteration, runs of n iterations
      // are captured once as a hipGraph and replayed; the last iterations (fewer than n + 1) are enqueued as usual.  n even: the
      // frame-strip schedule alternates two statistics buffers.  Not with the profiler's events in the stream.
      static const int graphN = [] { const char* e = fluhip::ab_getenv("FLUHIP_GRAPH_ITERS"); return e ? std::atoi(e) & ~1 : 0; }();
      int64_t i = 0;
      if (graphN >= 2 && !ctx->prof && iters >= 2 * (int64_t) graphN + 2)
      {
        enqueue_iteration(c, updateW, updateH, false);
        i = 1;
        // whatever fails in here, the stream leaves capture mode and the graph objects are released (an early
        // return between Begin- and EndCapture left the context's stream capturing, and every later call on it failed)
        struct Capture
        {
          hipStream_t s;
          hipGraph_t graph = nullptr;
          hipGraphExec_t exec = nullptr;
          bool capturing = false;
          ~Capture()
          {
            if (capturing) (void) hipStreamEndCapture(s, &graph);
            if (exec) (void) hipGraphExecDestroy(exec);
            if (graph) (void) hipGraphDestroy(graph);
          }
        } cap{ctx->stream};
        HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        cap.capturing = true;
        for (int g = 0; g < graphN; g++) enqueue_iteration(c, updateW, updateH, false);
        cap.capturing = false;
        HIPCHK(ctx, hipStreamEndCapture(ctx->stream, &cap.graph));
        HIPCHK(ctx, hipGraphInstantiate(&cap.exec, cap.graph, nullptr, nullptr, 0));
        for (; i + graphN < iters; i += graphN) HIPCHK(ctx, hipGraphLaunch(cap.exec, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // (experiment: the graph is dropped right away)
      }
      for (; i < iters; i++) enqueue_iteration(c, updateW, updateH, i + 1 == iters);
    }
    HIPCHK(ctx, hipGetLastError());
    if (c->planError) return fail(ctx, "internal error: an update launch did not take the form its dry run announced");
    return FLUHIP_OK;
  }
  // One event per iteration; callbacks are delivered in order as the events complete, and the host never runs more
  // than kLag iterations ahead of the last one it has reported: a cancel at iteration i stops the device after at
  // most kLag - 1 further iterations (alg/NMF.hpp:175-176 stops at i exactly; the client above never looks at the
  // factors of a cancelled job, clients/nrt/NMFClient.hpp:273-274).
  constexpr int kMaxLag = 64;
  const int kLag = std::max(1, std::min(kMaxLag, ctx->progressLag));
  hipEvent_t ev[kMaxLag];
  for (int i = 0; i < kLag; i++) ev[i] = take_event(ctx);
  auto give_back = [&] { for (int i = 0; i < kLag; i++) ctx->eventPool.push_back(ev[i]); };
  int64_t reported = 0;
  for (int64_t i = 0; i < iters; i++)
  {
    enqueue_iteration(c, updateW, updateH, i + 1 == iters);
    if (hipEventRecord(ev[i % kLag], ctx->stream) != hipSuccess) { give_back(); return fail(ctx, "HIP error: hipEventRecord"); }
    const int64_t enq = i + 1;
    while (reported < enq)
    {
      hipEvent_t e = ev[reported % kLag];
      const bool mustWait = enq - reported >= kLag || enq == iters;
      hipError_t q = mustWait ? hipEventSynchronize(e) : hipEventQuery(e);
      if (q == hipErrorNotReady) break;
      if (q != hipSuccess) { give_back(); return fail(ctx, std::string("HIP error: ") + hipGetErrorString(q) + " in the iteration loop"); }
      reported++;
      if (!progress(reported, user))
      {
        (void) hipStreamSynchronize(ctx->stream);
        give_back();
        return fail(ctx, "cancelled", FLUHIP_CANCELLED);
      }
    }
  }
  give_back();
  if (c->planError) return fail(ctx, "internal error: an update launch did not take the form its dry run announced");
  return FLUHIP_OK;
}

int corpus_iterate(fluhip_corpus* c, int64_t iters, bool updateW, bool updateH,
                          fluhip_progress_fn progress, void* user)
{
  c->loopTimed = false;
  if (!c->loopEv0) { c->loopEv0 = take_event(c->ctx); c->loopEv1 = take_event(c->ctx); }
  const bool ev0 = c->loopEv0 && c->loopEv1 && hipEventRecord(c->loopEv0, c->ctx->stream) == hipSuccess;
  const int rc = corpus_iterate_loop(c, iters, updateW, updateH, progress, user);
  c->loopTimed = ev0 && hipEventRecord(c->loopEv1, c->ctx->stream) == hipSuccess;
  if (c->strip)
  {
    c->stripReady = false; // H may change before the next call
    c->stripSideReady = false;
    if (c->wPending && !c->stripNormFresh)
    {
      // the last launch was a reduce: one workgroup per buffer recomputes the column norms of W'
      StripArgs a;
      a.V = c->mag.as<double>(); a.strideV = c->Tp * c->Fp; a.ldv = c->Fp;
      a.W = c->Wf.as<double>(); a.strideW = c->Fp * c->Kp;
      a.H = c->H1.as<double>(); a.strideH = c->Tp * c->Kp;
      a.part = c->stripPart.as<double>(); a.nrm = c->wnorm.as<double>();
      a.F = (int) c->F; a.T = (int) c->T; a.K = (int) c->K; a.B = (int) c->B;
      a.doH = a.doW = 0; a.wPend = 1; a.statGen = c->stripGen;
#ifdef FLUHIP_AB_SWITCHES
      if (c->stripTile)
      {
        BinTileArgs bt{};
        bt.work = c->tileWork.as<double>(); bt.F = a.F; bt.B = a.B;
        a.nRec = nmf_bintile_records(a.F); a.tileStat = nmf_bintile_records_ptr(bt, c->stripGen);
      }
#endif
      launch_nmf_strip(a, c->ctx->stream);
      c->stripNormFresh = true;
    }
    if (c->wPending) c->stripStatsValid = false; // the normalisation below rewrites W
  }
  if (c->wPending)
  {
    // leave the deferred form on every exit (also a cancelled run hands back W, alg/NMF.hpp:175-176)
    launch_wnorm_apply(c->Wf.as<double>(), c->Fp * c->Kp, (int) c->F, (int) c->Kp, (int) c->B, c->wnorm.as<double>(),
                       c->ctx->stream);
    c->wPending = false;
  }
  return rc;
}

// ---------------------------------------------------------------------------------------
// C ABI
// Host allocations inside the library (std::vector images of seeds, plans, staging) can throw: nothing may cross the C ABI.
// std::bad_alloc is classified like a device out-of-memory (fluhip_last_error_is_out_of_memory), anything else is an error.
template <typename Fn> static int guarded(fluhip_ctx* ctx, Fn&& fn)
{
  try { return fn(); }
  catch (const std::bad_alloc&) { return fail_oom(ctx, "out of host memory inside libflucoma_hip"); }
  catch (const std::exception& e) { return fail(ctx, std::string("internal error: ") + e.what()); }
}

extern "C" {

static int fluhip_corpus_create_impl(fluhip_ctx* ctx, int64_t count, int64_t n, int64_t win, int64_t fft,
                         int64_t hop, int64_t K, fluhip_corpus** out)
{
  if (!ctx || !out) return FLUHIP_ERROR;
  *out = nullptr;
  if (count < 1) return fail(ctx, "corpus must hold at least one buffer");
  if (count > 65535) return fail(ctx, "a corpus holds at most 65535 buffers (split larger corpora into several)");
  int rc = check_shape(ctx, n, win, fft, hop, K);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<fluhip_corpus> c(new fluhip_corpus);
  c->ctx = ctx; c->B = count; c->n = n; c->win = win; c->fft = fft; c->hop = hop; c->K = K;
  rc = corpus_alloc(ctx, c.get());
  if (rc) return rc;
  *out = c.release();
  return FLUHIP_OK;
}

int fluhip_corpus_create_ragged(fluhip_ctx* ctx, int64_t count, const int64_t* n, int64_t win, int64_t fft, int64_t hop,
                                int64_t K, fluhip_corpus** out)
{
  if (!ctx || !out) return FLUHIP_ERROR;
  *out = nullptr;
  if (!n || count < 1) return fail(ctx, "corpus must hold at least one buffer");
  if (count > 65535) return fail(ctx, "a corpus holds at most 65535 buffers (split larger corpora into several)");
  int64_t nmax = 0;
  for (int64_t i = 0; i < count; i++)
  {
    if (n[i] < 1) return fail(ctx, "buffer " + std::to_string(i) + ": not enough frames");
    nmax = std::max(nmax, n[i]);
  }
  int rc = check_shape(ctx, nmax, win, fft, hop, K);
  if (rc) return rc;
  // one set of launches over buffers of different lengths needs the work-list form of the factor-update kernel (padded
  // rank 16 / 32 / 64 / 128) and the block form of the STFT (it takes per-buffer lengths)
  if (update_variant((int) padded_rank(K)) != 5) return fail(ctx, "ragged corpora support ranks up to 128");
  if (!(fft == 1024 || fft == 2048 || fft == 4096) || (win % 2) != 0 || win > fft)
    return fail(ctx, "ragged corpora need an STFT shape with a block form (fft 1024 / 2048 / 4096, even window)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<fluhip_corpus> c(new fluhip_corpus);
  c->ctx = ctx; c->B = count; c->n = nmax; c->win = win; c->fft = fft; c->hop = hop; c->K = K;
  c->ragged = true;
  c->nOf.assign(n, n + count);
  c->tOf.resize((size_t) count);
  for (int64_t i = 0; i < count; i++) c->tOf[(size_t) i] = (int) ((n[i] + hop) / hop); // alg/STFT.hpp:98-99 per buffer
  rc = corpus_alloc(ctx, c.get());
  if (rc) return rc;
  *out = c.release();
  return FLUHIP_OK;
}

int64_t fluhip_corpus_frames_of(const fluhip_corpus* c, int64_t i)
{
  if (!c || i < 0 || i >= c->B) return 0;
  return c->ragged ? c->tOf[(size_t) i] : c->T;
}

int fluhip_corpus_set_audio_ragged_host(fluhip_corpus* c, const float* const* audio)
{
  if (!c || !audio) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  if (!c->ragged) return fail(ctx, "not a ragged corpus");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t) c->B * c->n * sizeof(float);
  if (c->audioOwn.bytes < bytes) HIPCHK(ctx, c->audioOwn.alloc(bytes, false, ctx->stream));
  // every buffer at its slot of the longest buffer's stride; the kernel never reads past a buffer's own length
  for (int64_t i = 0; i < c->B; i++)
  {
    if (!audio[i]) return fail(ctx, "buffer " + std::to_string(i) + ": null audio");
    HIPCHK(ctx, hipMemcpyAsync(c->audioOwn.as<float>() + i * c->n, audio[i], (size_t) c->nOf[(size_t) i] * sizeof(float),
                               hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  c->audioDev = c->audioOwn.as<float>();
  return FLUHIP_OK;
}

int fluhip_corpus_writeback_ragged_host(fluhip_corpus* c, float* const* bases, float* const* acts)
{
  if (!c) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf db, da;
  const size_t nb = (size_t) c->B * c->K * c->F * sizeof(float), na = (size_t) c->B * c->K * c->T * sizeof(float);
  if (bases) HIPCHK(ctx, db.alloc(nb, false, ctx->stream));
  if (acts) HIPCHK(ctx, da.alloc(na, false, ctx->stream));
  int rc = fluhip_corpus_writeback_dev(c, bases ? db.as<float>() : nullptr, acts ? da.as<float>() : nullptr);
  if (rc) return rc;
  for (int64_t i = 0; i < c->B; i++)
  {
    const int64_t Ti = fluhip_corpus_frames_of(c, i);
    if (bases && bases[i])
      HIPCHK(ctx, hipMemcpyAsync(bases[i], db.as<float>() + i * c->K * c->F, (size_t) c->K * c->F * sizeof(float),
                                 hipMemcpyDeviceToHost, ctx->stream));
    if (acts && acts[i]) // K rows of the buffer's own T_i frames out of rows of the longest buffer's length
      HIPCHK(ctx, hipMemcpy2DAsync(acts[i], (size_t) Ti * sizeof(float), da.as<float>() + i * c->K * c->T,
                                   (size_t) c->T * sizeof(float), (size_t) Ti * sizeof(float), (size_t) c->K,
                                   hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLUHIP_OK;
}

void fluhip_corpus_destroy(fluhip_corpus* c)
{
  if (!c) return;
  (void) hipSetDevice(c->ctx->device);
  (void) hipStreamSynchronize(c->ctx->stream);
  if (c->loopEv0) c->ctx->eventPool.push_back(c->loopEv0);
  if (c->loopEv1) c->ctx->eventPool.push_back(c->loopEv1);
  delete c;
}

int64_t fluhip_corpus_frames(const fluhip_corpus* c) { return c ? c->T : 0; }
int64_t fluhip_corpus_bins(const fluhip_corpus* c) { return c ? c->F : 0; }
int64_t fluhip_corpus_device_bytes(const fluhip_corpus* c) { return c ? c->device_bytes() : 0; }

static int fluhip_corpus_set_audio_host_impl(fluhip_corpus* c, const float* audio)
{
  if (!c || !audio) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t) c->B * c->n * sizeof(float);
  if (c->audioOwn.bytes < bytes) HIPCHK(ctx, c->audioOwn.alloc(bytes, false, ctx->stream));
  // The copy runs on the context's COPY stream: whatever another corpus of this context has in flight on the compute stream
  // (the previous slice of a pool job: its iterations) goes on beside it -- SURVEY section 7 step 5's double-buffered upload.
  // A corpus whose own audio may still be read by enqueued work drains the compute stream first.
  if (c->touched) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (!ctx->copyStream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copyStream, hipStreamNonBlocking));
  HIPCHK(ctx, hipMemcpyAsync(c->audioOwn.p, audio, bytes, hipMemcpyHostToDevice, ctx->copyStream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->copyStream)); // the caller's buffer is free again; the device copy is complete
  c->audioDev = c->audioOwn.as<float>();
  return FLUHIP_OK;
}

int fluhip_corpus_set_audio_dev(fluhip_corpus* c, const float* audio_dev)
{
  if (!c || !audio_dev) return FLUHIP_ERROR;
  c->audioDev = audio_dev;
  return FLUHIP_OK;
}

// the STFT phase with the frame-major magnitudes alone (every buffer's STFT::process + magnitude, nothing for the factor
// updates): what a spectrogram-only caller -- BufSTFT, the algorithm-level fluhip_stft_* -- pays per frame.  The corpus has no
// spectrogram for fluhip_corpus_nmf afterwards until fluhip_corpus_stft runs again.
int fluhip_corpus_stft_mag_only(fluhip_corpus* c)
{
  if (!c) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  if (!c->audioDev) return fail(ctx, "corpus has no audio: call fluhip_corpus_set_audio_* first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return corpus_stft(c, c->audioDev, nullptr, c->n, true);
}

static int fluhip_corpus_stft_impl(fluhip_corpus* c)
{
  if (!c) return FLUHIP_ERROR;
  if (!c->audioDev) return fail(c->ctx, "corpus has no audio");
  HIPCHK(c->ctx, hipSetDevice(c->ctx->device));
  return corpus_stft(c, c->audioDev, nullptr, c->n);
}

static int fluhip_corpus_nmf_impl(fluhip_corpus* c, int64_t iters, int update_w, int update_h, int64_t seed,
                      const int64_t* seeds, fluhip_progress_fn progress, void* user)
{
  if (!c) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  if (!c->haveMag) return fail(ctx, "corpus has no spectrogram: call fluhip_corpus_stft first");
  if (iters < 0) return fail(ctx, "negative iteration count");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  FactorInit fi;
  if (!c->seedW32.empty()) fi.W0f32 = c->seedW32.data(); // nrt/NMFClient.hpp:246-258 -> alg/NMF.hpp:102-112
  if (!c->seedH32.empty()) fi.H0f32 = c->seedH32.data(); // :113-124
  int rc = corpus_init_factors(c, seed, seeds, fi);
  if (rc) return rc;
  return corpus_iterate(c, iters, update_w != 0, update_h != 0, progress, user);
}

static int fluhip_corpus_set_factors_impl(fluhip_corpus* c, const float* bases_seed, const float* acts_seed)
{
  if (!c) return FLUHIP_ERROR;
  const size_t nw = (size_t) c->B * c->K * c->F, nh = (size_t) c->B * c->K * c->T;
  if (bases_seed) c->seedW32.assign(bases_seed, bases_seed + nw);
  else std::vector<float>().swap(c->seedW32);
  if (acts_seed) c->seedH32.assign(acts_seed, acts_seed + nh);
  else std::vector<float>().swap(c->seedH32);
  return FLUHIP_OK;
}

int fluhip_corpus_writeback_dev(fluhip_corpus* c, float* bases_dev, float* acts_dev)
{
  if (!c) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  if (!c->haveFactors) return fail(ctx, "corpus has no factors: call fluhip_corpus_nmf first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (bases_dev) // clients/nrt/NMFClient.hpp:277-283
    launch_gather_w_f32(c->Wf.as<double>(), c->Fp * c->Kp, bases_dev, c->K * c->F, (int) c->F,
                        (int) c->K, (int) c->Kp, (int) c->B, s);
  if (acts_dev) // :286-300
    launch_acts_f32(c->H1.as<double>(), c->Tp * c->Kp, acts_dev, c->K * c->T, (int) c->T, (int) c->K,
                    (int) c->Kp, (int) c->B, c->hmax.as<double>(), s);
  HIPCHK(ctx, hipGetLastError());
  return FLUHIP_OK;
}

// clients/nrt/NMFClient.hpp:302-334 for every buffer of the corpus: estimate -> ratio mask -> ISTFT per component
static int fluhip_corpus_keep_spectrum_impl(fluhip_corpus* c, int on)
{
  if (!c) return FLUHIP_ERROR;
  c->keepSpec = on != 0;
  if (!c->keepSpec) c->spec.release();
  c->haveMag = c->haveMag && !(c->keepSpec && !c->spec.p); // a later resynthesis needs the STFT to run again
  return FLUHIP_OK;
}

int fluhip_corpus_resynth_dev(fluhip_corpus* c, float* out_dev)
{
  if (!c || !out_dev) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  if (!c->haveFactors) return fail(ctx, "corpus has no factors: call fluhip_corpus_nmf first");
  if (!c->keepSpec || !c->spec.p)
    return fail(ctx, "resynthesis needs the complex spectrogram: fluhip_corpus_keep_spectrum(c, 1) before fluhip_corpus_stft");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const double *wtab = nullptr, *ttab = nullptr;
  int rc = get_window(ctx, c->win, c->fft, c->windowType, &wtab);
  if (rc) return rc;
  rc = get_twiddle(ctx, c->fft, &ttab);
  if (rc) return rc;
  // Batched form (kernels_stft2.hip resynth_seq_kernel): every component of a group of buffers in one launch, the
  // overlap-add in registers -- no windowed frames in memory (the per-buffer path below writes and re-reads K T win doubles
  // per buffer: 58 GB each way on the bench shard, 85 ms against 113 ms for 200 iterations).  FLUHIP_RESYNTH_BATCH=0 off.
  static const int batchEnv = [] { const char* e = fluhip::ab_getenv("FLUHIP_RESYNTH_BATCH"); return e ? std::atoi(e) : 1; }();
  if (batchEnv != 0 && resynth_batch_supported((int) c->win, (int) c->fft, (int) c->hop))
  {
    // buffers per launch: the reciprocal V-hat of a group within ~1 GiB
    const int64_t perBuf = c->T * c->F * (int64_t) sizeof(double);
    const int64_t group = std::max<int64_t>(1, std::min<int64_t>(c->B, ((int64_t) 1 << 30) / perBuf));
    DevBuf mult, wt, nrm;
    HIPCHK(ctx, mult.alloc((size_t) (group * perBuf), false, s));
    HIPCHK(ctx, wt.alloc((size_t) (group * c->Kp * c->F) * sizeof(double), false, s));
    HIPCHK(ctx, nrm.alloc((size_t) c->hop * sizeof(double), false, s));
    launch_resynth_normaliser(wtab, (int) c->win, (int) c->hop, nrm.as<double>(), s);
    for (int64_t b0 = 0; b0 < c->B; b0 += group)
    {
      const int nb = (int) std::min(group, c->B - b0);
      const double* Wb = c->Wf.as<double>() + b0 * c->Fp * c->Kp;
      const double* Hb = c->H1.as<double>() + b0 * c->Tp * c->Kp;
      launch_resynth_mult(Wb, c->Fp * c->Kp, Hb, c->Tp * c->Kp, wt.as<double>(), mult.as<double>(), (int) c->T, (int) c->F,
                          (int) c->K, (int) c->Kp, nb, s);
      ResynthBatchArgs ra;
      ra.spec = c->spec.as<double>() + b0 * c->T * c->F * 2; ra.specStride = c->T * c->F * 2;
      ra.mult = mult.as<double>(); ra.multStride = c->T * c->F;
      ra.Wt = wt.as<double>(); ra.wtStride = c->Kp * c->F;
      ra.H1 = Hb; ra.hStride = c->Tp * c->Kp;
      ra.Kp = (int) c->Kp; ra.K = (int) c->K;
      ra.win = (int) c->win; ra.fft = (int) c->fft; ra.hop = (int) c->hop; ra.T = (int) c->T; ra.F = (int) c->F; ra.B = nb;
      ra.window = wtab; ra.twiddle = ttab; ra.nrmTab = nrm.as<double>();
      ra.out32 = out_dev + b0 * c->K * c->n; ra.n = c->n; ra.outStride = c->n; ra.trim = c->win / 2;
      ra.nTab = c->ragged ? c->nTab.as<int64_t>() + b0 : nullptr;
      if (!launch_resynth_batch(ra, s)) return fail(ctx, "batched resynthesis refused a shape it had accepted");
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(s)); // the workspaces go out of scope
    return FLUHIP_OK;
  }
  DevBuf vhat, frames;
  HIPCHK(ctx, vhat.alloc((size_t) c->T * c->F * sizeof(double), false, s));
  const int64_t compsPerLaunch = std::max<int64_t>(1, std::min<int64_t>(c->K, ((int64_t) 1 << 30) / (c->T * c->win * 8)));
  HIPCHK(ctx, frames.alloc((size_t) compsPerLaunch * c->T * c->win * sizeof(double), false, s));
  for (int64_t b = 0; b < c->B; b++)
  {
    const double* Wb = c->Wf.as<double>() + b * c->Fp * c->Kp;
    const double* Hb = c->H1.as<double>() + b * c->Tp * c->Kp;
    // a ragged corpus: the buffer's own frames and samples (arrays are strided by the longest buffer's)
    const int Tb = c->ragged ? c->tOf[(size_t) b] : (int) c->T;
    const int64_t nb = c->ragged ? c->nOf[(size_t) b] : c->n;
    launch_vhat(Wb, 0, Hb, 0, vhat.as<double>(), c->F, 0, Tb, (int) c->F, (int) c->Kp, 1, s);
    ResynthArgs ra;
    ra.spec = c->spec.as<double>() + b * c->T * c->F * 2; ra.Wf = Wb; ra.H1 = Hb;
    ra.Vhat = vhat.as<double>(); ra.ldV = c->F; ra.Kp = (int) c->Kp;
    ra.win = (int) c->win; ra.fft = (int) c->fft; ra.hop = (int) c->hop; ra.T = Tb; ra.F = (int) c->F;
    ra.window = wtab; ra.twiddle = ttab; ra.frames = frames.as<double>(); ra.out = nullptr; ra.n = nb; ra.outStride = c->n;
    ra.trim = c->win / 2;
    for (int64_t k = 0; k < c->K; k += compsPerLaunch)
    {
      ra.k = (int) k;
      ra.nComp = (int) std::min(compsPerLaunch, c->K - k);
      ra.out32 = out_dev + (b * c->K + k) * c->n;
      ra.bigScratch = big_fft_scratch(ctx, ra.win, ra.fft, ra.T);
      if (stft_needs_scratch(ra.win, ra.fft) && !ra.bigScratch) return FLUHIP_ERROR;
      launch_resynth(ra, s);
    }
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(s)); // vhat / frames go out of scope
  return FLUHIP_OK;
}

static int fluhip_corpus_resynth_host_impl(fluhip_corpus* c, float* out)
{
  if (!c || !out) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf d;
  const size_t nb = (size_t) c->B * c->K * c->n * sizeof(float);
  HIPCHK(ctx, d.alloc(nb, false, ctx->stream));
  int rc = fluhip_corpus_resynth_dev(c, d.as<float>());
  if (rc) return rc;
  return copy_to_host(ctx, out, nb, d.p, nb, nb, 1, ctx->stream);
}

// The same, written the way an interleaved host buffer holds it (MemoryBufferAdaptor, SuperCollider and Max buffers:
// frames x channels): out[t * frame_stride + b * K + k] = component k of buffer b at sample t -- the layout of
// resynth.samps(i * rank + j) in clients/nrt/NMFClient.hpp:321-326 when the buffer's channels are interleaved.  The
// transposition happens on the device; the host sees one streaming copy instead of count x K strided passes over its buffer.
static int fluhip_corpus_resynth_interleaved_host_impl(fluhip_corpus* c, float* out, int64_t frame_stride)
{
  if (!c || !out) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  if (c->ragged) return fail(ctx, "interleaved resynthesis needs equal-length buffers");
  const int64_t chans = c->B * c->K;
  if (frame_stride < chans) return fail(ctx, "frame stride below count x K");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  DevBuf d, dt;
  const size_t nb = (size_t) chans * c->n * sizeof(float);
  HIPCHK(ctx, d.alloc(nb, false, s));
  HIPCHK(ctx, dt.alloc(nb, false, s));
  int rc = fluhip_corpus_resynth_dev(c, d.as<float>());
  if (rc) return rc;
  launch_transpose_f32(d.as<float>(), c->n, dt.as<float>(), chans, (int) chans, c->n, s); // [chans][n] -> [n][chans]
  HIPCHK(ctx, hipGetLastError());
  const size_t width = (size_t) chans * sizeof(float);
  if (frame_stride == chans) return copy_to_host(ctx, out, nb, dt.p, nb, nb, 1, s);
  return copy_to_host(ctx, out, (size_t) frame_stride * sizeof(float), dt.p, width, width, (size_t) c->n, s);
}

int fluhip_corpus_resynth_ragged_host(fluhip_corpus* c, float* const* out)
{
  if (!c || !out) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf d;
  const size_t nb = (size_t) c->B * c->K * c->n * sizeof(float);
  HIPCHK(ctx, d.alloc(nb, true, ctx->stream));
  int rc = fluhip_corpus_resynth_dev(c, d.as<float>());
  if (rc) return rc;
  for (int64_t i = 0; i < c->B; i++)
  {
    if (!out[i]) continue;
    const int64_t ni = c->ragged ? c->nOf[(size_t) i] : c->n; // K rows of the buffer's own samples out of rows of the longest
    HIPCHK(ctx, hipMemcpy2DAsync(out[i], (size_t) ni * sizeof(float), d.as<float>() + i * c->K * c->n, (size_t) c->n * sizeof(float),
                                 (size_t) ni * sizeof(float), (size_t) c->K, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLUHIP_OK;
}

static int fluhip_corpus_writeback_host_impl(fluhip_corpus* c, float* bases, float* acts)
{
  if (!c) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf db, da;
  const size_t nb = (size_t) c->B * c->K * c->F * sizeof(float), na = (size_t) c->B * c->K * c->T * sizeof(float);
  if (bases) HIPCHK(ctx, db.alloc(nb, false, ctx->stream));
  if (acts) HIPCHK(ctx, da.alloc(na, false, ctx->stream));
  int rc = fluhip_corpus_writeback_dev(c, bases ? db.as<float>() : nullptr, acts ? da.as<float>() : nullptr);
  if (rc) return rc;
  // (large corpora: through the pinned staging blocks, copy_to_host)
  if (bases && (rc = copy_to_host(ctx, bases, nb, db.p, nb, nb, 1, ctx->stream))) return rc;
  if (acts && (rc = copy_to_host(ctx, acts, na, da.p, na, na, 1, ctx->stream))) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return FLUHIP_OK;
}

int fluhip_corpus_read_f64(fluhip_corpus* c, double* mag, double* W1, double* H1)
{
  if (!c) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  if (mag)
  {
    if (!c->haveMag) return fail(ctx, "corpus has no spectrogram");
    for (int64_t b = 0; b < c->B; b++)
      HIPCHK(ctx, hipMemcpy2DAsync(mag + b * c->T * c->F, (size_t) c->F * sizeof(double),
                                   c->mag.as<double>() + b * c->Tp * c->Fp, (size_t) c->Fp * sizeof(double),
                                   (size_t) c->F * sizeof(double), (size_t) c->T, hipMemcpyDeviceToHost, s));
  }
  DevBuf dw, dh;
  if (W1 || H1)
    if (!c->haveFactors) return fail(ctx, "corpus has no factors");
  if (W1)
  {
    const size_t bytes = (size_t) c->B * c->K * c->F * sizeof(double);
    HIPCHK(ctx, dw.alloc(bytes, false, s));
    launch_gather_w_f64(c->Wf.as<double>(), c->Fp * c->Kp, dw.as<double>(), c->K * c->F, (int) c->F,
                        (int) c->K, (int) c->Kp, (int) c->B, s);
    HIPCHK(ctx, hipMemcpyAsync(W1, dw.p, bytes, hipMemcpyDeviceToHost, s));
  }
  if (H1)
  {
    const size_t bytes = (size_t) c->B * c->T * c->K * sizeof(double);
    HIPCHK(ctx, dh.alloc(bytes, false, s));
    launch_gather_h_f64(c->H1.as<double>(), c->Tp * c->Kp, dh.as<double>(), c->T * c->K, (int) c->T,
                        (int) c->K, (int) c->Kp, (int) c->B, s);
    HIPCHK(ctx, hipMemcpyAsync(H1, dh.p, bytes, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(ctx, hipStreamSynchronize(s));
  return FLUHIP_OK;
}

// ---- algorithm-level single-buffer entry points -----------------------------------------
int64_t fluhip_debug_plan_lists(int64_t count, const int64_t* frames, int64_t bins, int64_t K, int which, int32_t* desc,
                                int64_t cap, int32_t* info8)
{
  if (!frames || count < 1 || bins < 1 || K < 1 || K > 128) return -1;
  std::vector<int> tOf((size_t) count);
  int tmax = 1;
  for (int64_t i = 0; i < count; i++)
  {
    if (frames[i] < 1) return -1;
    tOf[(size_t) i] = (int) frames[i];
    tmax = std::max(tmax, tOf[(size_t) i]);
  }
  ListPlanHost plan;
  build_list_plan(tOf, tmax, (int) bins, (int) padded_rank(K), plan);
  const ListSide& sd = which ? plan.H : plan.W;
  if (info8)
  {
    info8[0] = sd.wgs; info8[1] = sd.ng; info8[2] = sd.partial; info8[3] = sd.maxSplit; info8[4] = sd.pieces;
    info8[5] = (int32_t) sd.nPartials; info8[6] = plan.sideW ? 1 : 0; info8[7] = sd.statParts;
  }
  const int64_t n = (int64_t) sd.list.size();
  static_assert(sizeof(WaveDesc) == 12 * sizeof(int32_t), "descriptor layout");
  if (desc) std::memcpy(desc, sd.list.data(), (size_t) std::min(n, cap) * sizeof(WaveDesc));
  return n;
}

int fluhip_debug_plan_kind(int64_t count, int64_t frames, int64_t bins, int64_t K)
{
  if (count < 1 || frames < 1 || bins < 1 || K < 1) return -1;
  const int Kp = (int) padded_rank(K);
  if (update_variant(Kp) != 5) return 0;
  const PlanShape ps{count, frames, bins, Kp};
  return list_plan_pays(&ps) ? 1 : 0;
}

int fluhip_debug_plan_h_update(int64_t count, int64_t frames, int64_t bins, int64_t K)
{
  if (count < 1 || frames < 1 || bins < 1 || K < 1) return -1;
  const int Kp = (int) padded_rank(K);
  if (update_variant(Kp) != 5) return 0;
  const PlanShape ps{count, frames, bins, Kp};
  int wA = 0;
  if (list_plan_pays(&ps) || plan_tail(count, (int) frames, (int) bins, Kp, &wA) > 1) return 0; // (work lists, two-launch H update: the plain forms only)
  if (!nmf_side_column_supported((int) frames, (int) bins, Kp)) return 0;                       // no side column, nothing to take over
  // the launcher's own answer for these arguments (nothing is launched, no device is touched: UpdateArgs::dryRun)
  double dummy = 0.0;
  UpdateArgs a;
  a.R = (int) bins; a.C = (int) frames; a.B = (int) count; a.Kp = Kp;
  a.nsplit = 1;
  a.nrm = &dummy; a.nrmMode = 2;
  a.sideOut = &dummy; a.sideWold = &dummy;
  a.cmbStat = &dummy; a.cmbSide = &dummy; a.cmbWold = &dummy; a.cmbNrmOut = &dummy; a.cmbRowOut = &dummy;
  a.dryRun = true;
  return launch_nmf_update5(a, nullptr);
}

int fluhip_debug_plan_tail(int64_t count, int64_t frames, int64_t bins, int64_t K, int64_t* out4)
{
  if (count < 1 || frames < 1 || bins < 1 || K < 1 || !out4) return FLUHIP_ERROR;
  const int Kp = (int) padded_rank(K);
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  if (update_variant(Kp) != 5) return FLUHIP_OK;
  int wA = 0;
  const int sp = plan_tail(count, (int) frames, (int) bins, Kp, &wA);
  if (sp > 1)
  {
    const int G = ((int) frames + 15) / 16, w = nmf_update5_strips((int) frames, Kp, (int) count);
    out4[0] = sp; out4[1] = wA; out4[2] = w - wA; out4[3] = (int64_t) wA * ((G + w - 1) / w) * 16;
  }
  return FLUHIP_OK;
}

// The whole plan of a shape without a device (decide_update_plan), for tests/test_plan_table.py and tools: out32 =
//  0 variant   1 nsplitW   2 nsplitH   3 lazy   4 sideW   5 stripsW   6 Kp   7 Kc   8 strip (0 no, 1 fused, 2 bin strips, 3 bin tiles)
//  9 work lists   10 tailSplitH   11 tailStripsH   12 tailRestH   13 tailColsH   14 stripsH
//  15 .. 21 workspaces in doubles: part, dpart, csum, wscratch, colPart, stripPart, wide
//  22 what the plain H update takes over (bit 0 side partials, bit 1 norm combine, bit 2 column sums from the W update; 0 when
//     the schedule has no such launch)   23 form of the norm-combine launch between the updates in the steady state (WnormForm;
//     -1: none, the H update does it)   24 slices of the side-column launch (0: no side column)
//  25 kSideFromHSlots   26 the first word of wscratch behind the statistics records (B stripsW 2 Kp)
int fluhip_debug_plan_shape(int64_t count, int64_t frames, int64_t bins, int64_t K, int64_t* out32)
{
  if (count < 1 || frames < 1 || bins < 1 || K < 1 || !out32) return FLUHIP_ERROR;
  UpdatePlan p;
  decide_update_plan(count, frames, bins, K, p);
  std::memset(out32, 0, 32 * sizeof(int64_t));
  out32[0] = p.variant; out32[1] = p.nsplitW; out32[2] = p.nsplitH; out32[3] = p.lazy; out32[4] = p.sideW; out32[5] = p.stripsW;
  out32[6] = p.Kp; out32[7] = p.Kc; out32[8] = p.strip ? (p.stripTile ? 3 : (p.stripBin ? 2 : 1)) : 0; out32[9] = p.useLists;
  out32[10] = p.tailSplitH; out32[11] = p.tailStripsH; out32[12] = p.tailRestH; out32[13] = p.tailColsH; out32[14] = p.stripsH;
  out32[15] = p.partDoubles; out32[16] = p.dpartDoubles; out32[17] = p.csumDoubles; out32[18] = p.wscratchDoubles;
  out32[19] = p.colPartDoubles; out32[20] = p.stripPartDoubles; out32[21] = p.wideDoubles;
  out32[23] = -1;
  out32[25] = kSideFromHSlots;
  out32[26] = count * p.stripsW * 2 * p.Kp;
  if (p.variant == 5 && p.lazy && !p.strip)
  {
    const int Kp = (int) p.Kp;
    int takes = 0;
    if (!p.useLists && p.sideW && p.nsplitH == 1 && p.tailSplitH <= 1)
    {
      // the launcher's own answer (nothing is launched, no device is touched: UpdateArgs::dryRun), asked as the iteration loop asks
      double dummy = 0.0;
      UpdateArgs a;
      a.R = (int) bins; a.C = (int) frames; a.B = (int) count; a.Kp = Kp; a.Kc = p.Kc;
      a.nsplit = 1;
      a.nrm = &dummy; a.nrmMode = 2;
      a.sideOut = &dummy; a.sideWold = &dummy;
      if (p.stripsW <= 16 && p.stripsH <= 16)
      {
        a.cmbStat = &dummy; a.cmbSide = &dummy; a.cmbWold = &dummy; a.cmbNrmOut = &dummy; a.cmbRowOut = &dummy;
        if (p.colPartDoubles) { a.colIn = &dummy; a.colInN = p.stripsW; }
      }
      a.dryRun = true;
      takes = launch_nmf_update5(a, nullptr);
    }
    out32[22] = takes;
    const int nsl = p.sideW ? wnorm_side_slices((int) frames, Kp) : 0;
    out32[24] = (takes & 1) ? 0 : nsl;
    if (!(takes & 2))
      out32[23] = (int) wnorm_combine_form(Kp, (int) count, p.stripsW, (takes & 1) ? p.stripsH : nsl, (int) frames, (takes & 1) ? 2 : 0, false);
  }
  return FLUHIP_OK;
}

int fluhip_debug_wnorm_form(int Kp, int count, int parts, int slices, int side_rows, int side_phase, int want_colsum)
{
  return (int) wnorm_combine_form(Kp, count, parts, slices, side_rows, side_phase, want_colsum != 0);
}

int fluhip_corpus_plan(const fluhip_corpus* c, int64_t* out8)
{
  if (!c || !out8) return FLUHIP_ERROR;
  out8[0] = update_variant((int) c->Kp);
  out8[1] = c->nsplitW;
  out8[2] = c->nsplitH | ((int64_t) c->tailSplitH << 16); // (tail split of the two-launch H update in the high half)
  out8[3] = c->lazy ? 1 : 0;
  out8[4] = c->sideW ? 1 : 0;
  out8[5] = c->stripsW;
  out8[6] = c->Kp | ((int64_t) (c->Kc > 0 ? c->Kc : c->Kp) << 16); // (the rank the factor updates compute in the high half)
  out8[7] = c->strip ? (c->stripTile ? 3 : (c->stripBin ? 2 : 1)) : 0; // 2: the W update as a bin-strip launch (A/B); 3: as the bin-tiled launch
  return FLUHIP_OK;
}

// ---- profiling ------------------------------------------------------------------------
// debugging aid for FLUHIP_K5_INSTR: first 8 words of the split-denominator scratch
int fluhip_corpus_debug_words(fluhip_corpus* c, int64_t* out32)
{
  if (!c || !out32 || !c->dpart.p) return FLUHIP_ERROR;
  HIPCHK(c->ctx, hipStreamSynchronize(c->ctx->stream));
  if (c->strip)
  {
    // FLUHIP_STRIP_INSTR: the 32 words behind the partials (kernels_nmf_strip.hip STRIP_STAMP)
    std::memset(out32, 0, 32 * sizeof(int64_t));
    const int64_t off = nmf_strip_part_doubles((int) c->F, (int) c->T, (int) c->B) - 32;
    HIPCHK(c->ctx, hipMemcpy(out32, c->stripPart.as<double>() + off, 32 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return FLUHIP_OK;
  }
  HIPCHK(c->ctx, hipMemcpy(out32, c->dpart.p, 32 * sizeof(int64_t), hipMemcpyDeviceToHost));
  return FLUHIP_OK;
}

int fluhip_corpus_last_loop_ms(fluhip_corpus* c, double* ms)
{
  if (!c || !ms) return FLUHIP_ERROR;
  if (!c->loopTimed) return fail(c->ctx, "no iteration loop has been timed on this corpus");
  float t = 0.f;
  HIPCHK(c->ctx, hipEventSynchronize(c->loopEv1));
  HIPCHK(c->ctx, hipEventElapsedTime(&t, c->loopEv0, c->loopEv1));
  *ms = (double) t;
  return FLUHIP_OK;
}

int fluhip_corpus_update_clocks(fluhip_corpus* c, int64_t* out8, int reset)
{
  if (!c || !c->clk.p) return FLUHIP_ERROR;
  fluhip_ctx* ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (out8) HIPCHK(ctx, hipMemcpy(out8, c->clk.p, 8 * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (reset) HIPCHK(ctx, hipMemset(c->clk.p, 0, 8 * sizeof(int64_t)));
  return FLUHIP_OK;
}


// the entry points of the clients' batched block (NMFClient.hpp), exception-tight
int fluhip_corpus_create(fluhip_ctx* ctx, int64_t count, int64_t n, int64_t win, int64_t fft,
                         int64_t hop, int64_t K, fluhip_corpus** out)
{
  return guarded(ctx, [&] { return fluhip_corpus_create_impl(ctx, count, n, win, fft, hop, K, out); });
}

int fluhip_corpus_set_audio_host(fluhip_corpus* c, const float* audio)
{
  return guarded((c ? c->ctx : nullptr), [&] { return fluhip_corpus_set_audio_host_impl(c, audio); });
}

int fluhip_corpus_stft(fluhip_corpus* c)
{
  return guarded((c ? c->ctx : nullptr), [&] { return fluhip_corpus_stft_impl(c); });
}

int fluhip_corpus_set_factors(fluhip_corpus* c, const float* bases_seed, const float* acts_seed)
{
  return guarded((c ? c->ctx : nullptr), [&] { return fluhip_corpus_set_factors_impl(c, bases_seed, acts_seed); });
}

int fluhip_corpus_nmf(fluhip_corpus* c, int64_t iters, int update_w, int update_h, int64_t seed,
                      const int64_t* seeds, fluhip_progress_fn progress, void* user)
{
  return guarded((c ? c->ctx : nullptr), [&] { return fluhip_corpus_nmf_impl(c, iters, update_w, update_h, seed, seeds, progress, user); });
}

int fluhip_corpus_writeback_host(fluhip_corpus* c, float* bases, float* acts)
{
  return guarded((c ? c->ctx : nullptr), [&] { return fluhip_corpus_writeback_host_impl(c, bases, acts); });
}

int fluhip_corpus_resynth_host(fluhip_corpus* c, float* out)
{
  return guarded((c ? c->ctx : nullptr), [&] { return fluhip_corpus_resynth_host_impl(c, out); });
}

int fluhip_corpus_resynth_interleaved_host(fluhip_corpus* c, float* out, int64_t frame_stride)
{
  return guarded((c ? c->ctx : nullptr), [&] { return fluhip_corpus_resynth_interleaved_host_impl(c, out, frame_stride); });
}

int fluhip_corpus_keep_spectrum(fluhip_corpus* c, int on)
{
  return guarded((c ? c->ctx : nullptr), [&] { return fluhip_corpus_keep_spectrum_impl(c, on); });
}

} // extern "C"
