// ga_plan_conv.cpp -- the convolver stages of a chunk: formulation D (coarse partitions, fused groups, pre-mix, carried tails), the shared-IR
// groups of formulation A, formulations B / C / R on private state, scratch sizing (see ga_chunk_internal.hpp).
#include "ga_chunk_internal.hpp"

namespace ga {

// ======================================================================================================
// convolver rows <-> groups
// ======================================================================================================
// make sure the group's state arrays cover all rows; new rows start from zero state
void Context::ensureGroupState(ConvGroup& g) {
  Context& c = *this;
  int need = (int)roundup(std::max<size_t>(g.rows.size(), 1), 128);
  if (need <= g.rp) return;
  const int hist = g.P - 1;
  size_t hBytes = (size_t)kBins * std::max(hist, 1) * need * sizeof(float);
  size_t oBytes = (size_t)need * kBlock * sizeof(float);
  float* nr = (float*)c.dalloc(hBytes);
  float* ni = (float*)c.dalloc(hBytes);
  float* o0 = (float*)c.dalloc(oBytes);
  float* o1 = (float*)c.dalloc(oBytes);
  GA_HIP(hipMemsetAsync(nr, 0, hBytes, c.stream));
  GA_HIP(hipMemsetAsync(ni, 0, hBytes, c.stream));
  GA_HIP(hipMemsetAsync(o0, 0, oBytes, c.stream));
  GA_HIP(hipMemsetAsync(o1, 0, oBytes, c.stream));
  if (g.rp > 0) {
    if (hist > 0 && !g.histZero) {
      GA_HIP(hipMemcpy2DAsync(nr, (size_t)need * 4, g.histR, (size_t)g.rp * 4, (size_t)g.rp * 4, (size_t)kBins * hist,
                              hipMemcpyDeviceToDevice, c.stream));
      GA_HIP(hipMemcpy2DAsync(ni, (size_t)need * 4, g.histI, (size_t)g.rp * 4, (size_t)g.rp * 4, (size_t)kBins * hist,
                              hipMemcpyDeviceToDevice, c.stream));
    }
    GA_HIP(hipMemcpyAsync(o0, g.overlap[g.ovCur], (size_t)g.rp * kBlock * 4, hipMemcpyDeviceToDevice, c.stream));
    GA_HIP(hipStreamSynchronize(c.stream));
    size_t oldH = (size_t)kBins * std::max(hist, 1) * g.rp * sizeof(float);
    c.dfree(g.histR, oldH);
    c.dfree(g.histI, oldH);
    c.dfree(g.overlap[0], (size_t)g.rp * kBlock * 4);
    c.dfree(g.overlap[1], (size_t)g.rp * kBlock * 4);
  }
  g.histR = nr;
  g.histI = ni;
  g.overlap[0] = o0;
  g.overlap[1] = o1;
  g.ovCur = 0;
  g.rp = need;
}

// ======================================================================================================
// formulation D (ga_coarse.hip): coarse partitions, consumer sums fused in the frequency domain
// ======================================================================================================
// Which convolver outputs may be summed as spectra?  A node whose single output feeds exactly ONE input (or parameter) of
// one consumer, and is a term of that input in every segment of the chunk.  Every mixing rule of AudioNodeInput.MixBuffer
// (equal counts, 1 -> N, N -> 1 down-mix, min(N, M), AudioNodeInput.cs:182-244) is linear in the term, so the consumer
// may receive the sum of the group as ONE term -- the leader's output -- and nothing from the other members.  What changes
// is only the association of the float32 additions (the reference adds the members one by one in connection order).
void Context::planCoarseFusion(const std::vector<int>& topo, const std::vector<Segment>& segs) {
  // The grouping is a function of the graph, of the convolvers' formulations and impulse responses and of the segments' control
  // state (who is a term of which input): while none of them moved since the previous chunk the leaders stand.
  {
    uint64_t key = hmix(graphVersion, (uint64_t)segs.size());
    for (const Segment& sg : segs) key = hmix(key, sg.hash);
    for (int id : topo) {
      const NodeS& nd = *nodes[id];
      if (nd.type == GA_NODE_CONVOLVER) key = hmix(hmix(key, ((uint64_t)id << 8) | (uint64_t)nd.convPath), (uint64_t)(uintptr_t)nd.ir.get());
    }
    if (fusionKeyValid && key == fusionKey) return;
    fusionKey = key;
    fusionKeyValid = true;
  }
  std::vector<int> cand;
  for (int id : topo) {
    NodeS& nd = *nodes[id];
    if (nd.type != GA_NODE_CONVOLVER) continue;
    nd.dLeader = -1;
    if (!nd.ir || nd.convPath != 4) continue;
    nd.dLeader = id;
    if (nd.outputs[0].connectedInputs.size() == 1) cand.push_back(id);
  }
  if (cand.size() < 2) return;
  std::unordered_map<int, int> seen;                 // candidate -> segments in which it is a term of its consumer's input
  std::unordered_map<int, std::vector<int>> byConsumer;
  for (int id : cand) {
    seen[id] = 0;
    byConsumer[nodes[id]->outputs[0].connectedInputs[0].node].push_back(id);
  }
  for (const Segment& sg : segs)
    for (const NodeSeg& ns : sg.nodes) {
      if (byConsumer.find(ns.id) == byConsumer.end()) continue;
      auto scan = [&](const InSeg& is, int inputIdx) {
        for (const TermS& t : is.terms) {
          auto it = seen.find(t.node);
          if (it == seen.end()) continue;
          const InRef& r = nodes[t.node]->outputs[0].connectedInputs[0];
          if (r.node == ns.id && r.input == inputIdx && t.out == 0) it->second++;
        }
      };
      for (int i = 0; i < (int)ns.ins.size(); i++) scan(ns.ins[i], i);
      for (int p = 0; p < (int)ns.pins.size(); p++) scan(ns.pins[p], -1 - p);
    }
  std::map<std::tuple<int, int, int, int, int>, int> leaderOf;   // (consumer, input, depth, output channels, partitions) -> leader
  for (int id : cand) {   // topo order: the leader is the first member the traversal reaches
    NodeS& nd = *nodes[id];
    if (seen[id] != (int)segs.size()) continue;
    const InRef& r = nd.outputs[0].connectedInputs[0];
    auto key = std::make_tuple(r.node, r.input, nd.depth, nd.effectiveOutCh, nd.ir->coarseP);
    auto it = leaderOf.find(key);
    if (it == leaderOf.end()) leaderOf.emplace(key, id);
    else nd.dLeader = it->second;
  }
}

// chunk-long view of input channel `c` of a convolver: the segment views when they agree, else a materialised copy
static const float* convChunkInput(Context& c, Exec& ex, const Exec::ConvInRow& ci, int ch) {
  const auto& segs = ex.segs;
  const float* stable = nullptr;
  bool same = true, first = true;
  for (size_t si = 0; si < segs.size(); si++) {
    const float* v = (ci[si].empty() || ch >= (int)ci[si].size()) ? nullptr : ci[si][ch];
    if (first) { stable = v; first = false; } else if (v != stable) same = false;
  }
  if (same) return stable;
  float* slab = getSlab(c);
  for (size_t si = 0; si < segs.size(); si++) {
    const float* v = (ci[si].empty() || ch >= (int)ci[si].size()) ? nullptr : ci[si][ch];
    MixJob mj;
    mj.out = slab;
    mj.term0 = (int)ex.terms.size();
    mj.nterms = v ? 1 : 0;
    mj.f0 = segs[si].b0 * kBlock;
    mj.n = (segs[si].b1 - segs[si].b0) * kBlock;
    if (v) {
      ex.terms.push_back(v);
      ex.noteAlign(v, mj.f0);
    }
    ex.mixJobs.push_back(mj);
  }
  return slab;
}

// the hist_len samples in front of the chunk of input channel `ch`: a span of a PlayableAudioBuffer, the node's own copy, or nothing yet
static const float* coarseHistory(const NodeS& nd, int ch) {
  if (ch < (int)nd.dHistExt.size() && nd.dHistExt[ch].first) return nd.dHistExt[ch].first;
  return nd.dHistZero ? nullptr : nd.dHist[nd.dHistCur] + (size_t)ch * nd.dHistLen;
}

// If the last hist_len samples of this chunk's input are device memory that stays (a PlayableAudioBuffer played zero-copy), the
// next chunk's history is that span and nothing has to be written; otherwise the span is forgotten and the caller copies.
static bool coarseHistoryStays(Context& c, NodeS& nd, int ch, const float* in, int64_t frames) {
  if ((int)nd.dHistExt.size() < nd.bInCh) nd.dHistExt.resize(nd.bInCh, {nullptr, -1});
  const int64_t hl = nd.dHistLen;
  int buf = -1;
  if (c.coarseExtHist && in && frames >= hl) buf = c.persistentBuffer(in + (frames - hl), hl);
  nd.dHistExt[ch] = buf >= 0 ? std::make_pair(in + (frames - hl), buf) : std::make_pair((const float*)nullptr, -1);
  return buf >= 0;
}

// One convolver stage of a chunk in formulation D, planned in five passes (Context::chunkPlanConvolvers calls planCoarseStage).
namespace {
struct CoarseStage {
  Context& c;
  Exec& ex;
  const std::vector<int>& dNodes;
  const int64_t n, frames;
  const int nT;
  static constexpr int kVoicesPerJob = kCoarseJobTerms;   // terms whose products one workgroup accumulates in registers
  struct Piece {   // <= 4 columns of one signal: (impulse-response channel, output channel of the group)
    int frame0, P, xrow, u0;
    IrSpectra* ir;
    int leader;
    int ncol;
    int irCh[16], outCh[16];
  };
  struct GroupInfo {   // a fused group (by leader), or a convolver on its own
    uint64_t sig = 1469598103934665603ull;
    int maxP = 0, nIn = 0, nOut = 0;
    const void* ir0 = nullptr;
    bool oneIr = true, tail = false, carried = false, fresh = true, noHist = false;
    // time-domain pre-mix (option "coarse_premix"): every member convolves with the same spectra and has the same channel
    // layout, so the group's inputs are added up in front of ONE set of transforms
    bool uniform = true, premix = false;
    int members = 0, nxr = 0, bInCh0 = 0, bSlots0 = 0;
    bool ts0 = false;
    int64_t hl0 = 0;
    struct Terms { std::vector<PremixTerm> in[32], hist[32]; };   // per input channel of the group
    std::unique_ptr<Terms> terms;                                 // (pre-mixed groups only)
  };
  std::vector<Views> chInOf;   // [position in dNodes]: chunk-long views of the node's input channels
  bool wideStrided = true;   // every 16-column term's spectra are h[0] + c x P x kCoarseBins
  std::vector<CoarseHandOver> fwdHandOver;   // a pending hand-over that rides in the first forward launch of this stage
  std::vector<PremixJob> pmJobs;
  std::vector<PremixTerm> pmTerms;
  size_t pmUsed = 0;       // bytes of the pre-mix arena handed out
  int64_t pmMaxN = 0;
  double pmBytes = 0;
  std::vector<CoarseXRow> xrows;
  std::vector<CoarseHistJob> hjobs;
  std::vector<Piece> pieces;
  std::map<int, GroupInfo> groups;   // by leader
  const bool tails;
  int frameNext = 0;
  int64_t maxHist = 0;
  double histBytes = 0;
  std::map<int, double> carryBytes;   // row -> bytes of next-chunk history its forward transform also writes
  int nxAll = 0, G = 1;
  int gBegin[9] = {};
  std::vector<CoarseTerm> terms;
  // launches: by column count (1, 2, 4) x (terms with their own impulse responses | one impulse response for all terms), and
  // by the group whose transforms complete the job's inputs.  Class index = 2 * column class + shared.
  static constexpr int kCwOf[4] = {1, 2, 4, 16};   // column classes of the multiply-accumulate launches
  std::vector<CoarseJob> jobs[8][8];
  int maxT[8] = {0, 0, 0, 0, 0, 0, 0, 0}, maxP[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int pbOf[8] = {4, 4, 4, 4, 4, 4, 4, 4};   // largest of 4, 2, 1 dividing every job's partition count (the sweep's register block)
  double macBytes[8][8] = {}, macFlops[8][8] = {};
  double pmFlops = 0, invFlops = 0;
  std::map<std::pair<int, int>, std::vector<int>> outRows;   // (leader, channel) -> Y rows to sum
  int yNext = 0;
  std::vector<CoarseOut> outs;
  std::vector<int> ylist;
  double invBytes = 0;
  int yFrames = 0, invBlocks = 0;

  CoarseStage(Context& c_, Exec& ex_, const std::vector<int>& d, int64_t n_)
      : c(c_), ex(ex_), dNodes(d), n(n_), frames(n_ * kBlock), nT((int)((n_ * kBlock + kCoarseBlock - 1) / kCoarseBlock)), tails(c_.coarseTail) {}
  void resolveInputs();    // chunk-long input views of every node (materialised where the segments disagree)
  void classifyGroups();   // which groups are pre-mixed, carry a tail, can use the one the previous chunk left
  void buildRows();        // signals to transform (pre-mixed groups: one per channel), history hand-over, pieces (signal x columns)
  void addPieces(NodeS& nd, int leader, int nxr, const int* xFrame, const int* xIndex);
  void buildJobs();        // multiply-accumulate jobs and their terms
  void buildOutputs();     // inverse-transform outputs, tail buffers
  void enqueue();          // tables into the plan, launches
};

void CoarseStage::resolveInputs() {
  chInOf.resize(dNodes.size());
  for (size_t di = 0; di < dNodes.size(); di++) {
    const int id = dNodes[di];
    NodeS& nd = *c.nodes[id];
    const int64_t hl = nd.dHistLen;
    const Exec::ConvInRow ci = ex.convIn[id];
    Views& chIn = chInOf[di];
    chIn.assign(nd.bInCh, nullptr);
    auto viewOf = [&](size_t si, int ch) { return (ci[si].empty() || ch >= (int)ci[si].size()) ? (const float*)nullptr : ci[si][ch]; };
    for (int ch = 0; ch < nd.bInCh; ch++) {
      // a channel that shows the same view as an earlier one in every segment IS that channel (a mono signal copied to all
      // channels of an explicit input, AudioNodeInput.cs:201-213): one chunk-long view -- one materialised copy -- serves both
      int same = -1;
      for (int e = 0; e < ch && same < 0; e++) {
        bool eq = true;
        for (size_t si = 0; si < ex.segs.size() && eq; si++) eq = viewOf(si, e) == viewOf(si, ch);
        if (eq) same = e;
      }
      chIn[ch] = same >= 0 ? chIn[same] : convChunkInput(c, ex, ci, ch);
    }
    bool allSame = true;
    for (int ch = 1; ch < nd.bInCh; ch++) allSame = allSame && (chIn[ch] == chIn[0]);
    if (nd.bShared && !allSame) {   // the channels start to differ: every channel inherits the (so far common) history
      // (a plan entry like every other device action of the chunk: ordered with the chunk's launches, nothing is issued at plan time)
      if (const float* h0 = coarseHistory(nd, 0))
        for (int ch = 1; ch < nd.bInCh; ch++) {
          float* dst = nd.dHist[nd.dHistCur] + (size_t)ch * hl;
          hipStream_t st = c.stream;
          ex.plan.add(LK_OTHER, [=](uint8_t*) { GA_HIP(hipMemcpyAsync(dst, h0, (size_t)hl * sizeof(float), hipMemcpyDeviceToDevice, st)); });
          if (ch < (int)nd.dHistExt.size()) nd.dHistExt[ch] = {nullptr, -1};
        }
      if (nd.dHistZero && coarseHistory(nd, 0)) nd.dHistZero = false;   // (the copies above are the channels' histories now)
      nd.bShared = false;
    }
  }
}

void CoarseStage::classifyGroups() {
  // ---- carried tails (option "coarse_tail"): every output of the stage keeps, from chunk to chunk, what the input so far adds to
  // the samples behind the chunk's end.  While a group of fused convolvers is the same as in the previous chunk its members need
  // no input history in front of the chunk: their windows start at the chunk (u = 0) and the previous chunk's tail is added to
  // the output instead -- P' - 1 fewer transforms per signal and chunk.  Any change (member set, impulse responses, channel
  // modes, a chunk in between that did not run this stage) falls back to the input histories, which are kept up to date either way.
  {
    auto mix = [](uint64_t& h, uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    for (int id : dNodes) {
      NodeS& nd = *c.nodes[id];
      GroupInfo& g = groups[nd.dLeader >= 0 ? nd.dLeader : id];
      mix(g.sig, (uint64_t)id);
      mix(g.sig, (uint64_t)(uintptr_t)nd.ir.get());
      mix(g.sig, (uint64_t)nd.bInCh | ((uint64_t)nd.bSlots << 8) | ((uint64_t)nd.isTrueStereo << 16) | ((uint64_t)nd.bShared << 17) |
                     ((uint64_t)nd.ir->coarseP << 24));
      g.maxP = std::max(g.maxP, nd.ir->coarseP);
      g.fresh = g.fresh && nd.dHistZero;   // no member has seen input yet: nothing in front of the chunk either
      g.nIn += nd.bShared ? 1 : nd.bInCh;
      if (!g.ir0) g.ir0 = nd.ir.get();
      g.oneIr = g.oneIr && g.ir0 == nd.ir.get();
      g.nOut = std::max(g.nOut, nd.isTrueStereo ? 2 : nd.bSlots);
      const int nxr = nd.bShared ? 1 : nd.bInCh;
      if (g.members++ == 0) {
        g.nxr = nxr;
        g.bInCh0 = nd.bInCh;
        g.bSlots0 = nd.bSlots;
        g.ts0 = nd.isTrueStereo;
        g.hl0 = nd.dHistLen;
      } else {
        g.uniform = g.uniform && g.nxr == nxr && g.bInCh0 == nd.bInCh && g.bSlots0 == nd.bSlots && g.ts0 == nd.isTrueStereo && g.hl0 == nd.dHistLen;
      }
    }
    for (auto& kv : groups) {
      NodeS& ld = *c.nodes[kv.first];
      // a tail costs P' more inverse transforms per output channel and chunk and saves P' - 2 forward transforms per input row:
      // worth it for sums of many signals, not for a convolver on its own.  The P' more output blocks are nearly free in the
      // reduction kernel (one impulse response for the whole group); the general kernel skips the partition blocks whose windows
      // lie behind the chunk (all zero), so a group of private impulse responses multiplies exactly the products it would have
      // multiplied with the histories in front -- but measured it does not pay (ga_engine.hpp): option "coarse_tail_private", off
      kv.second.premix = c.coarsePremix && kv.second.members >= 2 && kv.second.oneIr && kv.second.uniform && kv.second.nxr <= 32;
      // (a pre-mixed group always keeps its tail: reading every member's history again would cost members x (P' - 1) blocks)
      kv.second.tail = tails && (kv.second.oneIr || c.coarseTailPrivate) &&
                       (kv.second.premix || (int64_t)kv.second.nIn * (kv.second.maxP - 2) >= (int64_t)kv.second.nOut * kv.second.maxP);
      kv.second.carried = kv.second.tail && ld.dTail[0] && ld.dTailSeq + 1 == c.chunkSeq && ld.dTailSig == kv.second.sig &&
                          ld.dTailLen == (int64_t)(kv.second.maxP + 1) * kCoarseBlock;
      kv.second.noHist = kv.second.carried || (kv.second.tail && kv.second.fresh);
    }
  }
}

// columns of a node's input rows: discrete -> slot c reads input c, IR channel c, output c ; true stereo -> (L,h0,outL) (L,h1,outR)
// (R,h2,outL) (R,h3,outR)  (ConvolverNode.cs:127-151).  Pieces of 4, 2, 1 columns per row.
void CoarseStage::addPieces(NodeS& nd, int leader, int nxr, const int* xFrame, const int* xIndex) {
  IrSpectra& ir = *nd.ir;
  for (int xc = 0; xc < nxr; xc++) {
    int cols[32][2], ncols = 0;
    for (int slot = 0; slot < nd.bSlots; slot++) {
      const int inc = nd.isTrueStereo ? (slot >> 1) : slot;
      if (!(nd.bShared || inc == xc)) continue;
      cols[ncols][0] = slot;                               // slot index == IR channel index in both modes
      cols[ncols][1] = nd.isTrueStereo ? (slot & 1) : slot;
      ncols++;
    }
    // 16 columns of one signal at once where the group's terms have impulse responses of their own (the general kernel's 16-column
    // instance: the signal's frames are staged once for all of them) and its double-buffered spectra fit the LDS (P' <= 4)
    const GroupInfo& gi = groups[leader];
    const bool wide = !gi.oneIr && ir.coarseP <= 4 && c.coarseWide;
    for (int c0 = 0; c0 < ncols;) {
      const int left = ncols - c0;
      const int w = (wide && left >= 16) ? 16 : (left >= 4 ? 4 : (left >= 2 ? 2 : 1));
      Piece pc{};
      pc.frame0 = xFrame[xc];
      pc.xrow = xIndex[xc];
      pc.u0 = xrows[xIndex[xc]].u0;
      pc.P = ir.coarseP;
      pc.ir = &ir;
      pc.leader = leader;
      pc.ncol = w;
      for (int j = 0; j < w; j++) {
        pc.irCh[j] = cols[c0 + j][0];
        pc.outCh[j] = cols[c0 + j][1];
      }
      pieces.push_back(pc);
      c0 += w;
    }
  }
}

void CoarseStage::buildRows() {
  for (size_t di = 0; di < dNodes.size(); di++) {
    const int id = dNodes[di];
    NodeS& nd = *c.nodes[id];
    IrSpectra& ir = *nd.ir;
    const int P = ir.coarseP;
    if (P < 1 || P > kCoarseMaxP) fail(GA_ERR_INVALID_OPERATION, "internal: coarse partition count out of range");
    const int64_t hl = nd.dHistLen;
    GroupInfo& gi0 = groups[nd.dLeader >= 0 ? nd.dLeader : id];
    const bool carried = gi0.noHist;   // no windows in front of the chunk
    const Views& chIn = chInOf[di];
    const int nxr = nd.bShared ? 1 : nd.bInCh;
    c.stats.mac_flops_total += 8.0 * ir.P * kBins * (double)nd.bSlots * (double)n;
    c.stats.mac_bytes_total += ((double)ir.P * kBins * 8.0 + kBins * 8.0 + 512.0) * nd.bSlots * (double)n;
    if (gi0.premix) {
      // a member of a pre-mixed group: nothing to transform for it; its samples join the group's sum and its own history of the
      // next chunk is written on the way (or by a copy job where that is not possible)
      for (int ch = 0; ch < nxr; ch++) {
        const float* oldHist = coarseHistory(nd, ch);
        float* nextHist = nd.dHist[nd.dHistCur ^ 1] + (size_t)ch * hl;
        PremixTerm t{chIn[ch], nullptr};
        if (coarseHistoryStays(c, nd, ch, chIn[ch], frames)) {
          // (the next chunk's history is a span of the member's sample buffer)
        } else if (c.coarseCarry && chIn[ch] && frames >= hl && ((uintptr_t)nextHist & 15) == 0) {
          t.carry = nextHist;
          pmBytes += (double)hl * 4.0;
        } else {
          hjobs.push_back(CoarseHistJob{oldHist, chIn[ch], nextHist, hl, frames});
          maxHist = std::max(maxHist, hl);
          histBytes += 2.0 * (double)hl * 4.0;
        }
        if (!gi0.terms) gi0.terms = std::make_unique<GroupInfo::Terms>();
        gi0.terms->in[ch].push_back(t);
        if (!carried) gi0.terms->hist[ch].push_back(PremixTerm{oldHist, nullptr});
      }
      nd.dHistCur ^= 1;
      nd.dHistZero = false;
      continue;
    }
    int xFrame[32], xIndex[32];
    for (int ch = 0; ch < nxr; ch++) {
      CoarseXRow r;
      const float* oldHist = coarseHistory(nd, ch);   // (kept up to date in every mode)
      r.hist = carried ? nullptr : oldHist;
      r.in = chIn[ch];
      r.nvalid = frames;
      r.frame0 = frameNext;
      // windows u0 .. u_last: with carried tails the last one is u = nT ([last block | nothing yet]: it feeds the outputs behind
      // the chunk's end), and a group that continues needs none in front of the chunk
      r.u0 = carried ? 0 : -(P - 1);
      r.n_frames = (gi0.tail ? nT + 1 : nT) - r.u0;
      r.hist_len = (int)hl;
      r.flags = 0;
      r.scale = 1.0f;
      xFrame[ch] = frameNext;
      xIndex[ch] = (int)xrows.size();
      frameNext += r.n_frames;
      // the next chunk's history: the last hl samples of [history | input].  When they all come from this chunk's input the
      // forward kernel writes them while it holds the samples (ga_kernels.hpp, CoarseXRow::carry); otherwise a copy job.
      float* nextHist = nd.dHist[nd.dHistCur ^ 1] + (size_t)ch * hl;
      r.carry = nullptr;
      r.carry_from = 0;
      if (coarseHistoryStays(c, nd, ch, r.in, frames)) {
        // (the next chunk's history is a span of the sample buffer the input aliases)
      } else if (c.coarseCarry && r.in && frames >= hl && (((uintptr_t)r.in | (uintptr_t)nextHist) & 15) == 0) {
        r.carry = nextHist;
        r.carry_from = frames - hl;
        carryBytes[(int)xrows.size()] = (double)hl * 4.0;
      } else {
        hjobs.push_back(CoarseHistJob{oldHist, r.in, nextHist, hl, frames});
        maxHist = std::max(maxHist, hl);
        histBytes += 2.0 * (double)hl * 4.0;
      }
      xrows.push_back(r);
    }
    nd.dHistCur ^= 1;
    nd.dHistZero = false;
    addPieces(nd, nd.dLeader >= 0 ? nd.dLeader : id, nxr, xFrame, xIndex);
  }
  // ---- pre-mixed groups: one mixed signal [history | chunk] per input channel, transformed like a single convolver's input ----
  for (auto& kv : groups) {
    GroupInfo& g = kv.second;
    if (!g.premix) continue;
    NodeS& ld = *c.nodes[kv.first];
    const int64_t hl = g.hl0;
    int xFrame[32], xIndex[32];
    for (int ch = 0; ch < g.nxr; ch++) {
      const size_t bytes = (size_t)(hl + frames) * sizeof(float);
      if (pmUsed + bytes > c.coarseM.bytes) fail(GA_ERR_INVALID_OPERATION, "internal: the pre-mix arena is too small for the plan");
      float* mixed = (float*)((char*)c.coarseM.p + pmUsed);
      pmUsed += (bytes + 255) & ~(size_t)255;
      auto job = [&](float* out, const std::vector<PremixTerm>& tv, int64_t len, int64_t carryFrom) {
        PremixJob j{out, (int)pmTerms.size(), 0, len, carryFrom, 1, 0};
        for (const PremixTerm& t : tv) {
          if (!t.in) continue;   // (silent: adds nothing, and has no carry)
          pmTerms.push_back(t);
          j.nterms++;
          if ((uintptr_t)t.in & 15) j.flags &= ~1;
          if (t.carry) j.flags |= 2;
          pmBytes += (double)len * 4.0;
          pmFlops += (double)len * 4.0;   // (compensated summation: four operations per sample)
        }
        pmJobs.push_back(j);
        pmMaxN = std::max(pmMaxN, len);
        pmBytes += (double)len * 4.0;
      };
      if (!g.terms) g.terms = std::make_unique<GroupInfo::Terms>();
      if (!g.noHist) job(mixed, g.terms->hist[ch], hl, hl);
      job(mixed + hl, g.terms->in[ch], frames, std::max<int64_t>(0, frames - hl));
      CoarseXRow r{};
      r.hist = g.noHist ? nullptr : mixed;
      r.in = mixed + hl;
      r.nvalid = frames;
      r.frame0 = frameNext;
      r.u0 = g.noHist ? 0 : -(g.maxP - 1);
      r.n_frames = (g.tail ? nT + 1 : nT) - r.u0;
      r.hist_len = (int)hl;
      r.flags = 0;
      r.scale = 1.0f;
      r.carry = nullptr;
      r.carry_from = 0;
      xFrame[ch] = frameNext;
      xIndex[ch] = (int)xrows.size();
      frameNext += r.n_frames;
      xrows.push_back(r);
    }
    addPieces(ld, kv.first, g.nxr, xFrame, xIndex);
    c.stats.coarse_premixed_signals += (int64_t)g.members * g.nxr;
  }
  // the previous chunk's bus on its way to the caller's page-locked rows (Context::pendingHandOver): one-term jobs at the head of
  // this launch -- their workgroups write over PCIe while the others stream the members' samples from HBM
  if (!pmJobs.empty() && !c.pendingHandOver.empty()) {
    std::vector<PremixJob> head;
    for (const Context::HandOver& h : c.pendingHandOver) {
      head.push_back(PremixJob{h.dst_dev, (int)pmTerms.size(), 1, h.n, h.n, 1 | 4, 0});
      pmTerms.push_back(PremixTerm{h.src, nullptr});
      pmMaxN = std::max(pmMaxN, h.n);
      pmBytes += 2.0 * (double)h.n * 4.0;
    }
    pmJobs.insert(pmJobs.begin(), head.begin(), head.end());
    c.pendingHandOver.clear();
    c.stats.deferred_handovers++;
  } else if (!c.pendingHandOver.empty() && !xrows.empty()) {   // no pre-mix launch: they ride in the stage's first forward launch
    for (const Context::HandOver& h : c.pendingHandOver) fwdHandOver.push_back(CoarseHandOver{h.src, h.dst_dev, h.n});
    c.pendingHandOver.clear();
    c.stats.deferred_handovers++;
  }
  ex.flushLevel();   // (materialised inputs)
}

void CoarseStage::buildJobs() {
  // ---- jobs: pieces with the same (leader, output channels, partitions) accumulate into the same Y rows ----
  struct Key {
    int leader, P, ncol;
    std::array<int, 16> out;
    bool operator<(const Key& o) const { return std::tie(leader, P, ncol, out) < std::tie(o.leader, o.P, o.ncol, o.out); }
  };
  std::map<Key, std::vector<const Piece*>> byKey;
  for (const Piece& pc : pieces) {
    Key k{pc.leader, pc.P, pc.ncol, {}};
    k.out.fill(-1);
    for (int j = 0; j < pc.ncol; j++) k.out[j] = pc.outCh[j];
    byKey[k].push_back(&pc);
  }
  // groups of signals: the multiply-accumulate jobs of group g run (second stream) while group g + 1 is transformed
  nxAll = (int)xrows.size();
  G = !c.coarseOverlap ? 1 : (nxAll >= 256 ? 4 : (nxAll >= 64 ? 2 : 1));
  auto groupOf = [&](int xrow) { return std::min(G - 1, (int)((int64_t)xrow * G / std::max(nxAll, 1))); };
  for (int g = 0; g <= G; g++) gBegin[g] = 0;
  for (int x = 0; x < nxAll; x++) gBegin[groupOf(x) + 1] = x + 1;
  for (int g = 1; g <= G; g++) gBegin[g] = std::max(gBegin[g], gBegin[g - 1]);
  for (auto& kv : byKey) {
    const Key& k = kv.first;
    const int cw = k.ncol, ci = cw == 1 ? 0 : (cw == 2 ? 1 : (cw == 4 ? 2 : 3));
    const auto& pv = kv.second;
    for (size_t p0 = 0; p0 < pv.size(); p0 += kVoicesPerJob) {
      const size_t p1 = std::min(pv.size(), p0 + kVoicesPerJob);
      const int term0 = (int)terms.size();
      bool shared = true;
      int lastX = 0;
      for (size_t i = p0; i < p1; i++) {
        const Piece& pc = *pv[i];
        CoarseTerm t{};
        t.frame0 = pc.frame0 - (pc.u0 + (pc.P - 1));   // frame the window u = -(P - 1) would have (the kernels index from there)
        for (int j = 0; j < 16; j++) t.h[j] = nullptr;
        for (int j = 0; j < cw; j++) t.h[j] = pc.ir->coarse + (size_t)pc.irCh[j] * pc.P * kCoarseBins;
        if (i > p0)
          for (int j = 0; j < cw; j++) shared = shared && (t.h[j] == terms[term0].h[j]);
        terms.push_back(t);
        lastX = std::max(lastX, pc.xrow);
      }
      if (cw == 16) {
        shared = false;   // (the reduction kernel has no 16-column instance)
        for (size_t i = term0; i < terms.size(); i++)   // the matrix-core kernel addresses the columns' spectra from h[0]
          for (int j = 1; j < 16; j++) wideStrided = wideStrided && terms[i].h[j] == terms[i].h[0] + (size_t)j * k.P * kCoarseBins;
      }
      const int grp = groupOf(lastX);
      const int yrow0 = yNext;
      yNext += cw;
      for (int j = 0; j < cw; j++) outRows[{k.leader, k.out[j]}].push_back(yrow0 + j);
      const int jb = shared ? kCoarseSumJobBlocks(cw) : kCoarseJobBlocks(cw);   // (the two kernels' job sizes)
      const GroupInfo& gi = groups[k.leader];
      const int nTo = gi.tail ? nT + gi.maxP : nT;   // output blocks: the chunk's, and with a tail those the chunk's input still reaches
      for (int t0 = 0; t0 < nTo; t0 += jb) {
        CoarseJob jb_{};
        jb_.term0 = term0;
        jb_.n_terms = (int)(p1 - p0);
        jb_.P = k.P;
        jb_.t0 = t0;
        jb_.n_t = std::min(jb, nTo - t0);
        jb_.yrow0 = yrow0;
        jb_.shared_h = shared ? 1 : 0;
        jb_.u_lo = gi.noHist ? 0 : -(k.P - 1);
        jb_.u_hi = gi.tail ? nT : nT - 1;
        const int cj = 2 * ci + (shared ? 1 : 0);
        jobs[cj][grp].push_back(jb_);
        maxT[cj] = std::max(maxT[cj], jb_.n_t);
        maxP[cj] = std::max(maxP[cj], k.P);
        while (k.P % pbOf[cj]) pbOf[cj] >>= 1;
        const int fread = std::max(0, std::min(jb_.u_hi, t0 + jb_.n_t - 1) - std::max(jb_.u_lo, t0 - (k.P - 1)) + 1);   // frames that exist
        // (the spectra of the terms are necessary bytes ONCE: the block ranges of one group of terms are neighbours in the grid and
        // the second range finds them in the L2 -- PMC: profiles/r03_config5_pmc_hbm_traffic.json)
        macBytes[cj][grp] += (double)jb_.n_terms * fread * kCoarseBins * 8.0 +
                             (t0 == 0 ? (double)(shared ? 1 : jb_.n_terms) * k.P * cw * kCoarseBins * 8.0 : 0.0) + (double)cw * jb_.n_t * kCoarseBins * 8.0;
        // complex multiply-adds (8 flops): every term's products in the general kernel; in the reduction the terms' frames are
        // added up first (2 flops per complex value) and the sum is multiplied once
        macFlops[cj][grp] += shared ? ((double)jb_.n_terms * fread * 2.0 + (double)k.P * jb_.n_t * cw * 8.0) * kCoarseBins
                                    : (double)jb_.n_terms * k.P * jb_.n_t * cw * 8.0 * kCoarseBins;
      }
    }
  }
}

void CoarseStage::buildOutputs() {
  int maxPAll = 0;
  for (auto& kv : groups)
    if (kv.second.tail) maxPAll = std::max(maxPAll, kv.second.maxP);
  yFrames = nT + maxPAll;   // coarse blocks per Y row (rows of groups with a shorter or no tail leave their end unused)
  invBlocks = nT;
  if (tails) {   // tail buffers live with the group's leader: [2][channels][tail_len], read one, write the other
    std::map<int, int> chOf;
    for (auto& kv : outRows) chOf[kv.first.first] = std::max(chOf[kv.first.first], kv.first.second + 1);
    for (auto& kv : groups) {
      if (!kv.second.tail) continue;
      NodeS& ld = *c.nodes[kv.first];
      const int64_t len = (int64_t)(kv.second.maxP + 1) * kCoarseBlock;
      const int nch = chOf.count(kv.first) ? chOf[kv.first] : 0;
      if (nch == 0) continue;
      if (!ld.dTail[0] || ld.dTailLen != len || ld.dTailCh != nch) {
        if (kv.second.carried) fail(GA_ERR_INVALID_OPERATION, "internal: a carried tail changed its shape");
        GA_HIP(hipStreamSynchronize(c.stream));
        for (int b = 0; b < 2; b++) {
          if (ld.dTail[b]) c.dfree(ld.dTail[b], (size_t)ld.dTailLen * ld.dTailCh * sizeof(float));
          ld.dTail[b] = (float*)c.dalloc((size_t)len * nch * sizeof(float));
        }
        ld.dTailLen = len;
        ld.dTailCh = nch;
        ld.dTailCur = 0;
      }
    }
  }
  for (auto& kv : outRows) {
    CoarseOut o{};
    o.out = ex.nodeOut(kv.first.first, kv.first.second);
    o.nvalid = frames;
    o.y0 = (int)ylist.size();
    o.ny = (int)kv.second.size();
    o.n_y = nT;
    if (groups[kv.first.first].tail) {
      NodeS& ld = *c.nodes[kv.first.first];
      const GroupInfo& gi = groups[kv.first.first];
      o.n_y = nT + gi.maxP;
      o.tail_len = ld.dTailLen;
      o.tail_in = gi.carried ? ld.dTail[ld.dTailCur] + (size_t)kv.first.second * ld.dTailLen : nullptr;
      o.tail_out = ld.dTail[ld.dTailCur ^ 1] + (size_t)kv.first.second * ld.dTailLen;
      invBlocks = std::max(invBlocks, (int)((frames + o.tail_len + kCoarseBlock - 1) / kCoarseBlock));
      invBytes += (double)o.tail_len * 4.0 * (gi.carried ? 2.0 : 1.0);
      if (gi.carried) c.stats.coarse_carried_outputs++;
    }
    ylist.insert(ylist.end(), kv.second.begin(), kv.second.end());
    outs.push_back(o);
    invBytes += (double)o.ny * o.n_y * kCoarseBins * 8.0 + (double)frames * 4.0;
    invFlops += (double)o.n_y * (kCoarseTransformFlops + (double)o.ny * 2.0 * kCoarseBins);
  }
  if (tails)
    for (auto& kv : groups) {   // this chunk's tails are the next chunk's, if the group is still the same then
      NodeS& ld = *c.nodes[kv.first];
      if (!kv.second.tail || !ld.dTail[0]) continue;
      ld.dTailCur ^= 1;
      ld.dTailSig = kv.second.sig;
      ld.dTailSeq = c.chunkSeq;
    }
  if ((size_t)frameNext * kCoarseBins * sizeof(float2) > c.coarseX.bytes || (size_t)yNext * yFrames * kCoarseBins * sizeof(float2) > c.coarseY.bytes)
    fail(GA_ERR_INVALID_OPERATION, "internal: coarse spectra arenas are too small for the plan");

}

void CoarseStage::enqueue() {
  // (the launches below run after this object is gone: they capture locals, never members)
  const int G = this->G, yFrames = this->yFrames, invBlocks = this->invBlocks, nxAll = this->nxAll;
  const double invBytes = this->invBytes, histBytes = this->histBytes;
  const size_t xo = ex.plan.putv(xrows), ho = ex.plan.putv(hjobs), to = ex.plan.putv(terms), oo = ex.plan.putv(outs), yo = ex.plan.putv(ylist);
  const size_t pjo = ex.plan.putv(pmJobs), pto = ex.plan.putv(pmTerms), fho = ex.plan.putv(fwdHandOver);
  const int nfh = (int)fwdHandOver.size();
  const int npm = (int)pmJobs.size();
  const int64_t pmMaxN = this->pmMaxN;
  const double pmBytes = this->pmBytes, pmFlops = this->pmFlops, invFlops = this->invFlops;
  struct MacLaunch { size_t off; int nj, cw, mt, mp, pb, grp; bool ap; double bytes, flops; };
  const bool matrixCores = c.coarseMfma && wideStrided;
  std::vector<MacLaunch> macs;
  for (int g = 0; g < G; g++)
    for (int i = 0; i < 8; i++) {
      if (jobs[i][g].empty()) continue;
      macs.push_back(MacLaunch{ex.plan.putv(jobs[i][g]), (int)jobs[i][g].size(), kCwOf[i >> 1], maxT[i], maxP[i], pbOf[i], g, (i & 1) == 0,
                               macBytes[i][g], macFlops[i][g]});
      c.stats.mac_launches += 1;
    }
  hipStream_t st = c.stream;
  float2* X = (float2*)c.coarseX.p;
  float2* Y = (float2*)c.coarseY.p;
  const float2* tw16 = c.twiddles16(4096);
  const float2* twFwd = c.twiddles16pw();
  const float2* twab = c.coarseTwab();
  const int nh = (int)hjobs.size(), no = (int)outs.size();
  // per group: rows, longest row, bytes, windows per workgroup (long runs fetch every input sample once; keep >= ~4
  // workgroups per CU's worth of parallelism)
  struct FwdLaunch { int x0, nx, maxFrames, run; double bytes, flops; };
  std::vector<FwdLaunch> fwds;
  for (int g = 0; g < G; g++) {
    FwdLaunch f{this->gBegin[g], this->gBegin[g + 1] - this->gBegin[g], 0, 1, 0.0, 0.0};
    for (int x = f.x0; x < f.x0 + f.nx; x++) {
      f.maxFrames = std::max(f.maxFrames, xrows[x].n_frames);
      f.bytes += (double)(xrows[x].n_frames + 1) * kCoarseBlock * 4.0 + (double)xrows[x].n_frames * kCoarseBins * 8.0;
      f.flops += (double)xrows[x].n_frames * kCoarseTransformFlops;
      if (auto it = carryBytes.find(x); it != carryBytes.end()) f.bytes += it->second;
    }
    while (f.run < 16 && (int64_t)nxAll * ((f.maxFrames + 2 * f.run - 1) / (2 * f.run)) >= 1024) f.run *= 2;
    if (f.run >= 8) {   // equal runs of about 12 windows (measured on config 3: 10, 12 and 20 per run beat 16 + a short last run by 2.5 %)
      const int k = (f.maxFrames + 11) / 12;
      f.run = (f.maxFrames + k - 1) / k;
    }
    if (const char* e = expenv("GA_COARSE_RUN")) f.run = std::max(1, atoi(e));   // measurements only
    fwds.push_back(f);
  }
  if (G > 1) c.ensureOverlapStream();
  Context* cp = &c;
  ex.plan.add(GA_STAGE_COARSE_SECTION, [=](uint8_t* base) {
    // one piece of the section: a launch with its own profile events (the two stages overlap on two streams)
    auto timed = [&](hipStream_t sx, int kind, double bytes, double flops, const std::function<const char*()>& launch) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (cp->profileNow) {
        GA_HIP(hipEventCreate(&e0));
        GA_HIP(hipEventCreate(&e1));
        GA_HIP(hipEventRecord(e0, sx));
      }
      cp->noteKernel(kind, launch());
      if (cp->profileNow) {
        GA_HIP(hipEventRecord(e1, sx));
        cp->extraProf.push_back(Context::ExtraProf{e0, e1, kind, bytes});
      }
      cp->stats.kernel_launches++;
      cp->stats.stage_launches[kind]++;
      cp->stats.stage_bytes[kind] += bytes;
      cp->stats.stage_flops[kind] += flops;
    };
    hipStream_t s2 = G > 1 ? cp->stream2 : st;
    if (npm > 0)
      timed(st, LK_CPREMIX, pmBytes, pmFlops, [&] { return launch_coarse_premix(st, (const PremixJob*)(base + pjo), npm, (const PremixTerm*)(base + pto), pmMaxN); });
    for (int g = 0; g < G; g++) {
      const FwdLaunch& f = fwds[g];
      if (f.nx > 0)
        timed(st, LK_CFWD, f.bytes, f.flops, [&] {
          return launch_coarse_fwd(st, (const CoarseXRow*)(base + xo) + f.x0, f.nx, f.maxFrames, f.run, X, twFwd, twab,
                                   (const CoarseHandOver*)(base + fho), g == 0 ? nfh : 0);
        });
      if (G > 1) {
        GA_HIP(hipEventRecord(cp->dGroupEv[g], st));
        GA_HIP(hipStreamWaitEvent(s2, cp->dGroupEv[g], 0));
      }
      for (const MacLaunch& m : macs)
        if (m.grp == g)
          timed(s2, LK_CMAC, m.bytes, m.flops, [&] {
            return launch_coarse_mac(s2, (const CoarseJob*)(base + m.off), m.nj, (const CoarseTerm*)(base + to), X, Y, yFrames, m.cw, m.mt, m.mp, m.ap, m.pb,
                                     matrixCores);
          });
    }
    if (G > 1) {   // join: the inverse transforms (and the next chunk's forward transforms, which reuse X) wait for every job
      GA_HIP(hipEventRecord(cp->dJoinEv, s2));
      GA_HIP(hipStreamWaitEvent(st, cp->dJoinEv, 0));
    }
  });
  ex.plan.add(LK_CINV, [=](uint8_t* base) {
    cp->noteKernel(LK_CINV, launch_coarse_inv(st, (const CoarseOut*)(base + oo), no, invBlocks, (const int*)(base + yo), Y, yFrames, tw16, twab));
  }, invBytes, invFlops);
  const int64_t mh = maxHist;
  if (nh > 0) ex.plan.add(LK_CHIST, [=](uint8_t* base) { launch_coarse_hist(st, (const CoarseHistJob*)(base + ho), nh, mh); }, histBytes);
}
}  // namespace

static void planCoarseStage(Context& c, Exec& ex, const std::vector<int>& dNodes, int64_t n) {
  CoarseStage s(c, ex, dNodes, n);
  s.resolveInputs();
  s.classifyGroups();
  s.buildRows();
  s.buildJobs();
  s.buildOutputs();
  s.enqueue();
}

// When everything the destination receives in this chunk is ONE fused group of formulation D convolvers (or a single one) with
// the bus's channel count, the group's inverse transforms write the bus themselves: the leader's output slabs ARE the bus
// rows (the caller's device rows or page-locked host rows when Context::render set busTarget), and the destination's mix -- a
// copy of one term -- disappears (Exec::resolveInSeg skips a forced target that already holds its only term).
void Context::aliasBusToLeader(ChunkRun& r) {
  Exec& ex = *r.ex;
  int leader = -1, nch = 0;
  for (const Segment& sg : r.segs) {
    if (sg.nodes.empty() || sg.nodes.back().id != 0 || sg.nodes.back().ins.empty()) return;   // (node 0 is the destination)
    const InSeg& is = sg.nodes.back().ins[0];
    for (const TermS& t : is.terms) {
      const NodeS& nd = *nodes[t.node];
      if (nd.type != GA_NODE_CONVOLVER || !nd.ir || nd.convPath != 4 || nd.dLeader < 0 || t.out != 0 || t.ch != is.bufCh) return;
      if (leader < 0) {
        leader = nd.dLeader;
        nch = is.bufCh;
      } else if (leader != nd.dLeader || nch != is.bufCh) {
        return;
      }
    }
  }
  if (leader < 0 || nch < 1 || nch > (int)busSlabs.size()) return;
  const NodeS& ld = *nodes[leader];
  if (ld.outputs.empty() || ld.outputs[0].connectedInputs.size() != 1) return;
  const InRef& to = ld.outputs[0].connectedInputs[0];
  if (to.node != 0 || to.input != 0) return;
  for (int ch = 0; ch < nch; ch++) ex.setNodeOut(leader, ch, busTarget[ch] ? busTarget[ch] : busSlabs[ch]);
}

// pass 5: convolver formulations of new nodes, fusion groups, scratch arenas (sized before any recorded launch captures them)
void Context::chunkConvScratch(ChunkRun& r) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  // resampler trajectories used in this chunk go into one device table
  for (auto& kv : resamplers) kv.second->devOffset = -1;

  // ---- convolver scratch planes are shared by all groups: size them for the largest group BEFORE any recorded
  //      launch captures their address ----
  if (topoHasConvolvers) {   // (a graph without convolvers -- tens of thousands of nodes of config 4 -- skips these sweeps)
    refOrderSensitivity(topo);
    assignConvPaths(topo, n);
    for (int id : topo) {
      NodeS& nd = *nodes[id];
      if (nd.type == GA_NODE_CONVOLVER) nd.refOrder = nd.refSens && nd.ir && (nd.convPath == 2 || nd.convPath == 3);
    }
    planCoarseFusion(topo, segs);
    for (int id : topo)
      if (nodes[id]->type == GA_NODE_CONVOLVER) nodes[id]->dGroupSize = 0;
    for (int id : topo) {
      const NodeS& nd = *nodes[id];
      if (nd.type == GA_NODE_CONVOLVER && nd.ir && nd.convPath == 4 && nd.dLeader >= 0) nodes[nd.dLeader]->dGroupSize++;
    }
    aliasBusToLeader(r);
  }
  bHistMax = 0;
  if (topoHasConvolvers) {
    size_t xMax = 0, yMax = 0;
    size_t bx = 0, by = 0;
    for (int id : topo) {
      NodeS& nd = *nodes[id];
      if (nd.type != GA_NODE_CONVOLVER || !nd.ir) continue;
      if (nd.convPath == 2 || nd.convPath == 3) {
        bx += nd.bInCh;
        by += nd.bSlots;
        bHistMax = std::max(bHistMax, nd.ir->P - 1);
      }
      for (auto& rr : nd.convRows) {
        ConvGroup& g = *rr.group;
        ensureGroupState(g);
        const int ty_ = (int)roundup(n, 64), tx_ = ty_ + g.P + 128;
        xMax = std::max(xMax, (size_t)kBins * tx_ * g.rp * sizeof(float));
        yMax = std::max(yMax, (size_t)kBins * ty_ * g.rp * sizeof(float));
      }
    }
    if (xMax) {
      ensure(planes[0], xMax);
      ensure(planes[1], xMax);
      ensure(planes[2], yMax);
      ensure(planes[3], yMax);
    }
    {  // formulation D: the stages of a chunk run one after the other on the stream and share the two arenas
      std::map<int, std::pair<size_t, size_t>> perDepth;   // depth -> (X frames, Y frames upper bound)
      const int64_t nT = (n * kBlock + kCoarseBlock - 1) / kCoarseBlock;
      for (int id : topo) {
        NodeS& nd = *nodes[id];
        if (nd.type != GA_NODE_CONVOLVER || !nd.ir || nd.convPath != 4) continue;
        auto& pd = perDepth[nd.depth];
        pd.first += (size_t)nd.bInCh * (size_t)(nT + nd.ir->coarseP);   // (+ the window behind the chunk's last block: carried tails)
        // Y rows: one per slot unless fused; fused groups need (members / 32 + 1) x channels rows, never more than the slots
        pd.second += (size_t)nd.bSlots * (size_t)(nT + kCoarseMaxP);
      }
      size_t xf = 0, yf = 0;
      for (auto& kv : perDepth) {
        xf = std::max(xf, kv.second.first);
        yf = std::max(yf, kv.second.second);
      }
      if (xf) {
        ensure(coarseX, xf * kCoarseBins * sizeof(float2));
        ensure(coarseY, yf * kCoarseBins * sizeof(float2));
      }
      if (coarsePremix) {   // pre-mixed groups: [history | chunk] of the mixed signal per input channel of the group
        std::map<int, int> members;
        for (int id : topo) {
          NodeS& nd = *nodes[id];
          if (nd.type == GA_NODE_CONVOLVER && nd.ir && nd.convPath == 4) members[nd.dLeader >= 0 ? nd.dLeader : id]++;
        }
        std::map<int, size_t> pmDepth;
        for (auto& kv : members) {
          if (kv.second < 2) continue;
          const NodeS& ld = *nodes[kv.first];
          pmDepth[ld.depth] += (size_t)ld.bInCh * ((((size_t)(ld.dHistLen + n * kBlock) * sizeof(float)) + 255) & ~(size_t)255);
        }
        size_t pm = 0;
        for (auto& kv : pmDepth) pm = std::max(pm, kv.second);
        if (pm) ensure(coarseM, pm);
      }
    }
    bRowX = bRowY = 0;
    if (bx) {  // formulation B scratch: [row][bin][block]; x planes alternate between two pairs (flushPlaneHistories)
      const size_t txb = (size_t)roundup(bHistMax, 4) + roundup(n, 16) + 16, tyb = (size_t)roundup(n, 256);
      bPairWrite = bPairCur ^ 1;
      flushPlaneHistories(bPairWrite);
      // both pairs grow together (a render that continues reaches the other pair in its next chunk; growing it then would
      // put an allocation into the steady state), but only a pair without residents can be reallocated
      const size_t xb = bx * kBins * txb * sizeof(float);
      if (planesB[0].bytes < xb || planesBalt[0].bytes < xb) flushPlaneHistories(bPairCur);
      ensure(planesB[0], xb);
      ensure(planesB[1], xb);
      ensure(planesBalt[0], xb);
      ensure(planesBalt[1], xb);
      ensure(planesB[2], by * kBins * tyb * sizeof(float));
      ensure(planesB[3], by * kBins * tyb * sizeof(float));
    }
  }

  if (topoHasOscillators)
  for (int id : topo) {  // OscillatorNode._phase lives on the device (one double, zero at Start)
    NodeS& nd = *nodes[id];
    if (nd.type != GA_NODE_OSCILLATOR || nd.oscPhase) continue;
    nd.oscPhase = (double*)dalloc(64);
    GA_HIP(hipMemsetAsync(nd.oscPhase, 0, 64, stream));
  }
}

// formulation A: convolvers that share an impulse-response channel run as one group per (impulse response, channel) -- the
// banded-Toeplitz matrix-core kernel or the block-axis transforms over all their rows (PartitionedConvolver.cs:104-223)
void Context::planConvolversShared(ChunkRun& r, int d, ConvPlanCtx& k) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  Exec& ex = *r.ex;
  (void)d; (void)topo;
  auto& active = k.active; auto& tsTemps = k.tsTemps; auto& prevIns = k.prevIns; int& prevP = k.prevP; int& prevRp = k.prevRp; int& prevRows = k.prevRows;
  for (auto& kv : active) {
    ConvGroup& g = *kv.first;
    const int P = g.P, hist = P - 1;
    const int nrows = (int)g.rows.size();
    std::vector<ConvRowIO> rio(nrows, ConvRowIO{nullptr, nullptr});
    for (auto& ns_ : kv.second) {
      NodeS& nd = *nodes[ns_.first];
      const int slot = ns_.second;
      const int idx = nd.convRows[slot].idx;
      // which input channel feeds this row: discrete -> slot ; true stereo -> L,L,R,R for h0,h1,h2,h3 (ConvolverNode.cs:127-151)
      const int inCh = nd.isTrueStereo ? (slot >> 1) : slot;
      const Exec::ConvInRow ci = ex.convIn[ns_.first];
      const float* stable = nullptr;
      bool same = true, first = true;
      for (size_t si = 0; si < segs.size(); si++) {
        const float* v = (ci[si].empty() || inCh >= (int)ci[si].size()) ? nullptr : ci[si][inCh];
        if (first) { stable = v; first = false; } else if (v != stable) same = false;
      }
      const float* in = stable;
      if (!same) {  // materialise: per segment copy / zero fill into a row slab
        float* slab = getSlab(*this);
        for (size_t si = 0; si < segs.size(); si++) {
          const float* v = (ci[si].empty() || inCh >= (int)ci[si].size()) ? nullptr : ci[si][inCh];
          MixJob mj;
          mj.out = slab;
          mj.term0 = (int)ex.terms.size();
          mj.nterms = v ? 1 : 0;
          mj.f0 = segs[si].b0 * kBlock;
          mj.n = (segs[si].b1 - segs[si].b0) * kBlock;
          if (v) {
            ex.terms.push_back(v);
            ex.noteAlign(v, mj.f0);
          }
          ex.mixJobs.push_back(mj);
        }
        in = slab;
      }
      float* out;
      if (nd.isTrueStereo) {
        out = getSlab(*this);  // temp1 / temp2, summed below (ConvolverNode.cs:137-143)
      } else {
        out = ex.nodeOut(ns_.first, slot);
      }
      rio[idx] = ConvRowIO{in, out};
      if (nd.isTrueStereo) {
        auto it = tsTemps.find(ns_.first);
        if (it == tsTemps.end()) it = tsTemps.emplace(ns_.first, std::array<float*, 4>{nullptr, nullptr, nullptr, nullptr}).first;
        it->second[slot] = out;
      }
    }
    ex.flushLevel();
    const int rp = g.rp;
    const int ty = (int)roundup(n, 64);
    const int tx = ty + P + 128;
    ConvPlanes pl{(float*)planes[0].p, (float*)planes[1].p, (float*)planes[2].p, (float*)planes[3].p, tx, ty, rp};
    size_t rioOff = ex.plan.putv(rio);
    hipStream_t st = stream;
    Twiddles tw{w128, w256};
    const int nn = (int)n;
    float* hR = g.histR;
    float* hI = g.histI;
    const bool hz = g.histZero;
    const float* hr = g.ir->hr + (size_t)g.irCh * kBins * P;
    const float* hi = g.ir->hi + (size_t)g.irCh * kBins * P;
    float* ovIn = g.overlap[g.ovCur];
    float* ovOut = g.overlap[g.ovCur ^ 1];
    g.ovCur ^= 1;
    g.histZero = false;
    // forward spectra depend only on the inputs: a group fed by exactly the same signals as the previous one (e.g. the
    // channels of one stereo IR behind mono voices) reuses the X rows that are still in the scratch planes
    std::vector<const float*> ins(nrows);
    for (int r = 0; r < nrows; r++) ins[r] = rio[r].in;
    const bool skipFwd = (prevP == P && prevRp == rp && prevRows == nrows && prevIns == ins);
    prevIns = ins;
    prevP = P;
    prevRp = rp;
    prevRows = nrows;
    ex.plan.add(LK_FFT, [=](uint8_t* base) {
      // frequency-domain delay line of the previous chunk(s) in front of this chunk's spectra
      if (hist > 0) {
        launch_plane_copy(st, pl.xr, tx, 0, hz ? nullptr : hR, hist, 0, hist, rp);
        launch_plane_copy(st, pl.xi, tx, 0, hz ? nullptr : hI, hist, 0, hist, rp);
      }
      // rows beyond this chunk that the banded MAC may touch for its (discarded) padded outputs
      int tail = std::min(tx - (hist + nn), 256);
      launch_plane_copy(st, pl.xr, tx, hist + nn, nullptr, 0, 0, tail, rp);
      launch_plane_copy(st, pl.xi, tx, hist + nn, nullptr, 0, 0, tail, rp);
      if (!skipFwd) launch_rfft_fwd(st, (const ConvRowIO*)(base + rioOff), nrows, nn, hist, pl, tw);
    });
    ex.plan.add(LK_MAC, [=](uint8_t*) { launch_spectral_mac_shared(st, pl, hr, hi, P, nn, nrows); });
    ex.plan.add(LK_FFT, [=](uint8_t* base) {
      launch_irfft_ola(st, (const ConvRowIO*)(base + rioOff), nrows, nn, pl, ovIn, ovOut, tw);
      if (hist > 0) {  // keep the last P-1 spectra for the next chunk (the FDL, PartitionedConvolver.cs:122-128)
        launch_plane_copy(st, hR, hist, 0, pl.xr, tx, nn, hist, rp);
        launch_plane_copy(st, hI, hist, 0, pl.xi, tx, nn, hist, rp);
      }
    });
    stats.mac_flops_total += 8.0 * P * kBins * (double)kv.second.size() * (double)n;
    // streaming-formulation bytes: per channel-instance per block FDL read + write + input, IR once per block per channel
    stats.mac_bytes_total += ((double)P * kBins * 8.0 + kBins * 8.0 + 512.0) * (double)kv.second.size() * (double)n +
                             (double)P * kBins * 8.0 * (double)n;
    stats.mac_launches += 1;
  }
}

// formulations B / C: nodes with an impulse response of their own (per-node planes; block-axis FFT segments or the direct sum)
void Context::planConvolversPrivate(ChunkRun& r, int d, ConvPlanCtx& k, bool refOrder) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  Exec& ex = *r.ex;
  (void)d; (void)topo;
  const std::vector<int>& bNodes = k.bNodes; auto& tsTemps = k.tsTemps;
  const int hist = (int)roundup(bHistMax, 4);   // plane time origin, 16-byte aligned rows
  const int txb = hist + (int)roundup(n, 16) + 16, tyb = (int)roundup(n, 256);
  // rows of this depth start after the rows of the depths before it: a node's spectra stay intact for the next chunk
  const size_t rowX0 = bRowX, rowY0 = bRowY;
  ConvPlanesB plb{xPlane(bPairWrite, 0) + rowX0 * kBins * txb, xPlane(bPairWrite, 1) + rowX0 * kBins * txb,
                  (float*)planesB[2].p + rowY0 * kBins * tyb, (float*)planesB[3].p + rowY0 * kBins * tyb, txb, tyb};
  std::vector<ConvRowIO> xrows, yrows;
  std::vector<ConvSetB> sets;
  std::map<int, std::vector<ConvSetC>> setsC;   // by P: launches per distinct segment length
  struct SetTaps { IrSpectra* ir; int slot[16]; };
  std::map<int, std::vector<SetTaps>> setsCTaps;   // which taps spectra each column of a set needs (filled per FFT length)
  std::vector<HistJobB> restore;
  std::vector<const float*> ovIn;
  std::vector<float*> ovOut;
  double flops = 0;
  std::vector<const float*> chIn;   // (scratch vectors live outside the node loop: a thousand convolvers per chunk)
  std::vector<float*> slotOut;
  std::vector<int> cols;
  for (int id : bNodes) {
    NodeS& nd = *nodes[id];
    const int P = nd.ir->P, h = P - 1;
    const Exec::ConvInRow ci = ex.convIn[id];
    // chunk-long input pointer of every input channel (stable view, or a materialised copy)
    chIn.assign(nd.bInCh, nullptr);
    for (int c = 0; c < nd.bInCh; c++) {
      const float* stable = nullptr;
      bool same = true, first = true;
      for (size_t si = 0; si < segs.size(); si++) {
        const float* v = (ci[si].empty() || c >= (int)ci[si].size()) ? nullptr : ci[si][c];
        if (first) { stable = v; first = false; } else if (v != stable) same = false;
      }
      if (same) {
        chIn[c] = stable;
      } else {
        float* slab = getSlab(*this);
        for (size_t si = 0; si < segs.size(); si++) {
          const float* v = (ci[si].empty() || c >= (int)ci[si].size()) ? nullptr : ci[si][c];
          MixJob mj;
          mj.out = slab;
          mj.term0 = (int)ex.terms.size();
          mj.nterms = v ? 1 : 0;
          mj.f0 = segs[si].b0 * kBlock;
          mj.n = (segs[si].b1 - segs[si].b0) * kBlock;
          if (v) {
            ex.terms.push_back(v);
            ex.noteAlign(v, mj.f0);
          }
          ex.mixJobs.push_back(mj);
        }
        chIn[c] = slab;
      }
    }
    bool allSame = true;
    for (int c = 1; c < nd.bInCh; c++) allSame = allSame && (chIn[c] == chIn[0]);
    const size_t hstride = (size_t)kBins * std::max(h, 1);
    if (nd.bShared && !allSame) {
      // the channels start to differ: every channel inherits the (so far common) history of channel 0
      if (!nd.bHistZero && h > 0 && nd.bHistPlane < 0)   // (a plane-resident shared row is simply read by every channel)
        for (int c = 1; c < nd.bInCh; c++) {   // (plan entries: ordered with the chunk's launches)
          float *dr = nd.bHistR + c * hstride, *di = nd.bHistI + c * hstride;
          const float *sr = nd.bHistR, *sim = nd.bHistI;
          hipStream_t st = stream;
          ex.plan.add(LK_OTHER, [=](uint8_t*) {
            GA_HIP(hipMemcpyAsync(dr, sr, hstride * 4, hipMemcpyDeviceToDevice, st));
            GA_HIP(hipMemcpyAsync(di, sim, hstride * 4, hipMemcpyDeviceToDevice, st));
          });
        }
      nd.bShared = false;
    }
    const int nxr = nd.bShared ? 1 : nd.bInCh;
    const int x0 = (int)xrows.size();
    for (int c = 0; c < nxr; c++) {
      const int xi = x0 + c;
      xrows.push_back(ConvRowIO{chIn[c], nullptr});
      float* xr_row = plb.xr + (size_t)xi * kBins * txb;
      float* xi_row = plb.xi + (size_t)xi * kBins * txb;
      // [0, hist - h) zeros, [hist - h, hist) this channel's history, rows after the chunk zero (K padding reads them)
      if (hist - h > 0) {
        restore.push_back(HistJobB{xr_row, nullptr, txb, 0, hist - h, 0});
        restore.push_back(HistJobB{xi_row, nullptr, txb, 0, hist - h, 0});
      }
      if (h > 0) {
        const float *srcR = nullptr, *srcI = nullptr;
        int sstride = h;
        if (nd.bHistZero) {
        } else if (nd.bHistPlane >= 0) {   // the previous chunk's x planes (never the pair being written: flushed above)
          const size_t off = (size_t)(nd.bHistRow + (nd.bHistNx == 1 ? 0 : c)) * kBins * nd.bHistTxb + nd.bHistOff;
          srcR = xPlane(nd.bHistPlane, 0) + off;
          srcI = xPlane(nd.bHistPlane, 1) + off;
          sstride = nd.bHistTxb;
        } else {
          srcR = nd.bHistR + c * hstride;
          srcI = nd.bHistI + c * hstride;
        }
        restore.push_back(HistJobB{xr_row + (hist - h), srcR, txb, sstride, h, 0});
        restore.push_back(HistJobB{xi_row + (hist - h), srcI, txb, sstride, h, 0});
      }
      const int tailn = txb - (hist + (int)n);
      restore.push_back(HistJobB{xr_row + hist + (int)n, nullptr, txb, 0, tailn, 0});
      restore.push_back(HistJobB{xi_row + hist + (int)n, nullptr, txb, 0, tailn, 0});
    }
    // slots: discrete -> slot c reads input c, IR channel c ; true stereo -> (L,h0) (L,h1) (R,h2) (R,h3)
    slotOut.assign(nd.bSlots, nullptr);
    for (int slot = 0; slot < nd.bSlots; slot++) {
      if (nd.isTrueStereo) {
        float* tmp = getSlab(*this);
        slotOut[slot] = tmp;
        auto it = tsTemps.find(id);
        if (it == tsTemps.end()) it = tsTemps.emplace(id, std::array<float*, 4>{nullptr, nullptr, nullptr, nullptr}).first;
        it->second[slot] = tmp;
      } else {
        slotOut[slot] = ex.nodeOut(id, slot);
      }
    }
    // sets: columns grouped by the x-row they read, at most 16 per set, y rows consecutive per set
    for (int xc = 0; xc < nxr; xc++) {
      cols.clear();
      for (int slot = 0; slot < nd.bSlots; slot++) {
        int inc = nd.isTrueStereo ? (slot >> 1) : slot;
        if (nd.bShared || inc == xc) cols.push_back(slot);
      }
      for (size_t c0 = 0; c0 < cols.size(); c0 += 16) {
        ConvSetB st{};
        st.x = x0 + xc;
        st.y0 = (int)yrows.size();
        st.ncol = (int)std::min<size_t>(16, cols.size() - c0);
        st.P = P;
        ConvSetC sc{};
        sc.x = st.x;
        sc.y0 = st.y0;
        sc.ncol = st.ncol;
        sc.P = P;
        for (int j = 0; j < st.ncol; j++) {
          int slot = cols[c0 + j];
          st.hr[j] = nd.ir->hr + (size_t)slot * kBins * P;   // slot index == IR channel index in both modes
          st.hi[j] = nd.ir->hi + (size_t)slot * kBins * P;
          sc.hs[j] = nullptr;   // per FFT length, below
          yrows.push_back(ConvRowIO{nullptr, slotOut[slot]});
          ovIn.push_back(nd.bOverlap + ((size_t)slot * 2 + nd.bOvCur) * kBlock);
          ovOut.push_back(nd.bOverlap + ((size_t)slot * 2 + (nd.bOvCur ^ 1)) * kBlock);
        }
        if (refOrder) {
          sets.push_back(st);
          stats.ref_order_rows += st.ncol;
        } else if (nd.convPath == 3) {
          SetTaps tp{nd.ir.get(), {}};
          for (int j = 0; j < st.ncol; j++) tp.slot[j] = cols[c0 + j];
          setsC[P].push_back(sc);
          setsCTaps[P].push_back(tp);
        } else {
          sets.push_back(st);
        }
      }
    }
    nd.bOvCur ^= 1;
    nd.bHistZero = false;
    if (h > 0) {   // the history of the next chunk: the last h spectra of these rows
      if (nd.bHistPlane == bPairWrite) fail(GA_ERR_DEVICE, "internal: convolver history lives in the planes being written");
      nd.bHistPlane = bPairWrite;
      nd.bHistRow = (int)rowX0 + x0;
      nd.bHistNx = nxr;
      nd.bHistOff = hist + (int)n - h;
      nd.bHistTxb = txb;
      bResidents[bPairWrite].push_back(id);
    }
    flops += 8.0 * P * kBins * (double)nd.bSlots * (double)n;
  }
  bRowX += xrows.size();
  bRowY += yrows.size();
  bPairCur = bPairWrite;
  ex.flushLevel();
  size_t xo = ex.plan.putv(xrows), yo = ex.plan.putv(yrows), so = ex.plan.putv(sets), ro = ex.plan.putv(restore),
         oi = ex.plan.putv(ovIn), oo = ex.plan.putv(ovOut);
  const int nx = (int)xrows.size(), ny = (int)yrows.size(), ns_ = (int)sets.size(), nr = (int)restore.size();
  hipStream_t st = stream;
  Twiddles tw{w128, w256};
  const int nn = (int)n;
  const int maxn = std::max(hist, txb - hist - nn);
  const bool f64 = fft64 || refOrder;   // (formulation R: the reference's FftFlat precision around its own partition sum)
  ex.plan.add(LK_FFT, [=](uint8_t* base) {
    launch_hist_copy_b(st, (const HistJobB*)(base + ro), nr, std::max(maxn, 1));
    launch_rfft_fwd_b(st, (const ConvRowIO*)(base + xo), nx, nn, hist, plb, tw, f64);
  });
  Context* cp = this;
  // (formulation R: 8 separately rounded operations per complex multiply-add -- the planner's count of what the launch executes)
  if (ns_ > 0 && refOrder)
    ex.plan.add(LK_MAC, [=](uint8_t* base) {
      launch_refmac(st, (const ConvSetB*)(base + so), ns_, nn, hist, plb);
      cp->noteKernel(LK_MAC, "refmac_kernel");
    }, 0.0, flops);
  else if (ns_ > 0)
    ex.plan.add(LK_MAC, [=](uint8_t* base) {
      launch_spectral_mac_b(st, (const ConvSetB*)(base + so), ns_, nn, hist, plb);
      cp->noteKernel(LK_MAC, "spectral_mac_b_kernel");
    }, 0.0, flops);
  for (auto& kv : setsC) {
    const int Pc = kv.first;
    static const char* r16env = expenv("GA_TCONV_RADIX16");   // A/B switches for measurements
    static const char* planenv = expenv("GA_TCONV_MIXED");
    const bool r16 = r16env ? atoi(r16env) != 0 : useRadix16;
    std::vector<TconvLaunch> tplan;
    if (debugTconvN2 > 0) {   // tests: a length no kernel exists for must come back as an error code, not abort the host
      const int Lc = std::max(1, debugTconvN2 - (Pc - 1));
      tplan.push_back(TconvLaunch{debugTconvN2, 0, (nn + Lc - 1) / Lc});
    } else if (r16 && !(planenv && atoi(planenv) == 0)) {
      tplan = tconvPlan(nn, Pc);
    } else {   // one FFT length for the whole chunk
      const int N2 = tapFftSize(Pc), Lc = N2 - (Pc - 1);
      tplan.push_back(TconvLaunch{N2, 0, (nn + Lc - 1) / Lc});
    }
    const std::vector<SetTaps>& taps = setsCTaps[Pc];
    for (const TconvLaunch& tl : tplan) {
      std::vector<ConvSetC> sv = kv.second;
      for (size_t i = 0; i < sv.size(); i++) {
        const float2* hsp = ensureTapSpectra(*taps[i].ir, tl.N2);
        for (int j = 0; j < sv[i].ncol; j++) sv[i].hs[j] = hsp + (size_t)taps[i].slot[j] * kBins * tl.N2;
      }
      size_t co = ex.plan.putv(sv);
      const int nc = (int)sv.size();
      const int N2 = tl.N2, tbase = tl.tbase, nseg = tl.nseg;
      const float2* twc = r16 ? twiddles16(N2) : twiddlesC(N2);
      ex.plan.add(LK_MAC, [=](uint8_t* base) {
        if (r16) launch_tconv16(st, (const ConvSetC*)(base + co), nc, nn, hist, plb, N2, twc, nseg, tbase);
        else launch_tconv(st, (const ConvSetC*)(base + co), nc, nn, hist, plb, N2, twc, nseg);
      });
    }
  }
  ex.plan.add(LK_FFT, [=](uint8_t* base) {
    launch_irfft_ola_b(st, (const ConvRowIO*)(base + yo), ny, nn, plb, (const float* const*)(base + oi), (float* const*)(base + oo), tw, f64);
  });
  stats.mac_flops_total += flops;
  // streaming-formulation bytes: per channel-instance per block FDL read + write + input; every distinct
  // IR channel is counted once per block however many nodes share it
  {
    std::map<std::pair<IrSpectra*, int>, int> distinct;
    double bytes = 0;
    for (int id : bNodes) {
      NodeS& nd = *nodes[id];
      const int P = nd.ir->P;
      bytes += ((double)P * kBins * 8.0 + kBins * 8.0 + 512.0) * nd.bSlots * (double)n;
      for (int sl = 0; sl < nd.bSlots; sl++) distinct[{nd.ir.get(), sl}] = P;
    }
    for (auto& kv : distinct) bytes += (double)kv.second * kBins * 8.0 * (double)n;
    stats.mac_bytes_total += bytes;
  }
  stats.mac_launches += 1;
}

// pass 7 (per convolver depth d): the convolvers whose inputs are complete, once per chunk over all blocks
void Context::chunkPlanConvolvers(ChunkRun& r, int d) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  Exec& ex = *r.ex;
    // ---- convolvers whose inputs are complete (depth d): once per chunk over all blocks ----
    // group -> (node, slot); ordered by (IR buffer, IR channel) so groups fed by the same inputs are adjacent
    ConvPlanCtx k;
    auto& active = k.active;
    std::vector<int>& bNodes = k.bNodes;  // formulation B / C nodes of this depth
    std::vector<int>& dNodes = k.dNodes;  // formulation D nodes of this depth
    for (int id : topo) {
      NodeS& nd = *nodes[id];
      if (nd.type != GA_NODE_CONVOLVER || !nd.ir || nd.depth != d) continue;
      if (!ex.convIn.has(id)) continue;
      if (nd.convPath == 4) {
        dNodes.push_back(id);
        continue;
      }
      if (nd.convPath >= 2) {
        bNodes.push_back(id);
        continue;
      }
      for (int slot = 0; slot < (int)nd.convRows.size(); slot++) active[nd.convRows[slot].group].push_back({id, slot});
    }
    if (!active.empty()) planConvolversShared(r, d, k);
    // ---- formulation D: coarse partitions, consumer sums fused in the frequency domain ----
    if (!dNodes.empty()) planCoarseStage(*this, ex, dNodes, n);
    // ---- formulations B / C: nodes with a private impulse response ----
    // (those that this chunk evaluates in the reference's own order -- formulation R -- in a pass of their own: double-precision
    // transforms and launch_refmac instead of the matrix-core / block-axis-FFT partition sums)
    if (!bNodes.empty()) {
      std::vector<int> plain, ref;
      for (int id : bNodes) (nodes[id]->refOrder ? ref : plain).push_back(id);
      if (!plain.empty()) {
        bNodes = plain;
        planConvolversPrivate(r, d, k, false);
      }
      if (!ref.empty()) {
        bNodes = ref;
        planConvolversPrivate(r, d, k, true);
      }
    }
    auto& tsTemps = k.tsTemps;
    // true stereo: outL = conv0(L) + conv2(R) ; outR = conv1(L) + conv3(R)  (ConvolverNode.cs:127-144)
    for (auto& kv : tsTemps) {
      hipStream_t st = stream;
      int64_t fr = frames;
      for (int o = 0; o < 2; o++) {
        float* out = ex.nodeOut(kv.first, o);
        float *a = kv.second[o], *b2 = kv.second[o + 2];
        if (a && b2) ex.plan.add(LK_OTHER, [=](uint8_t*) { launch_pair_sum(st, out, a, b2, fr); });
      }
    }
}

}  // namespace ga
