// Host model of the tile-parallel numEntriesScannedInFilter (pinot_amd/csrc/pg_filter_stats_tiles.h) against a direct, doc-by-doc restatement of
// the reference's iterators: AndDocIdIterator.java:37-66, OrDocIdIterator.java:91-119 (advance() only, which is all an AND ever calls on its
// children), NotDocIdIterator.java:28-70 and SVScanDocIdIterator.java:76-112 (both entry points, with its batches of 256).  Random match bitmaps of
// every density; an AND whose children are scans, bitmaps, NOTs over a scan or over an OR of leaves, ORs of leaves, NOTs and ANDs, and ANDs of their own.
// Build: g++ -O2 -std=c++17 -I pinot_amd/csrc tests/filter_stats_tiles_test.cpp -o tests/_build/filter_stats_tiles_test
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <vector>

#include "pg_filter_stats_tiles.h"

typedef std::vector<uint64_t> Bits;

struct Leaf {
  Bits m;
  bool scan;            // a scan counts what it steps over; a bitmap iterator counts nothing
  int64_t counted = 0;  // by the sequential model
};
enum { LEAF, OR, NOT, AND };
struct Node {
  int kind = LEAF;
  int leaf = -1;            // LEAF; NOT: the scan below it (-1: the NOT is over an OR of the leaves in `kids`)
  std::vector<Node> kids;   // OR: leaves, NOTs and ANDs; AND: leaves, NOTs, ORs, ANDs
};

static int64_t next_set(const Bits& m, int64_t from, int64_t n) {
  for (int64_t w = from >> 6; w * 64 < n; w++) {
    const uint64_t cur = m[(size_t)w] & (w == (from >> 6) ? ~0ULL << (from & 63) : ~0ULL);
    if (cur) { const int64_t p = w * 64 + __builtin_ctzll(cur); return p < n ? p : -1; }
  }
  return -1;
}

// ---- the sequential model ---------------------------------------------------------------------------------------------------------------------
struct SeqIt {
  virtual ~SeqIt() {}
  virtual int64_t advance(int64_t target) = 0;   // -1: EOF
};
struct SeqLeaf : SeqIt {
  Leaf& l;
  int64_t n;
  int64_t pos = 0;
  SeqLeaf(Leaf& leaf, int64_t docs) : l(leaf), n(docs) {}
  int64_t advance(int64_t target) override {
    if (l.scan) {   // SVScanDocIdIterator#advance: doc by doc from the target
      if (target >= n) return -1;
      const int64_t p = next_set(l.m, target, n);
      l.counted += (p < 0 ? n : p + 1) - target;
      return p;
    }
    if (target > pos) pos = target;   // BitmapDocIdIterator: never backwards
    const int64_t p = next_set(l.m, pos, n);
    pos = p < 0 ? n : p + 1;
    return p;
  }
};
struct SeqOr : SeqIt {   // OrDocIdIterator#advance
  std::vector<std::unique_ptr<SeqIt>> its;
  std::vector<int64_t> next_ids;
  int64_t advance(int64_t target) override {
    int64_t best = INT64_MAX;
    for (size_t i = 0; i < its.size(); i++) {
      if (next_ids[i] == -2) continue;   // exhausted
      int64_t d = next_ids[i];
      if (d < target) {
        d = its[i]->advance(target);
        next_ids[i] = d < 0 ? -2 : d;
        if (d < 0) continue;
      }
      if (d < best) best = d;
    }
    return best == INT64_MAX ? -1 : best;
  }
};
// SVScanDocIdIterator with both of its entry points, under NotDocIdIterator
struct SeqScan {
  Leaf& l;
  int64_t n, next_doc = 0;
  std::vector<int64_t> batch;
  size_t cursor = 0;
  SeqScan(Leaf& leaf, int64_t docs) : l(leaf), n(docs) {}
  int64_t next() {   // SVScanDocIdIterator.java:76-98: whole batches of 256 docs until one holds a match
    while (cursor >= batch.size()) {
      batch.clear();
      cursor = 0;
      if (next_doc >= n) return -1;
      const int64_t limit = std::min<int64_t>(n - next_doc, 256);
      for (int64_t d = next_doc; d < next_doc + limit; d++)
        if ((l.m[(size_t)(d >> 6)] >> (d & 63)) & 1) batch.push_back(d);
      next_doc += limit;
      l.counted += limit;
    }
    return batch[cursor++];
  }
  int64_t advance(int64_t target) {   // :101-112
    batch.clear();
    cursor = 0;
    next_doc = target;
    while (next_doc < n) {
      const int64_t d = next_doc++;
      l.counted++;
      if ((l.m[(size_t)(d >> 6)] >> (d & 63)) & 1) return d;
    }
    return -1;
  }
};
struct SeqNot : SeqIt {   // NotDocIdIterator.java:28-70
  SeqScan child;
  int64_t n, next_doc = 0, next_non_matching;
  SeqNot(Leaf& leaf, int64_t docs) : child(leaf, docs), n(docs) {
    const int64_t cur = child.next();
    next_non_matching = cur < 0 ? n : cur;
  }
  int64_t next() {
    if (next_doc >= n) return -1;
    while (next_doc == next_non_matching) {
      next_doc++;
      const int64_t d = child.next();
      next_non_matching = d < 0 ? n : d;
    }
    if (next_doc >= n) return -1;
    return next_doc++;
  }
  int64_t advance(int64_t target) override {
    next_doc = target;
    if (target > next_non_matching) {
      const int64_t d = child.advance(target);
      next_non_matching = d < 0 ? n : d;
    }
    return next();
  }
};
// NotDocIdIterator over an OrDocIdIterator of leaves: both entry points all the way down (OrDocIdIterator.java:50-119)
struct SeqFullLeaf {
  Leaf& l;
  int64_t n, pos = 0;
  SeqScan scan;
  SeqFullLeaf(Leaf& leaf, int64_t docs) : l(leaf), n(docs), scan(leaf, docs) {}
  int64_t next() {
    if (l.scan) return scan.next();
    const int64_t p = next_set(l.m, pos, n);   // BitmapDocIdIterator
    pos = p < 0 ? n : p + 1;
    return p;
  }
  int64_t advance(int64_t target) {
    if (l.scan) return scan.advance(target);
    if (target > pos) pos = target;
    return next();
  }
};
struct SeqNotOr : SeqIt {
  std::vector<SeqFullLeaf> its;
  std::vector<int64_t> next_ids;   // -1 before the first call, -2 exhausted
  int64_t previous = -1;
  int64_t n, next_doc = 0, next_non_matching;
  int64_t or_next() {
    int64_t best = INT64_MAX;
    for (size_t i = 0; i < its.size(); i++) {
      if (next_ids[i] == -2) continue;
      int64_t d = next_ids[i];
      if (d == previous) {
        d = its[i].next();
        next_ids[i] = d < 0 ? -2 : d;
        if (d < 0) continue;
      }
      best = std::min(best, d);
    }
    if (best == INT64_MAX) return -1;
    previous = best;
    return best;
  }
  int64_t or_advance(int64_t target) {
    int64_t best = INT64_MAX;
    for (size_t i = 0; i < its.size(); i++) {
      if (next_ids[i] == -2) continue;
      int64_t d = next_ids[i];
      if (d < target) {
        d = its[i].advance(target);
        next_ids[i] = d < 0 ? -2 : d;
        if (d < 0) continue;
      }
      best = std::min(best, d);
    }
    if (best == INT64_MAX) return -1;
    previous = best;
    return best;
  }
  SeqNotOr(const Node& nd, std::vector<Leaf>& leaves, int64_t docs) : n(docs) {
    for (auto& c : nd.kids) { its.emplace_back(leaves[(size_t)c.leaf], docs); next_ids.push_back(-1); }
    const int64_t cur = or_next();
    next_non_matching = cur < 0 ? n : cur;
  }
  int64_t next() {
    if (next_doc >= n) return -1;
    while (next_doc == next_non_matching) {
      next_doc++;
      const int64_t d = or_next();
      next_non_matching = d < 0 ? n : d;
    }
    if (next_doc >= n) return -1;
    return next_doc++;
  }
  int64_t advance(int64_t target) override {
    next_doc = target;
    if (target > next_non_matching) {
      const int64_t d = or_advance(target);
      next_non_matching = d < 0 ? n : d;
    }
    return next();
  }
};
struct SeqAnd : SeqIt {   // AndDocIdIterator.java:37-66
  std::vector<std::unique_ptr<SeqIt>> its;
  int64_t next_doc = 0;
  int64_t next() {
    int64_t max_doc = next_doc;
    int max_idx = -1, index = 0;
    const int k = (int)its.size();
    while (index < k) {
      if (index == max_idx) { index++; continue; }
      const int64_t d = its[(size_t)index]->advance(max_doc);
      if (d < 0) return -1;
      if (d == max_doc) index++;
      else { max_doc = d; max_idx = index; index = 0; }
    }
    next_doc = max_doc;
    return next_doc++;
  }
  int64_t advance(int64_t target) override { next_doc = target; return next(); }
};
static std::unique_ptr<SeqIt> seq_of(const Node& nd, std::vector<Leaf>& leaves, int64_t n) {
  switch (nd.kind) {
    case LEAF: return std::make_unique<SeqLeaf>(leaves[(size_t)nd.leaf], n);
    case NOT:
      if (nd.leaf < 0) return std::make_unique<SeqNotOr>(nd, leaves, n);
      return std::make_unique<SeqNot>(leaves[(size_t)nd.leaf], n);
    case OR: {
      auto o = std::make_unique<SeqOr>();
      for (auto& c : nd.kids) { o->its.push_back(seq_of(c, leaves, n)); o->next_ids.push_back(-1); }
      return o;
    }
    default: {
      auto a = std::make_unique<SeqAnd>();
      for (auto& c : nd.kids) a->its.push_back(seq_of(c, leaves, n));
      return a;
    }
  }
}

// ---- the tile model: what the device runs, with loops where it has kernels and scans ----------------------------------------------------------
struct Tiles {
  const std::vector<Leaf>& leaves;
  int64_t n, n_words;
  std::vector<int64_t> counts;

  Bits match_of(const Node& nd) const {   // the docs the node's iterator returns
    Bits out((size_t)n_words + 1, 0);
    const uint64_t last = n & 63 ? ~0ULL >> (64 - (n & 63)) : ~0ULL;
    if (nd.kind == NOT && nd.leaf < 0) {   // the docs outside the union
      for (auto& c : nd.kids) {
        const Bits m = match_of(c);
        for (int64_t w = 0; w < n_words; w++) out[(size_t)w] |= m[(size_t)w];
      }
      for (int64_t w = 0; w < n_words; w++) out[(size_t)w] = ~out[(size_t)w] & (w == n_words - 1 ? last : ~0ULL);
      return out;
    }
    if (nd.kind == LEAF || nd.kind == NOT) {
      out = leaves[(size_t)nd.leaf].m;
      if (nd.kind == NOT)
        for (int64_t w = 0; w < n_words; w++) out[(size_t)w] = ~out[(size_t)w] & (w == n_words - 1 ? last : ~0ULL);
      return out;
    }
    if (nd.kind == AND) std::fill(out.begin(), out.begin() + n_words, ~0ULL);
    for (auto& c : nd.kids) {
      const Bits m = match_of(c);
      for (int64_t w = 0; w < n_words; w++) out[(size_t)w] = nd.kind == AND ? out[(size_t)w] & m[(size_t)w] : out[(size_t)w] | m[(size_t)w];
    }
    if (nd.kind == AND && n_words) out[(size_t)n_words - 1] &= last;
    return out;
  }
  void latch(const Bits& targets, int leaf) {
    const uint64_t* tw = targets.data();
    const uint64_t* mw = leaves[(size_t)leaf].m.data();
    uint32_t state = 2;
    for (int64_t t = 0; t * FS_LATCH_WORDS < n_words; t++) {
      const int64_t w_lo = t * FS_LATCH_WORDS, w_hi = std::min<int64_t>(n_words, w_lo + FS_LATCH_WORDS);
      counts[(size_t)leaf] += fs_latch_count(tw, mw, w_lo, w_hi, state == 1, n);
      state = fs_latch_then(state, fs_latch_summary(tw, mw, w_lo, w_hi));
    }
  }
  // a NOT inside an OR: the OR's targets that reach it
  void not_scan_in_or(const Bits& targets, const Bits& others, int leaf) {
    struct Look {
      const Bits &nm, &t;
      static int64_t prev(const Bits& b, int64_t x) { for (; x >= 0; x--) if ((b[(size_t)(x >> 6)] >> (x & 63)) & 1) return x; return -1; }
      int64_t prev_target(int64_t x) const { return prev(t, x); }
      int64_t prev_non_match(int64_t x) const { return prev(nm, x); }
    } look{others, targets};
    Bits received((size_t)n_words + 1, 0);
    for (int64_t t = next_set(targets, 0, n); t >= 0; t = next_set(targets, t + 1, n))
      if (fs_not_in_or_receives(look, t)) received[(size_t)(t >> 6)] |= 1ULL << (t & 63);
    not_scan(received, others, leaf);
  }
  // a NOT over an OR of leaves: `others` = the docs the NOT returns
  void not_or(const Bits& targets, const Bits& others, const Node& nd) {
    Bits uni((size_t)n_words + 1, 0);
    const uint64_t last = n & 63 ? ~0ULL >> (64 - (n & 63)) : ~0ULL;
    for (int64_t w = 0; w < n_words; w++) uni[(size_t)w] = ~others[(size_t)w] & (w == n_words - 1 ? last : ~0ULL);
    struct LookU {
      const Bits &m, &nm, &t;
      int64_t n;
      static int64_t prev(const Bits& b, int64_t x) { for (; x >= 0; x--) if ((b[(size_t)(x >> 6)] >> (x & 63)) & 1) return x; return -1; }
      int64_t next_match(int64_t x) const { return x >= n ? -1 : next_set(m, x, n); }
      int64_t next_non_match(int64_t x) const { return next_set(nm, x, n); }
      int64_t prev_target(int64_t x) const { return prev(t, std::min(x, n - 1)); }
    } u{uni, others, targets, n};
    Bits resets((size_t)n_words + 1, 0);
    for (int64_t t = next_set(targets, 0, n); t >= 0; t = next_set(targets, t + 1, n))
      if (fs_not_is_reset(u, t)) resets[(size_t)(t >> 6)] |= 1ULL << (t & 63);
    for (auto& kid : nd.kids) {
      if (!leaves[(size_t)kid.leaf].scan) continue;
      struct LookC {
        const Bits& m;
        Bits a;
        int64_t n;
        int64_t next_match(int64_t x) const { return x < 0 || x >= n ? -1 : next_set(m, x, n); }
        int64_t prev_advance(int64_t x) const { return LookU::prev(a, std::min(x, n - 1)); }
        int64_t span(int64_t x, int64_t y) const { return y - x; }
        bool batched() const { return true; }
      } c{leaves[(size_t)kid.leaf].m, Bits((size_t)n_words + 1, 0), n};
      for (int64_t r = next_set(resets, 0, n); r >= 0; r = next_set(resets, r + 1, n))
        if (fs_notor_child_advanced(u, c, r)) c.a[(size_t)(r >> 6)] |= 1ULL << (r & 63);
      int64_t total = fs_notor_episode(u, c, n, n);
      for (int64_t a = next_set(c.a, 0, n); a >= 0; a = next_set(c.a, a + 1, n)) total += fs_notor_advance_cost(u, c, a, n);
      counts[(size_t)kid.leaf] += total; if (getenv("FS_DEBUG")) printf("not_or leaf %d total %lld seq %lld\n", kid.leaf, (long long)total, (long long)leaves[(size_t)kid.leaf].counted);
    }
  }
  void not_scan(const Bits& targets, const Bits& others, int leaf) {
    struct Look {   // the bitmaps scanned directly (the device: two-level indexes)
      const Bits &m, &nm, &t;
      Bits r, c;
      int64_t n;
      int64_t next_match(int64_t x) const { return next_set(m, x, n); }
      int64_t next_non_match(int64_t x) const { return next_set(nm, x, n); }
      int64_t next_reset(int64_t x) const { return next_set(r, x, n); }
      int64_t next_consume(int64_t x) const { return next_set(c, x, n); }
      static int64_t prev(const Bits& b, int64_t x) { for (; x >= 0; x--) if ((b[(size_t)(x >> 6)] >> (x & 63)) & 1) return x; return -1; }
      int64_t prev_target(int64_t x) const { return prev(t, x); }
      int64_t prev_reset(int64_t x) const { return prev(r, x); }
      int64_t prev_non_match(int64_t x) const { return prev(nm, x); }
      int64_t span(int64_t a, int64_t b) const { return b - a; }
      bool batched() const { return true; }
    } look{leaves[(size_t)leaf].m, others, targets, Bits((size_t)n_words + 1, 0), Bits((size_t)n_words + 1, 0), n};
    int64_t total = 0;
    for (int64_t t = next_set(look.t, 0, n); t >= 0; t = next_set(look.t, t + 1, n)) {
      if (fs_not_is_reset(look, t)) { look.r[(size_t)(t >> 6)] |= 1ULL << (t & 63); total += fs_not_advance_cost(look, t, n); }
      if ((look.m[(size_t)(t >> 6)] >> (t & 63)) & 1) look.c[(size_t)(t >> 6)] |= 1ULL << (t & 63);
    }
    for (int64_t t = next_set(look.c, 0, n); t >= 0; t = next_set(look.c, t + 1, n)) total += fs_not_episode_cost(look, t, n);
    total += fs_not_ctor_cost(look, n);
    counts[(size_t)leaf] += total;
  }
  // an AND started at the docs of `active` (nullptr: drained — every doc)
  void run_and(const Node& nd, const Bits* active) {
    const int64_t n_tiles = (n_words + FS_TILE_WORDS - 1) / FS_TILE_WORDS;
    const int k = (int)nd.kids.size();
    std::vector<Bits> match, targets((size_t)k, Bits((size_t)n_words + 1, 0));
    for (auto& c : nd.kids) match.push_back(match_of(c));
    struct Io {   // one tile: positions relative to its first word
      std::vector<Bits>&m, &t;
      const Bits* act;
      int64_t w_lo;
      uint64_t match(int c, int32_t w) const { return m[(size_t)c][(size_t)(w_lo + w)]; }
      uint64_t active(int32_t w) const { return act ? (*act)[(size_t)(w_lo + w)] : ~0ULL; }
      void target(int c, int32_t doc) { t[(size_t)c][(size_t)(w_lo + (doc >> 6))] |= 1ULL << (doc & 63); }
    };
    std::vector<uint32_t> maps((size_t)n_tiles), entry((size_t)n_tiles);
    for (int64_t t = 0; t < n_tiles; t++) {
      const int64_t lo = t * FS_TILE_WORDS * 64, hi = std::min<int64_t>(n, lo + FS_TILE_WORDS * 64);
      Io io{match, targets, active, t * FS_TILE_WORDS};
      maps[(size_t)t] = fs_and_tile_exits(k, io, (int32_t)(hi - lo));
      uint32_t map = 0;   // ... and without the shortcut behind the first common doc
      for (int s = 0; s <= k; s++) map |= fs_and_tile(k, io, 0, (int32_t)(hi - lo), (uint32_t)s, false) << (4 * s);
      if (map != maps[(size_t)t]) { printf("exit maps differ: tile %lld %08x %08x\n", (long long)t, map, maps[(size_t)t]); exit(1); }
    }
    uint32_t run = FS_MAP_IDENTITY;
    for (int64_t t = 0; t < n_tiles; t++) {
      entry[(size_t)t] = run & 15u;   // the state an idle start of the segment has become
      run = fs_map_then(run, maps[(size_t)t]);
    }
    for (int64_t t = 0; t < n_tiles; t++) {
      const int64_t lo = t * FS_TILE_WORDS * 64, hi = std::min<int64_t>(n, lo + FS_TILE_WORDS * 64);
      Io io{match, targets, active, t * FS_TILE_WORDS};
      fs_and_tile(k, io, 0, (int32_t)(hi - lo), entry[(size_t)t], true);
    }
    for (int j = 0; j < k; j++) {
      const Node& c = nd.kids[(size_t)j];
      const Bits& tj = targets[(size_t)j];
      if (c.kind == LEAF) { if (leaves[(size_t)c.leaf].scan) latch(tj, c.leaf); }
      else if (c.kind == NOT && c.leaf < 0) not_or(tj, match[(size_t)j], c);
      else if (c.kind == NOT) not_scan(tj, match[(size_t)j], c.leaf);
      else if (c.kind == AND) run_and(c, &tj);
      else
        for (auto& g : c.kids) {   // an OR hands its targets on to every child
          if (g.kind == LEAF) { if (leaves[(size_t)g.leaf].scan) latch(tj, g.leaf); }
          else if (g.kind == NOT) not_scan_in_or(tj, match_of(g), g.leaf);
          else run_and(g, &tj);
        }
    }
  }
};

static Node random_leaf(std::vector<Leaf>& leaves, std::mt19937_64& rng, int64_t n, bool must_scan) {
  const int64_t n_words = (n + 63) / 64;
  Leaf l;
  l.scan = must_scan || rng() % 4 != 0;
  l.m.assign((size_t)n_words + 1, 0);
  static const double dens[] = {0.0, 0.0005, 0.01, 0.1, 0.5, 0.9, 0.999, 1.0};
  const double p = dens[rng() % 8];
  const bool runs = rng() % 4 == 0;   // long runs of equal bits (sorted-ish columns)
  bool cur = false;
  for (int64_t d = 0; d < n; d++) {
    if (!runs || rng() % 257 == 0 || d == 0) cur = std::uniform_real_distribution<double>(0, 1)(rng) < p;
    if (cur) l.m[(size_t)(d >> 6)] |= 1ULL << (d & 63);
  }
  Node nd;
  nd.leaf = (int)leaves.size();
  leaves.push_back(std::move(l));
  return nd;
}
static Node random_and(std::vector<Leaf>& leaves, std::mt19937_64& rng, int64_t n, int depth) {
  Node a;
  a.kind = AND;
  const int k = 1 + (int)(rng() % 4);
  for (int j = 0; j < k; j++) {
    const int what = (int)(rng() % 9);
    if (what < 4) a.kids.push_back(random_leaf(leaves, rng, n, false));
    else if (what == 4) { Node c = random_leaf(leaves, rng, n, true); c.kind = NOT; a.kids.push_back(c); }
    else if (what == 5) {   // a NOT over an OR of leaves
      Node c;
      c.kind = NOT;
      const int nl = 1 + (int)(rng() % 3);
      for (int i = 0; i < nl; i++) c.kids.push_back(random_leaf(leaves, rng, n, false));
      a.kids.push_back(c);
    }
    else if (what < 8 || depth == 0) {
      Node o;
      o.kind = OR;
      const int nl = 1 + (int)(rng() % 3);
      for (int i = 0; i < nl; i++) {
        const int pick = (int)(rng() % 6);
        if (pick < 2 && depth > 0) o.kids.push_back(random_and(leaves, rng, n, depth - 1));
        else if (pick == 2) { Node c = random_leaf(leaves, rng, n, true); c.kind = NOT; o.kids.push_back(c); }
        else o.kids.push_back(random_leaf(leaves, rng, n, false));
      }
      a.kids.push_back(o);
    } else {
      a.kids.push_back(random_and(leaves, rng, n, depth - 1));   // an AND directly under an AND
    }
  }
  return a;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 400;
  std::mt19937_64 rng(20260930);
  int64_t checked = 0;
  for (int round = 0; round < rounds; round++) {
    static const int64_t sizes[] = {1, 63, 64, 65, 255, 256, 257, 511, 513, 2047, 2048, 2049, 4096, 10000, 70001, 200003};
    const int64_t n = sizes[rng() % (sizeof(sizes) / sizeof(sizes[0]))];
    std::vector<Leaf> leaves;
    const Node root = random_and(leaves, rng, n, 2);
    {
      std::unique_ptr<SeqIt> it = seq_of(root, leaves, n);
      SeqAnd* top = static_cast<SeqAnd*>(it.get());
      while (top->next() >= 0) {}   // DocIdSetOperator drains the iterator
    }
    Tiles tiles{leaves, n, (n + 63) / 64, std::vector<int64_t>(leaves.size(), 0)};
    tiles.run_and(root, nullptr);
    for (size_t i = 0; i < leaves.size(); i++) {
      if (!leaves[i].scan) continue;
      checked++;
      if (tiles.counts[i] != leaves[i].counted) {
        printf("MISMATCH round %d n %lld leaf %zu: tiles %lld sequential %lld\n", round, (long long)n, i, (long long)tiles.counts[i], (long long)leaves[i].counted);
        return 1;
      }
    }
  }
  printf("OK %d rounds, %lld scan leaves\n", rounds, (long long)checked);
  return 0;
}
