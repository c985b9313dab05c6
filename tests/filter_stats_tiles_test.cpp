// Host model of the tile-parallel numEntriesScannedInFilter (pinot_amd/csrc/pg_filter_stats_tiles.h) against a direct, doc-by-doc restatement of
// the reference's iterators (AndDocIdIterator.java:37-66, OrDocIdIterator.java:91-119, SVScanDocIdIterator.java:101-112 — advance() only, which
// is all an AND ever calls on its children).  Random match bitmaps of every density, AND children that are scans, bitmaps or ORs of both.
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
struct Child {          // a child of the AND: one leaf, an OR over several, or a NOT over one scan
  std::vector<int> leaves;
  bool is_or;
  bool is_not = false;
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

static void run_sequential(std::vector<Leaf>& leaves, const std::vector<Child>& children, int64_t n) {
  std::vector<std::unique_ptr<SeqIt>> its;
  for (auto& c : children) {
    if (c.is_not) { its.push_back(std::make_unique<SeqNot>(leaves[(size_t)c.leaves[0]], n)); continue; }
    if (!c.is_or) { its.push_back(std::make_unique<SeqLeaf>(leaves[(size_t)c.leaves[0]], n)); continue; }
    auto o = std::make_unique<SeqOr>();
    for (int li : c.leaves) { o->its.push_back(std::make_unique<SeqLeaf>(leaves[(size_t)li], n)); o->next_ids.push_back(-1); }
    its.push_back(std::move(o));
  }
  int64_t next_doc = 0;
  const int k = (int)its.size();
  for (;;) {   // AndDocIdIterator#next, drained
    int64_t max_doc = next_doc;
    int max_idx = -1, index = 0;
    bool eof = false;
    while (index < k) {
      if (index == max_idx) { index++; continue; }
      const int64_t d = its[(size_t)index]->advance(max_doc);
      if (d < 0) { eof = true; break; }
      if (d == max_doc) index++;
      else { max_doc = d; max_idx = index; index = 0; }
    }
    if (eof) break;
    next_doc = max_doc + 1;
  }
}

// ---- the tile model: what the device runs, with loops where it has kernels and scans ----------------------------------------------------------
static std::vector<int64_t> run_tiles(const std::vector<Leaf>& leaves, const std::vector<Child>& children, int64_t n) {
  const int64_t n_words = (n + 63) / 64, n_tiles = (n_words + FS_TILE_WORDS - 1) / FS_TILE_WORDS;
  const int k = (int)children.size();
  std::vector<Bits> match((size_t)k, Bits((size_t)n_words + 1, 0)), targets((size_t)k, Bits((size_t)n_words + 1, 0));
  for (int j = 0; j < k; j++) {
    for (int li : children[(size_t)j].leaves)
      for (int64_t w = 0; w < n_words; w++) match[(size_t)j][(size_t)w] |= leaves[(size_t)li].m[(size_t)w];
    if (children[(size_t)j].is_not)   // the docs a NOT returns: the others, of those that exist
      for (int64_t w = 0; w < n_words; w++)
        match[(size_t)j][(size_t)w] = ~match[(size_t)j][(size_t)w] & (w * 64 + 64 > n ? ~0ULL >> (64 - (n - w * 64)) : ~0ULL);
  }
  struct Io {   // one tile: positions relative to its first word
    std::vector<Bits>&m, &t;
    int64_t w_lo;
    uint64_t match(int c, int32_t w) const { return m[(size_t)c][(size_t)(w_lo + w)]; }
    void target(int c, int32_t doc) { t[(size_t)c][(size_t)(w_lo + (doc >> 6))] |= 1ULL << (doc & 63); }
  };
  std::vector<uint32_t> maps((size_t)n_tiles), entry((size_t)n_tiles);
  for (int64_t t = 0; t < n_tiles; t++) {
    const int64_t lo = t * FS_TILE_WORDS * 64, hi = std::min<int64_t>(n, lo + FS_TILE_WORDS * 64);
    Io io{match, targets, t * FS_TILE_WORDS};
    maps[(size_t)t] = fs_and_tile_exits(k, io, (int32_t)(hi - lo));
    uint32_t map = 0;   // ... and without the shortcut behind the first common doc
    for (int s = 0; s <= k; s++) map |= fs_and_tile(k, io, 0, (int32_t)(hi - lo), (uint32_t)s, false) << (4 * s);
    if (map != maps[(size_t)t]) { printf("exit maps differ: tile %lld %08x %08x\n", (long long)t, map, maps[(size_t)t]); exit(1); }
  }
  uint32_t run = FS_MAP_IDENTITY;
  for (int64_t t = 0; t < n_tiles; t++) {
    entry[(size_t)t] = run & 15u;   // the state a clean start of the segment has become
    run = fs_map_then(run, maps[(size_t)t]);
  }
  for (int64_t t = 0; t < n_tiles; t++) {
    const int64_t lo = t * FS_TILE_WORDS * 64, hi = std::min<int64_t>(n, lo + FS_TILE_WORDS * 64);
    Io io{match, targets, t * FS_TILE_WORDS};
    fs_and_tile(k, io, 0, (int32_t)(hi - lo), entry[(size_t)t], true);
  }
  std::vector<int64_t> counts(leaves.size(), 0);
  for (int j = 0; j < k; j++) {
    if (!children[(size_t)j].is_not) continue;
    const int li = children[(size_t)j].leaves[0];
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
    } look{leaves[(size_t)li].m, match[(size_t)j], targets[(size_t)j], Bits((size_t)n_words + 1, 0), Bits((size_t)n_words + 1, 0), n};
    int64_t total = 0;
    for (int64_t t = next_set(look.t, 0, n); t >= 0; t = next_set(look.t, t + 1, n)) {
      if (fs_not_is_reset(look, t)) { look.r[(size_t)(t >> 6)] |= 1ULL << (t & 63); total += fs_not_advance_cost(look, t, n); }
      if ((look.m[(size_t)(t >> 6)] >> (t & 63)) & 1) look.c[(size_t)(t >> 6)] |= 1ULL << (t & 63);
    }
    for (int64_t t = next_set(look.c, 0, n); t >= 0; t = next_set(look.c, t + 1, n)) total += fs_not_episode_cost(look, t, n);
    total += fs_not_ctor_cost(look, n);
    counts[(size_t)li] = total;
  }
  for (int j = 0; j < k; j++)
    for (int li : children[(size_t)j].leaves) {
      if (!leaves[(size_t)li].scan || children[(size_t)j].is_not) continue;
      const uint64_t* tw = targets[(size_t)j].data();
      const uint64_t* mw = leaves[(size_t)li].m.data();
      uint32_t state = 2;
      for (int64_t t = 0; t * FS_LATCH_WORDS < n_words; t++) {
        const int64_t w_lo = t * FS_LATCH_WORDS, w_hi = std::min<int64_t>(n_words, w_lo + FS_LATCH_WORDS);
        counts[(size_t)li] += fs_latch_count(tw, mw, w_lo, w_hi, state == 1, n);
        state = fs_latch_then(state, fs_latch_summary(tw, mw, w_lo, w_hi));
      }
    }
  return counts;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 400;
  std::mt19937_64 rng(20260930);
  int64_t checked = 0;
  for (int round = 0; round < rounds; round++) {
    static const int64_t sizes[] = {1, 63, 64, 65, 255, 256, 257, 511, 513, 2047, 2048, 2049, 4096, 10000, 70001, 200003};
    const int64_t n = sizes[rng() % (sizeof(sizes) / sizeof(sizes[0]))];
    const int64_t n_words = (n + 63) / 64;
    const int k = 1 + (int)(rng() % 4);
    std::vector<Leaf> leaves;
    std::vector<Child> children;
    for (int j = 0; j < k; j++) {
      Child c;
      c.is_or = rng() % 3 == 0;
      c.is_not = !c.is_or && rng() % 3 == 0;
      const int nl = c.is_or ? 1 + (int)(rng() % 3) : 1;
      for (int i = 0; i < nl; i++) {
        Leaf l;
        l.scan = c.is_not || rng() % 4 != 0;
        l.m.assign((size_t)n_words + 1, 0);
        static const double dens[] = {0.0, 0.0005, 0.01, 0.1, 0.5, 0.9, 0.999, 1.0};
        const double p = dens[rng() % 8];
        const bool runs = rng() % 4 == 0;   // long runs of equal bits (sorted-ish columns)
        bool cur = false;
        for (int64_t d = 0; d < n; d++) {
          if (!runs || rng() % 257 == 0 || d == 0) cur = std::uniform_real_distribution<double>(0, 1)(rng) < p;
          if (cur) l.m[(size_t)(d >> 6)] |= 1ULL << (d & 63);
        }
        c.leaves.push_back((int)leaves.size());
        leaves.push_back(std::move(l));
      }
      children.push_back(c);
    }
    run_sequential(leaves, children, n);
    const std::vector<int64_t> got = run_tiles(leaves, children, n);
    for (size_t i = 0; i < leaves.size(); i++) {
      if (!leaves[i].scan) continue;
      checked++;
      if (got[i] != leaves[i].counted) {
        printf("MISMATCH round %d n %lld k %d leaf %zu: tiles %lld sequential %lld\n", round, (long long)n, k, i, (long long)got[i], (long long)leaves[i].counted);
        return 1;
      }
    }
  }
  printf("OK %d rounds, %lld scan leaves\n", rounds, (long long)checked);
  return 0;
}
