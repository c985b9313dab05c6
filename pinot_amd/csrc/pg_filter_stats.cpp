// numEntriesScannedInFilter for filter shapes whose count depends on how the reference's docId iterators drive each other.
//
// For a flat AND the count has a closed form the kernels produce on the fly (every restricted scan evaluates exactly the
// surviving candidates: ScanBasedDocIdIterator#applyAnd, SVScanDocIdIterator.java:115-142).  Under OR / NOT, or an AND without an
// index-based child, the reference leapfrogs iterators (AndDocIdIterator.java:37-66, OrDocIdIterator.java:57-119,
// NotDocIdIterator.java:45-70) and a scan iterator counts every doc it steps over: _numEntriesScanned++ per doc in advance(),
// whole 256-doc batches in next() (SVScanDocIdIterator.java:76-112).  That count is a property of the iterator automaton, not of
// the doc sets, so it is reproduced by running the same automaton — but over the leaves' match BITMAPS, which the GPU produces
// (one filter launch per leaf), instead of over column values: next() / advance() become find-next-set-bit on 64-bit words and a
// scan iterator counts distances instead of visiting docs, so the cost is proportional to the number of iterator calls, not to
// the docs scanned.  The doc set itself (and every aggregate) still comes from the kernels; only this statistic is derived here.
//
// Mirrors, node for node: FilterOperator#getTrues / getFalses (AndFilterOperator.java:52-90, OrFilterOperator.java:49-87,
// NotFilterOperator.java:52-58, BaseFilterOperator.java:96-112), AndDocIdSet#iterator (AndDocIdSet.java:72-186),
// OrDocIdSet#iterator (OrDocIdSet.java:62-125, with the deviation DESIGN.md §2 documents: bitmap children are OR-ed into the merged
// bitmap), NotDocIdSet, and DocIdSetOperator's drain by next() (DocIdSetOperator.java:59-86).
#include <algorithm>
#include <deque>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include "pg_filter_stats_tiles.h"
#include "pg_internal.hpp"

namespace pg {

static const int32_t kEof = -1;   // Constants.EOF
static const int kScanBatch = 256;   // BlockDocIdIterator.OPTIMAL_ITERATOR_BATCH_SIZE

int64_t HostBits::next_set(int64_t from) const {
  if (from < 0) from = 0;
  if (from >= n_docs) return -1;
  size_t wi = (size_t)(from >> 6);
  uint64_t cur = w[wi] & (~0ULL << (from & 63));
  const size_t nw = w.size();
  while (!cur) {
    if (++wi >= nw) return -1;
    cur = w[wi];
  }
  const int64_t p = (int64_t)wi * 64 + __builtin_ctzll(cur);
  return p < n_docs ? p : -1;
}
int64_t HostBits::cardinality() const {
  int64_t c = 0;
  for (uint64_t x : w) c += __builtin_popcountll(x);
  return c;
}
void HostBits::resize_for(int64_t docs) {
  n_docs = docs;
  w.assign((size_t)((docs + 63) / 64) + 1, 0);
}
void HostBits::add_range(int64_t lo, int64_t hi_inclusive) {
  for (int64_t d = std::max<int64_t>(lo, 0); d <= hi_inclusive && d < n_docs;) {
    const int64_t wi = d >> 6, last = std::min<int64_t>(std::min(hi_inclusive, n_docs - 1), wi * 64 + 63);
    w[(size_t)wi] |= (~0ULL << (d & 63)) & (~0ULL >> (63 - (last & 63)));
    d = last + 1;
  }
}

namespace {

enum class ItKind { Sorted, Bitmap, Scan, Other };

struct It {
  ItKind kind = ItKind::Other;
  virtual ~It() = default;
  virtual int32_t next() = 0;
  virtual int32_t advance(int32_t target) = 0;
  virtual const HostBits* doc_ids() const { return nullptr; }                 // BitmapBasedDocIdIterator#getDocIds
  virtual const std::vector<std::pair<int32_t, int32_t>>* ranges() const { return nullptr; }   // SortedDocIdIterator#getDocIdRanges
};
using ItPtr = std::unique_ptr<It>;

struct Counter { int64_t entries = 0; };

// SVScanDocIdIterator over the leaf's match bitmap
struct ScanIt : It {
  const HostBits& m;
  Counter& c;
  int64_t next_doc = 0;
  std::vector<int32_t> batch;
  size_t cursor = 0;
  // MVScanDocIdIterator (MVScanDocIdIterator.java:65-117) when `mv_off` is set — the docs' first entries: it steps doc by doc, no
  // batches, and counts every ENTRY of every doc it evaluates
  const int32_t* mv_off = nullptr;
  ScanIt(const HostBits& bits, Counter& counter, const int32_t* mv_offsets = nullptr) : m(bits), c(counter), mv_off(mv_offsets) { kind = ItKind::Scan; }
  int32_t mv_next() {
    const int64_t n = m.n_docs;
    if (next_doc >= n) return kEof;
    const int64_t p = m.next_set(next_doc);
    const int64_t last = p < 0 ? n : p + 1;   // docs [next_doc, last) are evaluated
    c.entries += (int64_t)mv_off[last] - (int64_t)mv_off[next_doc];
    next_doc = last;
    return p < 0 ? kEof : (int32_t)p;
  }
  int32_t next() override {   // :76-98 — whole batches of up to 256 docs until one holds a match
    if (mv_off) return mv_next();
    if (cursor >= batch.size()) {
      batch.clear();
      cursor = 0;
      const int64_t n = m.n_docs;
      if (next_doc >= n) return kEof;
      const int64_t p = m.next_set(next_doc);
      if (p < 0) {   // every remaining batch is empty: they are all scanned
        c.entries += n - next_doc;
        next_doc = n;
        return kEof;
      }
      const int64_t skipped = (p - next_doc) / kScanBatch;   // empty batches before the one holding p
      c.entries += skipped * kScanBatch;
      next_doc += skipped * kScanBatch;
      const int64_t limit = std::min<int64_t>(n - next_doc, kScanBatch);
      for (int64_t q = p; q >= 0 && q < next_doc + limit; q = m.next_set(q + 1)) batch.push_back((int32_t)q);
      next_doc += limit;
      c.entries += limit;
    }
    return batch[cursor++];
  }
  int32_t advance(int32_t target) override {   // :101-112 — doc by doc from the target to the first match
    if (mv_off) { next_doc = target; return mv_next(); }
    batch.clear();
    cursor = 0;
    next_doc = target;
    const int64_t n = m.n_docs;
    if (next_doc >= n) return kEof;
    const int64_t p = m.next_set(next_doc);
    if (p < 0) {
      c.entries += n - next_doc;
      next_doc = n;
      return kEof;
    }
    c.entries += p - next_doc + 1;
    next_doc = p + 1;
    return (int32_t)p;
  }
  void apply_and(HostBits& doc_ids_io) {   // :115-142 — every candidate is evaluated once
    if (mv_off) {
      for (size_t i = 0; i < doc_ids_io.w.size(); i++)
        for (uint64_t x = doc_ids_io.w[i]; x; x &= x - 1) {
          const int64_t d = (int64_t)i * 64 + __builtin_ctzll(x);
          if (d < m.n_docs) c.entries += mv_off[d + 1] - mv_off[d];
        }
    } else {
      c.entries += doc_ids_io.cardinality();
    }
    for (size_t i = 0; i < doc_ids_io.w.size(); i++) doc_ids_io.w[i] &= m.w[i];
  }
};

// BitmapDocIdIterator / RangelessBitmapDocIdIterator (PeekableIntIterator#advanceIfNeeded never moves backwards)
struct BitmapIt : It {
  std::shared_ptr<HostBits> bits;
  int64_t pos = 0;
  explicit BitmapIt(std::shared_ptr<HostBits> b) : bits(std::move(b)) { kind = ItKind::Bitmap; }
  int32_t next() override {
    const int64_t p = bits->next_set(pos);
    if (p < 0) { pos = bits->n_docs; return kEof; }
    pos = p + 1;
    return (int32_t)p;
  }
  int32_t advance(int32_t target) override {
    if (target > pos) pos = target;
    return next();
  }
  const HostBits* doc_ids() const override { return bits.get(); }
};

struct SortedIt : It {   // SortedDocIdIterator.java
  std::vector<std::pair<int32_t, int32_t>> r;   // inclusive ranges, ascending
  size_t cur = 0;
  int64_t next_doc;
  explicit SortedIt(std::vector<std::pair<int32_t, int32_t>> rs) : r(std::move(rs)), next_doc(r[0].first) { kind = ItKind::Sorted; }
  int32_t next() override {
    if (next_doc <= r[cur].second) return (int32_t)next_doc++;
    if (cur + 1 < r.size()) {
      cur++;
      next_doc = r[cur].first;
      return (int32_t)next_doc++;
    }
    return kEof;
  }
  int32_t advance(int32_t target) override {
    if (target <= r[cur].second) {
      next_doc = std::max<int64_t>(target, r[cur].first);
      return (int32_t)next_doc++;
    }
    while (cur + 1 < r.size()) {
      cur++;
      if (target <= r[cur].second) {
        next_doc = std::max<int64_t>(target, r[cur].first);
        return (int32_t)next_doc++;
      }
    }
    return kEof;
  }
  const std::vector<std::pair<int32_t, int32_t>>* ranges() const override { return &r; }
};

struct MatchAllIt : It {
  int64_t n, next_doc = 0;
  explicit MatchAllIt(int64_t docs) : n(docs) {}
  int32_t next() override { return next_doc < n ? (int32_t)next_doc++ : kEof; }
  int32_t advance(int32_t t) override { next_doc = t; return next(); }
};
struct EmptyIt : It {
  int32_t next() override { return kEof; }
  int32_t advance(int32_t) override { return kEof; }
};

struct AndIt : It {   // AndDocIdIterator.java:37-66
  std::vector<ItPtr> its;
  int32_t next_doc = 0;
  int32_t next() override {
    int32_t max_doc = next_doc;
    int max_idx = -1;
    const int n = (int)its.size();
    int index = 0;
    while (index < n) {
      if (index == max_idx) { index++; continue; }
      const int32_t d = its[(size_t)index]->advance(max_doc);
      if (d == kEof) return kEof;
      if (d == max_doc) index++;
      else { max_doc = d; max_idx = index; index = 0; }
    }
    next_doc = max_doc;
    return next_doc++;
  }
  int32_t advance(int32_t t) override { next_doc = t; return next(); }
};

struct OrIt : It {   // OrDocIdIterator.java:33-139
  std::vector<ItPtr> its;
  std::vector<int32_t> next_ids;
  int n_live = 0;
  int32_t prev = -1;
  void init() { n_live = (int)its.size(); next_ids.assign(its.size(), -1); }
  void remove_exhausted() {
    int i = 0;
    while (i < n_live) {
      if (next_ids[(size_t)i] == kEof) {
        n_live--;
        std::swap(its[(size_t)i], its[(size_t)n_live]);      // the reference overwrites slot i with the last live one
        next_ids[(size_t)i] = next_ids[(size_t)n_live];
      } else {
        i++;
      }
    }
  }
  int32_t next() override {
    int32_t best = INT32_MAX;
    bool exhausted = false;
    for (int i = 0; i < n_live; i++) {
      int32_t d = next_ids[(size_t)i];
      if (d == prev) {
        d = its[(size_t)i]->next();
        next_ids[(size_t)i] = d;
        if (d == kEof) { exhausted = true; continue; }
      }
      best = std::min(best, d);
    }
    if (exhausted) remove_exhausted();
    if (best != INT32_MAX) { prev = best; return best; }
    return kEof;
  }
  int32_t advance(int32_t target) override {
    int32_t best = INT32_MAX;
    bool exhausted = false;
    for (int i = 0; i < n_live; i++) {
      int32_t d = next_ids[(size_t)i];
      if (d < target) {
        d = its[(size_t)i]->advance(target);
        next_ids[(size_t)i] = d;
        if (d == kEof) { exhausted = true; continue; }
      }
      best = std::min(best, d);
    }
    if (exhausted) remove_exhausted();
    if (best != INT32_MAX) { prev = best; return best; }
    return kEof;
  }
};

struct NotIt : It {   // NotDocIdIterator.java:28-70
  ItPtr child;
  int32_t n, next_doc = 0, next_non_matching;
  NotIt(ItPtr c, int32_t docs) : child(std::move(c)), n(docs) {
    const int32_t cur = child->next();
    next_non_matching = cur == kEof ? n : cur;
  }
  int32_t next() override {
    if (next_doc >= n) return kEof;
    while (next_doc == next_non_matching) {
      next_doc++;
      const int32_t d = child->next();
      next_non_matching = d == kEof ? n : d;
    }
    if (next_doc >= n) return kEof;
    return next_doc++;
  }
  int32_t advance(int32_t target) override {
    next_doc = target;
    if (target > next_non_matching) {
      const int32_t d = child->advance(target);
      next_non_matching = d == kEof ? n : d;
    }
    return next();
  }
};

// ---- doc-id SETS (BlockDocIdSet): what getTrues / getFalses return; iterator() builds the automaton ------------------------------
enum class SetKind { Empty, MatchAll, Scan, Bitmap, Sorted, And, Or, Not };
struct Set {
  SetKind kind = SetKind::Empty;
  const HostBits* leaf_bits = nullptr;                       // Scan: match bitmap; Bitmap: the doc set
  const uint64_t* dev_words = nullptr;                       // the same bitmap in HBM (device evaluation; a Bitmap without it is built from `ranges`)
  const int32_t* mv_off = nullptr;                           // Scan over a multi-value column: the docs' first entries
  const int32_t* mv_off_dev = nullptr;                       // ... in HBM (device evaluation)
  std::shared_ptr<HostBits> owned;                           // Bitmap built here (flips, range lists)
  std::vector<std::pair<int32_t, int32_t>> ranges;           // Sorted
  std::vector<std::unique_ptr<Set>> children;                // And / Or / Not
};
using SetPtr = std::unique_ptr<Set>;

struct Emu {
  const StatLeafBits& leaves;
  int32_t n_docs;
  std::vector<std::unique_ptr<Counter>> counters;
  const StatLeafWords* dev_leaves = nullptr;   // device evaluation: the leaves' bitmaps stay in HBM (entries may be missing while only the shape is judged)

  void bind_leaf(Set& s, const FilterOp& op) {
    if (!dev_leaves) { s.leaf_bits = &leaves.at(&op); return; }
    auto it = dev_leaves->find(&op);
    s.dev_words = it == dev_leaves->end() ? nullptr : it->second;
  }

  SetPtr mk(SetKind k) { auto s = std::make_unique<Set>(); s->kind = k; return s; }

  static std::vector<std::pair<int32_t, int32_t>> sorted_ranges(const FilterOp& op, int32_t n_docs) {   // SortedIndexBasedFilterOperator#getTrues
    const Column& c = *op.col;
    const PredEval& e = op.eval;
    std::vector<std::pair<int32_t, int32_t>> out;
    if (e.is_range) { out.push_back({c.sorted_start[(size_t)e.start_dict_id], c.sorted_end[(size_t)e.end_dict_id - 1]}); return out; }
    const std::vector<int32_t>& ids = e.exclusive ? e.non_matching : e.matching;
    std::vector<std::pair<int32_t, int32_t>> r;
    for (int32_t id : ids) {
      const int32_t s = c.sorted_start[(size_t)id], en = c.sorted_end[(size_t)id];
      if (!r.empty() && s == r.back().second + 1) r.back().second = en;
      else r.push_back({s, en});
    }
    if (!e.exclusive) return r;
    if (r[0].first > 0) out.push_back({0, r[0].first - 1});
    for (size_t i = 0; i + 1 < r.size(); i++) out.push_back({r[i].second + 1, r[i + 1].first - 1});
    if (r.back().second < n_docs - 1) out.push_back({r.back().second + 1, n_docs - 1});
    return out;
  }

  SetPtr trues(const FilterOp& op) {
    switch (op.kind) {
      case OpKind::Empty: return mk(SetKind::Empty);
      case OpKind::MatchAll: return mk(SetKind::MatchAll);
      case OpKind::Scan: {
        auto s = mk(SetKind::Scan);
        bind_leaf(*s, op);
        if (op.col && op.col->is_mv) { s->mv_off = op.col->mv_offsets_host.data(); s->mv_off_dev = op.col->mv_offsets_dev.as<int32_t>(); }
        return s;
      }
      case OpKind::Inverted: {
        const std::vector<int32_t>& ids = op.eval.exclusive ? op.eval.non_matching : op.eval.matching;
        if (ids.empty()) return mk(SetKind::Empty);   // InvertedIndexFilterOperator: no dictId to look up
        auto s = mk(SetKind::Bitmap);
        bind_leaf(*s, op);
        return s;
      }
      case OpKind::RangeIdx: {   // RangeIndexBasedFilterOperator over an exact index: a BitmapDocIdSet
        auto s = mk(SetKind::Bitmap);
        bind_leaf(*s, op);
        return s;
      }
      case OpKind::Sorted: {
        auto s = mk(SetKind::Sorted);
        s->ranges = sorted_ranges(op, n_docs);
        if (s->ranges.empty()) return mk(SetKind::Empty);
        return s;
      }
      case OpKind::Bitmap: {
        auto s = mk(SetKind::Bitmap);
        if (dev_leaves) {   // filled on the device from the range list
          for (size_t i = 0; i < op.range_lo.size(); i++) s->ranges.push_back({op.range_lo[i], op.range_hi[i]});
          return s;
        }
        s->owned = std::make_shared<HostBits>();
        s->owned->resize_for(n_docs);
        for (size_t i = 0; i < op.range_lo.size(); i++) s->owned->add_range(op.range_lo[i], op.range_hi[i]);
        s->leaf_bits = s->owned.get();
        return s;
      }
      case OpKind::And:
      case OpKind::Or: {
        auto s = mk(op.kind == OpKind::And ? SetKind::And : SetKind::Or);
        for (auto& c : op.children) s->children.push_back(trues(*c));
        return s;
      }
      case OpKind::Not:   // NotFilterOperator#getTrues
        if (op.children[0]->kind == OpKind::Empty) return mk(SetKind::MatchAll);
        return falses(*op.children[0]);
    }
    return mk(SetKind::Empty);
  }

  SetPtr not_of(SetPtr t) { auto s = mk(SetKind::Not); s->children.push_back(std::move(t)); return s; }

  SetPtr falses(const FilterOp& op) {
    switch (op.kind) {
      case OpKind::Not: return trues(*op.children[0]);
      case OpKind::And: {   // AndFilterOperator#getFalses
        std::vector<SetPtr> sets;
        for (auto& c : op.children) {
          SetPtr t = trues(*c);
          if (t->kind == SetKind::Empty) return mk(SetKind::MatchAll);
          if (t->kind == SetKind::MatchAll) continue;
          sets.push_back(std::move(t));
        }
        if (sets.empty()) return mk(SetKind::Empty);
        if (sets.size() == 1) return not_of(std::move(sets[0]));
        auto a = mk(SetKind::And);
        a->children = std::move(sets);
        return not_of(std::move(a));
      }
      case OpKind::Or: {    // OrFilterOperator#getFalses
        std::vector<SetPtr> sets;
        for (auto& c : op.children) {
          SetPtr t = trues(*c);
          if (t->kind == SetKind::MatchAll) return mk(SetKind::Empty);
          if (t->kind == SetKind::Empty) continue;
          sets.push_back(std::move(t));
        }
        if (sets.empty()) return mk(SetKind::MatchAll);
        if (sets.size() == 1) return not_of(std::move(sets[0]));
        auto o = mk(SetKind::Or);
        o->children = std::move(sets);
        return not_of(std::move(o));
      }
      default: {            // BaseFilterOperator#getFalses
        SetPtr t = trues(op);
        if (t->kind == SetKind::MatchAll) return mk(SetKind::Empty);
        if (t->kind == SetKind::Empty) return mk(SetKind::MatchAll);
        return not_of(std::move(t));
      }
    }
  }

  std::shared_ptr<HostBits> bits_of_ranges(const std::vector<std::pair<int32_t, int32_t>>& r) {
    auto b = std::make_shared<HostBits>();
    b->resize_for(n_docs);
    for (auto& p : r) b->add_range(p.first, p.second);
    return b;
  }
  std::shared_ptr<HostBits> clone_bits(const HostBits& src) {
    auto b = std::make_shared<HostBits>();
    b->n_docs = src.n_docs;
    b->w = src.w;
    b->w.resize((size_t)((n_docs + 63) / 64) + 1, 0);
    return b;
  }

  ItPtr iterator(Set& s) {
    switch (s.kind) {
      case SetKind::Empty: return std::make_unique<EmptyIt>();
      case SetKind::MatchAll: return std::make_unique<MatchAllIt>(n_docs);
      case SetKind::Scan:
        counters.push_back(std::make_unique<Counter>());
        return std::make_unique<ScanIt>(*s.leaf_bits, *counters.back(), s.mv_off);
      case SetKind::Bitmap: return std::make_unique<BitmapIt>(s.owned ? s.owned : clone_bits(*s.leaf_bits));
      case SetKind::Sorted: return std::make_unique<SortedIt>(s.ranges);
      case SetKind::Not: return std::make_unique<NotIt>(iterator(*s.children[0]), n_docs);
      case SetKind::And: {   // AndDocIdSet#iterator
        std::vector<ItPtr> all;
        for (auto& c : s.children) all.push_back(iterator(*c));
        int n_sorted = 0, n_bitmap = 0, n_scan = 0;
        for (auto& it : all) { n_sorted += it->kind == ItKind::Sorted; n_bitmap += it->kind == ItKind::Bitmap; n_scan += it->kind == ItKind::Scan; }
        const int n_index = n_sorted + n_bitmap;
        auto a = std::make_unique<AndIt>();
        if ((n_index > 0 && n_scan > 0) || n_index > 1) {
          std::shared_ptr<HostBits> docs;
          for (auto& it : all)   // sorted ranges first (intersected), then the bitmaps
            if (it->kind == ItKind::Sorted) {
              auto b = bits_of_ranges(*it->ranges());
              if (!docs) docs = b;
              else for (size_t i = 0; i < docs->w.size(); i++) docs->w[i] &= b->w[i];
            }
          for (auto& it : all)
            if (it->kind == ItKind::Bitmap) {
              if (!docs) docs = clone_bits(*it->doc_ids());
              else for (size_t i = 0; i < docs->w.size(); i++) docs->w[i] &= i < it->doc_ids()->w.size() ? it->doc_ids()->w[i] : 0;
            }
          for (auto& it : all)
            if (it->kind == ItKind::Scan && docs->next_set(0) >= 0) static_cast<ScanIt*>(it.get())->apply_and(*docs);   // applyAnd: an empty candidate set is not scanned
            else if (it->kind == ItKind::Scan) std::fill(docs->w.begin(), docs->w.end(), 0);
          auto merged = std::make_unique<BitmapIt>(docs);
          std::vector<ItPtr> remaining;
          for (auto& it : all) if (it->kind == ItKind::Other) remaining.push_back(std::move(it));
          if (remaining.empty()) return merged;
          a->its.push_back(std::move(merged));
          for (auto& it : remaining) a->its.push_back(std::move(it));
          return a;
        }
        a->its = std::move(all);
        return a;
      }
      case SetKind::Or: {    // OrDocIdSet#iterator
        std::vector<ItPtr> all;
        for (auto& c : s.children) all.push_back(iterator(*c));
        int n_sorted = 0;
        for (auto& it : all) n_sorted += it->kind == ItKind::Sorted;
        auto o = std::make_unique<OrIt>();
        if (n_sorted > 1) {
          auto docs = std::make_shared<HostBits>();
          docs->resize_for(n_docs);
          for (auto& it : all) {
            if (it->kind == ItKind::Sorted) for (auto& p : *it->ranges()) docs->add_range(p.first, p.second);
            else if (it->kind == ItKind::Bitmap) for (size_t i = 0; i < docs->w.size() && i < it->doc_ids()->w.size(); i++) docs->w[i] |= it->doc_ids()->w[i];
          }
          auto merged = std::make_unique<BitmapIt>(docs);
          std::vector<ItPtr> remaining;
          for (auto& it : all) if (it->kind != ItKind::Sorted && it->kind != ItKind::Bitmap) remaining.push_back(std::move(it));
          if (remaining.empty()) return merged;
          o->its.push_back(std::move(merged));
          for (auto& it : remaining) o->its.push_back(std::move(it));
        } else {
          o->its = std::move(all);
        }
        o->init();
        return o;
      }
    }
    return std::make_unique<EmptyIt>();
  }
};


// ---- device evaluation (pg_filter_stats_tiles.h) --------------------------------------------------------------------------------------
static const int kFsBlock = 256;

struct FsAndProg {
  int32_t k;                                  // children of the AndDocIdIterator, in its order
  const uint64_t* active;                     // the docs the AND is (re)started at by its parent's advance(); nullptr: drained by next() — every doc
  const uint64_t* match[FS_MAX_CHILDREN];     // the docs each child's iterator returns
  uint64_t* targets[FS_MAX_CHILDREN];         // out (second simulation): the targets each child was advanced to; nullptr: not wanted
};
static const int kFsStride = FS_TILE_WORDS + 1;   // a lane's tile in LDS: 9 words apart (lanes walk their tiles independently)
static size_t fs_and_lds_bytes(int k, bool emit) { return (size_t)((emit ? 2 : 1) * k + 1) * 64 * kFsStride * 8; }   // matches (+ targets) + the activation docs

// The AND automaton, one wavefront per workgroup over 64 consecutive tiles, a tile per lane.  The children's words of the 64 tiles (512 consecutive
// words each) are staged in LDS by coalesced loads: read from HBM in place, every lane walked its own 64 bytes of every child word by word
// (58 ms per 10^9 docs for two scans; profiles/r06_filter_stats_device.txt).  EMIT = false: the exit state of every entry state (`out`: the
// tiles' maps); EMIT = true: the simulation from the true entry state (`in`: inclusive prefix of the maps), targets collected in LDS and stored whole.
template <bool EMIT>
__global__ void __launch_bounds__(64) fs_and_kernel(const FsAndProg p, int64_t n_docs, int64_t n_words, int64_t n_tiles, uint32_t* __restrict__ out,
                                                     const uint32_t* __restrict__ in) {
  extern __shared__ __attribute__((aligned(16))) uint64_t fs_lds[];
  const int lane = (int)threadIdx.x;
  const int64_t tile0 = (int64_t)blockIdx.x * 64, word0 = tile0 * FS_TILE_WORDS;
  uint64_t* const la = fs_lds;
  uint64_t* const lm = la + 64 * kFsStride;
  uint64_t* const lt = lm + (size_t)p.k * 64 * kFsStride;
  for (int i = lane; i < 64 * FS_TILE_WORDS; i += 64) {
    const int64_t w = word0 + i;
    const int at = (i / FS_TILE_WORDS) * kFsStride + i % FS_TILE_WORDS;
    la[at] = p.active ? (w < n_words ? p.active[w] : 0) : ~0ULL;
    for (int c = 0; c < p.k; c++) {
      lm[c * 64 * kFsStride + at] = w < n_words ? p.match[c][w] : 0;
      if (EMIT) lt[c * 64 * kFsStride + at] = 0;
    }
  }
  __syncthreads();
  const int64_t tile = tile0 + lane;
  struct Io {   // the lane's tile: positions relative to its first doc
    const uint64_t* a;
    const uint64_t* m;
    uint64_t* t;
    int lane;
    __device__ __forceinline__ uint64_t active(int32_t w) const { return a[lane * kFsStride + w]; }
    __device__ __forceinline__ uint64_t match(int c, int32_t w) const { return m[(c * 64 + lane) * kFsStride + w]; }
    __device__ __forceinline__ void target(int c, int32_t doc) { t[(c * 64 + lane) * kFsStride + (doc >> 6)] |= 1ULL << (doc & 63); }
  } io{la, lm, lt, lane};
  if (tile < n_tiles) {
    const int64_t lo = tile * (FS_TILE_WORDS * 64);
    const int32_t end = (int32_t)(lo + FS_TILE_WORDS * 64 < n_docs ? FS_TILE_WORDS * 64 : n_docs - lo);
    if (!EMIT) out[tile] = fs_and_tile_exits(p.k, io, end);
    else (void)fs_and_tile(p.k, io, 0, end, tile ? (in[tile - 1] & 15u) : FS_IDLE, true);   // the entry: what an idle start of the segment has become by here
  }
  if (EMIT) {
    __syncthreads();
    for (int c = 0; c < p.k; c++) {
      if (!p.targets[c]) continue;
      for (int i = lane; i < 64 * FS_TILE_WORDS; i += 64)
        if (word0 + i < n_words) p.targets[c][word0 + i] = lt[(c * 64 + i / FS_TILE_WORDS) * kFsStride + i % FS_TILE_WORDS];
    }
  }
}
// entries of the docs in `v` (a word of docs from `base` on): a multi-value scan counts every entry of a doc it evaluates (MVScanDocIdIterator.java:65-117);
// mv_off: the docs' first entries (numDocs + 1 values) — runs of docs, two look-ups each
__device__ __forceinline__ unsigned long long fs_entries_of(uint64_t v, int64_t base, const int32_t* __restrict__ mv_off) {
  unsigned long long sum = 0;
  while (v) {
    const int a = __builtin_ctzll(v);
    const uint64_t rest = ~(v >> a);
    const int len = rest ? __builtin_ctzll(rest) : 64 - a;
    sum += (unsigned long long)(mv_off[base + a + len] - mv_off[base + a]);
    v = a + len >= 64 ? 0 : v & (~0ULL << (a + len));
  }
  return sum;
}
__device__ __forceinline__ void fs_block_add(unsigned long long v, unsigned long long* total) {
  __shared__ unsigned long long s_sum;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  if (v) atomicAdd(&s_sum, v);
  __syncthreads();
  if (threadIdx.x == 0 && s_sum) atomicAdd(total, s_sum);
}
// The visited latch, a word per lane and a tile (FS_LATCH_WORDS = 64 words) per wavefront: the words' summaries combine through two ballots
__global__ void __launch_bounds__(kFsBlock) fs_latch_summary_kernel(const uint64_t* __restrict__ t, const uint64_t* __restrict__ m, int64_t n_words, uint8_t* __restrict__ out) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t s = w < n_words ? fs_latch_summary(t, m, w, w + 1) : 0u;
  const unsigned long long ev = __ballot(s != 0), set = __ballot(s == 1);
  if ((threadIdx.x & 63) == 0) out[w >> 6] = ev ? (uint8_t)(((set >> (63 - __clzll((long long)ev))) & 1ULL) ? 1 : 2) : (uint8_t)0;
}
__global__ void __launch_bounds__(kFsBlock) fs_latch_count_kernel(const uint64_t* __restrict__ t, const uint64_t* __restrict__ m, int64_t n_words,
                                                                  const uint8_t* __restrict__ prefix, int64_t n_docs, const int32_t* __restrict__ mv_off, unsigned long long* total) {
  const int lane = (int)(threadIdx.x & 63);
  const int64_t n_latch_tiles = (n_words + 63) >> 6, waves = (int64_t)gridDim.x * (kFsBlock / 64);
  unsigned long long v = 0;
  for (int64_t tile = (int64_t)blockIdx.x * (kFsBlock / 64) + (threadIdx.x >> 6); tile < n_latch_tiles; tile += waves) {
    const int64_t w = tile * 64 + lane;
    const uint32_t s = w < n_words ? fs_latch_summary(t, m, w, w + 1) : 0u;
    const unsigned long long ev = __ballot(s != 0), set = __ballot(s == 1);
    const unsigned long long below = ev & ((1ULL << lane) - 1ULL);   // the words of this tile before the lane's
    const bool carry = below ? ((set >> (63 - __clzll((long long)below))) & 1ULL) != 0 : (tile > 0 && prefix[tile - 1] == 1);
    if (w < n_words) {
      const uint64_t visited = fs_latch_visited(t, m, w, carry, n_docs, nullptr);
      v += mv_off ? fs_entries_of(visited, w * 64, mv_off) : (unsigned long long)__popcll(visited);
    }
  }
  for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
  if (lane == 0 && v) atomicAdd(total, v);
}
// dst = a (op) b word by word — op 0: AND, 1: OR, 2: copy of a, 3: the docs not in a; with `total`: the docs of `a` (before the operation; their entries with `mv_off`) are added to it
__global__ void fs_words_kernel(uint64_t* __restrict__ dst, const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, int64_t n_words, int64_t n_docs, int op,
                                const int32_t* __restrict__ mv_off, unsigned long long* total) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long c = 0;
  if (w < n_words) {
    uint64_t x = a[w];
    if (w * 64 + 64 > n_docs) x &= ~0ULL >> (64 - (n_docs - w * 64));   // (the last word: docs that exist)
    c = mv_off ? fs_entries_of(x, w * 64, mv_off) : (unsigned long long)__popcll(x);
    dst[w] = op == 0 ? (x & b[w]) : op == 1 ? (x | b[w]) : op == 2 ? x : (~a[w] & (w * 64 + 64 > n_docs ? ~0ULL >> (64 - (n_docs - w * 64)) : ~0ULL));
  }
  if (total) fs_block_add(c, total);
}
__global__ void fs_fill_ranges_kernel(uint64_t* __restrict__ dst, const int32_t* __restrict__ lo, const int32_t* __restrict__ hi, int32_t n_ranges, int64_t n_words) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const int64_t first = w * 64, last = first + 63;
  // ascending disjoint inclusive ranges: the first one that ends at or behind this word's first doc
  int32_t a = 0, b = n_ranges;
  while (a < b) {
    const int32_t mid = (a + b) >> 1;
    if ((int64_t)hi[mid] < first) a = mid + 1; else b = mid;
  }
  uint64_t x = 0;
  for (int32_t r = a; r < n_ranges && (int64_t)lo[r] <= last; r++) {
    const int64_t s = (int64_t)lo[r] > first ? (int64_t)lo[r] : first, e = (int64_t)hi[r] < last ? (int64_t)hi[r] : last;
    x |= (~0ULL << (s - first)) & (~0ULL >> (63 - (e - first)));
  }
  dst[w] = x;
}

// ---- first / last set bit queries on a bitmap: groups of FS_GROUP_WORDS words with the first set position from each group on (suffix minimum)
// and the last one up to it (prefix maximum) — the look-ups of the NOT-over-scan count (pg_filter_stats_tiles.h)
#define FS_GROUP_WORDS 8
struct FsIndex {
  const uint64_t* words;
  const int32_t* first_rev;   // [n_groups - 1 - g]: first set position in groups >= g, INT32_MAX: none
  const int32_t* last_upto;   // [g]: last set position in groups <= g, -1: none
  int64_t n_docs, n_words, n_groups;
  __device__ __forceinline__ int64_t next(int64_t x) const {   // first set bit >= x, -1: none
    if (x < 0) x = 0;
    if (x >= n_docs) return -1;
    const int64_t g = x >> 9;
    int64_t w = x >> 6;
    const int64_t w_end = (g + 1) * FS_GROUP_WORDS < n_words ? (g + 1) * FS_GROUP_WORDS : n_words;
    uint64_t cur = words[w] & (~0ULL << (x & 63));
    for (;;) {
      if (cur) { const int64_t p = w * 64 + __builtin_ctzll(cur); return p < n_docs ? p : -1; }
      if (++w >= w_end) break;
      cur = words[w];
    }
    if (g + 1 >= n_groups) return -1;
    const int32_t v = first_rev[n_groups - 2 - g];
    return v == INT32_MAX ? -1 : (int64_t)v;
  }
  __device__ __forceinline__ int64_t prev(int64_t x) const {   // last set bit <= x, -1: none
    if (x < 0) return -1;
    if (x >= n_docs) x = n_docs - 1;
    const int64_t g = x >> 9;
    int64_t w = x >> 6;
    uint64_t cur = words[w] & (~0ULL >> (63 - (x & 63)));
    for (;;) {
      if (cur) return w * 64 + 63 - __builtin_clzll(cur);
      if (--w < g * FS_GROUP_WORDS) break;
      cur = words[w];
    }
    return g > 0 ? (int64_t)last_upto[g - 1] : -1;
  }
};
__global__ void __launch_bounds__(kFsBlock) fs_index_groups_kernel(const uint64_t* __restrict__ words, int64_t n_words, int64_t n_groups, int32_t* __restrict__ first_rev,
                                                                   int32_t* __restrict__ last) {   // a word per lane, a group per 8 lanes
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t x = w < n_words ? words[w] : 0;
  int32_t f = x ? (int32_t)(w * 64 + __builtin_ctzll(x)) : INT32_MAX, l = x ? (int32_t)(w * 64 + 63 - __builtin_clzll(x)) : -1;
  for (int off = 1; off < FS_GROUP_WORDS; off <<= 1) {
    const int32_t of = __shfl_xor(f, off), ol = __shfl_xor(l, off);
    f = of < f ? of : f;
    l = ol > l ? ol : l;
  }
  const int64_t g = w / FS_GROUP_WORDS;
  if ((threadIdx.x & (FS_GROUP_WORDS - 1)) == 0 && g < n_groups) { first_rev[n_groups - 1 - g] = f; last[g] = l; }
}
struct FsNotLook {   // the look-ups the count's formulas ask for
  FsIndex m, nm, t, r, c;
  const int32_t* mv_off;   // the scan is over a multi-value column: the docs' first entries (it counts entries, and its next() knows no batches)
  __device__ __forceinline__ int64_t span(int64_t a, int64_t b) const { return mv_off ? (int64_t)(mv_off[b] - mv_off[a]) : b - a; }
  __device__ __forceinline__ bool batched() const { return mv_off == nullptr; }
  __device__ __forceinline__ int64_t prev_non_match(int64_t x) const { return nm.prev(x); }
  __device__ __forceinline__ int64_t next_match(int64_t x) const { return m.next(x); }
  __device__ __forceinline__ int64_t next_non_match(int64_t x) const { return nm.next(x); }
  __device__ __forceinline__ int64_t prev_target(int64_t x) const { return t.prev(x); }
  __device__ __forceinline__ int64_t next_reset(int64_t x) const { return r.next(x); }
  __device__ __forceinline__ int64_t prev_reset(int64_t x) const { return r.prev(x); }
  __device__ __forceinline__ int64_t next_consume(int64_t x) const { return c.next(x); }
};
__device__ __forceinline__ void fs_wave_add(unsigned long long v, unsigned long long* total) {
  for (int off = 32; off; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(total, v);
}
// a word of the NOT's targets per lane: which of them advance() the scan (R), which are matches the NOT steps over (C); the advance() costs
__global__ void __launch_bounds__(kFsBlock) fs_not_resets_kernel(const FsNotLook look, int64_t n_words, int64_t n_docs, uint64_t* __restrict__ resets, uint64_t* __restrict__ consumes,
                                                                 unsigned long long* total) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long cost = 0;
  if (w < n_words) {
    uint64_t r = 0;
    for (uint64_t bits = look.t.words[w]; bits; bits &= bits - 1) {
      const int64_t t = w * 64 + __builtin_ctzll(bits);
      if (fs_not_is_reset(look, t)) {
        r |= 1ULL << (t & 63);
        cost += (unsigned long long)fs_not_advance_cost(look, t, n_docs);
      }
    }
    resets[w] = r;
    consumes[w] = look.t.words[w] & look.m.words[w];
  }
  fs_wave_add(cost, total);
}
// a NOT inside an OR: of the OR's targets, those OrDocIdIterator#advance hands on to the NOT (fs_not_in_or_receives)
__global__ void __launch_bounds__(kFsBlock) fs_not_received_kernel(const FsNotLook look, int64_t n_words, uint64_t* __restrict__ received) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  uint64_t r = 0;
  for (uint64_t bits = look.t.words[w]; bits; bits &= bits - 1) {
    const int64_t t = w * 64 + __builtin_ctzll(bits);
    if (fs_not_in_or_receives(look, t)) r |= 1ULL << (t & 63);
  }
  received[w] = r;
}
// ... then the batches of every episode, charged to its last target that is a match; the constructor's next() by the first lane
__global__ void __launch_bounds__(kFsBlock) fs_not_episodes_kernel(const FsNotLook look, int64_t n_words, int64_t n_docs, unsigned long long* total) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long cost = 0;
  if (w < n_words)
    for (uint64_t bits = look.c.words[w]; bits; bits &= bits - 1) cost += (unsigned long long)fs_not_episode_cost(look, w * 64 + __builtin_ctzll(bits), n_docs);
  if (w == 0) cost += (unsigned long long)fs_not_ctor_cost(look, n_docs);
  fs_wave_add(cost, total);
}
// ---- a NOT over an OR of leaves (pg_filter_stats_tiles.h, fs_notor_*): the NOT's resets from (targets, union); per scan child the resets that
// advance() it, then its advances' costs with the episodes of next() calls between them
struct FsNotOrChild {
  FsIndex m, a;            // the child's matches, its advances
  const int32_t* mv_off;
  __device__ __forceinline__ int64_t next_match(int64_t x) const { return x < 0 ? -1 : m.next(x); }
  __device__ __forceinline__ int64_t prev_advance(int64_t x) const { return a.prev(x); }
  __device__ __forceinline__ int64_t span(int64_t x, int64_t y) const { return mv_off ? (int64_t)(mv_off[y] - mv_off[x]) : y - x; }
  __device__ __forceinline__ bool batched() const { return mv_off == nullptr; }
};
__global__ void __launch_bounds__(kFsBlock) fs_notor_resets_kernel(const FsNotLook u, int64_t n_words, uint64_t* __restrict__ resets) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  uint64_t r = 0;
  for (uint64_t bits = u.t.words[w]; bits; bits &= bits - 1) {
    const int64_t t = w * 64 + __builtin_ctzll(bits);
    if (fs_not_is_reset(u, t)) r |= 1ULL << (t & 63);
  }
  resets[w] = r;
}
__global__ void __launch_bounds__(kFsBlock) fs_notor_advances_kernel(const FsNotLook u, const FsNotOrChild c, const uint64_t* __restrict__ resets, int64_t n_words,
                                                                     uint64_t* __restrict__ advances) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  uint64_t a = 0;
  for (uint64_t bits = resets[w]; bits; bits &= bits - 1) {
    const int64_t r = w * 64 + __builtin_ctzll(bits);
    if (fs_notor_child_advanced(u, c, r)) a |= 1ULL << (r & 63);
  }
  advances[w] = a;
}
__global__ void __launch_bounds__(kFsBlock) fs_notor_costs_kernel(const FsNotLook u, const FsNotOrChild c, int64_t n_words, int64_t n_docs, unsigned long long* total) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long cost = 0;
  if (w < n_words)
    for (uint64_t bits = c.a.words[w]; bits; bits &= bits - 1) cost += (unsigned long long)fs_notor_advance_cost(u, c, w * 64 + __builtin_ctzll(bits), n_docs);
  if (w == 0) cost += (unsigned long long)fs_notor_episode(u, c, n_docs, n_docs);   // the episode behind the child's last advance
  fs_wave_add(cost, total);
}
struct FsMaxI32 { __host__ __device__ int32_t operator()(int32_t a, int32_t b) const { return a > b ? a : b; } };
struct FsMinI32 { __host__ __device__ int32_t operator()(int32_t a, int32_t b) const { return a < b ? a : b; } };
struct FsMapThen { __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return fs_map_then(a, b); } };
struct FsLatchThen { __host__ __device__ uint8_t operator()(uint8_t a, uint8_t b) const { return b ? b : a; } };

// The walk over the doc-id SETS that Emu::iterator makes on the host, with bitmaps in HBM and the AND automaton in tiles.  Two passes over the same
// code: a dry one that only sizes the scratch, then the launches.
struct DevEval {
  Emu& emu;
  int64_t n_docs, n_words, n_tiles, n_latch_tiles;
  hipStream_t stream;
  bool dry = true;
  uint8_t* base = nullptr;
  size_t off = 0;
  int64_t closed_form = 0;               // drained scans: every doc once
  unsigned long long* total = nullptr;   // device counter of everything else
  std::deque<std::vector<int32_t>> keep_alive;

  template <typename T> T* take(size_t count) {
    off = (off + 255) & ~(size_t)255;
    T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
    off += count * sizeof(T);
    return p;
  }
  uint64_t* take_words() { return take<uint64_t>((size_t)n_words + 1); }
  dim3 grid_for(int64_t lanes) const { return dim3((unsigned)((lanes + kFsBlock - 1) / kFsBlock)); }

  enum class Kind { Scan, Bitmap, Sorted, Other, Unfit };
  // AndDocIdSet#iterator: index-based children and scans merge into one bitmap-based iterator when there is an index-based child beside a scan,
  // or two of them; with nothing else left that iterator IS the AND's
  struct AndShape { bool fits, merged; int n_other; };
  static AndShape and_shape(const Set& s) {
    int n_index = 0, n_scan = 0, n_other = 0;
    for (auto& c : s.children) {
      const Kind k = child_kind(*c);
      if (k == Kind::Unfit) return {false, false, 0};
      n_index += k == Kind::Bitmap || k == Kind::Sorted;
      n_scan += k == Kind::Scan;
      n_other += k == Kind::Other;
    }
    const bool merged = (n_index > 0 && n_scan > 0) || n_index > 1;
    return {(merged ? 1 + n_other : (int)s.children.size()) <= FS_MAX_CHILDREN, merged, n_other};
  }
  static bool fits_and(const Set& s) { return and_shape(s).fits; }
  // what Emu::iterator would hand an AND for this child
  static Kind child_kind(const Set& s) {
    switch (s.kind) {
      case SetKind::Scan: return Kind::Scan;   // (over a multi-value column: entries instead of docs)
      case SetKind::Bitmap: return Kind::Bitmap;
      case SetKind::Sorted: return Kind::Sorted;
      case SetKind::And: {   // an AND under an AND / an OR: advance()d by its parent — the tile automaton started at the parent's targets
        const AndShape a = and_shape(s);
        if (!a.fits) return Kind::Unfit;
        return a.merged && a.n_other == 0 ? Kind::Bitmap : Kind::Other;
      }
      case SetKind::Or: {
        int n_sorted = 0, n_other = 0;
        for (auto& c : s.children) {
          if (c->kind == SetKind::Scan) n_other++;
          else if (c->kind == SetKind::Sorted) n_sorted++;
          else if (c->kind == SetKind::And) { const Kind k = child_kind(*c); if (k == Kind::Unfit) return Kind::Unfit; n_other += k == Kind::Other; }
          else if (c->kind == SetKind::Not) { if (child_kind(*c) == Kind::Unfit || c->children[0]->kind == SetKind::Or) return Kind::Unfit; n_other++; }   // (the NOT — over one leaf — receives the OR's targets its cursor lies before)
          else if (c->kind != SetKind::Bitmap) return Kind::Unfit;   // an OR inside an OR
        }
        return n_sorted > 1 && n_other == 0 ? Kind::Bitmap : Kind::Other;   // OrDocIdSet#iterator merges index-based children only beside >= 2 sorted ones
      }
      case SetKind::Not: {   // NotDocIdIterator over one leaf or over an OR of leaves (other compound children: the host walk)
        const Set& c = *s.children[0];
        if (c.kind == SetKind::Scan) return Kind::Other;
        if (c.kind == SetKind::Or) {
          for (auto& l : c.children)
            if (l->kind != SetKind::Scan && l->kind != SetKind::Bitmap && l->kind != SetKind::Sorted) return Kind::Unfit;
          return Kind::Other;
        }
        return c.kind == SetKind::Bitmap || c.kind == SetKind::Sorted ? Kind::Other : Kind::Unfit;
      }
      default: return Kind::Unfit;   // Empty / MatchAll under an AND
    }
  }
  static bool fits_drained(const Set& s) {
    switch (s.kind) {
      case SetKind::Scan: return true;
      case SetKind::Not: return fits_drained(*s.children[0]);
      case SetKind::Or: for (auto& c : s.children) if (!fits_drained(*c)) return false; return true;
      case SetKind::And: return fits_and(s);
      default: return true;
    }
  }

  const uint64_t* ranges_words(const std::vector<std::pair<int32_t, int32_t>>& r) {
    uint64_t* w = take_words();
    int32_t* lo = take<int32_t>(r.size() + 1);
    int32_t* hi = take<int32_t>(r.size() + 1);
    if (dry) return w;
    keep_alive.emplace_back();
    std::vector<int32_t>& h = keep_alive.back();
    for (auto& p : r) h.push_back(p.first);
    for (auto& p : r) h.push_back(p.second);
    if (!r.empty()) {
      PG_HIP(hipMemcpyAsync(lo, h.data(), r.size() * 4, hipMemcpyHostToDevice, stream));
      PG_HIP(hipMemcpyAsync(hi, h.data() + r.size(), r.size() * 4, hipMemcpyHostToDevice, stream));
    }
    hipLaunchKernelGGL(fs_fill_ranges_kernel, grid_for(n_words), dim3(kFsBlock), 0, stream, w, lo, hi, (int32_t)r.size(), n_words);
    return w;
  }
  const uint64_t* leaf_words(const Set& s) {   // Scan / Bitmap / Sorted
    if (s.kind == SetKind::Sorted || (s.kind == SetKind::Bitmap && !s.dev_words)) return ranges_words(s.ranges);
    if (!dry && !s.dev_words) fail(PG_ERR_INTERNAL, "filter statistics: a leaf's match bitmap is missing");
    return s.dev_words;
  }
  void words_op(uint64_t* dst, const uint64_t* a, const uint64_t* b, int op, bool count_a, const int32_t* mv_off = nullptr) {
    if (dry) return;
    hipLaunchKernelGGL(fs_words_kernel, grid_for(n_words), dim3(kFsBlock), 0, stream, dst, a, b ? b : a, n_words, n_docs, op, mv_off, count_a ? total : nullptr);
  }
  const uint64_t* or_words(const Set& s, std::vector<std::pair<const Set*, const uint64_t*>>* rests = nullptr) {   // the docs an OR's iterator returns
    uint64_t* acc = take_words();
    bool first = true;
    for (auto& c : s.children) {
      const uint64_t* w;
      if (c->kind == SetKind::Not) {
        uint64_t* rest = take_words();
        words_op(rest, leaf_words(*c->children[0]), nullptr, 3, false);
        if (rests) rests->push_back({c.get(), rest});
        w = rest;
      } else {
        w = c->kind == SetKind::And ? plan_and(*c).docs() : leaf_words(*c);
      }
      words_op(acc, first ? w : acc, first ? nullptr : w, first ? 2 : 1, false);
      first = false;
    }
    return acc;
  }

  // a leaf's count from the targets of the AND child it sits under
  void latch_count(const uint64_t* targets, const uint64_t* match, const int32_t* mv_off) {
    const size_t padded = (size_t)grid_for(n_words).x * (kFsBlock / 64);   // (every wavefront of the grid writes its tile's summary)
    uint8_t* summary = take<uint8_t>(padded);
    uint8_t* prefix = take<uint8_t>(padded);
    size_t tmp_bytes = 0;
    PG_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, (uint8_t*)nullptr, (uint8_t*)nullptr, (size_t)n_latch_tiles, FsLatchThen(), stream));
    uint8_t* tmp = take<uint8_t>(tmp_bytes + 256);
    if (dry) return;
    hipLaunchKernelGGL(fs_latch_summary_kernel, grid_for(n_words), dim3(kFsBlock), 0, stream, targets, match, n_words, summary);
    PG_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, summary, prefix, (size_t)n_latch_tiles, FsLatchThen(), stream));
    hipLaunchKernelGGL(fs_latch_count_kernel, dim3((unsigned)std::min<int64_t>((n_latch_tiles + kFsBlock / 64 - 1) / (kFsBlock / 64), 4096)), dim3(kFsBlock), 0, stream, targets, match, n_words, prefix, n_docs, mv_off, total);
  }

  FsIndex index_of(const uint64_t* words) {
    const int64_t n_groups = (n_words + FS_GROUP_WORDS - 1) / FS_GROUP_WORDS;
    int32_t* first_rev = take<int32_t>((size_t)n_groups);
    int32_t* first_scan = take<int32_t>((size_t)n_groups);
    int32_t* last = take<int32_t>((size_t)n_groups);
    int32_t* last_scan = take<int32_t>((size_t)n_groups);
    size_t tmp_bytes = 0, tmp2 = 0;
    PG_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)n_groups, FsMinI32(), stream));
    PG_HIP(rocprim::inclusive_scan(nullptr, tmp2, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)n_groups, FsMaxI32(), stream));
    uint8_t* tmp = take<uint8_t>(std::max(tmp_bytes, tmp2) + 256);
    if (!dry) {
      hipLaunchKernelGGL(fs_index_groups_kernel, grid_for(n_words), dim3(kFsBlock), 0, stream, words, n_words, n_groups, first_rev, last);
      PG_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, first_rev, first_scan, (size_t)n_groups, FsMinI32(), stream));
      PG_HIP(rocprim::inclusive_scan(tmp, tmp2, last, last_scan, (size_t)n_groups, FsMaxI32(), stream));
    }
    return FsIndex{words, first_scan, last_scan, n_docs, n_words, n_groups};
  }
  // the scan under a NOT: `targets` of the NOT, the scan's matches, the docs the NOT returns (the others)
  void not_count(const uint64_t* targets, const uint64_t* match, const uint64_t* others, const int32_t* mv_off) {
    uint64_t* resets = take_words();
    uint64_t* consumes = take_words();
    FsNotLook look{};
    look.mv_off = mv_off;
    look.m = index_of(match);
    look.nm = index_of(others);
    look.t = index_of(targets);
    if (!dry) hipLaunchKernelGGL(fs_not_resets_kernel, grid_for(n_words), dim3(kFsBlock), 0, stream, look, n_words, n_docs, resets, consumes, total);
    look.r = index_of(resets);
    look.c = index_of(consumes);
    if (!dry) hipLaunchKernelGGL(fs_not_episodes_kernel, grid_for(n_words), dim3(kFsBlock), 0, stream, look, n_words, n_docs, total);
  }

  // the scans of an OR of leaves under a NOT: `targets` of the NOT, the union of the OR's children, the docs the NOT returns (the others)
  template <typename CountedList>
  void not_or_count(const uint64_t* targets, const uint64_t* uni, const uint64_t* others, const CountedList& scans) {
    uint64_t* resets = take_words();
    FsNotLook u{};
    u.m = index_of(uni);
    u.nm = index_of(others);
    u.t = index_of(targets);
    if (!dry) hipLaunchKernelGGL(fs_notor_resets_kernel, grid_for(n_words), dim3(kFsBlock), 0, stream, u, n_words, resets);
    for (auto& leaf : scans) {
      uint64_t* advances = take_words();
      FsNotOrChild c{};
      c.mv_off = leaf.mv_off;
      c.m = index_of(leaf.match);
      if (!dry) hipLaunchKernelGGL(fs_notor_advances_kernel, grid_for(n_words), dim3(kFsBlock), 0, stream, u, c, (const uint64_t*)resets, n_words, advances);
      c.a = index_of(advances);
      if (!dry) hipLaunchKernelGGL(fs_notor_costs_kernel, grid_for(n_words), dim3(kFsBlock), 0, stream, u, c, n_words, n_docs, total);
    }
  }

  // AndDocIdSet#iterator of one AND: its children's iterators in the AND's order (index-based children and scans merged, with the applyAnd counts —
  // once, when the iterator is built), and the docs the AND itself returns (asked for by a parent)
  struct Counted { const uint64_t* match; const int32_t* mv_off; };   // a scan leaf; mv_off: over a multi-value column
  struct AndChild {
    const uint64_t* match;
    std::vector<Counted> counted;            // the scan leaves that receive this child's targets
    const uint64_t* not_scan = nullptr;      // the scan under a NOT
    const int32_t* not_mv_off = nullptr;     // ... over a multi-value column
    struct NotInOr { const uint64_t* scan; const uint64_t* rest; const int32_t* mv_off; };
    std::vector<NotInOr> nots;               // the scans under NOTs inside the OR this child is
    std::vector<const Set*> inner;           // the ANDs started at this child's targets: the child itself, or children of the OR it is
    const uint64_t* not_or = nullptr;        // a NOT over an OR of leaves: the union of their matches
    std::vector<Counted> not_or_scans;       // ... and the scans among them
  };
  struct AndPlan {
    DevEval* ev = nullptr;
    std::vector<AndChild> its;
    const uint64_t* all = nullptr;
    bool have_all = false;
    const uint64_t* docs() {   // the intersection of the children
      if (have_all) return all;
      have_all = true;
      if (its.size() == 1) return all = its[0].match;
      uint64_t* acc = ev->take_words();
      for (size_t j = 1; j < its.size(); j++) ev->words_op(acc, j == 1 ? its[0].match : acc, its[j].match, 0, false);
      return all = acc;
    }
  };
  std::deque<std::pair<const Set*, AndPlan>> plans;
  AndPlan& plan_and(const Set& s) {
    for (auto& p : plans) if (p.first == &s) return p.second;
    plans.emplace_back(&s, AndPlan{});
    AndPlan& P = plans.back().second;   // (a deque: references survive the plans of nested ANDs)
    P.ev = this;
    std::vector<const Set*> sorted, bitmaps, scans, others;
    for (auto& c : s.children) {
      switch (child_kind(*c)) {
        case Kind::Sorted: sorted.push_back(c.get()); break;
        case Kind::Bitmap: bitmaps.push_back(c.get()); break;
        case Kind::Scan: scans.push_back(c.get()); break;
        default: others.push_back(c.get()); break;
      }
    }
    const int n_index = (int)(sorted.size() + bitmaps.size());
    auto words_of = [&](const Set& c) { return c.kind == SetKind::Or ? or_words(c) : c.kind == SetKind::And ? plan_and(c).docs() : leaf_words(c); };
    auto child_of = [&](const Set& c) {
      AndChild a;
      if (c.kind == SetKind::Not && c.children[0]->kind == SetKind::Or) {
        const Set& o = *c.children[0];
        const uint64_t* uni = or_words(o);
        uint64_t* rest = take_words();
        words_op(rest, uni, nullptr, 3, false);
        a.match = rest;
        a.not_or = uni;
        for (auto& l : o.children)
          if (l->kind == SetKind::Scan) a.not_or_scans.push_back({leaf_words(*l), l->mv_off_dev});
      } else if (c.kind == SetKind::Not) {
        const uint64_t* inner = leaf_words(*c.children[0]);
        uint64_t* rest = take_words();
        words_op(rest, inner, nullptr, 3, false);
        a.match = rest;
        if (c.children[0]->kind == SetKind::Scan) { a.not_scan = inner; a.not_mv_off = c.children[0]->mv_off_dev; }
      } else if (c.kind == SetKind::Or) {
        std::vector<std::pair<const Set*, const uint64_t*>> rests;   // the docs the NOTs inside it return
        a.match = or_words(c, &rests);
        for (auto& l : c.children) {
          if (l->kind == SetKind::Scan) a.counted.push_back({leaf_words(*l), l->mv_off_dev});
          else if (l->kind == SetKind::And) a.inner.push_back(l.get());
        }
        for (auto& r : rests)
          if (r.first->children[0]->kind == SetKind::Scan) a.nots.push_back({leaf_words(*r.first->children[0]), r.second, r.first->children[0]->mv_off_dev});
      } else if (c.kind == SetKind::And) {
        a.match = plan_and(c).docs();
        a.inner.push_back(&c);
      } else {
        a.match = leaf_words(c);
        if (c.kind == SetKind::Scan) a.counted.push_back({a.match, c.mv_off_dev});
      }
      return a;
    };
    if ((n_index > 0 && !scans.empty()) || n_index > 1) {
      uint64_t* docs = take_words();
      bool first = true;
      for (auto* list : {&sorted, &bitmaps})
        for (const Set* c : *list) {
          const uint64_t* w = words_of(*c);
          words_op(docs, first ? w : docs, first ? nullptr : w, first ? 2 : 0, false);
          first = false;
        }
      for (const Set* c : scans) words_op(docs, docs, leaf_words(*c), 0, true, c->mv_off_dev);   // applyAnd: every surviving candidate is evaluated once
      P.its.push_back({docs, {}, nullptr, {}});
      for (const Set* c : others) P.its.push_back(child_of(*c));
    } else {
      for (auto& c : s.children) P.its.push_back(child_of(*c));
    }
    return P;
  }
  // AndDocIdIterator started at the docs of `active` (nullptr: drained by next())
  void sim_and(AndPlan& P, const uint64_t* active) {
    std::vector<AndChild>& its = P.its;
    bool any = false;
    for (auto& a : its) any = any || !a.counted.empty() || a.not_scan || !a.inner.empty() || !a.nots.empty() || !a.not_or_scans.empty();
    if (!any) return;   // bitmap-based iterators all the way down: nothing is scanned
    FsAndProg prog{};
    prog.k = (int32_t)its.size();
    prog.active = active;
    for (int j = 0; j < prog.k; j++) {
      const AndChild& a = its[(size_t)j];
      prog.match[j] = a.match;
      prog.targets[j] = a.counted.empty() && !a.not_scan && a.inner.empty() && a.nots.empty() && a.not_or_scans.empty() ? nullptr : take_words();
    }
    uint32_t* maps = take<uint32_t>((size_t)n_tiles);
    uint32_t* prefix = take<uint32_t>((size_t)n_tiles);
    size_t tmp_bytes = 0;
    PG_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n_tiles, FsMapThen(), stream));
    uint8_t* tmp = take<uint8_t>(tmp_bytes + 256);
    if (!dry) {
      const dim3 grid((unsigned)((n_tiles + 63) / 64));
      hipLaunchKernelGGL(fs_and_kernel<false>, grid, dim3(64), fs_and_lds_bytes(prog.k, false), stream, prog, n_docs, n_words, n_tiles, maps, (const uint32_t*)nullptr);
      PG_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, maps, prefix, (size_t)n_tiles, FsMapThen(), stream));
      hipLaunchKernelGGL(fs_and_kernel<true>, grid, dim3(64), fs_and_lds_bytes(prog.k, true), stream, prog, n_docs, n_words, n_tiles, (uint32_t*)nullptr, (const uint32_t*)prefix);
    }
    for (int j = 0; j < prog.k; j++) {
      for (const Counted& leaf : its[(size_t)j].counted) latch_count(prog.targets[j], leaf.match, leaf.mv_off);
      if (its[(size_t)j].not_scan) not_count(prog.targets[j], its[(size_t)j].not_scan, its[(size_t)j].match, its[(size_t)j].not_mv_off);
      for (auto& nt : its[(size_t)j].nots) {   // a NOT inside the OR: the targets that reach it, then as under the AND itself
        uint64_t* received = take_words();
        FsNotLook look{};
        look.nm = index_of(nt.rest);
        look.t = index_of(prog.targets[j]);
        if (!dry) hipLaunchKernelGGL(fs_not_received_kernel, grid_for(n_words), dim3(kFsBlock), 0, stream, look, n_words, received);
        not_count(received, nt.scan, nt.rest, nt.mv_off);
      }
      if (!its[(size_t)j].not_or_scans.empty()) not_or_count(prog.targets[j], its[(size_t)j].not_or, its[(size_t)j].match, its[(size_t)j].not_or_scans);
      for (const Set* in : its[(size_t)j].inner) sim_and(plan_and(*in), prog.targets[j]);
    }
  }
  void run_and(const Set& s) { sim_and(plan_and(s), nullptr); }

  void run_drained(const Set& s) {   // the iterator is drained by next(): every child of an OR / NOT is drained in turn
    switch (s.kind) {
      case SetKind::Scan: closed_form += s.mv_off ? (int64_t)s.mv_off[n_docs] : n_docs; break;   // SVScanDocIdIterator#next: whole batches to the end; over a multi-value column every entry
      case SetKind::Not: run_drained(*s.children[0]); break;
      case SetKind::Or: for (auto& c : s.children) run_drained(*c); break;
      case SetKind::And: run_and(s); break;
      default: break;
    }
  }
};
}  // namespace

// `root`: the physical filter tree of the plan; `leaves`: the match bitmap of every Scan / Inverted leaf in it (from the GPU).
int64_t emulate_entries_scanned_in_filter(const FilterOp& root, const StatLeafBits& leaves, int32_t n_docs) {
  Emu emu{leaves, n_docs, {}};
  SetPtr set = emu.trues(root);
  ItPtr it = emu.iterator(*set);
  if (it->kind != ItKind::Bitmap && it->kind != ItKind::Sorted)   // an index-only iterator scans nothing while it is drained
    while (it->next() != kEof) {}   // DocIdSetOperator drains the iterator
  int64_t total = 0;
  for (auto& c : emu.counters) total += c->entries;
  return total;
}

bool filter_stats_on_device(const FilterOp& root, int32_t n_docs) {
  if (n_docs <= 0) return false;
  static const StatLeafBits no_host_bits;
  StatLeafWords none;
  Emu emu{no_host_bits, n_docs, {}};
  emu.dev_leaves = &none;
  SetPtr set = emu.trues(root);
  return DevEval::fits_drained(*set);
}

int64_t entries_scanned_on_device(const FilterOp& root, const StatLeafWords& leaves, int32_t n_docs, DeviceBuffer& arena, void* hip_stream) {
  static const StatLeafBits no_host_bits;
  Emu emu{no_host_bits, n_docs, {}};
  emu.dev_leaves = &leaves;
  SetPtr set = emu.trues(root);
  const int64_t n_words = ((int64_t)n_docs + 63) / 64;
  DevEval ev{emu, n_docs, n_words, (n_words + FS_TILE_WORDS - 1) / FS_TILE_WORDS, (n_words + FS_LATCH_WORDS - 1) / FS_LATCH_WORDS, (hipStream_t)hip_stream};
  ev.total = ev.take<unsigned long long>(1);
  ev.run_drained(*set);   // dry: sizes
  if (arena.size < ev.off + 256) arena.alloc(ev.off + ev.off / 4 + 256);
  ev.dry = false;
  ev.plans.clear();
  ev.base = arena.as<uint8_t>();
  ev.off = 0;
  ev.closed_form = 0;
  ev.total = ev.take<unsigned long long>(1);
  PG_HIP(hipMemsetAsync(ev.total, 0, 8, ev.stream));
  ev.run_drained(*set);
  PG_HIP(hipGetLastError());
  unsigned long long counted = 0;
  PG_HIP(hipMemcpyAsync(&counted, ev.total, 8, hipMemcpyDeviceToHost, ev.stream));
  PG_HIP(hipStreamSynchronize(ev.stream));
  return ev.closed_form + (int64_t)counted;
}

}  // namespace pg
