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
  const int32_t* mv_off = nullptr;                           // Scan over a multi-value column: the docs' first entries
  std::shared_ptr<HostBits> owned;                           // Bitmap built here (flips, range lists)
  std::vector<std::pair<int32_t, int32_t>> ranges;           // Sorted
  std::vector<std::unique_ptr<Set>> children;                // And / Or / Not
};
using SetPtr = std::unique_ptr<Set>;

struct Emu {
  const StatLeafBits& leaves;
  int32_t n_docs;
  std::vector<std::unique_ptr<Counter>> counters;

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
        s->leaf_bits = &leaves.at(&op);
        if (op.col && op.col->is_mv) s->mv_off = op.col->mv_offsets_host.data();
        return s;
      }
      case OpKind::Inverted: {
        const std::vector<int32_t>& ids = op.eval.exclusive ? op.eval.non_matching : op.eval.matching;
        if (ids.empty()) return mk(SetKind::Empty);   // InvertedIndexFilterOperator: no dictId to look up
        auto s = mk(SetKind::Bitmap);
        s->leaf_bits = &leaves.at(&op);
        return s;
      }
      case OpKind::RangeIdx: {   // RangeIndexBasedFilterOperator over an exact index: a BitmapDocIdSet
        auto s = mk(SetKind::Bitmap);
        s->leaf_bits = &leaves.at(&op);
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

}  // namespace pg
