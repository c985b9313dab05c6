// numEntriesScannedInFilter of leapfrogged AND shapes, evaluated tile by tile (device: pg_filter_stats.cpp; host model: tests/filter_stats_tiles_test.cpp).
//
// The reference's AndDocIdIterator (AndDocIdIterator.java:37-66) is a sequential automaton over its children's advance() calls, and a scan child
// counts every doc it steps over (SVScanDocIdIterator.java:101-112: advance(target) visits [target, first match >= target]).  Two facts make it
// parallel:
//
//  1. A scan iterator that is only ever ADVANCED visits the union of [t, next_match(t)] over the targets t it was handed (targets ascend; a call
//     that finds its cursor already at or past the target was covered by an earlier interval).  The same holds for every scan child of an
//     OrDocIdIterator (OrDocIdIterator.java:91-119: advance() forwards the target to each child whose cursor lies before it).  So the count of a
//     leaf is a function of (its match bitmap M, the target bitmap T of the AND child it sits under): doc d is visited iff the last target <= d
//     lies after the last match < d — a set / reset latch over the doc positions, evaluated with carries (fs_latch_*).
//  2. Cut into tiles of docs, the AND automaton enters a tile in one of K + 1 states only: `idle` (a match was emitted, nothing is pending: it
//     resumes where next() / the parent's advance() starts it again) or `child c is scanning` (its advance() was called in an earlier tile and has
//     found nothing yet; whatever it returns becomes max_doc with max_idx = c, index = 0).  Every tile is simulated from every entry state
//     (fs_and_tile), the K + 1 -> K + 1 maps compose associatively (a device scan), and a second simulation from the tile's TRUE entry state
//     writes the targets.
//
// Match bitmaps: bit d of word d / 64; bits at and beyond n_docs are clear.
#pragma once   // (C++: the tile automaton is a template over how the bitmaps are read)
#include <stdint.h>

#if defined(__HIPCC__)
#define FS_HD __host__ __device__ __forceinline__
#else
#define FS_HD static inline
#endif

#define FS_MAX_CHILDREN 7          // children of the AND (K + 1 entry states in 4-bit fields of one 32-bit map)
#define FS_TILE_WORDS 8            // 512 docs per tile of the AND automaton (a lane's tile: 64 bytes per child, staged in LDS on the device)
#define FS_LATCH_WORDS 64          // words per tile of the visited latch (one wavefront, a word per lane, on the device)
#define FS_IDLE 0u                 // entry / exit state: nothing pending, the AND resumes at the next doc it is started at; c + 1: child c is scanning

#define FS_MERGED 15u              // fs_and_tile(stop_at): the simulation emitted `stop_at` — from there on every entry state runs the same course

// first set bit in [from, end) of a tile's bitmap read through `word(i)` (positions relative to the tile: < 64 * FS_TILE_WORDS), -1 when none
template <typename WordAt>
FS_HD int32_t fs_next_set(WordAt word, int32_t from, int32_t end) {
  if (from >= end) return -1;
  int32_t wi = from >> 6;
  const int32_t last = (end - 1) >> 6;
  uint64_t cur = word(wi) & (~0ULL << (from & 63));
  while (!cur) {
    if (++wi > last) return -1;
    cur = word(wi);
  }
  const int32_t p = wi * 64 + __builtin_ctzll(cur);
  return p < end ? p : -1;
}

// One tile of the AND automaton over k children from entry state `entry`, docs [from, end) RELATIVE to the tile (`from` = 0 unless the caller
// resumes the idle state inside the tile); returns the exit state.  `io.match(child, word)`: a word of the docs the child's iterator returns;
// `io.active(word)`: the docs at which the AND may be (re)started — every doc for an AND that is drained by next(); for an AND that sits under
// an OR or another AND, the targets its parent advance()s it to (the parent only calls once its cursor lies behind the AND's last answer: the
// first such target behind an emitted doc).  With `emit`, `io.target(child, doc)` records every advance() call's target (the caller owns the
// tile's words).  `stop_at` >= 0: return FS_MERGED when that doc is emitted.  (An AND emits every doc all its children match, whatever state
// it came from, unless it idles across it: behind the tile's first such doc the simulations that emitted it coincide, so the exits pass runs
// each of them up to that doc and the common course once.)
template <typename Io>
FS_HD uint32_t fs_and_tile(int32_t k, Io& io, int32_t from, int32_t end, uint32_t entry, bool emit, int32_t stop_at = -1) {
  int32_t x, max_idx = -1, index = 0;
  if (entry != FS_IDLE) {
    const int32_t c = (int32_t)entry - 1;
    const int32_t d = fs_next_set([&](int32_t w) { return io.match(c, w); }, from, end);
    if (d < 0) return entry;   // still scanning
    x = d;
    max_idx = c;
  } else {
    x = fs_next_set([&](int32_t w) { return io.active(w); }, from, end);
    if (x < 0) return FS_IDLE;
  }
  for (;;) {
    while (index < k) {
      if (index == max_idx) { index++; continue; }
      if (emit) io.target(index, x);
      const int32_t c = index;
      const int32_t d = fs_next_set([&](int32_t w) { return io.match(c, w); }, x, end);
      if (d < 0) return (uint32_t)index + 1;   // the call returns in a later tile (or never: EOF ends the AND)
      if (d == x) {
        index++;
      } else {
        x = d;
        max_idx = index;
        index = 0;
      }
    }
    if (x == stop_at) return FS_MERGED;
    // every child sits on x: emitted; the AND resumes at the next doc it is started at
    x = fs_next_set([&](int32_t w) { return io.active(w); }, x + 1, end);
    if (x < 0) return FS_IDLE;
    max_idx = -1;
    index = 0;
  }
}

// the exit state of every entry state of one tile (`end` docs), as 4-bit fields
template <typename Io>
FS_HD uint32_t fs_and_tile_exits(int32_t k, Io& io, int32_t end) {
  int32_t first_common = -1;   // the first doc every child matches
  for (int32_t w = 0; w * 64 < end && first_common < 0; w++) {
    uint64_t all = ~0ULL;
    for (int32_t c = 0; c < k; c++) all &= io.match(c, w);
    if (all) first_common = w * 64 + __builtin_ctzll(all);
  }
  if (first_common >= end) first_common = -1;
  uint32_t common = FS_MERGED, map = 0;
  for (int32_t s = 0; s <= k; s++) {
    uint32_t r = fs_and_tile(k, io, 0, end, (uint32_t)s, false, first_common);
    if (r == FS_MERGED) {
      if (common == FS_MERGED) common = fs_and_tile(k, io, first_common + 1, end, FS_IDLE, false);
      r = common;
    }
    map |= r << (4 * s);
  }
  return map;
}

// the K + 1 exit states of a tile as 4-bit fields; composition `then(a, b)`: a's tile first
FS_HD uint32_t fs_map_then(uint32_t a, uint32_t b) {
  uint32_t out = 0;
  for (int s = 0; s < 8; s++) out |= ((b >> (4 * ((a >> (4 * s)) & 15u))) & 15u) << (4 * s);
  return out;
}
#define FS_MAP_IDENTITY 0x76543210u

// ---- the visited latch: state(d) = T(d) | (!M(d - 1) & state(d - 1)); a leaf's count is the number of docs with state(d) set ------------------
// summary of a run of words: 0 keeps the state that enters, 1 leaves it set, 2 leaves it clear (as seen by the doc AFTER the run)
FS_HD uint32_t fs_latch_summary(const uint64_t* t, const uint64_t* m, int64_t w_lo, int64_t w_hi) {
  for (int64_t w = w_hi - 1; w >= w_lo; w--) {
    const uint64_t tw = t[w], mw = m[w];
    if (!(tw | mw)) continue;
    // the last event decides: a target at p sets from p on, a match at p clears from p + 1 on (a doc that is both: cleared behind it)
    const int pt = tw ? 63 - __builtin_clzll(tw) : -1, pm = mw ? 63 - __builtin_clzll(mw) : -1;
    return pt > pm ? 1u : 2u;
  }
  return 0u;
}
FS_HD uint32_t fs_latch_then(uint32_t a, uint32_t b) { return b ? b : a; }

// the visited docs of word w given the state entering it (`carry`), docs at and beyond n_docs masked out; `top`: the state leaving the word
FS_HD uint64_t fs_latch_visited(const uint64_t* t, const uint64_t* m, int64_t w, bool carry, int64_t n_docs, bool* top) {
  const uint64_t s = t[w], mw = m[w];
  const uint64_t r = (mw << 1) | (w > 0 ? m[w - 1] >> 63 : 0);
  // bits that set (generate), bits that hand the previous state on (propagate): a prefix network over the 64 positions
  uint64_t g = s, pp = ~(s | r);
  g |= pp & (g << 1);  pp &= pp << 1;
  g |= pp & (g << 2);  pp &= pp << 2;
  g |= pp & (g << 4);  pp &= pp << 4;
  g |= pp & (g << 8);  pp &= pp << 8;
  g |= pp & (g << 16); pp &= pp << 16;
  g |= pp & (g << 32);
  // positions below the first set / reset of the word take the entering state
  const uint64_t ev = s | r;
  const uint64_t below = ev ? ((ev & (~ev + 1)) - 1) : ~0ULL;
  uint64_t v = g | (carry ? below : 0);
  if (top) *top = (v >> 63) != 0;
  const int64_t base = w * 64;
  if (base + 64 > n_docs) v &= n_docs > base ? (~0ULL >> (64 - (n_docs - base))) : 0;
  return v;
}
// visited docs of words [w_lo, w_hi) given the state entering w_lo
FS_HD int64_t fs_latch_count(const uint64_t* t, const uint64_t* m, int64_t w_lo, int64_t w_hi, bool carry, int64_t n_docs) {
  int64_t total = 0;
  for (int64_t w = w_lo; w < w_hi && w * 64 < n_docs; w++) total += __builtin_popcountll(fs_latch_visited(t, m, w, carry, n_docs, &carry));
  return total;
}

// ---- NOT over a scan, under an AND (NotDocIdIterator.java:45-70 over SVScanDocIdIterator.java:76-112) -----------------------------------------
// The NOT keeps `nm`, the scan's next match; the AND advances it to ascending targets t.  With r(t) = the doc the NOT returned (the first
// non-match >= t), nm before a call is always the scan's first match behind the doc returned before.  Three kinds of call:
//   t < nm   nothing is scanned;
//   t > nm   a RESET: the scan is advance()d — it drops its batch and counts [t, first match >= t] — and a new episode of next() calls starts
//            right behind that match (B0);
//   t == nm  (also a reset that lands on a match) the NOT steps over the run of matches at t with one next() per match.
// Within an episode next() streams: batches of 256 docs from B0 on, whole batches counted, up to the batch that holds the last match handed out —
// the scan's first match behind the run of the episode's last target that is a match.  (Batches overlap what a later reset scans again: the count
// is a sum over calls, not a set of docs.)  So with T the child's target bitmap, M the scan's matches, R the resets and C = T & M:
//   count = sum over R of the advance() cost + sum over the episodes' last C target of the batches + the constructor's episode (B0 = 0, one
//   next() whatever follows).
// `L` answers first / last set bit queries on those bitmaps (-1: none): the host model scans, the device keeps two-level indexes.
#define FS_SCAN_BATCH 256   // BlockDocIdIterator.OPTIMAL_ITERATOR_BATCH_SIZE

// the child's next match the NOT knows of before it is advanced to target t (t = n_docs: behind its last target), -1: the child is exhausted
template <typename L>
FS_HD int64_t fs_not_known_match(L& l, int64_t t) {
  const int64_t t_prev = l.prev_target(t - 1);
  int64_t from = 0;   // the first doc behind the one the NOT returned last
  if (t_prev >= 0) {
    const int64_t r = l.next_non_match(t_prev);
    if (r < 0) return -1;   // the matches ran to the end: the child is exhausted
    from = r + 1;
  }
  return l.next_match(from);
}
template <typename L>
FS_HD bool fs_not_is_reset(L& l, int64_t t) {   // is the scan advance()d when the NOT is advanced to target t
  const int64_t nm = fs_not_known_match(l, t);
  return nm >= 0 && t > nm;
}
// `l.span(a, b)`: what a scan counts for evaluating the docs [a, b) — b - a, or their entries over a multi-value column;
// `l.batched()`: does next() scan whole batches (SVScanDocIdIterator) or doc by doc up to the match (MVScanDocIdIterator.java:65-117)
template <typename L>
FS_HD int64_t fs_not_advance_cost(L& l, int64_t t, int64_t n_docs) {
  const int64_t p = l.next_match(t);
  return l.span(t, p < 0 ? n_docs : p + 1);
}
template <typename L>
FS_HD int64_t fs_not_batches(L& l, int64_t b0, int64_t q, int64_t n_docs) {   // what next() counts from b0 until it has handed out match q (-1: until EOF)
  if (q < 0) return l.span(b0, n_docs);
  if (!l.batched()) return l.span(b0, q + 1);
  const int64_t end = b0 + FS_SCAN_BATCH * ((q - b0) / FS_SCAN_BATCH + 1);
  return l.span(b0, end < n_docs ? end : n_docs);
}
// `t`: a target that is a match (in C): the batches of its episode, if it is the episode's last such target (else 0)
template <typename L>
FS_HD int64_t fs_not_episode_cost(L& l, int64_t t, int64_t n_docs) {
  const int64_t c_next = l.next_consume(t + 1), r_next = l.next_reset(t + 1);
  if (c_next >= 0 && !(r_next >= 0 && r_next <= c_next)) return 0;   // a later target of the same episode hands out more matches
  const int64_t t_e = l.prev_reset(t);
  const int64_t b0 = t_e < 0 ? 0 : l.next_match(t_e) + 1;
  const int64_t run_end = l.next_non_match(t);
  return fs_not_batches(l, b0, run_end < 0 ? -1 : l.next_match(run_end + 1), n_docs);
}
// the constructor's next() when no target of its episode is a match
template <typename L>
FS_HD int64_t fs_not_ctor_cost(L& l, int64_t n_docs) {
  const int64_t c_first = l.next_consume(0), r_first = l.next_reset(0);
  if (c_first >= 0 && !(r_first >= 0 && r_first <= c_first)) return 0;   // counted with that target's episode (B0 = 0)
  return fs_not_batches(l, 0, l.next_match(0), n_docs);
}

// A NOT inside an OR (under an AND): OrDocIdIterator#advance forwards a target only to a child whose cursor lies before it, and a NOT's cursor is
// the doc it returned last — the first non-match at or behind its last target.  Of the OR's targets inside one run of the scan's matches (and on
// the non-match that ends it) only the first reaches the NOT.
template <typename L>
FS_HD bool fs_not_in_or_receives(L& l, int64_t t) {
  if (t == 0) return true;
  return !(l.prev_target(t - 1) > l.prev_non_match(t - 1));
}

// ---- NOT over an OR of leaves, under an AND (NotDocIdIterator over OrDocIdIterator.java:50-119, both entry points) -----------------------------
// Towards the NOT the OR is one child whose matches are the union U of its children's: the NOT's resets R and its known next match follow from
// (targets, U) as above.  Child i of the OR holds a cursor — its first match at or behind the OR's current doc (OrDocIdIterator keeps every
// child on its next doc and asks it for another only once that doc has been returned) — and so, with `known` = the union's match the NOT knows of:
//   * at a reset r the OR advance()s child i iff its cursor lies before r: next_match_i(known(r)) < r — the scan drops its batch and counts
//     [r, its first match >= r];
//   * between two of ITS advances (A_i, a subset of R) the scan streams through next(): whole batches from behind the match its advance
//     returned (B0; 0 for the constructor's episode, whose first next() always happens) up to the batch that holds its cursor at the end of the
//     episode, next_match_i(known(the next advance)) — nothing when the cursor has not moved; after the last advance: up to
//     next_match_i(known(behind the last target)); an exhausted cursor has scanned to the end.
// `U`: prev_target / next_match / next_non_match over the union; `C`: next_match / span / batched of the child, prev_advance over A_i.
template <typename U, typename C>
FS_HD bool fs_notor_child_advanced(U& u, C& c, int64_t r) {   // r: a reset of the NOT
  const int64_t cursor = c.next_match(fs_not_known_match(u, r));
  return cursor >= 0 && cursor < r;
}
template <typename U, typename C>
FS_HD int64_t fs_notor_episode(U& u, C& c, int64_t a, int64_t n_docs) {   // the episode that ends at the child's advance a (n_docs: the last one)
  const int64_t prev = c.prev_advance(a - 1);
  int64_t b0 = 0;
  if (prev >= 0) {
    const int64_t p = c.next_match(prev);
    if (p < 0) return 0;   // that advance ran to the end: the OR dropped the child
    b0 = p + 1;
  }
  const int64_t known = fs_not_known_match(u, a);
  const int64_t q = known < 0 ? -1 : c.next_match(known);
  if (prev >= 0 && q >= 0 && q < b0) return 0;   // the cursor is still the match the advance returned
  return fs_not_batches(c, b0, q, n_docs);
}
template <typename U, typename C>
FS_HD int64_t fs_notor_advance_cost(U& u, C& c, int64_t a, int64_t n_docs) {   // a in A_i: its advance and the episode it ends
  return fs_not_advance_cost(c, a, n_docs) + fs_notor_episode(u, c, a, n_docs);
}
