/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY (see po_internal.h).
 * Filter operators, BlockDocIdSets and BlockDocIdIterators: SURVEY.md §8a rows a1, a2, a4, a6, a8, a9.
 * Evaluation order, batching and the numEntriesScannedInFilter bookkeeping follow the reference exactly so that the
 * ExecutionStatistics goldens (e.g. 63064 in InnerSegmentAggregationSingleValueQueriesTest.java:55-58) reproduce.
 *
 * Deliberate deviation (documented in DESIGN.md): OrDocIdSet.iterator() in this reference snapshot never adds
 * BitmapBasedDocIdIterators to its merge list (core/operator/docidsets/OrDocIdSet.java:74-84), so (a) the merge into one
 * BitmapDocIdIterator happens only with >= 2 SORTED children (`numSorted + 0 > 1`) and (b) in that case the bitmap children
 * would be dropped from the union.  The oracle follows (a) exactly — the iterator type decides whether an enclosing AND
 * restricts its scans by the OR (numEntriesScannedInFilter) — and deviates in (b) only: the bitmap children are ORed in, since
 * dropping them would change query results.
 */
#include <stdio.h>

#include "po_internal.h"

/* =====================================================================================================================
 * iterators
 * ===================================================================================================================== */
static po_iter* iter_new(int kind, int32_t (*next)(po_iter*), int32_t (*advance)(po_iter*, int32_t), void* st) {
  po_iter* it = (po_iter*)po_xcalloc(1, sizeof(po_iter));
  it->kind = kind;
  it->next = next;
  it->advance = advance;
  it->state = st;
  return it;
}

/* ---- SVScanDocIdIterator, core/operator/dociditerators/SVScanDocIdIterator.java ---------------------------------------- */
typedef struct scan_state {
  const po_pred_eval* eval;
  const po_column* col;
  int32_t num_docs;
  int32_t batch[PO_SCAN_BATCH];
  int32_t buf_i[PO_SCAN_BATCH];
  int32_t first_mismatch, cursor, next_doc_id;
  int64_t num_entries_scanned;
  /* MVScanDocIdIterator: DictIdMatcher's buffer of _maxNumValuesPerMVEntry ints and the reader context */
  int32_t* mv_buf;
  po_mv_ctx mv_ctx;
} scan_state;

/* ValueMatcher#doesValueMatch (:213-291) */
static int scan_does_value_match(scan_state* s, int32_t doc_id) {
  const po_column* c = s->col;
  if (c->has_dictionary) {
    int32_t d = (c->fwd_encoding == PG_FWD_DICT_SORTED) ? po_sorted_get_dict_id(c, doc_id) : po_fixedbit_read(c, doc_id);
    return po_pred_apply_dict(s->eval, d);
  }
  if (c->data_type == PG_TYPE_STRING) { int32_t len = 0; const uint8_t* v = po_raw_get_bytes(c, doc_id, &len); return po_pred_apply_string(s->eval, v, len); }
  switch (c->data_type) {
    case PG_TYPE_INT: return po_pred_apply_int(s->eval, po_raw_get_int(c, doc_id));
    case PG_TYPE_LONG: return po_pred_apply_long(s->eval, po_raw_get_long(c, doc_id));
    case PG_TYPE_FLOAT: return po_pred_apply_float(s->eval, po_raw_get_float(c, doc_id));
    default: return po_pred_apply_double(s->eval, po_raw_get_double(c, doc_id));
  }
}

/* ValueMatcher#matchValues: read the block, then PredicateEvaluator#applySV(limit, docIds, values) compaction */
static int scan_match_values(scan_state* s, int limit, int32_t* doc_ids) {
  const po_column* c = s->col;
  int matches = 0;
  if (c->has_dictionary) {
    po_fwd_read_dict_ids(c, doc_ids, limit, s->buf_i);
    for (int i = 0; i < limit; i++)
      if (po_pred_apply_dict(s->eval, s->buf_i[i])) doc_ids[matches++] = doc_ids[i];
    return matches;
  }
  for (int i = 0; i < limit; i++) {
    int32_t d = doc_ids[i];
    int ok;
    if (c->data_type == PG_TYPE_STRING) { int32_t len = 0; const uint8_t* v = po_raw_get_bytes(c, d, &len); ok = po_pred_apply_string(s->eval, v, len); }
    else switch (c->data_type) {
      case PG_TYPE_INT: ok = po_pred_apply_int(s->eval, po_raw_get_int(c, d)); break;
      case PG_TYPE_LONG: ok = po_pred_apply_long(s->eval, po_raw_get_long(c, d)); break;
      case PG_TYPE_FLOAT: ok = po_pred_apply_float(s->eval, po_raw_get_float(c, d)); break;
      default: ok = po_pred_apply_double(s->eval, po_raw_get_double(c, d)); break;
    }
    if (ok) doc_ids[matches++] = d;
  }
  return matches;
}

static int32_t scan_next(po_iter* it) { /* :76-98 */
  scan_state* s = (scan_state*)it->state;
  if (s->cursor >= s->first_mismatch) {
    int limit, batch_size = 0;
    do {
      limit = s->num_docs - s->next_doc_id;
      if (limit > PO_SCAN_BATCH) limit = PO_SCAN_BATCH;
      if (limit > 0) {
        for (int i = 0; i < limit; i++) s->batch[i] = s->next_doc_id + i;
        batch_size = scan_match_values(s, limit, s->batch);
        s->next_doc_id += limit;
        s->num_entries_scanned += limit;
      }
    } while ((limit > 0) & (batch_size == 0));
    s->first_mismatch = batch_size;
    s->cursor = 0;
    if (s->first_mismatch == 0) return PO_EOF;
  }
  return s->batch[s->cursor++];
}

static int32_t scan_advance(po_iter* it, int32_t target) { /* :101-112 */
  scan_state* s = (scan_state*)it->state;
  s->next_doc_id = target;
  s->first_mismatch = 0;
  while (s->next_doc_id < s->num_docs) {
    int32_t d = s->next_doc_id++;
    s->num_entries_scanned++;
    if (scan_does_value_match(s, d)) return d;
  }
  return PO_EOF;
}

/* ScanBasedDocIdIterator#applyAnd(ImmutableRoaringBitmap) → :115-142: candidate docIds in batches of _batch.length */
static po_bitmap* scan_apply_and(po_iter* it, const po_bitmap* doc_ids) {
  scan_state* s = (scan_state*)it->state;
  po_bitmap* result = po_bitmap_new(doc_ids->universe);
  int32_t buffer[PO_SCAN_BATCH];
  int64_t pos = po_bitmap_next_set(doc_ids, 0);
  while (pos >= 0) {
    int limit = 0;
    /* RoaringBatchIterator#nextBatch never crosses a container (65536-doc) boundary */
    int64_t container_end = ((pos >> 16) + 1) << 16;
    while (pos >= 0 && pos < container_end && limit < PO_SCAN_BATCH) {
      buffer[limit++] = (int32_t)pos;
      pos = po_bitmap_next_set(doc_ids, pos + 1);
    }
    if (limit > 0) {
      int first_mismatch = scan_match_values(s, limit, buffer);
      for (int i = 0; i < first_mismatch; i++) po_bitmap_add(result, buffer[i]);
    }
    s->num_entries_scanned += limit;
  }
  return result;
}

/* ---- MVScanDocIdIterator, core/operator/dociditerators/MVScanDocIdIterator.java -------------------------------------------------
 * DictIdMatcher#doesValueMatch (:184-192): read the doc's dictIds, count ALL of them, then PredicateEvaluator#applyMV
 * (BaseDictionaryBasedPredicateEvaluator.java:164-180: exclusive predicates need every value to pass, the others any value) */
static int mvscan_does_value_match(scan_state* s, int32_t doc_id) {
  const int32_t length = po_mv_get_dict_ids(s->col, doc_id, s->mv_buf, &s->mv_ctx);
  s->num_entries_scanned += length;
  if (s->eval->exclusive) {
    for (int32_t i = 0; i < length; i++)
      if (!po_pred_apply_dict(s->eval, s->mv_buf[i])) return 0;
    return 1;
  }
  for (int32_t i = 0; i < length; i++)
    if (po_pred_apply_dict(s->eval, s->mv_buf[i])) return 1;
  return 0;
}
static int32_t mvscan_next(po_iter* it) { /* :65-77 */
  scan_state* s = (scan_state*)it->state;
  while (s->next_doc_id < s->num_docs) {
    const int32_t d = s->next_doc_id++;
    if (mvscan_does_value_match(s, d)) return d;
  }
  return PO_EOF;
}
static int32_t mvscan_advance(po_iter* it, int32_t target) { /* :80-83 */
  ((scan_state*)it->state)->next_doc_id = target;
  return mvscan_next(it);
}
static po_bitmap* mvscan_apply_and(po_iter* it, const po_bitmap* doc_ids) { /* applyAnd :86-117: every candidate doc, one by one */
  scan_state* s = (scan_state*)it->state;
  po_bitmap* result = po_bitmap_new(doc_ids->universe);
  for (int64_t pos = po_bitmap_next_set(doc_ids, 0); pos >= 0; pos = po_bitmap_next_set(doc_ids, pos + 1))
    if (mvscan_does_value_match(s, (int32_t)pos)) po_bitmap_add(result, (int32_t)pos);
  return result;
}

/* ---- BitmapDocIdIterator / RangelessBitmapDocIdIterator ----------------------------------------------------------------- */
typedef struct bitmap_it_state {
  po_bitmap* doc_ids;   /* borrowed */
  int64_t pos;          /* PeekableIntIterator position: next candidate */
  int32_t num_docs;     /* BitmapDocIdIterator only */
  int rangeless;
} bitmap_it_state;

static int32_t bitmap_next(po_iter* it) {
  bitmap_it_state* s = (bitmap_it_state*)it->state;
  int64_t d = po_bitmap_next_set(s->doc_ids, s->pos);
  if (d < 0) {
    s->pos = s->doc_ids->n_words * 64;
    return PO_EOF;
  }
  s->pos = d + 1;
  if (!s->rangeless && d >= s->num_docs) return PO_EOF;
  return (int32_t)d;
}
static int32_t bitmap_advance(po_iter* it, int32_t target) {
  bitmap_it_state* s = (bitmap_it_state*)it->state;
  if (s->pos < target) s->pos = target; /* advanceIfNeeded */
  return bitmap_next(it);
}
static po_iter* bitmap_iter_new(po_bitmap* b, int32_t num_docs, int rangeless) {
  bitmap_it_state* s = (bitmap_it_state*)po_xcalloc(1, sizeof(*s));
  s->doc_ids = b;
  s->num_docs = num_docs;
  s->rangeless = rangeless;
  return iter_new(rangeless ? PO_IT_RANGELESS_BITMAP : PO_IT_BITMAP, bitmap_next, bitmap_advance, s);
}

/* ---- SortedDocIdIterator, core/operator/dociditerators/SortedDocIdIterator.java:30-90 ------------------------------------ */
typedef struct ranges { int n; int32_t* lo; int32_t* hi; } ranges;
typedef struct sorted_it_state { ranges* r; int cur; int32_t next_doc_id; } sorted_it_state;

static int32_t sorted_next(po_iter* it) {
  sorted_it_state* s = (sorted_it_state*)it->state;
  if (s->r->n == 0) return PO_EOF;
  if (s->next_doc_id <= s->r->hi[s->cur]) return s->next_doc_id++;
  if (s->cur < s->r->n - 1) {
    s->cur++;
    s->next_doc_id = s->r->lo[s->cur];
    return s->next_doc_id++;
  }
  return PO_EOF;
}
static int32_t sorted_advance(po_iter* it, int32_t target) {
  sorted_it_state* s = (sorted_it_state*)it->state;
  if (s->r->n == 0) return PO_EOF;
  if (target <= s->r->hi[s->cur]) {
    s->next_doc_id = target > s->r->lo[s->cur] ? target : s->r->lo[s->cur];
    return s->next_doc_id++;
  }
  while (s->cur < s->r->n - 1) {
    s->cur++;
    if (target <= s->r->hi[s->cur]) {
      s->next_doc_id = target > s->r->lo[s->cur] ? target : s->r->lo[s->cur];
      return s->next_doc_id++;
    }
  }
  return PO_EOF;
}

/* ---- MatchAll / Empty ---------------------------------------------------------------------------------------------------- */
typedef struct matchall_state { int32_t num_docs, next_doc_id; } matchall_state;
static int32_t matchall_next(po_iter* it) {
  matchall_state* s = (matchall_state*)it->state;
  return s->next_doc_id < s->num_docs ? s->next_doc_id++ : PO_EOF;
}
static int32_t matchall_advance(po_iter* it, int32_t t) {
  ((matchall_state*)it->state)->next_doc_id = t;
  return matchall_next(it);
}
static int32_t empty_next(po_iter* it) { (void)it; return PO_EOF; }
static int32_t empty_advance(po_iter* it, int32_t t) { (void)it; (void)t; return PO_EOF; }

/* ---- AndDocIdIterator, core/operator/dociditerators/AndDocIdIterator.java:41-68 ----------------------------------------- */
typedef struct and_it_state { int n; po_iter** its; int32_t next_doc_id; } and_it_state;
static int32_t and_next(po_iter* it) {
  and_it_state* s = (and_it_state*)it->state;
  int32_t max_doc = s->next_doc_id;
  int max_idx = -1;
  int index = 0;
  while (index < s->n) {
    if (index == max_idx) {
      index++;
      continue;
    }
    int32_t d = s->its[index]->advance(s->its[index], max_doc);
    if (d != PO_EOF) {
      if (d == max_doc) {
        index++;
      } else {
        max_doc = d;
        max_idx = index;
        index = 0;
      }
    } else {
      return PO_EOF;
    }
  }
  s->next_doc_id = max_doc;
  return s->next_doc_id++;
}
static int32_t and_advance(po_iter* it, int32_t t) {
  ((and_it_state*)it->state)->next_doc_id = t;
  return and_next(it);
}

/* ---- OrDocIdIterator, core/operator/dociditerators/OrDocIdIterator.java:52-125 ------------------------------------------- */
typedef struct or_it_state { int n_live; po_iter** its; int32_t* next_ids; int32_t prev; } or_it_state;
static void or_remove_exhausted(or_it_state* s) {
  int i = 0;
  while (i < s->n_live) {
    if (s->next_ids[i] == PO_EOF) {
      s->n_live--;
      s->its[i] = s->its[s->n_live];
      s->next_ids[i] = s->next_ids[s->n_live];
    } else {
      i++;
    }
  }
}
static int32_t or_next(po_iter* it) {
  or_it_state* s = (or_it_state*)it->state;
  int32_t next = INT32_MAX;
  int exhausted = 0;
  for (int i = 0; i < s->n_live; i++) {
    int32_t d = s->next_ids[i];
    if (d == s->prev) {
      d = s->its[i]->next(s->its[i]);
      s->next_ids[i] = d;
      if (d == PO_EOF) {
        exhausted = 1;
        continue;
      }
    }
    if (d < next) next = d;
  }
  if (exhausted) or_remove_exhausted(s);
  if (next != INT32_MAX) {
    s->prev = next;
    return next;
  }
  return PO_EOF;
}
static int32_t or_advance(po_iter* it, int32_t target) {
  or_it_state* s = (or_it_state*)it->state;
  int32_t next = INT32_MAX;
  int exhausted = 0;
  for (int i = 0; i < s->n_live; i++) {
    int32_t d = s->next_ids[i];
    if (d < target) {
      d = s->its[i]->advance(s->its[i], target);
      s->next_ids[i] = d;
      if (d == PO_EOF) {
        exhausted = 1;
        continue;
      }
    }
    if (d < next) next = d;
  }
  if (exhausted) or_remove_exhausted(s);
  if (next != INT32_MAX) {
    s->prev = next;
    return next;
  }
  return PO_EOF;
}
static po_iter* or_iter_new(int n, po_iter** its) {
  or_it_state* s = (or_it_state*)po_xcalloc(1, sizeof(*s));
  s->n_live = n;
  s->its = (po_iter**)po_xmalloc(sizeof(po_iter*) * (size_t)(n + 1));
  s->next_ids = (int32_t*)po_xmalloc(sizeof(int32_t) * (size_t)(n + 1));
  for (int i = 0; i < n; i++) {
    s->its[i] = its[i];
    s->next_ids[i] = -1;
  }
  s->prev = -1;
  return iter_new(PO_IT_OR, or_next, or_advance, s);
}

/* ---- NotDocIdIterator, core/operator/dociditerators/NotDocIdIterator.java:28-80 ------------------------------------------- */
typedef struct not_it_state { po_iter* child; int32_t num_docs, next_doc_id, next_non_matching; } not_it_state;
static int32_t not_next(po_iter* it) {
  not_it_state* s = (not_it_state*)it->state;
  if (s->next_doc_id >= s->num_docs) return PO_EOF;
  while (s->next_doc_id == s->next_non_matching) {
    s->next_doc_id++;
    int32_t n = s->child->next(s->child);
    s->next_non_matching = (n == PO_EOF) ? s->num_docs : n;
  }
  if (s->next_doc_id >= s->num_docs) return PO_EOF;
  return s->next_doc_id++;
}
static int32_t not_advance(po_iter* it, int32_t target) {
  not_it_state* s = (not_it_state*)it->state;
  s->next_doc_id = target;
  if (target > s->next_non_matching) {
    int32_t n = s->child->advance(s->child, target);
    s->next_non_matching = (n == PO_EOF) ? s->num_docs : n;
  }
  return not_next(it);
}

/* =====================================================================================================================
 * BlockDocIdSets
 * ===================================================================================================================== */
static po_docidset* set_new(int kind, po_iter* (*iterator)(po_docidset*), int64_t (*entries)(po_docidset*), void* st) {
  po_docidset* s = (po_docidset*)po_xcalloc(1, sizeof(*s));
  s->kind = kind;
  s->iterator = iterator;
  s->num_entries_scanned = entries;
  s->state = st;
  return s;
}
static int64_t zero_entries(po_docidset* s) { (void)s; return 0; }

/* SVScanDocIdSet: the iterator is created with the set */
static po_iter* scanset_iterator(po_docidset* s) { return (po_iter*)s->state; }
static int64_t scanset_entries(po_docidset* s) { return ((scan_state*)((po_iter*)s->state)->state)->num_entries_scanned; }
static po_docidset* scanset_new(const po_pred_eval* eval, const po_column* col, int32_t num_docs) {
  scan_state* st = (scan_state*)po_xcalloc(1, sizeof(*st));
  st->eval = eval;
  st->col = col;
  st->num_docs = num_docs;
  po_iter* it;
  if (col->is_mv) {   /* ScanBasedFilterOperator#getTrues :61-65: MVScanDocIdSet for a multi-value column */
    st->mv_buf = (int32_t*)po_xmalloc(sizeof(int32_t) * (size_t)(col->mv_max_values + 1));
    st->mv_ctx = (po_mv_ctx)PO_MV_CTX_INIT;
    it = iter_new(PO_IT_SCAN, mvscan_next, mvscan_advance, st);
  } else {
    it = iter_new(PO_IT_SCAN, scan_next, scan_advance, st);
  }
  return set_new(PO_SET_SCAN, scanset_iterator, scanset_entries, it);
}

typedef struct bitmapset_state { po_bitmap* b; int32_t num_docs; } bitmapset_state;
static po_iter* bitmapset_iterator(po_docidset* s) {
  bitmapset_state* st = (bitmapset_state*)s->state;
  return bitmap_iter_new(st->b, st->num_docs, 0);
}
static po_docidset* bitmapset_new(po_bitmap* b, int32_t num_docs) {
  bitmapset_state* st = (bitmapset_state*)po_xcalloc(1, sizeof(*st));
  st->b = b;
  st->num_docs = num_docs;
  return set_new(PO_SET_BITMAP, bitmapset_iterator, zero_entries, st);
}

static po_iter* sortedset_iterator(po_docidset* s) {
  sorted_it_state* st = (sorted_it_state*)po_xcalloc(1, sizeof(*st));
  st->r = (ranges*)s->state;
  st->cur = 0;
  st->next_doc_id = st->r->n ? st->r->lo[0] : 0;
  return iter_new(PO_IT_SORTED, sorted_next, sorted_advance, st);
}

static po_iter* matchallset_iterator(po_docidset* s) {
  matchall_state* st = (matchall_state*)po_xcalloc(1, sizeof(*st));
  st->num_docs = (int32_t)(intptr_t)s->state;
  return iter_new(PO_IT_MATCH_ALL, matchall_next, matchall_advance, st);
}
static po_docidset* matchallset_new(int32_t num_docs) {
  return set_new(PO_SET_MATCH_ALL, matchallset_iterator, zero_entries, (void*)(intptr_t)num_docs);
}
static po_iter* emptyset_iterator(po_docidset* s) { (void)s; return iter_new(PO_IT_EMPTY, empty_next, empty_advance, NULL); }
static po_docidset* emptyset_new(void) { return set_new(PO_SET_EMPTY, emptyset_iterator, zero_entries, NULL); }

/* ---- AndDocIdSet, core/operator/docidsets/AndDocIdSet.java:72-186 ---------------------------------------------------------- */
typedef struct compound_state {
  int n; po_docidset** sets; int32_t num_docs;
  int n_scan_based; po_docidset** scan_based;   /* _scanBasedDocIdSets */
  int64_t entries_non_scan;
} compound_state;

static int cmp_bitmap_card(const void* a, const void* b) {
  int64_t x = po_bitmap_cardinality(((bitmap_it_state*)(*(po_iter* const*)a)->state)->doc_ids);
  int64_t y = po_bitmap_cardinality(((bitmap_it_state*)(*(po_iter* const*)b)->state)->doc_ids);
  return x < y ? -1 : (x > y ? 1 : 0);
}

static po_iter* andset_iterator(po_docidset* set) {
  compound_state* cs = (compound_state*)set->state;
  int n = cs->n;
  po_iter** all = (po_iter**)po_xcalloc((size_t)n + 1, sizeof(po_iter*));
  po_iter** sorted_its = (po_iter**)po_xcalloc((size_t)n + 1, sizeof(po_iter*));
  po_iter** bitmap_its = (po_iter**)po_xcalloc((size_t)n + 1, sizeof(po_iter*));
  po_iter** scan_its = (po_iter**)po_xcalloc((size_t)n + 1, sizeof(po_iter*));
  po_iter** remaining = (po_iter**)po_xcalloc((size_t)n + 2, sizeof(po_iter*));
  int n_sorted = 0, n_bitmap = 0, n_scan = 0, n_rem = 0;
  cs->scan_based = (po_docidset**)po_xcalloc((size_t)n + 1, sizeof(po_docidset*));
  cs->n_scan_based = 0;
  cs->entries_non_scan = 0;
  for (int i = 0; i < n; i++) {
    po_docidset* ds = cs->sets[i];
    po_iter* it = ds->iterator(ds);
    all[i] = it;
    if (it->kind == PO_IT_SORTED) {
      sorted_its[n_sorted++] = it;
      cs->entries_non_scan += ds->num_entries_scanned(ds);
    } else if (it->kind == PO_IT_BITMAP || it->kind == PO_IT_RANGELESS_BITMAP) {
      bitmap_its[n_bitmap++] = it;
      cs->entries_non_scan += ds->num_entries_scanned(ds);
    } else if (it->kind == PO_IT_SCAN) {
      scan_its[n_scan++] = it;
      cs->scan_based[cs->n_scan_based++] = ds;
    } else {
      remaining[n_rem++] = it;
      cs->scan_based[cs->n_scan_based++] = ds;
    }
  }
  /* bitmaps: lowest cardinality first (:110); Java's List.sort is stable, as is this insertion sort */
  for (int i = 1; i < n_bitmap; i++) {
    po_iter* key = bitmap_its[i];
    int j = i - 1;
    while (j >= 0 && cmp_bitmap_card(&bitmap_its[j], &key) > 0) {
      bitmap_its[j + 1] = bitmap_its[j];
      j--;
    }
    bitmap_its[j + 1] = key;
  }
  int n_index = n_sorted + n_bitmap;
  po_iter* result;
  if ((n_index > 0 && n_scan > 0) || n_index > 1) {
    po_bitmap* doc_ids;
    if (n_sorted > 0) {
      doc_ids = po_bitmap_new(cs->num_docs);
      /* SortedRangeIntersection.intersectSortedRangeSets == intersection of the range sets */
      ranges* r0 = ((sorted_it_state*)sorted_its[0]->state)->r;
      for (int k = 0; k < r0->n; k++) po_bitmap_add_range(doc_ids, r0->lo[k], (int64_t)r0->hi[k] + 1);
      for (int s = 1; s < n_sorted; s++) {
        ranges* rs = ((sorted_it_state*)sorted_its[s]->state)->r;
        po_bitmap* tmp = po_bitmap_new(cs->num_docs);
        for (int k = 0; k < rs->n; k++) po_bitmap_add_range(tmp, rs->lo[k], (int64_t)rs->hi[k] + 1);
        po_bitmap_and(doc_ids, tmp);
        po_bitmap_free(tmp);
      }
      for (int b = 0; b < n_bitmap; b++) po_bitmap_and(doc_ids, ((bitmap_it_state*)bitmap_its[b]->state)->doc_ids);
    } else {
      doc_ids = po_bitmap_clone(((bitmap_it_state*)bitmap_its[0]->state)->doc_ids);
      for (int b = 1; b < n_bitmap; b++) po_bitmap_and(doc_ids, ((bitmap_it_state*)bitmap_its[b]->state)->doc_ids);
    }
    for (int s = 0; s < n_scan; s++) {
      po_bitmap* next;
      if (po_bitmap_next_set(doc_ids, 0) < 0) next = po_bitmap_new(cs->num_docs); /* docIds.isEmpty() */
      else next = ((scan_state*)scan_its[s]->state)->col->is_mv ? mvscan_apply_and(scan_its[s], doc_ids) : scan_apply_and(scan_its[s], doc_ids);
      po_bitmap_free(doc_ids);
      doc_ids = next;
    }
    po_iter* rangeless = bitmap_iter_new(doc_ids, cs->num_docs, 1);
    if (n_rem == 0) {
      result = rangeless;
    } else {
      and_it_state* st = (and_it_state*)po_xcalloc(1, sizeof(*st));
      st->n = n_rem + 1;
      st->its = (po_iter**)po_xmalloc(sizeof(po_iter*) * (size_t)st->n);
      st->its[0] = rangeless;
      for (int i = 0; i < n_rem; i++) st->its[i + 1] = remaining[i];
      result = iter_new(PO_IT_AND, and_next, and_advance, st);
    }
  } else {
    and_it_state* st = (and_it_state*)po_xcalloc(1, sizeof(*st));
    st->n = n;
    st->its = all;
    all = NULL;
    result = iter_new(PO_IT_AND, and_next, and_advance, st);
  }
  free(all);
  free(sorted_its);
  free(bitmap_its);
  free(scan_its);
  free(remaining);
  return result;
}

static int64_t compound_entries(po_docidset* set) {
  compound_state* cs = (compound_state*)set->state;
  int64_t t = cs->entries_non_scan;
  for (int i = 0; i < cs->n_scan_based; i++) t += cs->scan_based[i]->num_entries_scanned(cs->scan_based[i]);
  return t;
}

static po_docidset* compoundset_new(int kind, po_iter* (*iterator)(po_docidset*), int n, po_docidset** sets,
                                    int32_t num_docs) {
  compound_state* cs = (compound_state*)po_xcalloc(1, sizeof(*cs));
  cs->n = n;
  cs->sets = (po_docidset**)po_xmalloc(sizeof(po_docidset*) * (size_t)(n + 1));
  memcpy(cs->sets, sets, sizeof(po_docidset*) * (size_t)n);
  cs->num_docs = num_docs;
  return set_new(kind, iterator, compound_entries, cs);
}

/* ---- OrDocIdSet, core/operator/docidsets/OrDocIdSet.java:58-125 ----------------------------------------------------------- */
static po_iter* orset_iterator(po_docidset* set) {
  compound_state* cs = (compound_state*)set->state;
  int n = cs->n;
  po_iter** all = (po_iter**)po_xcalloc((size_t)n + 1, sizeof(po_iter*));
  po_iter** sorted_its = (po_iter**)po_xcalloc((size_t)n + 1, sizeof(po_iter*));
  po_iter** bitmap_its = (po_iter**)po_xcalloc((size_t)n + 1, sizeof(po_iter*));
  po_iter** remaining = (po_iter**)po_xcalloc((size_t)n + 2, sizeof(po_iter*));
  int n_sorted = 0, n_bitmap = 0, n_rem = 0;
  cs->scan_based = (po_docidset**)po_xcalloc((size_t)n + 1, sizeof(po_docidset*));
  cs->n_scan_based = 0;
  cs->entries_non_scan = 0;
  for (int i = 0; i < n; i++) {
    po_docidset* ds = cs->sets[i];
    po_iter* it = ds->iterator(ds);
    all[i] = it;
    if (it->kind == PO_IT_SORTED) {
      sorted_its[n_sorted++] = it;
      cs->entries_non_scan += ds->num_entries_scanned(ds);
    } else if (it->kind == PO_IT_BITMAP || it->kind == PO_IT_RANGELESS_BITMAP) {
      bitmap_its[n_bitmap++] = it;   /* see the deviation note in the file header */
      cs->entries_non_scan += ds->num_entries_scanned(ds);
    } else {
      remaining[n_rem++] = it;
      cs->scan_based[cs->n_scan_based++] = ds;
    }
  }
  po_iter* result;
  if (n_sorted > 1) {   /* numSortedDocIdIterators + numBitmapBasedDocIdIterators > 1 with the latter always 0 (see the file header) */
    po_bitmap* doc_ids = po_bitmap_new(cs->num_docs);
    for (int s = 0; s < n_sorted; s++) {
      ranges* r = ((sorted_it_state*)sorted_its[s]->state)->r;
      for (int k = 0; k < r->n; k++) po_bitmap_add_range(doc_ids, r->lo[k], (int64_t)r->hi[k] + 1);
    }
    for (int b = 0; b < n_bitmap; b++) po_bitmap_or(doc_ids, ((bitmap_it_state*)bitmap_its[b]->state)->doc_ids);
    po_iter* bit = bitmap_iter_new(doc_ids, cs->num_docs, 0);
    if (n_rem == 0) {
      result = bit;
    } else {
      po_iter** its = (po_iter**)po_xmalloc(sizeof(po_iter*) * (size_t)(n_rem + 1));
      its[0] = bit;
      for (int i = 0; i < n_rem; i++) its[i + 1] = remaining[i];
      result = or_iter_new(n_rem + 1, its);
      free(its);
    }
  } else {
    result = or_iter_new(n, all);
  }
  free(all);
  free(sorted_its);
  free(bitmap_its);
  free(remaining);
  return result;
}

/* ---- NotDocIdSet ---------------------------------------------------------------------------------------------------------- */
typedef struct notset_state { po_docidset* child; int32_t num_docs; } notset_state;
static po_iter* notset_iterator(po_docidset* set) {
  notset_state* ns = (notset_state*)set->state;
  not_it_state* st = (not_it_state*)po_xcalloc(1, sizeof(*st));
  st->child = ns->child->iterator(ns->child);
  st->next_doc_id = 0;
  int32_t cur = st->child->next(st->child);
  st->next_non_matching = (cur == PO_EOF) ? ns->num_docs : cur;
  st->num_docs = ns->num_docs;
  return iter_new(PO_IT_NOT, not_next, not_advance, st);
}
static int64_t notset_entries(po_docidset* set) {
  notset_state* ns = (notset_state*)set->state;
  return ns->child->num_entries_scanned(ns->child);
}
static po_docidset* notset_new(po_docidset* child, int32_t num_docs) {
  notset_state* ns = (notset_state*)po_xcalloc(1, sizeof(*ns));
  ns->child = child;
  ns->num_docs = num_docs;
  return set_new(PO_SET_NOT, notset_iterator, notset_entries, ns);
}

/* =====================================================================================================================
 * filter operators
 * ===================================================================================================================== */
po_filter_op* po_op_new(int kind, int32_t num_docs) {
  po_filter_op* op = (po_filter_op*)po_xcalloc(1, sizeof(*op));
  op->kind = kind;
  op->num_docs = num_docs;
  return op;
}
static int op_is_empty(const po_filter_op* op) { return op->kind == PO_OP_EMPTY; }
static int op_is_match_all(const po_filter_op* op) { return op->kind == PO_OP_MATCH_ALL; }

/* FilterOperatorUtils.DefaultImplementation#getLeafFilterOperator, core/operator/filter/FilterOperatorUtils.java:74-133 */
po_filter_op* po_leaf_filter_operator(po_pred_eval* eval, const po_column* col, int32_t num_docs) {
  if (eval->always_false) return po_op_new(PO_OP_EMPTY, num_docs);
  if (eval->always_true) return po_op_new(PO_OP_MATCH_ALL, num_docs);
  po_filter_op* op;
  int sorted_ok = col->is_sorted && col->has_dictionary;
  if (eval->pred_type == PG_PRED_RANGE) {
    /* range: Sorted > RangeIndex > Scan — the inverted index is NOT used for RANGE */
    op = po_op_new(sorted_ok ? PO_OP_SORTED : (col->range_idx ? PO_OP_RANGE_INDEX : PO_OP_SCAN), num_docs);
  } else {
    if (sorted_ok) op = po_op_new(PO_OP_SORTED, num_docs);
    else if (col->inv_len > 0) op = po_op_new(PO_OP_INVERTED, num_docs);
    /* RangeIndexBasedFilterOperator.canEvaluate (:58-63): EQ over an exact range index */
    else if (col->range_idx && eval->pred_type == PG_PRED_EQ) op = po_op_new(PO_OP_RANGE_INDEX, num_docs);
    else op = po_op_new(PO_OP_SCAN, num_docs);
  }
  op->eval = eval;
  op->col = col;
  return op;
}

/* priorities: core/operator/filter/PrioritizedFilterOperator.java:31-38, getPriority FilterOperatorUtils.java:213-252 */
static int op_priority(const po_filter_op* op) {
  switch (op->kind) {
    case PO_OP_SORTED: return 0;
    case PO_OP_INVERTED: return 100;
    case PO_OP_BITMAP: return 100;   /* BitmapBasedFilterOperator: MEDIUM_PRIORITY */
    case PO_OP_RANGE_INDEX: return 200;   /* LOW_PRIORITY, FilterOperatorUtils.java:224-230 */
    case PO_OP_AND: return 300;
    case PO_OP_OR: return 400;
    case PO_OP_NOT: return op_priority(op->children[0]);
    case PO_OP_SCAN: return op->col && op->col->is_mv ? 550 : 500;   /* getScanBasedFilterPriority :253-265: multi-value scans last */
    default: return 10000;
  }
}

po_filter_op* po_and_filter_operator(int n, po_filter_op** ops, int32_t num_docs) { /* :136-158 */
  po_filter_op** ch = (po_filter_op**)po_xcalloc((size_t)n + 1, sizeof(po_filter_op*));
  int m = 0;
  for (int i = 0; i < n; i++) {
    if (op_is_empty(ops[i])) { free(ch); return po_op_new(PO_OP_EMPTY, num_docs); }
    if (!op_is_match_all(ops[i])) ch[m++] = ops[i];
  }
  if (m == 0) { free(ch); return po_op_new(PO_OP_MATCH_ALL, num_docs); }
  if (m == 1) { po_filter_op* r = ch[0]; free(ch); return r; }
  /* reorderAndFilterChildOperators: List.sort by priority (stable) */
  for (int i = 1; i < m; i++) {
    po_filter_op* key = ch[i];
    int kp = op_priority(key);
    int j = i - 1;
    while (j >= 0 && op_priority(ch[j]) > kp) { ch[j + 1] = ch[j]; j--; }
    ch[j + 1] = key;
  }
  po_filter_op* op = po_op_new(PO_OP_AND, num_docs);
  op->n_children = m;
  op->children = ch;
  return op;
}

po_filter_op* po_or_filter_operator(int n, po_filter_op** ops, int32_t num_docs) { /* :161-183 */
  po_filter_op** ch = (po_filter_op**)po_xcalloc((size_t)n + 1, sizeof(po_filter_op*));
  int m = 0;
  for (int i = 0; i < n; i++) {
    if (op_is_match_all(ops[i])) { free(ch); return po_op_new(PO_OP_MATCH_ALL, num_docs); }
    if (!op_is_empty(ops[i])) ch[m++] = ops[i];
  }
  if (m == 0) { free(ch); return po_op_new(PO_OP_EMPTY, num_docs); }
  if (m == 1) { po_filter_op* r = ch[0]; free(ch); return r; }
  po_filter_op* op = po_op_new(PO_OP_OR, num_docs);
  op->n_children = m;
  op->children = ch;
  return op;
}

po_filter_op* po_not_filter_operator(po_filter_op* child, int32_t num_docs) { /* :186-196 */
  if (op_is_match_all(child)) return po_op_new(PO_OP_EMPTY, num_docs);
  if (op_is_empty(child)) return po_op_new(PO_OP_MATCH_ALL, num_docs);
  po_filter_op* op = po_op_new(PO_OP_NOT, num_docs);
  op->n_children = 1;
  op->children = (po_filter_op**)po_xcalloc(1, sizeof(po_filter_op*));
  op->children[0] = child;
  return op;
}

/* FilterPlanNode#constructPhysicalOperator, core/plan/FilterPlanNode.java:195-320 */
static void set_null_handling(po_filter_op* op, int on) {   /* every operator of the tree is built with the query's flag */
  if (!op) return;
  op->null_handling = on;
  for (int i = 0; i < op->n_children; i++) set_null_handling(op->children[i], on);
}

static po_filter_op* construct_physical_operator(po_segment* seg, const pg_filter_node* f, int32_t num_docs, int nh) {
  switch (f->type) {
    case PG_FILTER_AND: {
      po_filter_op** ch = (po_filter_op**)po_xcalloc((size_t)f->n_children + 1, sizeof(po_filter_op*));
      int m = 0;
      for (int i = 0; i < f->n_children; i++) {
        po_filter_op* c = construct_physical_operator(seg, &f->children[i], num_docs, nh);
        if (!c) { free(ch); return NULL; }
        if (op_is_empty(c)) { free(ch); return po_op_new(PO_OP_EMPTY, num_docs); }
        if (!op_is_match_all(c)) ch[m++] = c;
      }
      po_filter_op* r = po_and_filter_operator(m, ch, num_docs);
      free(ch);
      return r;
    }
    case PG_FILTER_OR: {
      po_filter_op** ch = (po_filter_op**)po_xcalloc((size_t)f->n_children + 1, sizeof(po_filter_op*));
      int m = 0;
      for (int i = 0; i < f->n_children; i++) {
        po_filter_op* c = construct_physical_operator(seg, &f->children[i], num_docs, nh);
        if (!c) { free(ch); return NULL; }
        if (op_is_match_all(c)) { free(ch); return po_op_new(PO_OP_MATCH_ALL, num_docs); }
        if (!op_is_empty(c)) ch[m++] = c;
      }
      po_filter_op* r = po_or_filter_operator(m, ch, num_docs);
      free(ch);
      return r;
    }
    case PG_FILTER_NOT: {
      po_filter_op* c = construct_physical_operator(seg, &f->children[0], num_docs, nh);
      if (!c) return NULL;
      return po_not_filter_operator(c, num_docs);
    }
    case PG_FILTER_PREDICATE: {
      po_column* col = po_segment_column(seg, f->column);
      if (!col) {
        po_set_error("column not found: %s", f->column ? f->column : "(null)");
        return NULL;
      }
      if (f->predicate_type == PG_PRED_IS_NULL || f->predicate_type == PG_PRED_IS_NOT_NULL) { /* :298-312 */
        const int not_null = f->predicate_type == PG_PRED_IS_NOT_NULL;
        if (!col->null_bitmap) return po_op_new(not_null ? PO_OP_MATCH_ALL : PO_OP_EMPTY, num_docs);
        po_filter_op* op = po_op_new(PO_OP_BITMAP, num_docs);
        op->bitmap = po_bitmap_clone(col->null_bitmap);
        op->bitmap_exclusive = not_null;
        return op;
      }
      po_pred_eval* eval = po_pred_eval_create(f, col);
      if (!eval) return NULL;
      /* FilterOperatorUtils.java:78-88: an always-true predicate under null handling matches the docs that hold a value */
      if (nh && eval->always_true && !eval->always_false && col->null_bitmap && po_bitmap_cardinality(col->null_bitmap) > 0) {
        po_filter_op* op = po_op_new(PO_OP_BITMAP, num_docs);
        op->bitmap = po_bitmap_clone(col->null_bitmap);
        op->bitmap_exclusive = 1;
        return op;
      }
      return po_leaf_filter_operator(eval, col, num_docs);
    }
    case PG_FILTER_CONSTANT_TRUE: return po_op_new(PO_OP_MATCH_ALL, num_docs);
    case PG_FILTER_CONSTANT_FALSE: return po_op_new(PO_OP_EMPTY, num_docs);
    default:
      po_set_error("bad filter node type %d", f->type);
      return NULL;
  }
}

po_filter_op* po_filter_plan(po_segment* seg, const pg_filter_node* filter, int null_handling) { /* FilterPlanNode.run :88-106 */
  po_filter_op* valid = NULL;
  if (seg->queryable_doc_ids) {
    valid = po_op_new(PO_OP_BITMAP, seg->total_docs);
    valid->bitmap = po_bitmap_clone(seg->queryable_doc_ids);
  }
  if (!filter) return valid ? valid : po_op_new(PO_OP_MATCH_ALL, seg->total_docs);
  po_filter_op* op = construct_physical_operator(seg, filter, seg->total_docs, null_handling);
  if (op && valid) {
    po_filter_op* both[2] = {op, valid};
    op = po_and_filter_operator(2, both, seg->total_docs);
  }
  set_null_handling(op, null_handling);
  return op;
}

/* ---- getTrues / getFalses -------------------------------------------------------------------------------------------------- */
static po_docidset* op_get_falses(po_filter_op* op);

static ranges* ranges_new(int cap) {
  ranges* r = (ranges*)po_xcalloc(1, sizeof(*r));
  r->lo = (int32_t*)po_xcalloc((size_t)cap + 2, sizeof(int32_t));
  r->hi = (int32_t*)po_xcalloc((size_t)cap + 2, sizeof(int32_t));
  return r;
}

/* SortedIndexBasedFilterOperator#getTrues, core/operator/filter/SortedIndexBasedFilterOperator.java:57-130 */
static po_docidset* sorted_get_trues(po_filter_op* op) {
  const po_pred_eval* e = op->eval;
  const po_column* c = op->col;
  ranges* out;
  if (e->is_range) {
    int32_t s, e1, s2, e2;
    po_sorted_get_doc_ids(c, e->start_dict_id, &s, &e1);
    po_sorted_get_doc_ids(c, e->end_dict_id - 1, &s2, &e2);
    out = ranges_new(1);
    out->n = 1;
    out->lo[0] = s;
    out->hi[0] = e2;
    return set_new(PO_SET_SORTED, sortedset_iterator, zero_entries, out);
  }
  int exclusive = e->exclusive;
  const int32_t* ids = exclusive ? e->non_matching_dict_ids : e->matching_dict_ids;
  int n = exclusive ? e->n_non_matching : e->n_matching;
  /* merge adjacent ranges (dictIds ascending) */
  ranges* r = ranges_new(n);
  int32_t ls, le;
  po_sorted_get_doc_ids(c, ids[0], &ls, &le);
  for (int i = 1; i < n; i++) {
    int32_t s, e1;
    po_sorted_get_doc_ids(c, ids[i], &s, &e1);
    if (s == le + 1) {
      le = e1;
    } else {
      r->lo[r->n] = ls; r->hi[r->n] = le; r->n++;
      ls = s; le = e1;
    }
  }
  r->lo[r->n] = ls; r->hi[r->n] = le; r->n++;
  if (!exclusive) return set_new(PO_SET_SORTED, sortedset_iterator, zero_entries, r);
  out = ranges_new(r->n + 1);
  if (r->lo[0] > 0) { out->lo[out->n] = 0; out->hi[out->n] = r->lo[0] - 1; out->n++; }
  for (int i = 0; i < r->n - 1; i++) { out->lo[out->n] = r->hi[i] + 1; out->hi[out->n] = r->lo[i + 1] - 1; out->n++; }
  if (r->hi[r->n - 1] < op->num_docs - 1) { out->lo[out->n] = r->hi[r->n - 1] + 1; out->hi[out->n] = op->num_docs - 1; out->n++; }
  free(r->lo); free(r->hi); free(r);
  return set_new(PO_SET_SORTED, sortedset_iterator, zero_entries, out);
}

/* InvertedIndexFilterOperator#getNextBlockWithoutNullHandling, core/operator/filter/InvertedIndexFilterOperator.java:60-96 */
static po_docidset* inverted_get_trues(po_filter_op* op) {
  const po_pred_eval* e = op->eval;
  const int32_t* ids = e->exclusive ? e->non_matching_dict_ids : e->matching_dict_ids;
  int n = e->exclusive ? e->n_non_matching : e->n_matching;
  if (n == 0) return emptyset_new();
  po_bitmap* b = po_bitmap_new(op->num_docs);
  for (int i = 0; i < n; i++)
    if (po_inv_get_doc_ids_or(op->col, ids[i], b)) { po_bitmap_free(b); return NULL; }
  if (e->exclusive) po_bitmap_flip(b, 0, op->num_docs);
  return bitmapset_new(b, op->num_docs);
}

/* BaseColumnFilterOperator (Scan / Inverted / Sorted / RangeIndex leaves), core/operator/filter/BaseColumnFilterOperator.java:45-72: under
 * null handling the docs whose value is null are neither true nor false */
static int is_column_leaf(const po_filter_op* op) {
  return op->kind == PO_OP_SCAN || op->kind == PO_OP_INVERTED || op->kind == PO_OP_SORTED || op->kind == PO_OP_RANGE_INDEX;
}
static const po_bitmap* op_null_bitmap(const po_filter_op* op) {   /* getNullBitmap: non-null and non-empty, else none */
  if (!is_column_leaf(op) || !op->col) return NULL;
  const po_column* c = op->col;
  return c->null_bitmap && po_bitmap_cardinality(c->null_bitmap) > 0 ? c->null_bitmap : NULL;
}
static po_docidset* op_get_nulls(po_filter_op* op) {   /* getNulls: BaseFilterOperator.java:98-100 (empty), BaseColumnFilterOperator.java:56-64 */
  const po_bitmap* nb = op_null_bitmap(op);
  return nb ? bitmapset_new(po_bitmap_clone(nb), op->num_docs) : emptyset_new();
}
static po_docidset* trues_without_null_handling(po_filter_op* op);

po_docidset* po_filter_get_trues(po_filter_op* op) {
  const po_bitmap* nb = op->null_handling ? op_null_bitmap(op) : NULL;
  if (nb) {   /* excludeNulls: AndDocIdSet(block, flip(nullBitmap)) */
    po_docidset* sets[2];
    sets[0] = trues_without_null_handling(op);
    if (!sets[0]) return NULL;
    po_bitmap* b = po_bitmap_clone(nb);
    po_bitmap_flip(b, 0, op->num_docs);
    sets[1] = bitmapset_new(b, op->num_docs);
    return compoundset_new(PO_SET_AND, andset_iterator, 2, sets, op->num_docs);
  }
  return trues_without_null_handling(op);
}

static po_docidset* trues_without_null_handling(po_filter_op* op) {
  switch (op->kind) {
    case PO_OP_EMPTY: return emptyset_new();
    case PO_OP_MATCH_ALL: return matchallset_new(op->num_docs);
    case PO_OP_SCAN: return scanset_new(op->eval, op->col, op->num_docs);
    case PO_OP_INVERTED: return inverted_get_trues(op);
    case PO_OP_SORTED: return sorted_get_trues(op);
    case PO_OP_RANGE_INDEX: { /* RangeIndexBasedFilterOperator#getNextBlockWithoutNullHandling :75-82: exact index -> BitmapDocIdSet */
      po_bitmap* b = po_range_index_matching(op->col, op->eval, op->num_docs);
      if (!b) return NULL;
      return bitmapset_new(b, op->num_docs);
    }
    case PO_OP_BITMAP: { /* BitmapBasedFilterOperator#getTrues :42-49 */
      po_bitmap* b = po_bitmap_clone(op->bitmap);
      if (op->bitmap_exclusive) po_bitmap_flip(b, 0, op->num_docs);
      return bitmapset_new(b, op->num_docs);
    }
    case PO_OP_AND:
    case PO_OP_OR: {
      po_docidset** sets = (po_docidset**)po_xcalloc((size_t)op->n_children + 1, sizeof(po_docidset*));
      for (int i = 0; i < op->n_children; i++) {
        sets[i] = po_filter_get_trues(op->children[i]);
        if (!sets[i]) { free(sets); return NULL; }
      }
      po_docidset* r = (op->kind == PO_OP_AND)
                           ? compoundset_new(PO_SET_AND, andset_iterator, op->n_children, sets, op->num_docs)
                           : compoundset_new(PO_SET_OR, orset_iterator, op->n_children, sets, op->num_docs);
      free(sets);
      return r;
    }
    case PO_OP_NOT: /* NotFilterOperator#getTrues, core/operator/filter/NotFilterOperator.java:52-58 */
      if (op_is_empty(op->children[0])) return matchallset_new(op->num_docs);
      return op_get_falses(op->children[0]);
    default: return NULL;
  }
}

static po_docidset* op_get_falses(po_filter_op* op) {
  switch (op->kind) {
    case PO_OP_NOT: return po_filter_get_trues(op->children[0]);
    case PO_OP_AND: { /* AndFilterOperator#getFalses :63-90 (null handling off) */
      po_docidset** sets = (po_docidset**)po_xcalloc((size_t)op->n_children + 1, sizeof(po_docidset*));
      int m = 0;
      for (int i = 0; i < op->n_children; i++) {
        po_docidset* t = po_filter_get_trues(op->children[i]);
        if (!t) { free(sets); return NULL; }
        if (t->kind == PO_SET_EMPTY) { free(sets); return matchallset_new(op->num_docs); }
        if (t->kind == PO_SET_MATCH_ALL) continue;
        if (op->null_handling) {   /* :72-78: the child's nulls are not false either */
          po_docidset* nu = op_get_nulls(op->children[i]);
          if (nu->kind != PO_SET_EMPTY) {
            po_docidset* both[2] = {t, nu};
            sets[m++] = compoundset_new(PO_SET_OR, orset_iterator, 2, both, op->num_docs);
            continue;
          }
        }
        sets[m++] = t;
      }
      po_docidset* r;
      if (m == 0) r = emptyset_new();
      else if (m == 1) r = notset_new(sets[0], op->num_docs);
      else r = notset_new(compoundset_new(PO_SET_AND, andset_iterator, m, sets, op->num_docs), op->num_docs);
      free(sets);
      return r;
    }
    case PO_OP_OR: { /* OrFilterOperator#getFalses :60-87 */
      po_docidset** sets = (po_docidset**)po_xcalloc((size_t)op->n_children + 1, sizeof(po_docidset*));
      int m = 0;
      for (int i = 0; i < op->n_children; i++) {
        po_docidset* t = po_filter_get_trues(op->children[i]);
        if (!t) { free(sets); return NULL; }
        if (t->kind == PO_SET_MATCH_ALL) { free(sets); return emptyset_new(); }
        if (t->kind == PO_SET_EMPTY) continue;
        if (op->null_handling) {   /* :71-77 */
          po_docidset* nu = op_get_nulls(op->children[i]);
          if (nu->kind != PO_SET_EMPTY) {
            po_docidset* both[2] = {t, nu};
            sets[m++] = compoundset_new(PO_SET_OR, orset_iterator, 2, both, op->num_docs);
            continue;
          }
        }
        sets[m++] = t;
      }
      po_docidset* r;
      if (m == 0) r = matchallset_new(op->num_docs);
      else if (m == 1) r = notset_new(sets[0], op->num_docs);
      else r = notset_new(compoundset_new(PO_SET_OR, orset_iterator, m, sets, op->num_docs), op->num_docs);
      free(sets);
      return r;
    }
    default: { /* BaseFilterOperator#getFalses :96-112 */
      po_docidset* t = po_filter_get_trues(op);
      if (!t) return NULL;
      if (t->kind == PO_SET_MATCH_ALL) return emptyset_new();
      if (op->null_handling) {   /* :110-116 */
        po_docidset* nu = op_get_nulls(op);
        if (nu->kind != PO_SET_EMPTY) {
          po_docidset* both[2] = {t, nu};
          return notset_new(compoundset_new(PO_SET_OR, orset_iterator, 2, both, op->num_docs), op->num_docs);
        }
      }
      if (t->kind == PO_SET_EMPTY) return matchallset_new(op->num_docs);
      return notset_new(t, op->num_docs);
    }
  }
}

/* canOptimizeCount / canProduceBitmaps: Inverted / Sorted / MatchAll / Empty leaves; AND, OR iff every child can;
 * NOT iff its child can (BaseFilterOperator.java:56-82 and overrides). */
int po_filter_can_optimize_count(po_filter_op* op) {
  switch (op->kind) {
    case PO_OP_EMPTY:
    case PO_OP_MATCH_ALL:
    case PO_OP_INVERTED:
    case PO_OP_BITMAP:
    case PO_OP_RANGE_INDEX:
    case PO_OP_SORTED: return 1;
    case PO_OP_SCAN: return 0;
    default:
      for (int i = 0; i < op->n_children; i++)
        if (!po_filter_can_optimize_count(op->children[i])) return 0;
      return 1;
  }
}

/* getNumMatchingDocs: and/or cardinalities of the children's bitmaps (BitmapCollection) == cardinality of the set */
int32_t po_filter_num_matching_docs(po_filter_op* op) {
  /* getNumMatchingDocs / getBitmaps answer from the operators' bitmaps, which know no nulls (InvertedIndexFilterOperator.java:103-131,
   * AndFilterOperator.java:99-110): under null handling FastFilteredCountOperator still counts the docs whose stored default value matches */
  const int nh = op->null_handling;
  set_null_handling(op, 0);
  po_docidset* t = po_filter_get_trues(op);
  set_null_handling(op, nh);
  if (!t) return -1;
  po_iter* it = t->iterator(t);
  int32_t n = 0;
  while (it->next(it) != PO_EOF) n++;
  return n;
}
