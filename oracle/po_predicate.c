/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY (see po_internal.h).
 * Predicate evaluators: SURVEY.md §8a row a3.
 *   PredicateEvaluatorProvider.getPredicateEvaluator     core/operator/filter/predicate/PredicateEvaluatorProvider.java:45-96
 *   EqualsPredicateEvaluatorFactory                      .../EqualsPredicateEvaluatorFactory.java:95-140 (dict), raw variants
 *   NotEqualsPredicateEvaluatorFactory, InPredicateEvaluatorFactory.java:158-188, NotInPredicateEvaluatorFactory.java:158-207
 *   RangePredicateEvaluatorFactory.java:68-117 (raw bounds), :119-246 (sorted dictionary → [startDictId, endDictId))
 */
#include <math.h>
#include <stdio.h>

#include "po_internal.h"

int po_parse_int(const char* s, int32_t* out);
int po_parse_long(const char* s, int64_t* out);
int po_parse_float(const char* s, float* out);
int po_parse_double(const char* s, double* out);

static int cmp_i32(const void* a, const void* b) {
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}
static int cmp_i64(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}
static int cmp_f64(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

static void build_match_arrays(po_pred_eval* e, int32_t card) {
  /* getMatchingDictIds / getNonMatchingDictIds (BaseDictionaryBasedPredicateEvaluator), ascending */
  e->matching_dict_ids = (int32_t*)po_xmalloc(sizeof(int32_t) * (size_t)(card + 1));
  e->non_matching_dict_ids = (int32_t*)po_xmalloc(sizeof(int32_t) * (size_t)(card + 1));
  e->n_matching = e->n_non_matching = 0;
  for (int32_t d = 0; d < card; d++) {
    if (e->dict_id_match[d]) e->matching_dict_ids[e->n_matching++] = d;
    else e->non_matching_dict_ids[e->n_non_matching++] = d;
  }
}

static po_pred_eval* create_dict_based(const pg_filter_node* p, const po_column* col) {
  po_pred_eval* e = (po_pred_eval*)po_xcalloc(1, sizeof(po_pred_eval));
  int32_t card = col->cardinality;
  e->pred_type = p->predicate_type;
  e->dictionary_based = 1;
  e->data_type = col->data_type;
  e->dict_id_match = (uint8_t*)po_xcalloc((size_t)card + 1, 1);
  switch (p->predicate_type) {
    case PG_PRED_EQ: {
      int32_t idx = po_dict_insertion_index_of(col, p->values[0]);
      if (idx == INT32_MIN) goto fail;
      if (idx >= 0) {                       /* Dictionary#indexOf = normalizeIndex(insertionIndexOf) */
        e->dict_id_match[idx] = 1;
        if (card == 1) e->always_true = 1;
      } else {
        e->always_false = 1;
      }
      e->num_matching_items = 1;
      break;
    }
    case PG_PRED_NOT_EQ: {
      int32_t idx = po_dict_insertion_index_of(col, p->values[0]);
      if (idx == INT32_MIN) goto fail;
      memset(e->dict_id_match, 1, (size_t)card);
      e->exclusive = 1;
      if (idx >= 0) {
        e->dict_id_match[idx] = 0;
        if (card == 1) e->always_false = 1;
      } else {
        e->always_true = 1;
      }
      e->num_matching_items = -1;
      break;
    }
    case PG_PRED_IN:
    case PG_PRED_NOT_IN: {
      int32_t n_found = 0;
      for (int i = 0; i < p->n_values; i++) {   /* PredicateUtils.getDictIdSet: values absent from the dictionary are dropped */
        int32_t idx = po_dict_insertion_index_of(col, p->values[i]);
        if (idx == INT32_MIN) goto fail;
        if (idx >= 0 && !e->dict_id_match[idx]) {
          e->dict_id_match[idx] = 1;
          n_found++;
        }
      }
      if (p->predicate_type == PG_PRED_IN) {
        if (n_found == 0) e->always_false = 1;
        else if (n_found == card) e->always_true = 1;
        e->num_matching_items = n_found;
      } else {
        for (int32_t d = 0; d < card; d++) e->dict_id_match[d] = !e->dict_id_match[d];
        e->exclusive = 1;
        if (n_found == 0) e->always_true = 1;
        else if (n_found == card) e->always_false = 1;
        e->num_matching_items = -n_found;
      }
      break;
    }
    case PG_PRED_RANGE: {
      /* SortedDictionaryBasedRangePredicateEvaluator, RangePredicateEvaluatorFactory.java:126-167 */
      e->is_range = 1;
      if (strcmp(p->lower, PG_RANGE_UNBOUNDED) == 0) {
        e->start_dict_id = 0;
      } else {
        int32_t ins = po_dict_insertion_index_of(col, p->lower);
        if (ins == INT32_MIN) goto fail;
        if (ins < 0) e->start_dict_id = -(ins + 1);
        else e->start_dict_id = p->lower_inclusive ? ins : ins + 1;
      }
      if (strcmp(p->upper, PG_RANGE_UNBOUNDED) == 0) {
        e->end_dict_id = card;
      } else {
        int32_t ins = po_dict_insertion_index_of(col, p->upper);
        if (ins == INT32_MIN) goto fail;
        if (ins < 0) e->end_dict_id = -(ins + 1);
        else e->end_dict_id = p->upper_inclusive ? ins + 1 : ins;
      }
      int32_t n = e->end_dict_id - e->start_dict_id;
      if (n < 0) n = 0;
      e->num_matching_items = n;
      if (n == 0) e->always_false = 1;
      else if (n == card) e->always_true = 1;
      for (int32_t d = e->start_dict_id; d < e->end_dict_id; d++) e->dict_id_match[d] = 1;
      break;
    }
    default:
      po_set_error("unsupported predicate type %d", p->predicate_type);
      goto fail;
  }
  build_match_arrays(e, card);
  return e;
fail:
  po_pred_eval_free(e);
  return NULL;
}

static float next_up_f(float v) { return nextafterf(v, INFINITY); }
static float next_down_f(float v) { return nextafterf(v, -INFINITY); }

static char* po_xstrdup(const char* s) { const size_t n = strlen(s) + 1; char* d = (char*)po_xcalloc(n, 1); memcpy(d, s, n); return d; }
static po_pred_eval* create_raw_based(const pg_filter_node* p, const po_column* col) {
  po_pred_eval* e = (po_pred_eval*)po_xcalloc(1, sizeof(po_pred_eval));
  e->pred_type = p->predicate_type;
  e->data_type = col->data_type;
  e->num_matching_items = INT32_MIN;
  int t = col->data_type;
  if (t == PG_TYPE_STRING && !col->is_mv) {
    /* StringRawValueBased{Equals,NotEquals,In,NotIn,Range}PredicateEvaluator (EqualsPredicateEvaluatorFactory.java, InPredicateEvaluatorFactory.java,
     * RangePredicateEvaluatorFactory.java: value.equals / set.contains / String#compareTo against the bounds) */
    if (p->predicate_type == PG_PRED_RANGE) {
      if (strcmp(p->lower, PG_RANGE_UNBOUNDED) != 0) e->lo_s = po_xstrdup(p->lower);
      if (strcmp(p->upper, PG_RANGE_UNBOUNDED) != 0) e->hi_s = po_xstrdup(p->upper);
      e->lo_inc = p->lower_inclusive; e->hi_inc = p->upper_inclusive;
      return e;
    }
    if (p->predicate_type != PG_PRED_EQ && p->predicate_type != PG_PRED_NOT_EQ && p->predicate_type != PG_PRED_IN && p->predicate_type != PG_PRED_NOT_IN) {
      po_set_error("predicate type %d on raw STRING column %s", p->predicate_type, col->name);
      goto fail;
    }
    e->exclusive = p->predicate_type == PG_PRED_NOT_EQ || p->predicate_type == PG_PRED_NOT_IN;
    e->n_raw_s = p->n_values;
    e->raw_s = (char**)po_xcalloc((size_t)p->n_values + 1, sizeof(char*));
    for (int i = 0; i < p->n_values; i++) e->raw_s[i] = po_xstrdup(p->values[i]);
    e->num_matching_items = (p->predicate_type == PG_PRED_EQ) ? 1 : (p->predicate_type == PG_PRED_NOT_EQ) ? -1
                            : (p->predicate_type == PG_PRED_IN) ? p->n_values : -p->n_values;
    return e;
  }
  if (t != PG_TYPE_INT && t != PG_TYPE_LONG && t != PG_TYPE_FLOAT && t != PG_TYPE_DOUBLE) {
    po_set_error("raw predicate on column %s of type %d is outside the hot path", col->name, t);
    goto fail;
  }
  if (p->predicate_type == PG_PRED_RANGE) {
    /* newRawValueBasedEvaluator: unbounded => inclusive MIN/MAX of the type; exclusive => +-1 (nextUp/nextDown) */
    int lo_unb = strcmp(p->lower, PG_RANGE_UNBOUNDED) == 0, hi_unb = strcmp(p->upper, PG_RANGE_UNBOUNDED) == 0;
    int lo_inc = lo_unb || p->lower_inclusive, hi_inc = hi_unb || p->upper_inclusive;
    if (t == PG_TYPE_INT) {
      int32_t lo = INT32_MIN, hi = INT32_MAX;
      if (!lo_unb && po_parse_int(p->lower, &lo)) goto fail;
      if (!hi_unb && po_parse_int(p->upper, &hi)) goto fail;
      if (!lo_inc) { if (lo == INT32_MAX) { po_set_error("Invalid range"); goto fail; } lo += 1; }
      if (!hi_inc) { if (hi == INT32_MIN) { po_set_error("Invalid range"); goto fail; } hi -= 1; }
      e->lo_i = lo; e->hi_i = hi;
    } else if (t == PG_TYPE_LONG) {
      int64_t lo = INT64_MIN, hi = INT64_MAX;
      if (!lo_unb && po_parse_long(p->lower, &lo)) goto fail;
      if (!hi_unb && po_parse_long(p->upper, &hi)) goto fail;
      if (!lo_inc) { if (lo == INT64_MAX) { po_set_error("Invalid range"); goto fail; } lo += 1; }
      if (!hi_inc) { if (hi == INT64_MIN) { po_set_error("Invalid range"); goto fail; } hi -= 1; }
      e->lo_i = lo; e->hi_i = hi;
    } else if (t == PG_TYPE_FLOAT) {
      float lo = -INFINITY, hi = INFINITY;
      if (!lo_unb && po_parse_float(p->lower, &lo)) goto fail;
      if (!hi_unb && po_parse_float(p->upper, &hi)) goto fail;
      /* FloatRawValueBasedRangePredicateEvaluator (:449-456): checkArgument(nextUp(lower) > lower) / (nextDown(upper) < upper) —
       * false for an exclusive bound at its infinity and for NaN */
      if (!lo_inc) { const float n = next_up_f(lo); if (!(n > lo)) { po_set_error("Invalid range"); goto fail; } lo = n; }
      if (!hi_inc) { const float n = next_down_f(hi); if (!(n < hi)) { po_set_error("Invalid range"); goto fail; } hi = n; }
      e->lo_f = lo; e->hi_f = hi;
    } else {
      double lo = -INFINITY, hi = INFINITY;
      if (!lo_unb && po_parse_double(p->lower, &lo)) goto fail;
      if (!hi_unb && po_parse_double(p->upper, &hi)) goto fail;
      /* DoubleRawValueBasedRangePredicateEvaluator: the same checkArgument pair */
      if (!lo_inc) { const double n = nextafter(lo, INFINITY); if (!(n > lo)) { po_set_error("Invalid range"); goto fail; } lo = n; }
      if (!hi_inc) { const double n = nextafter(hi, -INFINITY); if (!(n < hi)) { po_set_error("Invalid range"); goto fail; } hi = n; }
      e->lo_d = lo; e->hi_d = hi;
    }
    return e;
  }
  /* EQ / NOT_EQ / IN / NOT_IN on raw values: value set membership */
  e->exclusive = (p->predicate_type == PG_PRED_NOT_EQ || p->predicate_type == PG_PRED_NOT_IN);
  e->n_raw_values = p->n_values;
  e->raw_i = (int64_t*)po_xcalloc((size_t)p->n_values + 1, 8);
  e->raw_d = (double*)po_xcalloc((size_t)p->n_values + 1, 8);
  for (int i = 0; i < p->n_values; i++) {
    if (t == PG_TYPE_INT) { int32_t v; if (po_parse_int(p->values[i], &v)) goto fail; e->raw_i[i] = v; }
    else if (t == PG_TYPE_LONG) { int64_t v; if (po_parse_long(p->values[i], &v)) goto fail; e->raw_i[i] = v; }
    else if (t == PG_TYPE_FLOAT) { float v; if (po_parse_float(p->values[i], &v)) goto fail; e->raw_d[i] = v; }
    else { double v; if (po_parse_double(p->values[i], &v)) goto fail; e->raw_d[i] = v; }
  }
  qsort(e->raw_i, (size_t)p->n_values, 8, cmp_i64);
  qsort(e->raw_d, (size_t)p->n_values, 8, cmp_f64);
  e->num_matching_items = (p->predicate_type == PG_PRED_EQ) ? 1 : (p->predicate_type == PG_PRED_NOT_EQ) ? -1
                          : (p->predicate_type == PG_PRED_IN) ? p->n_values : -p->n_values;
  return e;
fail:
  po_pred_eval_free(e);
  return NULL;
}

po_pred_eval* po_pred_eval_create(const pg_filter_node* p, const po_column* col) {
  /* PredicateEvaluatorProvider: dictionary != null → dictionary based, else raw value based */
  (void)cmp_i32;
  if (col->has_dictionary) return create_dict_based(p, col);
  return create_raw_based(p, col);
}

void po_pred_eval_free(po_pred_eval* e) {
  if (!e) return;
  free(e->matching_dict_ids);
  free(e->non_matching_dict_ids);
  free(e->dict_id_match);
  free(e->raw_i);
  free(e->raw_d);
  for (int i = 0; i < e->n_raw_s; i++) free(e->raw_s[i]);
  free(e->raw_s);
  free(e->lo_s);
  free(e->hi_s);
  free(e);
}

/* String.compareTo orders UTF-16 code units; UTF-8 byte order differs only where a supplementary character (lead byte F0..F4: surrogates
 * D800..DFFF) meets U+E000..U+FFFF (lead byte EE / EF), which sort behind it in UTF-16 */
int po_utf16_unit_order(const uint8_t* a, int32_t alen, const uint8_t* b, int32_t blen) {
  const int32_t n = alen < blen ? alen : blen;
  for (int32_t i = 0; i < n; i++) {
    if (a[i] == b[i]) continue;
    const int x = a[i] == 0xEE || a[i] == 0xEF ? a[i] + 0x10 : a[i], y = b[i] == 0xEE || b[i] == 0xEF ? b[i] + 0x10 : b[i];
    return x < y ? -1 : 1;
  }
  return alen < blen ? -1 : (alen > blen ? 1 : 0);
}
int po_pred_apply_string(const po_pred_eval* e, const uint8_t* v, int32_t len) {
  if (e->pred_type == PG_PRED_RANGE) {
    if (e->lo_s) { const int c = po_utf16_unit_order(v, len, (const uint8_t*)e->lo_s, (int32_t)strlen(e->lo_s)); if (c < 0 || (c == 0 && !e->lo_inc)) return 0; }
    if (e->hi_s) { const int c = po_utf16_unit_order(v, len, (const uint8_t*)e->hi_s, (int32_t)strlen(e->hi_s)); if (c > 0 || (c == 0 && !e->hi_inc)) return 0; }
    return 1;
  }
  int found = 0;
  for (int i = 0; i < e->n_raw_s && !found; i++) found = (int32_t)strlen(e->raw_s[i]) == len && memcmp(e->raw_s[i], v, (size_t)len) == 0;
  return e->exclusive ? !found : found;
}

int po_pred_apply_dict(const po_pred_eval* e, int32_t d) {
  if (e->is_range) return e->start_dict_id <= d && e->end_dict_id > d;
  return e->dict_id_match[d];
}

static int set_has_i(const po_pred_eval* e, int64_t v) {
  int lo = 0, hi = e->n_raw_values - 1;
  while (lo <= hi) {
    int m = (lo + hi) >> 1;
    if (e->raw_i[m] < v) lo = m + 1; else if (e->raw_i[m] > v) hi = m - 1; else return 1;
  }
  return 0;
}
static int set_has_d(const po_pred_eval* e, double v) {
  for (int i = 0; i < e->n_raw_values; i++)
    if (e->raw_d[i] == v) return 1;
  return 0;
}

int po_pred_apply_int(const po_pred_eval* e, int32_t v) {
  if (e->pred_type == PG_PRED_RANGE) return v >= e->lo_i && v <= e->hi_i;
  return set_has_i(e, v) ^ e->exclusive;
}
int po_pred_apply_long(const po_pred_eval* e, int64_t v) {
  if (e->pred_type == PG_PRED_RANGE) return v >= e->lo_i && v <= e->hi_i;
  return set_has_i(e, v) ^ e->exclusive;
}
int po_pred_apply_float(const po_pred_eval* e, float v) {
  if (e->pred_type == PG_PRED_RANGE) return v >= e->lo_f && v <= e->hi_f;
  return set_has_d(e, (double)v) ^ e->exclusive;
}
int po_pred_apply_double(const po_pred_eval* e, double v) {
  if (e->pred_type == PG_PRED_RANGE) return v >= e->lo_d && v <= e->hi_d;
  return set_has_d(e, v) ^ e->exclusive;
}
