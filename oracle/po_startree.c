/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY (see po_internal.h).
 * Star-tree index: loading, the "is this query fit for the star-tree" test, the predicate map and
 * the tree traversal, each following one reference function (paths relative to /root/reference):
 *   load            pinot-segment-local/.../startree/v2/store/StarTreeLoaderUtils.java:53-128
 *                   pinot-segment-local/.../startree/OffHeapStarTree.java:38-85 (little-endian file), OffHeapStarTreeNode.java:30-155
 *   fit + predicates pinot-core/.../core/startree/StarTreeUtils.java:66-86 (pairs), :98-170 (extractPredicateEvaluatorsMap),
 *                   :179-211 (isFitForStarTree), :220-300 (OR clauses), :357-436 (createStarTreeBasedProjectOperator)
 *   traversal       pinot-core/.../core/startree/operator/StarTreeFilterOperator.java:155-200 (getFilterOperator),
 *                   :208-370 (traverseStarTree), :386-470 (getMatchingDictIds), CompositePredicateEvaluator.java:48-57
 * Pinned by the reference's own star-tree file (tests/golden/startree_airline, built by the reference's builder): the
 * traversal's answers on it equal a brute-force aggregation over its base docs (tests/test_startree.py).
 */
#include <stdio.h>

#include "po_internal.h"

int po_raw_parse_header(po_column* c);

#define STAR_ALL (-1) /* StarTreeNode.ALL */

static inline int32_t le32(const uint8_t* p) {
  return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}
static inline int64_t le64(const uint8_t* p) { return (int64_t)((uint64_t)(uint32_t)le32(p) | ((uint64_t)(uint32_t)le32(p + 4) << 32)); }

/* OffHeapStarTreeNode field offsets */
enum { N_DIMENSION_ID, N_DIMENSION_VALUE, N_START_DOC_ID, N_END_DOC_ID, N_AGGREGATED_DOC_ID, N_FIRST_CHILD_ID, N_LAST_CHILD_ID };
static inline int32_t node_get(const po_star_tree* st, int32_t node, int field) {
  return le32(st->nodes + ((int64_t)node * 7 + field) * 4);
}
static inline int node_is_leaf(const po_star_tree* st, int32_t node) { return node_get(st, node, N_FIRST_CHILD_ID) == -1; }
static inline int32_t node_num_children(const po_star_tree* st, int32_t node) {
  int32_t f = node_get(st, node, N_FIRST_CHILD_ID);
  return f == -1 ? 0 : node_get(st, node, N_LAST_CHILD_ID) - f + 1;
}
/* getChildForDimensionValue(ALL): the star child is the first child if it exists (children sorted by value) */
static int32_t node_star_child(const po_star_tree* st, int32_t node) {
  if (node_is_leaf(st, node)) return -1;
  int32_t f = node_get(st, node, N_FIRST_CHILD_ID);
  return node_get(st, f, N_DIMENSION_VALUE) == STAR_ALL ? f : -1;
}

static const char* pair_function_name(int32_t fn) { /* AggregationFunctionType#getName */
  switch (fn) {
    case PG_AGG_COUNT: return "count";
    case PG_AGG_SUM: return "sum";
    case PG_AGG_MIN: return "min";
    case PG_AGG_MAX: return "max";
    case PG_AGG_DISTINCTCOUNTHLL: return "distinctCountHLL";
    case PG_AGG_AVG: return "avg";
    case PG_AGG_MINMAXRANGE: return "minMaxRange";
    default: return NULL;
  }
}

int32_t po_star_tree_pair_index(const po_star_tree* st, int32_t function, const char* column) {
  const char* col = function == PG_AGG_COUNT ? "*" : column;
  if (!col) return -1;
  for (int i = 0; i < st->n_pairs; i++)
    if (st->pair_functions[i] == function && strcmp(st->pair_columns[i], col) == 0) return i;
  return -1;
}

/* ---- StarTreeLoaderUtils#loadStarTreeV2 -------------------------------------------------------------------------------------- */
int32_t po_segment_add_star_tree(void* segp, const pg_star_tree_desc* d) {
  po_segment* seg = (po_segment*)segp;
  if (!d || d->n_dimensions <= 0 || !d->star_tree.addr) { po_set_error("bad star-tree descriptor"); return PG_ERR_INVALID_ARGUMENT; }
  const uint8_t* t = (const uint8_t*)d->star_tree.addr;
  uint64_t len = d->star_tree.size;
  if (len < 24 || (uint64_t)le64(t) != 0xBADDA55B00DAD00DULL) { po_set_error("Invalid magic marker in star-tree data buffer"); return PG_ERR_INVALID_ARGUMENT; }
  if (le32(t + 8) != 1) { po_set_error("Invalid version in star-tree data buffer"); return PG_ERR_INVALID_ARGUMENT; }
  int32_t root_offset = le32(t + 12), n_dims = le32(t + 16);
  if (n_dims != d->n_dimensions) { po_set_error("star-tree has %d dimensions, descriptor %d", n_dims, d->n_dimensions); return PG_ERR_INVALID_ARGUMENT; }
  po_star_tree* st = (po_star_tree*)po_xcalloc(1, sizeof(*st));
  st->n_dims = n_dims;
  st->dims = (char**)po_xcalloc((size_t)n_dims, sizeof(char*));
  uint64_t off = 20;
  for (int i = 0; i < n_dims; i++) {
    if (off + 8 > len) { po_set_error("star-tree header truncated"); return PG_ERR_INVALID_ARGUMENT; }
    int32_t id = le32(t + off), nb = le32(t + off + 4);
    off += 8;
    if (id < 0 || id >= n_dims || nb < 0 || off + (uint64_t)nb > len) { po_set_error("star-tree header corrupt"); return PG_ERR_INVALID_ARGUMENT; }
    st->dims[id] = (char*)po_xcalloc((size_t)nb + 1, 1);
    memcpy(st->dims[id], t + off, (size_t)nb);
    off += (uint64_t)nb;
  }
  st->n_nodes = le32(t + off);
  off += 4;
  if ((int64_t)off != root_offset) { po_set_error("Error loading star-tree, header length mis-match"); return PG_ERR_INVALID_ARGUMENT; }
  if (off + (uint64_t)st->n_nodes * 28 != len) { po_set_error("Error loading star-tree, buffer size mis-match"); return PG_ERR_INVALID_ARGUMENT; }
  st->nodes = t + off;
  st->num_docs = d->num_docs;

  po_segment* sp = (po_segment*)po_xcalloc(1, sizeof(*sp));
  sp->name = strdup(seg->name);
  sp->total_docs = d->num_docs;
  sp->columns = (po_column**)po_xcalloc((size_t)(n_dims + d->n_pairs), sizeof(po_column*));
  for (int i = 0; i < n_dims; i++) {
    if (strcmp(d->dimensions[i], st->dims[i]) != 0) { po_set_error("dimension %d is %s in the tree, %s in the descriptor", i, st->dims[i], d->dimensions[i]); return PG_ERR_INVALID_ARGUMENT; }
    po_column* parent = po_segment_column(seg, st->dims[i]);
    if (!parent || !parent->has_dictionary) { po_set_error("star-tree dimension %s is not a dictionary column of the segment", st->dims[i]); return PG_ERR_INVALID_ARGUMENT; }
    po_column* c = (po_column*)po_xcalloc(1, sizeof(*c));
    *c = *parent;                                  /* same FieldSpec and Dictionary (StarTreeDataSource) */
    c->name = strdup(parent->name);
    c->fwd_encoding = PG_FWD_DICT_FIXED_BIT;       /* FixedBitSVForwardIndexReaderV2(buffer, numDocs, bitsPerElement) */
    c->is_sorted = 0;
    c->fwd = (const uint8_t*)d->dimension_forward_indexes[i].addr;
    c->fwd_len = d->dimension_forward_indexes[i].size;
    c->inv = NULL; c->inv_len = 0;
    c->num_docs = d->num_docs;
    if (c->fwd_len < ((uint64_t)d->num_docs * (uint64_t)c->bits_per_value + 7) / 8) { po_set_error("star-tree forward index of %s too short", c->name); return PG_ERR_INVALID_ARGUMENT; }
    sp->columns[sp->n_columns++] = c;
  }
  st->n_pairs = d->n_pairs;
  st->pair_functions = (int32_t*)po_xcalloc((size_t)d->n_pairs + 1, sizeof(int32_t));
  st->pair_columns = (char**)po_xcalloc((size_t)d->n_pairs + 1, sizeof(char*));
  st->pair_cols = (po_column**)po_xcalloc((size_t)d->n_pairs + 1, sizeof(po_column*));
  for (int i = 0; i < d->n_pairs; i++) {
    const pg_star_tree_pair* p = &d->pairs[i];
    const char* fname = pair_function_name(p->function);
    if (!fname) { po_set_error("unsupported star-tree function %d", p->function); return PG_ERR_UNSUPPORTED; }
    st->pair_functions[i] = p->function;
    st->pair_columns[i] = strdup(p->function == PG_AGG_COUNT ? "*" : p->column);
    po_column* c = (po_column*)po_xcalloc(1, sizeof(*c));
    size_t nl = strlen(fname) + 2 + strlen(st->pair_columns[i]) + 1;
    c->name = (char*)po_xcalloc(nl, 1);
    snprintf(c->name, nl, "%s__%s", fname, st->pair_columns[i]);   /* AggregationFunctionColumnPair#toColumnName */
    c->data_type = p->data_type;
    c->fwd_encoding = PG_FWD_RAW_FIXED_BYTE_CHUNK;
    c->fwd = (const uint8_t*)p->forward_index.addr;
    c->fwd_len = p->forward_index.size;
    c->num_docs = d->num_docs;
    if (po_raw_parse_header(c)) return PG_ERR_UNSUPPORTED;
    st->pair_cols[i] = c;
    sp->columns[sp->n_columns++] = c;
  }
  st->space = sp;
  seg->star_trees = (po_star_tree**)po_xrealloc(seg->star_trees, sizeof(po_star_tree*) * (size_t)(seg->n_star_trees + 1));
  seg->star_trees[seg->n_star_trees++] = st;
  return PG_OK;
}

/* ---- predicate map ----------------------------------------------------------------------------------------------------------- */
typedef struct composite_eval {   /* CompositePredicateEvaluator: predicate evaluators conjoined with OR, each maybe negated */
  int n;
  po_pred_eval** evals;
  int* negated;
  const pg_filter_node** preds;
} composite_eval;

typedef struct column_preds {     /* one entry of Map<String, List<CompositePredicateEvaluator>> */
  const char* column;
  int n;
  composite_eval* list;
} column_preds;

typedef struct pred_map {
  int n;
  column_preds* cols;             /* in first-insertion order */
} pred_map;

static column_preds* pred_map_get(pred_map* m, const char* column, int create) {
  for (int i = 0; i < m->n; i++) if (strcmp(m->cols[i].column, column) == 0) return &m->cols[i];
  if (!create) return NULL;
  m->cols = (column_preds*)po_xrealloc(m->cols, sizeof(column_preds) * (size_t)(m->n + 1));
  column_preds* c = &m->cols[m->n++];
  c->column = column; c->n = 0; c->list = NULL;
  return c;
}
static void column_preds_add(column_preds* c, composite_eval ce) {
  c->list = (composite_eval*)po_xrealloc(c->list, sizeof(composite_eval) * (size_t)(c->n + 1));
  c->list[c->n++] = ce;
}
static composite_eval composite_single(po_pred_eval* e, int negated, const pg_filter_node* p) {
  composite_eval ce;
  ce.n = 1;
  ce.evals = (po_pred_eval**)po_xcalloc(1, sizeof(void*));
  ce.negated = (int*)po_xcalloc(1, sizeof(int));
  ce.preds = (const pg_filter_node**)po_xcalloc(1, sizeof(void*));
  ce.evals[0] = e; ce.negated[0] = negated; ce.preds[0] = p;
  return ce;
}
static int composite_apply(const composite_eval* ce, int32_t dict_id) { /* CompositePredicateEvaluator#apply */
  for (int i = 0; i < ce->n; i++)
    if ((po_pred_apply_dict(ce->evals[i], dict_id) != 0) != (ce->negated[i] != 0)) return 1;
  return 0;
}

/* StarTreeUtils#getPredicateEvaluator: NULL when the predicate cannot be solved with the star-tree (no dictionary); *err on
 * a failed evaluator (literal parse error) */
static po_pred_eval* star_pred_eval(po_segment* seg, const pg_filter_node* p, int* err) {
  po_column* col = po_segment_column(seg, p->column);
  if (!col) { po_set_error("column not found: %s", p->column ? p->column : "(null)"); *err = PG_ERR_NOT_FOUND; return NULL; }
  if (!col->has_dictionary) return NULL;
  if (p->predicate_type == PG_PRED_IS_NULL || p->predicate_type == PG_PRED_IS_NOT_NULL) return NULL; /* StarTreeUtils.java:333-341 */
  po_pred_eval* e = po_pred_eval_create(p, col);
  if (!e) *err = PG_ERR_INVALID_ARGUMENT;
  return e;
}

/* unwraps NOT(NOT(...PREDICATE)) → predicate + parity; returns NULL for a nested AND/OR under NOT */
static const pg_filter_node* unwrap_not(const pg_filter_node* f, int* negated) {
  *negated = 0;
  while (f->type == PG_FILTER_NOT) { *negated = !*negated; f = &f->children[0]; }
  return f->type == PG_FILTER_PREDICATE ? f : NULL;
}

/* extractOrClausePredicates :262-300 */
static int or_clause_predicates(const pg_filter_node* f, const pg_filter_node*** preds, int** negs, int* n) {
  for (int i = 0; i < f->n_children; i++) {
    const pg_filter_node* c = &f->children[i];
    if (c->type == PG_FILTER_AND) return 0;
    if (c->type == PG_FILTER_OR) { if (!or_clause_predicates(c, preds, negs, n)) return 0; continue; }
    int neg = 0;
    const pg_filter_node* p = c;
    if (c->type == PG_FILTER_NOT) { p = unwrap_not(c, &neg); if (!p) return 0; }
    else if (c->type != PG_FILTER_PREDICATE) return 0;   /* constants are outside the reference's FilterContext.Type here */
    *preds = (const pg_filter_node**)po_xrealloc((void*)*preds, sizeof(void*) * (size_t)(*n + 1));
    *negs = (int*)po_xrealloc(*negs, sizeof(int) * (size_t)(*n + 1));
    (*preds)[*n] = p; (*negs)[*n] = neg; (*n)++;
  }
  return 1;
}

/* extractPredicateEvaluatorsMap :98-170.  1 = ok, 0 = the filter cannot be solved by the star-tree, <0 error */
static int extract_pred_map(po_segment* seg, const pg_filter_node* filter, pred_map* m) {
  if (!filter) return 1;
  int cap = 16, head = 0, tail = 0;
  const pg_filter_node** queue = (const pg_filter_node**)po_xcalloc((size_t)cap, sizeof(void*));
  queue[tail++] = filter;
  while (head < tail) {
    const pg_filter_node* f = queue[head++];
    switch (f->type) {
      case PG_FILTER_AND:
        for (int i = 0; i < f->n_children; i++) {
          if (tail == cap) { cap *= 2; queue = (const pg_filter_node**)po_xrealloc((void*)queue, sizeof(void*) * (size_t)cap); }
          queue[tail++] = &f->children[i];
        }
        break;
      case PG_FILTER_OR: { /* isOrClauseValidForStarTree :220-258 */
        const pg_filter_node** preds = NULL; int* negs = NULL; int n = 0;
        if (!or_clause_predicates(f, &preds, &negs, &n)) return 0;
        const char* identifier = NULL;
        composite_eval ce; ce.n = 0;
        ce.evals = (po_pred_eval**)po_xcalloc((size_t)n + 1, sizeof(void*));
        ce.negated = (int*)po_xcalloc((size_t)n + 1, sizeof(int));
        ce.preds = (const pg_filter_node**)po_xcalloc((size_t)n + 1, sizeof(void*));
        int always_true = 0;
        for (int i = 0; i < n; i++) {
          int err = 0;
          po_pred_eval* e = star_pred_eval(seg, preds[i], &err);
          if (err) return err;
          if (!e) return 0;
          int neg = negs[i];
          if ((e->always_true && !neg) || (e->always_false && neg)) { always_true = 1; break; }
          if ((e->always_true && neg) || (e->always_false && !neg)) continue;
          if (!identifier) identifier = preds[i]->column;
          else if (strcmp(identifier, preds[i]->column) != 0) return 0;
          ce.evals[ce.n] = e; ce.negated[ce.n] = neg; ce.preds[ce.n] = preds[i]; ce.n++;
        }
        if (always_true) break;              /* pair of nulls: always true, nothing to add */
        if (ce.n == 0) return 0;             /* all predicates always false: do not use the star-tree */
        column_preds_add(pred_map_get(m, identifier, 1), ce);
        break;
      }
      case PG_FILTER_NOT: {
        int neg = 0;
        const pg_filter_node* p = unwrap_not(f, &neg);
        if (!p) return 0;
        int err = 0;
        po_pred_eval* e = star_pred_eval(seg, p, &err);
        if (err) return err;
        if (!e) return 0;
        if ((e->always_true && neg) || (e->always_false && !neg)) return 0;
        if ((e->always_true && !neg) || (e->always_false && neg)) break;
        column_preds_add(pred_map_get(m, p->column, 1), composite_single(e, neg, p));
        break;
      }
      case PG_FILTER_PREDICATE: {
        int err = 0;
        po_pred_eval* e = star_pred_eval(seg, f, &err);
        if (err) return err;
        if (!e || e->always_false) return 0;
        if (!e->always_true) column_preds_add(pred_map_get(m, f->column, 1), composite_single(e, 0, f));
        break;
      }
      default: return 0;   /* constant filters never reach the star-tree path in the reference */
    }
  }
  return 1;
}

/* ---- java.util.HashSet<String> iteration order (the order the remaining predicate columns are turned into filters) ------- */
static uint32_t java_string_hash(const char* s) {
  uint32_t h = 0;
  for (const unsigned char* p = (const unsigned char*)s; *p; p++) h = 31u * h + *p;   /* ASCII column names */
  return h;
}
static void java_hashset_order(const char** names, int n) {
  int cap = 16;
  while (n > cap * 3 / 4) cap *= 2;
  for (int i = 1; i < n; i++) {   /* stable insertion sort by bucket */
    const char* key = names[i];
    uint32_t hk = java_string_hash(key); hk = (hk ^ (hk >> 16)) & (uint32_t)(cap - 1);
    int j = i - 1;
    while (j >= 0) {
      uint32_t hj = java_string_hash(names[j]); hj = (hj ^ (hj >> 16)) & (uint32_t)(cap - 1);
      if (hj <= hk) break;
      names[j + 1] = names[j];
      j--;
    }
    names[j + 1] = key;
  }
}

/* ---- traverseStarTree :208-370 ------------------------------------------------------------------------------------------------- */
typedef struct name_set { int n; const char** names; } name_set;
static int name_set_contains(const name_set* s, const char* x) {
  for (int i = 0; i < s->n; i++) if (strcmp(s->names[i], x) == 0) return 1;
  return 0;
}
static void name_set_remove(name_set* s, const char* x) {
  for (int i = 0; i < s->n; i++)
    if (strcmp(s->names[i], x) == 0) { s->names[i] = s->names[--s->n]; return; }
}
static name_set name_set_copy(const name_set* s) {
  name_set c; c.n = s->n;
  c.names = (const char**)po_xcalloc((size_t)s->n + 1, sizeof(char*));
  memcpy((void*)c.names, s->names, sizeof(char*) * (size_t)s->n);
  return c;
}

/* getMatchingDictIds(List<CompositePredicateEvaluator>) :386-470: the ids every composite evaluator accepts */
static uint8_t* matching_dict_ids(const column_preds* cp, int32_t cardinality, int32_t* n_matching) {
  uint8_t* match = (uint8_t*)po_xcalloc((size_t)cardinality + 1, 1);
  int32_t n = 0;
  for (int32_t d = 0; d < cardinality; d++) {
    int ok = 1;
    for (int i = 0; i < cp->n && ok; i++) ok = composite_apply(&cp->list[i], d);
    match[d] = (uint8_t)ok;
    n += ok;
  }
  *n_matching = n;
  return match;
}

/* returns the matched docs bitmap (NULL: a predicate column has no matching dictId → empty result); *remaining = the
 * predicate columns the tree could not resolve (they become scan filters over the star-tree docs) */
static po_bitmap* traverse_star_tree(po_segment* seg, const po_star_tree* st, const pred_map* pm, const name_set* group_by,
                                     name_set* remaining_out) {
  po_bitmap* docs = po_bitmap_new(st->num_docs);
  name_set remaining_pred; remaining_pred.n = pm->n;
  remaining_pred.names = (const char**)po_xcalloc((size_t)pm->n + 1, sizeof(char*));
  for (int i = 0; i < pm->n; i++) remaining_pred.names[i] = pm->cols[i].column;
  name_set remaining_gb = name_set_copy(group_by);
  int have_global = 0;
  name_set global_remaining; global_remaining.n = 0; global_remaining.names = NULL;
  int found_leaf = node_is_leaf(st, 0);
  if (found_leaf) { global_remaining = name_set_copy(&remaining_pred); have_global = 1; }

  int32_t* queue = (int32_t*)po_xmalloc(sizeof(int32_t) * (size_t)(st->n_nodes + 1));
  int head = 0, tail = 0;
  queue[tail++] = 0;
  int32_t current_dim = -1;
  uint8_t* matching = NULL;
  int32_t n_matching = 0;
  while (head < tail) {
    int32_t node = queue[head++];
    int32_t dim = node_get(st, node, N_DIMENSION_ID);
    if (dim > current_dim) {   /* previous level finished */
      name_set_remove(&remaining_pred, st->dims[dim]);
      name_set_remove(&remaining_gb, st->dims[dim]);
      if (found_leaf && !have_global) { global_remaining = name_set_copy(&remaining_pred); have_global = 1; }
      free(matching);
      matching = NULL;
      current_dim = dim;
    }
    if (remaining_pred.n == 0 && remaining_gb.n == 0) {   /* everything matched: the aggregated document */
      po_bitmap_add(docs, node_get(st, node, N_AGGREGATED_DOC_ID));
      continue;
    }
    if (node_is_leaf(st, node)) {
      po_bitmap_add_range(docs, node_get(st, node, N_START_DOC_ID), node_get(st, node, N_END_DOC_ID));
      continue;
    }
    const char* child_dim = st->dims[dim + 1];
    int32_t star = -1;
    if ((!have_global || !name_set_contains(&global_remaining, child_dim)) && !name_set_contains(&remaining_gb, child_dim))
      star = node_star_child(st, node);
    int32_t first = node_get(st, node, N_FIRST_CHILD_ID), last = node_get(st, node, N_LAST_CHILD_ID);
    if (name_set_contains(&remaining_pred, child_dim)) {
      if (!matching) {
        po_column* col = po_segment_column(seg, child_dim);
        const column_preds* cp = NULL;
        for (int i = 0; i < pm->n; i++) if (strcmp(pm->cols[i].column, child_dim) == 0) cp = &pm->cols[i];
        matching = matching_dict_ids(cp, col->cardinality, &n_matching);
        if (n_matching == 0) { po_bitmap_free(docs); free(queue); free(matching); return NULL; }
      }
      int32_t n_children = last - first + 1;
      /* (binary search and scan give the same child set; only the scan branch may substitute the star-node, and its
       *  condition numMatching >= numChildren - 1 implies the scan branch) */
      int use_star = 0;
      if (star >= 0 && n_matching >= n_children - 1) {
        int32_t hits = 0;
        for (int32_t c = first; c <= last; c++) {
          int32_t v = node_get(st, c, N_DIMENSION_VALUE);
          if (v != STAR_ALL && matching[v]) hits++;
        }
        use_star = hits == n_children - 1;
      }
      if (use_star) {
        queue[tail++] = star;
        found_leaf |= node_is_leaf(st, star);
      } else {
        for (int32_t c = first; c <= last; c++) {
          int32_t v = node_get(st, c, N_DIMENSION_VALUE);
          if (v != STAR_ALL && matching[v]) { queue[tail++] = c; found_leaf |= node_is_leaf(st, c); }
        }
      }
    } else if (star >= 0) {
      queue[tail++] = star;
      found_leaf |= node_is_leaf(st, star);
    } else {
      for (int32_t c = first; c <= last; c++)
        if (node_get(st, c, N_DIMENSION_VALUE) != STAR_ALL) { queue[tail++] = c; found_leaf |= node_is_leaf(st, c); }
    }
  }
  free(queue);
  free(matching);
  if (have_global) *remaining_out = global_remaining;
  else { remaining_out->n = 0; remaining_out->names = NULL; }
  return docs;
}

/* ---- createStarTreeBasedProjectOperator + StarTreeFilterOperator#getFilterOperator ------------------------------------------ */
int po_star_tree_plan(po_segment* seg, po_star_tree* st, const pg_query* q, po_filter_op** out_op) {
  /* extractAggregationFunctionPairs + isFitForStarTree: every aggregation's stored pair must be in the tree */
  for (int i = 0; i < q->n_aggregations; i++) {
    const pg_agg_spec* s = &q->aggregations[i];
    if (s->function == PG_AGG_DISTINCTCOUNT) return 0;                     /* no star-tree value aggregator */
    int32_t pi = po_star_tree_pair_index(st, s->function, s->column);
    if (pi < 0) return 0;
    if (s->function == PG_AGG_DISTINCTCOUNTHLL && st->num_docs > 0) {
      /* DistinctCountHLLAggregationFunction#canUseStarTree (:373-383): the tree's log2m must equal the query's */
      int32_t len = 0;
      const uint8_t* blob = po_raw_get_bytes(st->pair_cols[pi], 0, &len);
      if (len < 8 || (int32_t)po_be32(blob) != (s->log2m > 0 ? s->log2m : 8)) return 0;
    }
  }
  pred_map pm; pm.n = 0; pm.cols = NULL;
  int r = extract_pred_map(seg, q->filter, &pm);
  if (r <= 0) return r;
  name_set gb; gb.n = 0;
  gb.names = (const char**)po_xcalloc((size_t)q->n_group_by + 1, sizeof(char*));
  for (int j = 0; j < q->n_group_by; j++) {
    const char* g = q->group_by_columns[j];
    int is_dim = 0;
    for (int k = 0; k < st->n_dims; k++) is_dim |= strcmp(st->dims[k], g) == 0;
    if (!is_dim) return 0;
    if (!name_set_contains(&gb, g)) gb.names[gb.n++] = g;
  }
  for (int i = 0; i < pm.n; i++) {
    int is_dim = 0;
    for (int k = 0; k < st->n_dims; k++) is_dim |= strcmp(st->dims[k], pm.cols[i].column) == 0;
    if (!is_dim) return 0;
  }

  name_set remaining;
  po_bitmap* docs = traverse_star_tree(seg, st, &pm, &gb, &remaining);
  int32_t num_docs = st->num_docs;
  if (!docs) { *out_op = po_op_new(PO_OP_EMPTY, num_docs); return 1; }   /* EmptyFilterOperator */
  for (int i = 1; i < remaining.n; i++) {   /* deterministic order inside a hash bucket: by name */
    const char* key = remaining.names[i];
    int j = i - 1;
    while (j >= 0 && strcmp(remaining.names[j], key) > 0) { remaining.names[j + 1] = remaining.names[j]; j--; }
    remaining.names[j + 1] = key;
  }
  java_hashset_order(remaining.names, remaining.n);
  int cap = 1;
  for (int i = 0; i < remaining.n; i++) for (int k = 0; k < pm.n; k++) if (strcmp(pm.cols[k].column, remaining.names[i]) == 0) cap += pm.cols[k].n;
  po_filter_op** children = (po_filter_op**)po_xcalloc((size_t)cap + 1, sizeof(void*));
  int n = 0;
  po_filter_op* bm = po_op_new(PO_OP_BITMAP, num_docs);
  bm->bitmap = docs;
  children[n++] = bm;
  for (int i = 0; i < remaining.n; i++) {
    const column_preds* cp = NULL;
    for (int k = 0; k < pm.n; k++) if (strcmp(pm.cols[k].column, remaining.names[i]) == 0) cp = &pm.cols[k];
    const po_column* col = po_segment_column(st->space, remaining.names[i]);   /* star-tree DataSource */
    for (int c = 0; c < cp->n; c++) {
      const composite_eval* ce = &cp->list[c];
      po_filter_op** ors = (po_filter_op**)po_xcalloc((size_t)ce->n + 1, sizeof(void*));
      for (int e = 0; e < ce->n; e++) {
        po_filter_op* leaf = po_leaf_filter_operator(ce->evals[e], col, num_docs);
        ors[e] = ce->negated[e] ? po_not_filter_operator(leaf, num_docs) : leaf;
      }
      children[n++] = ce->n == 1 ? ors[0] : po_or_filter_operator(ce->n, ors, num_docs);
    }
  }
  *out_op = po_and_filter_operator(n, children, num_docs);
  return 1;
}
