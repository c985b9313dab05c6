/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY (see po_internal.h).
 *
 * HyperLogLog of com.clearspring.analytics:stream 2.9.8 (pom.xml:1411-1414; source NOT under /root/reference), as used
 * by DistinctCountHLLAggregationFunction (core/query/aggregation/function/DistinctCountHLLAggregationFunction.java:
 * 152-222 offer per value, :333-350 merge = addAll, :363-365 cardinality, :457-466 dictId bitmap → HLL) with the default
 * log2m = 8 (pinot-spi/.../utils/CommonConstants.java:117).  Restated from the published algorithm:
 *   offer(o):  x = MurmurHash.hash(o); j = x >>> (32-log2m); r = nlz((x << log2m) | (1 << (log2m-1)) + 1) + 1;
 *              reg[j] = max(reg[j], r)
 *   MurmurHash.hash(Integer/Long) = hashLong(value); Double → raw long bits; Float → raw int bits (sign-extended);
 *              String/byte[] → MurmurHash2(bytes, seed -1)   [string path: parity unpinned — no golden in the tree]
 *   cardinality(): alphaMM / sum(2^-reg); linear counting m*ln(m/zeros) when estimate <= 2.5 m; Math.round
 * Pinned by the reference goldens DISTINCTCOUNTHLL(column1/column3) = 5977 / 23825 / 1886 / 4492 on test_data-sv.avro
 * (InterSegmentAggregationSingleValueQueriesTest.java:261-274), reproduced in tests/test_oracle_goldens.py.
 */
#include <math.h>

#include "po_internal.h"

po_hll* po_hll_new(int32_t log2m) {
  po_hll* h = (po_hll*)po_xcalloc(1, sizeof(*h));
  h->log2m = log2m;
  h->m = 1 << log2m;
  h->regs = (uint8_t*)po_xcalloc((size_t)h->m, 1);
  return h;
}
void po_hll_free(po_hll* h) {
  if (!h) return;
  free(h->regs);
  free(h);
}

int32_t po_murmur_hash_long(int64_t data) {
  const uint32_t m = 0x5bd1e995u;
  uint32_t h = 0;
  uint32_t k = (uint32_t)(uint64_t)data * m;
  k ^= k >> 24;
  h ^= k * m;
  k = (uint32_t)((uint64_t)data >> 32) * m;
  k ^= k >> 24;
  h *= m;
  h ^= k * m;
  h ^= h >> 13;
  h *= m;
  h ^= h >> 15;
  return (int32_t)h;
}

int32_t po_murmur_hash_bytes(const uint8_t* data, int32_t length) {
  const uint32_t m = 0x5bd1e995u;
  uint32_t h = (uint32_t)(-1) ^ (uint32_t)length;
  int len4 = length >> 2;
  for (int i = 0; i < len4; i++) {
    int i4 = i << 2;
    /* sign-extending byte loads as in the Java code: only the top byte's sign matters and it is shifted out */
    uint32_t k = (uint32_t)data[i4] | ((uint32_t)data[i4 + 1] << 8) | ((uint32_t)data[i4 + 2] << 16) |
                 ((uint32_t)data[i4 + 3] << 24);
    k *= m;
    k ^= k >> 24;
    k *= m;
    h *= m;
    h ^= k;
  }
  int left = length - (len4 << 2);
  if (left != 0) {
    if (left >= 3) h ^= (uint32_t)((int32_t)(int8_t)data[length - 3] << 16);
    if (left >= 2) h ^= (uint32_t)((int32_t)(int8_t)data[length - 2] << 8);
    if (left >= 1) h ^= (uint32_t)(int32_t)(int8_t)data[length - 1];
    h *= m;
  }
  h ^= h >> 13;
  h *= m;
  h ^= h >> 15;
  return (int32_t)h;
}

void po_hll_offer_hash(po_hll* h, int32_t hash) {
  uint32_t x = (uint32_t)hash;
  uint32_t j = x >> (32 - h->log2m);
  uint32_t w = (x << h->log2m) | ((1u << (h->log2m - 1)) + 1u);
  int r = __builtin_clz(w) + 1;   /* w != 0 */
  if (h->regs[j] < r) h->regs[j] = (uint8_t)r;
}

void po_hll_offer_int(po_hll* h, int32_t v) { po_hll_offer_hash(h, po_murmur_hash_long((int64_t)v)); }
void po_hll_offer_long(po_hll* h, int64_t v) { po_hll_offer_hash(h, po_murmur_hash_long(v)); }
void po_hll_offer_float(po_hll* h, float v) {
  int32_t bits;
  memcpy(&bits, &v, 4);
  po_hll_offer_hash(h, po_murmur_hash_long((int64_t)bits));
}
void po_hll_offer_double(po_hll* h, double v) {
  int64_t bits;
  memcpy(&bits, &v, 8);
  po_hll_offer_hash(h, po_murmur_hash_long(bits));
}
void po_hll_offer_string(po_hll* h, const uint8_t* s, int32_t len) { po_hll_offer_hash(h, po_murmur_hash_bytes(s, len)); }

void po_hll_merge(po_hll* d, const po_hll* s) {
  for (int i = 0; i < d->m; i++)
    if (s->regs[i] > d->regs[i]) d->regs[i] = s->regs[i];
}

int64_t po_hll_cardinality(const po_hll* h) {
  int m = h->m;
  double alpha_mm;
  switch (h->log2m) {
    case 4: alpha_mm = 0.673 * m * m; break;
    case 5: alpha_mm = 0.697 * m * m; break;
    case 6: alpha_mm = 0.709 * m * m; break;
    default: alpha_mm = (0.7213 / (1 + 1.079 / m)) * m * m; break;
  }
  double sum = 0;
  double zeros = 0;
  for (int j = 0; j < m; j++) {
    int v = h->regs[j];
    sum += 1.0 / (double)(1ULL << v);
    if (v == 0) zeros++;
  }
  double estimate = alpha_mm * (1 / sum);
  if (estimate <= (5.0 / 2.0) * m) {
    /* linearCounting(m, V) = m * Math.log(m / V); no empty register: m / 0.0 = Infinity, Math.round(Infinity) = Long.MAX_VALUE — reachable
     * (every register 1 or 2 keeps the estimate under 2.5 m: a quarter of the 40-value sets at log2m 4) */
    if (zeros == 0) return INT64_MAX;
    return (int64_t)floor(m * log(m / zeros) + 0.5);   /* Math.round */
  }
  return (int64_t)floor(estimate + 0.5);
}

/* HyperLogLog.Builder.build(byte[]) via ObjectSerDeUtils.HYPER_LOG_LOG_SER_DE (core/common/ObjectSerDeUtils.java:733-767):
 * BE int log2m, BE int byte size, RegisterSet ints; register i = (word[i / 6] >>> (5 * (i % 6))) & 0x1f */
po_hll* po_hll_deserialize(const uint8_t* blob, int32_t len) {
  if (len < 8) return NULL;
  int32_t log2m = (int32_t)po_be32(blob), nbytes = (int32_t)po_be32(blob + 4);
  if (log2m < 1 || log2m > 30 || nbytes < 0 || 8 + nbytes > len) return NULL;
  po_hll* h = po_hll_new(log2m);
  int32_t n_words = nbytes / 4;
  for (int32_t i = 0; i < h->m; i++) {
    int32_t w = i / 6;
    if (w >= n_words) break;
    h->regs[i] = (uint8_t)((po_be32(blob + 8 + (int64_t)w * 4) >> (5 * (i % 6))) & 0x1f);
  }
  return h;
}
