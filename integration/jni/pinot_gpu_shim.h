/*
 * pinot_gpu_shim.h — the part of the JNI binding that is plain C: the wire format in which the Java side (NativeQuery.java) hands a
 * query over.  Java cannot build the pointer graph of pg_query, so it writes the query into a direct ByteBuffer as a flat
 * little-endian record; pgshim_query_parse() turns that record into a pg_query (+ pg_filter_node tree, strings) owned by one
 * allocation.  Compiled and tested without a JVM (integration/jni/jni_sequence_test.c); pinot_gpu_jni.c adds the JNIEnv glue.
 *
 * record := int32 magic 0x32514750 ("PGQ2"), int32 flags, int32 numGroupsLimit, int32 maxInitialResultHolderCapacity,
 *           int32 nGroupBy, int32 nAggregations, int32 hasFilter, int32 nOrderBy,
 *           int32 limit, int32 minSegmentGroupTrimSize,
 *           nGroupBy x string, nAggregations x { int32 function, int32 log2m, string column },
 *           nOrderBy x { int32 kind (pg_order_by_kind), int32 index, int32 ascending, int32 nullsLast }, [node]
 *           (PGQ1 had neither the order-by block nor limit / minSegmentGroupTrimSize: segment-level group trim, GroupByOperator.java:120-133)
 * node   := int32 type (pg_filter_type), int32 nChildren,
 *           type == PREDICATE: int32 predicateType, int32 nValues, string column, nValues x string, string lower, string upper,
 *                              int32 lowerInclusive, int32 upperInclusive
 *           then nChildren x node
 * string := int32 length (-1: null), bytes (UTF-8, no terminator), zero padding to a multiple of 4
 */
#ifndef PINOT_GPU_SHIM_H_
#define PINOT_GPU_SHIM_H_
#include <stddef.h>
#include <stdint.h>

#include "pinot_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PGSHIM_QUERY_MAGIC 0x32514750

typedef struct pgshim_query pgshim_query;
/* PG_OK or PG_ERR_INVALID_ARGUMENT (message in err, NUL terminated).  The record may be released after the call. */
int32_t pgshim_query_parse(const void* record, uint64_t size, pgshim_query** out_query, char* err, size_t err_cap);
const pg_query* pgshim_query_get(const pgshim_query* q);
void pgshim_query_free(pgshim_query* q);

#ifdef __cplusplus
}
#endif
#endif
