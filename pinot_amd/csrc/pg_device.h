// Device-side plan layout shared by the host planner (pg_plan.cpp) and the HIP kernels (pg_kernels.hip).
//
// Execution model: a segment is cut into wave tiles of PG_WAVE_DOCS consecutive docIds.  One workgroup of
// PG_BLOCK threads (16 wavefronts) is resident per CU; its wavefronts walk wave tiles independently (no barrier in the
// main loop).  Per wave tile a wavefront (1) runs the compiled filter program over a register stack of match masks
// (one dword per lane = 32 docs per lane in "quad layout": lane L owns quads k*64+L, k = 0..7, of 4 consecutive docs),
// (2) optionally stores the resulting match words / popcounts, and (3) streams the group-by and metric columns of the
// tile, aggregating the matching docs into the workgroup's LDS accumulator table.  PG_TILE_DOCS (8 wave tiles) is
// only the padding / docId-expansion granule.
#pragma once
#include <stdint.h>

#define PG_TILE_DOCS 16384
#define PG_TILE_WORDS 256          // 64-bit words per tile
#define PG_TILE_QUADS 4096         // 4-doc quads per tile
#define PG_CHUNK_DOCS 65536        // RoaringBitmap container span
#define PG_TILES_PER_CHUNK 4
#define PG_WAVE_DOCS 2048          // docs per wave tile: 64 lanes x 8 quads x 4 docs
#define PG_WAVE_QUADS 512
#define PG_WTILES_PER_TILE 8
#define PG_WTILES_PER_CHUNK 32
#ifndef PG_WAVES_PER_BLOCK
#define PG_WAVES_PER_BLOCK 16
#endif
#define PG_MAX_STACK 6
#define PG_MAX_GROUP_COLS 8
#define PG_MAX_SRCS 8
#define PG_MAX_OPS 32
#define PG_MAX_STATS 16
#define PG_MAX_FAST_SCANS 4
#define PG_GENERIC_BLOCK 512      // interpreter kernel workgroup
#define PG_BLOCK (PG_WAVES_PER_BLOCK * 64)   // one 16-wave workgroup per CU: every wave shares one LDS accumulator table

// ---- filter program --------------------------------------------------------------------------------------------------
enum PgFOp : int32_t {
  PG_F_PUSH_POSTINGS = 0,   // push OR of the leaf's posting lists (xor valid mask when exclusive)
  PG_F_PUSH_RANGES = 1,     // push docId ranges (sorted index / match-all)
  PG_F_PUSH_SCAN = 2,       // push predicate(column) evaluated for every doc of the tile
  PG_F_AND_SCAN = 3,        // top &= predicate(column), evaluated only where top has candidates (counts candidates)
  PG_F_AND = 4,             // pop b, top &= b
  PG_F_OR = 5,              // pop b, top |= b
  PG_F_NOT = 6,             // top = ~top & valid
  PG_F_PUSH_NONE = 7,       // push empty set
  PG_F_PUSH_ALL = 8,        // push every doc of the tile (MatchAllFilterOperator)
  PG_F_PUSH_WORDS = 9,      // push a precomputed docId set given as match words (range leaf `arg`, one dword per 32 docs)
  PG_F_PUSH_RANGEIDX = 10   // push the rows a bit-sliced range index selects (PgRangeIdxLeaf `arg`): interpreter kernels only
};

struct PgFInstr {
  int32_t op;
  int32_t arg;   // leaf index
};

enum PgColKind : int32_t {
  PG_COL_FIXED_BIT = 0, PG_COL_RAW32 = 1, PG_COL_RAW64 = 2,
  PG_COL_HLL_REGS = 3,  // star-tree DISTINCTCOUNTHLL pair: one byte per register, 2^bits registers per doc (transcoded at upload)
  PG_COL_VAR_BYTES = 4  // raw STRING / BYTES column: values back to back + int64 offsets (no query kernel reads it: GROUP BY goes through
                        // its virtual dictionary, pg_vdict.hip)
};
enum PgValType : int32_t { PG_V_I32 = 0, PG_V_I64 = 1, PG_V_F32 = 2, PG_V_F64 = 3 };

enum PgPredKind : int32_t {
  PG_P_RANGE = 0,       // inclusive [lo, hi] on the value (typed by val_type) or on the dictId (dictionary columns)
  PG_P_DICT_LUT = 1,    // dictId bitset lookup (IN / NOT_IN / NOT_EQ on dictionary columns)
  PG_P_SET = 2          // raw value in sorted set (xor exclusive)
};

struct PgScanLeaf {
  const uint8_t* data;      // column bytes, doc 0 at offset 0 (big-endian values / MSB-first bit stream)
  const uint32_t* lut;      // PG_P_DICT_LUT: ceil(cardinality/32) words
  const void* set_values;   // PG_P_SET: sorted values in native layout of val_type (int64 / double widened)
  int64_t lo, hi;           // PG_P_RANGE: integer bounds, or the bit patterns of double bounds
  int32_t col_kind;
  int32_t bits;             // PG_COL_FIXED_BIT
  int32_t val_type;         // PgValType of a raw column
  int32_t pred_kind;
  int32_t exclusive;
  int32_t n_set;
  int32_t stat_slot;        // PG_F_AND_SCAN: index into stats[] receiving the number of candidates evaluated
  int32_t mv;               // 1: a multi-value dictionary column (pg_mv_query_* only): `data` is the bit stream of ALL entries, `set_values`
                            // points at the docs' first entries (int32 [numDocs + 1]); a doc passes when ANY entry passes the dictId test,
                            // ALL entries when `exclusive` (applyMV); stats[stat_slot] receives the ENTRIES of the candidates evaluated
};

// One RoaringBitmap container re-laid out in HBM: payload 16-byte aligned inside the column's container buffer.
struct PgContainer {
  uint64_t offset;   // byte offset into PgPostingLeaf::containers
  uint32_t n;        // array: cardinality; run: number of runs; bitmap: cardinality
  uint16_t key;      // docId >> 16
  uint16_t type;     // 0 array, 1 bitmap, 2 run
};

#define PG_MAX_DENSE 8
struct PgPostingLeaf {
  const uint8_t* containers;
  const uint32_t* chunk_start;    // CSR over chunks: entries [chunk_start[c], chunk_start[c+1]) of `entries`
  const PgContainer* entries;     // the leaf's containers grouped by chunk (copied inline: one dependent load less)
  int32_t exclusive;
  int32_t has_csr;                // 0: every container of the leaf is served by the dense pointers below
  // Dense postings: a dictId whose containers of chunks [0, dense_chunks) are all bitmap containers laid out back to
  // back (8 KB stride) is addressed arithmetically — no descriptor loads on the per-tile path.
  const uint8_t* dense[PG_MAX_DENSE];
  int32_t n_dense;
  int32_t dense_chunks;
};

struct PgRangeLeaf {
  const int32_t* lo;   // inclusive
  const int32_t* hi;   // inclusive
  const uint32_t* words;   // non-null: the doc set as match words instead (one dword per 32 docs, whole wave tiles)
  int32_t n;
  int32_t pad;
};

// RangeIndexBasedFilterOperator over a bit-sliced range index (RoaringBitmap RangeBitmap): slice s = the rows whose stored value has
// bit s CLEAR, one container per (2^16-row chunk, slice).  rows with value <= t:  L = all;  for s = 0 .. slices-1:
// L = bit s of t ? L | slice_s : L & slice_s.  The leaf selects  lte(hi) AND NOT lte(lo_m1).
struct PgRangeIdxLeaf {
  const uint8_t* containers;
  const PgContainer* descs;   // [chunks][n_slices]; type 3: the chunk has no container for the slice (no row there has the bit clear)
  uint64_t hi;                // has_hi: rows with stored value <= hi
  uint64_t lo_m1;             // has_lo: minus the rows with stored value <= lo_m1
  int32_t n_slices;
  int32_t has_hi, has_lo;
  int32_t pad;
};

// ---- aggregation plan --------------------------------------------------------------------------------------------------
enum PgAccFn : int32_t { PG_ACC_COUNT = 0, PG_ACC_SUM = 1, PG_ACC_MIN = 2, PG_ACC_MAX = 3 };
enum PgAggMode : int32_t {
  PG_AGG_NONE = 0,
  PG_AGG_SINGLE = 1,   // no GROUP BY: one private LDS slot per thread, one flush per workgroup
  PG_AGG_LDS = 2,      // accumulator table [n_acc][G*R] in LDS, flushed to per-workgroup partials
  PG_AGG_GLOBAL = 3,   // accumulator table [n_acc][G] in HBM, device-scope atomics (memory-side: ~24 G atomics/s on MI355X)
  // Key space beyond one LDS table: the raw-key range is cut into n_parts ranges of part_groups keys; workgroup b (XCD
  // b % 8, index i = b / 8 inside it) owns range i % n_parts and walks the tile chunks of ITS XCD together with the other
  // ranges' workgroups of that XCD — every chunk is fetched from HBM once per XCD and re-read from that XCD's L2 by the other
  // ranges — aggregating only the docs whose key falls in its range into an LDS table [n_acc][part_groups].
  PG_AGG_LDS_PART = 4,
  // Large key spaces, one visit per doc: the docs that pass the filter are radix-partitioned by key range into (local key,
  // values) tuples in HBM — pass 1 counts per (workgroup, bucket), a scan turns the counts into exact offsets, pass 2 scatters —
  // and every bucket (2^radix_shift keys = one LDS table) is then aggregated from its contiguous tuple range with LDS atomics.
  PG_AGG_RADIX = 5,
  // Key spaces too large for any dense table (Π cardinalities > 64 M: the reference's LongMapBasedHolder territory): the same
  // radix pipeline on 64-bit raw keys with bucket = hash(key); every bucket is aggregated into an open-addressing hash table in
  // LDS (keys claimed with ds_cmpst, accumulators updated with LDS atomics) and its occupied slots are appended to the result.
  PG_AGG_RADIX_HASH = 6
};
#define PG_MAX_RADIX_BUCKETS 2048
#define PG_RADIX_INVALID_KEY 0xFFFFFFFFu   // local key of a padding tuple (staged scatter pads every flush to whole lines)
#define PG_MAX_RADIX_SRCS 4
// Partition pipeline v2 (pg_kernels_part.hip): count-free radix partitioning into chunked per-(workgroup, bucket) streams of
// bit-packed tuples, whole 128-byte lines only; see the header of pg_kernels_part.hip.
#define PG_P2_CHUNK 256            // tuples per chunk (1 KB per plane)
#define PG_P2_LINE 32              // tuples per 128-byte line of a plane
#define PG_P2_MAX_PLANES 4         // dwords per tuple
#define PG_P2_MAX_BUCKETS 256
#define PG_P2_WAVES 4              // wavefronts per scatter workgroup
#define PG_P2_POOL 512             // chunk ids a scatter workgroup holds in LDS (ring)
#define PG_P2_BATCH 128            // chunk ids claimed per global atomic
#define PG_P2_LIST 2048            // chunk-list window of an aggregation work item (LDS)
#define PG_P2_MAX_LINES (PG_P2_WAVES * 4 * 256 / PG_P2_LINE + PG_P2_MAX_BUCKETS)   // whole lines one round can complete
// p2_ctrl (dwords): [0] chunks claimed, [1] error flag, then the chunk index built between the scatter and the aggregation pass
#define PG_P2_CTRL_COUNTS 16                                   // [buckets] chunks per bucket
#define PG_P2_CTRL_STARTS (PG_P2_CTRL_COUNTS + PG_P2_MAX_BUCKETS)        // [buckets + 1] exclusive prefix
#define PG_P2_CTRL_CURSOR (PG_P2_CTRL_STARTS + PG_P2_MAX_BUCKETS + 16)   // [buckets] fill cursors
#define PG_P2_CTRL_STRIPE0 (PG_P2_CTRL_CURSOR + PG_P2_MAX_BUCKETS + 16)   // chunk-id cursors, one per stripe, each in a 128-byte line of its own
#define PG_P2_STRIPES 64
#define PG_P2_STRIPE_DWORDS 32
#define PG_P2_CTRL_TIMING (PG_P2_CTRL_STRIPE0 + PG_P2_STRIPES * PG_P2_STRIPE_DWORDS)   // 16 x uint64: cycles per phase (PG_P2_TIMING variant)
#define PG_P2_CTRL_DWORDS (PG_P2_CTRL_TIMING + 32)
#define PG_P2_AGG_THREADS 1024
// pruned-offer passes (pg_kernels_oct.hip): the survivor stream's control block (dwords)
#define PG_OCT_MAX_REGIONS 1024
#define PG_OCT_CTRL_COUNTS 16
#define PG_OCT_CTRL_TILE_START (PG_OCT_CTRL_COUNTS + PG_OCT_MAX_REGIONS + 16)
#define PG_OCT_CTRL_DWORDS (PG_OCT_CTRL_TILE_START + PG_OCT_MAX_REGIONS + 16)
enum PgP2FieldKind : int32_t {
  PG_P2_F_HLL = 0,      // (register index | rank << log2m) of the value a DISTINCTCOUNTHLL offers
  PG_P2_F_DICTID = 1,   // dictId of a dictionary-encoded source (value looked up in the aggregation pass)
  PG_P2_F_RAW32 = 2,    // raw 32-bit value minus the column's minimum (INT), or the bits of a FLOAT
  PG_P2_F_RAW64 = 3     // raw 64-bit value: two whole planes (low dword, high dword)
};

struct PgGroupCol {
  const uint8_t* data;   // fixed-bit dictIds (or the raw big-endian values of a no-dictionary group column)
  int64_t mult;          // prod of the cardinalities of the previous group columns (col 0 least significant)
  int32_t bits;
  int32_t col_kind;      // PG_COL_FIXED_BIT (0), or PG_COL_RAW32 / PG_COL_RAW64: the value itself is the key (hash group-by only),
                         // stored as value ^ 2^63 so that the table's empty marker ~0 is Long.MAX_VALUE
};

struct PgValueSrc {
  const uint8_t* data;
  const void* dict;      // native-endian dictionary values (val_type) for dictionary columns, else null
  int32_t col_kind;
  int32_t bits;
  int32_t val_type;
  int32_t fx_q;          // FLOAT / DOUBLE sources summed in fixed point: digit j of a value x is digit j (base 2^32) of trunc(|x| * 2^-fx_q)
};

// How a SUM accumulator takes its values (PgAccOp::is_float).  The reference adds every value to a double in docId order
// (SumAggregationFunction.java:160-179); any other order of floating additions gives other roundings, so floating SUMs are kept
// EXACT instead: each value is cut into base-2^32 digits of a fixed-point number whose scale the column's largest magnitude
// fixes, every digit is summed in its own int64 accumulator ("limb": |digit| < 2^32, < 2^31 docs, so no limb overflows), and the
// limbs are combined and rounded to double ONCE on the host.  Integer additions commute, so the result does not depend on the
// order of execution, on the number of workgroups, or on how partial tables are merged (including across GPUs: ncclSum on int64).
// LONG sources whose sum could leave int64 use two such digits of the value itself.
enum PgAccValueKind : int32_t {
  PG_ACCV_INT = 0,         // int64 value (INT / LONG sign-extended); MIN / MAX of integers
  PG_ACCV_DOUBLE = 1,      // double: MIN / MAX through order-preserving int64 keys; SUM in IEEE double only for columns holding NaN / Inf
  PG_ACCV_FIXED_DIGIT = 2, // SUM: digit `limb` of the fixed-point image of a FLOAT / DOUBLE value
  PG_ACCV_LONG_DIGIT = 3   // SUM: digit `limb` (0: low 32 bits unsigned, 1: high 32 bits signed) of a LONG value
};

struct PgAccOp {
  int32_t fn;         // PgAccFn
  int32_t src;        // index into srcs, -1 for COUNT
  int32_t is_float;   // PgAccValueKind
  int32_t limb;       // digit index of a PG_ACCV_FIXED_DIGIT / PG_ACCV_LONG_DIGIT accumulator
};

// Auxiliary (non-scalar) accumulators, always in HBM, one region per op, zero-initialised per execution:
//   PG_AUX_DICT_SET  DISTINCTCOUNT over a dictionary column: bitset of dictIds per group, `stride` 32-bit words per group
//                    (BaseDistinctAggregateAggregationFunction.java:306-345 keeps a RoaringBitmap of dictIds per group)
//   PG_AUX_HLL_DICT  DISTINCTCOUNTHLL over a dictionary column: 2^log2m one-byte registers per group; (index, rank) of every
//                    dictionary value is precomputed on the host (`lut`, DistinctCountHLLAggregationFunction.java:457-466
//                    offers dictionary.get(dictId) for each dictId of the group's bitmap — register max is idempotent, so
//                    offering every doc's value gives the same registers)
//   PG_AUX_HLL_RAW   DISTINCTCOUNTHLL over a raw column: hll.offer(value) per doc (:188-221), MurmurHash.hashLong on device
//   PG_AUX_HLL_BYTES DISTINCTCOUNTHLL over a star-tree pair column of serialized HyperLogLogs: register-wise max of the doc's
//                    registers into the group's (HyperLogLog#addAll, DistinctCountHLLAggregationFunction.java:158-175)
enum PgAuxKind : int32_t { PG_AUX_DICT_SET = 0, PG_AUX_HLL_DICT = 1, PG_AUX_HLL_RAW = 2, PG_AUX_HLL_BYTES = 3 };
#define PG_MAX_AUX 4
struct PgAuxOp {
  int32_t kind;
  int32_t src;          // index into srcs (column read for the op)
  int32_t stride;       // DICT_SET: words per group; HLL: registers (bytes) per group = 1 << log2m
  int32_t log2m;
  int32_t n_rep;        // replicas of the region (power of two): workgroup b updates replica b & (n_rep-1); small states
  int32_t lds_offset;   //   (few groups) would otherwise funnel every update of the chip into a handful of L2 lines
                        // lds_offset >= 0: the state of ONE workgroup lives in its LDS at this byte offset of the dynamic LDS
                        // (states up to ~128 KB: LDS atomics instead of memory-side ones); `base` is then the workgroups'
                        // partial area [grid][rep_bytes] in HBM, merged by pg_reduce_aux_kernel
  int64_t rep_bytes;    // bytes of one replica
  uint32_t* base;       // region of this op (patched per execution)
  const uint32_t* lut;  // HLL_DICT: per dictId (register index | rank << 16)
};

struct PgQueryPlan {
  int32_t num_docs;
  int32_t n_tiles;                  // 16 384-doc tiles (allocation / expansion granule)
  int32_t n_wtiles;                 // 2 048-doc wave tiles
  int32_t pad_w;
  int32_t n_instr;
  int32_t stack_depth;
  const PgFInstr* instrs;
  const PgScanLeaf* scans;
  const PgPostingLeaf* postings;
  const PgRangeLeaf* ranges;
  const PgRangeIdxLeaf* rangeidx;
  // outputs of the filter stage
  uint64_t* out_words;              // nullable: ceil(num_docs/64) match words (tile padded)
  uint32_t* out_tile_counts;        // nullable: matches per tile
  unsigned long long* stats;        // [PG_MAX_STATS]: slot 0 = matched docs, slots 1.. = candidates per AND_SCAN leaf
  // aggregation stage
  // fast path (pg_fast_query_kernel<SK, AGG>): instrs[0, n_index_instr) is an index-only program (postings / ranges /
  // AND / OR / NOT), optionally followed by ONE scan leaf of kind fast_scan_kind restricted to its result
  int32_t n_index_instr;
  int32_t fast_scan;                // index into scans, -1: none
  int32_t fast_scan_pushed;         // the scan is the whole filter (PUSH_SCAN): candidates = every doc
  int32_t fast_agg_shape;           // the aggregation has the fast shape (LDS table, <= 8-bit group columns, 32-bit sources)
  int32_t agg_mode;
  int32_t n_group_cols;
  int32_t n_srcs;
  int32_t n_ops;
  int32_t n_groups;                 // G = product of group cardinalities (1 without GROUP BY)
  int32_t replicas;                 // R: LDS copies per group (power of two) to spread atomic conflicts
  int32_t replica_shift;            // log2(R): group index = slot >> replica_shift
  int32_t n_aux;
  // PG_AGG_RADIX (pointers patched per execution)
  int32_t radix_shift;              // bucket = key >> radix_shift, local key = key & ((1 << radix_shift) - 1)
  int32_t radix_buckets;
  int32_t radix_slices;             // workgroups (slices) per bucket in the aggregation pass
  int32_t radix_stage;              // > 0: tuples per line of the staged scatter (pg_radix_scatter_packed_kernel: 32): ranges of the tuple area
                                    // hold whole lines, padded with PG_RADIX_INVALID_KEY; 0: one store per tuple, no padding
  const uint32_t* match_words;      // filter result, one dword per 32 docs, whole wave tiles
  uint32_t* radix_hist;             // [grid][radix_buckets] tuple counts, then exact offsets (bucket major)
  uint32_t* radix_bucket_start;     // [radix_buckets + 1]
  int32_t hash_cap;                 // PG_AGG_RADIX_HASH: slots of the per-bucket LDS hash table (power of two)
  int32_t pad_h;
  unsigned long long* hash_out_count;   // [0] groups appended, [1] overflow flag (a bucket had more distinct keys than slots)
  int64_t* hash_out_keys;           // [hash_out_cap] raw keys
  int64_t* hash_out_acc;            // [n_ops][hash_out_cap]
  int64_t hash_out_cap;
  uint8_t* radix_tuples;            // [matched] x radix_stride bytes: {u32 local key, u32 docId, 8 bytes per source (int64 / double bits)}
  int64_t radix_stride;             // 8 without sources, else 8 + 8 * n_srcs rounded up to 16 (hash: {u64 key, u32 docId, u32 0} + 8 per source)
  // packed 4-byte tuples (radix_packed): the local key in bits [0, radix_shift), then one bit field per source
  int32_t radix_packed;
  int32_t pk_pad;
  int32_t pk_shift[PG_MAX_RADIX_SRCS];   // first bit of source i's field
  int32_t pk_bits[PG_MAX_RADIX_SRCS];    // its width: log2m + 5 for a HyperLogLog source, the dictionary column's bits otherwise
  int32_t pk_hll[PG_MAX_RADIX_SRCS];     // > 0: the source feeds a DISTINCTCOUNTHLL of this log2m (field = register index | rank << log2m)
  const uint32_t* pk_lut[PG_MAX_RADIX_SRCS];   // per dictId (index | rank << 16) of a dictionary-encoded HyperLogLog source
  int32_t pk_affine[PG_MAX_RADIX_SRCS];        // 1: that source's dictionary is arithmetic, value = pk_base + pk_step x dictId (INT / LONG)
  int64_t pk_base[PG_MAX_RADIX_SRCS], pk_step[PG_MAX_RADIX_SRCS];
  // partition pipeline v2 (p2 = 1; pg_kernels_part.hip): a tuple is p2_planes dwords, stored as planes (structure of arrays) of
  // chunked streams; plane 0 holds the local key in bits [0, radix_shift) and never uses bit 31 (0xFFFFFFFF = padding); source i's
  // field sits in plane p2_fplane[i] at bit pk_shift[i], pk_bits[i] wide (PG_P2_F_RAW64: planes p2_fplane[i], p2_fplane[i] + 1)
  int32_t p2;
  int32_t p2_planes;
  int32_t p2_docid_plane;           // plane carrying the docId (MIN(docId) of numGroupsLimit trimming), -1: none
  int32_t p2_capacity;              // chunks of the tuple area (= PG_P2_STRIPES x p2_stripe_cap); chunk index p2_capacity is the spill chunk
                                    // (overflow = internal error).  Chunk ids are handed out per STRIPE: scatter workgroup b claims from the
                                    // cursor of stripe b % PG_P2_STRIPES, ids stripe x p2_stripe_cap + ... — one cursor for the whole grid
                                    // serialised the claims at ~60 ns each (profiles/r04_c_kernels__cfg5_.txt)
  int32_t p2_fplane[PG_MAX_RADIX_SRCS];
  int32_t p2_fkind[PG_MAX_RADIX_SRCS];
  int64_t p2_fbias[PG_MAX_RADIX_SRCS];   // PG_P2_F_RAW32 over INT: stored field = value - bias
  uint32_t* p2_tuples;              // [p2_planes][(p2_capacity + 1) * PG_P2_CHUNK] dwords
  int64_t p2_plane_stride;          // dwords per plane
  uint32_t* p2_meta;                // [p2_capacity]: owner bucket | filled lines << 16; 0xFFFFFFFF: chunk never handed out
  uint32_t* p2_ctrl;                // [0] chunks claimed so far, [1] error flag, [PG_P2_CTRL_*] the chunk index's counters
  uint32_t* p2_list;                // [p2_capacity] chunk records grouped by bucket: chunk id | filled lines << 27
  int32_t p2_fast_a;                // 1: the scatter's batched loader fits (<= 4 group columns: the first <= 24 bits, the others <= 8;
  int32_t p2_oct_a;                 //    at most one source, bit-packed <= 24 bits or raw 32-bit).  p2_oct_a: the oct-layout phase A fits
                                    //    (pg_p2_scatter_o*: one plane, fixed-bit group columns <= 8 bits — the first <= 24 —, no source / one
                                    //    raw INT / one <= 24-bit dictId field) — taken by plans without a filter pass in front
  int32_t n_lin_prefix;             // interpreter kernels: instrs[0, n_lin_prefix) is index-only and leaves one stack entry
  int32_t n_fast_scans;             // pg_fast_multi_*: instrs[n_index_instr, n_index_instr + n_fast_scans) are scan leaves ANDed in order
  int32_t tail_posting;             // pg_fast_multi_*: posting leaf ANDed in AFTER the scans (-1: none) — the queryableDocIds bitmap of
  int32_t tail_pad;                 //   FilterPlanNode.run's outer AND, which must not restrict the inner AND's scans (exact scan counts)
  int32_t n_parts;                  // PG_AGG_LDS_PART: key ranges (the grid is 8 x a multiple of it)
  int32_t part_groups;              // PG_AGG_LDS_PART: keys per range
  PgAuxOp aux[PG_MAX_AUX];
  PgGroupCol gcols[PG_MAX_GROUP_COLS];
  PgValueSrc srcs[PG_MAX_SRCS];
  PgAccOp ops[PG_MAX_OPS];
  int64_t* partials;                // LDS/SINGLE mode: [gridDim][n_ops][G]; GLOBAL mode: [n_ops][G]
  // Fused dense index program (pg_fast_i32range_d): the whole index-only program is an AND of up to 4 groups, each the OR of dense
  // bitmap postings (base + 8 KB x chunk; optionally complemented: NOT_IN / NOT_EQ) — 8 pointers in all, unused slots repeat a
  // pointer of their group.  One load per pointer and tile, issued one tile ahead.
  int32_t dense_fused;              // 1: dense_ptr / dense_group describe the index program
  int32_t dense_groups;             // number of AND-ed groups (1..4)
  int32_t dense_excl;               // bit g: group g is complemented
  int32_t dense_pad;
  const uint8_t* dense_ptr[8];
  int32_t dense_group[8];
  // pg_fast_i32range_p (pg_kernels_pipe.hip): dense_fused, one or two <= 8-bit group columns and every value accumulator over ONE raw
  // 32-bit column (srcs[pipe_src])
  int32_t pipe_fit;
  int32_t pipe_src;
  // the pipeline's other shapes (pg_pipe_*, round 3): pipe_general = 1 + which stages the filter has; pipe_tail: the dense bitmap ANDed
  // in after the scan (upsert queryableDocIds snapshot), chunk stride 8 KB like the dense postings
  int32_t pipe_general;
  int32_t pipe_has_index, pipe_has_scan;
  int32_t pipe_vscan;               // >= 0: index of the second scan leaf, a range over the value column tested on the value quads (-1: none)
  const uint8_t* pipe_tail;
  int32_t pipe_wide;                // pg_pipe_w*: value kind of the one source column (1: raw INT, 2: raw LONG, 3: raw DOUBLE); 0: not this family
  int32_t mv_no_windows;            // measurement knob (PG_MV_NO_WINDOWS): multi-value scan leaves walk doc by doc as in round 3
  // Interpreter kernels over small doc spaces with expensive per-doc state updates (a star-tree's serialized HyperLogLogs: one
  // wavefront-wide register merge per matching doc): every wave tile is visited by 2^tile_split_shift wavefronts, each evaluating
  // the tile's filter and then keeping only its share of the matching docs (a quad slot and a lane class), so that a 13 617-doc
  // star-tree occupies 224 wavefronts instead of 7.  Share 0 reports the tile's filter statistics and match words.
  int32_t tile_split_shift;
  int32_t tile_split_pad;
  // Multi-value columns (pg_kernels_mv.hip, appended so that the single-value kernels see the layout they were tuned on): per group
  // column / value source the docs' first entries (int32 [numDocs + 1]) when it is a multi-value column — its `data` is then the bit
  // stream of all entries — else null.  mv_src_len[i] = 1: the source's value is the doc's NUMBER of entries (COUNTMV / AVGMV's count).
  const int32_t* mv_gcol_offsets[PG_MAX_GROUP_COLS];
  const int32_t* mv_src_offsets[PG_MAX_SRCS];
  int32_t mv_src_len[PG_MAX_SRCS];
  int32_t mv;                       // 1: the plan touches a multi-value column: pg_mv_query_* run it
  int32_t mvg;                      // pg_mv_group_<mvg> (pg_kernels_mvg.hip): GROUP BY one multi-value column, integer accumulators over srcs[pipe_src] (or COUNT only), no filter; 4 / 8 = the entries requested per doc up front.  16 + 4 / 16 + 8: pg_mv_aggr_* — the *MV functions over the ONE multi-value INT source srcs[pipe_src], grouped by one or two single-value dictionary columns
  // Oct-layout kernels (pg_kernels_oct.hip, round 4): <= 4 group columns of <= 8 bits, COUNT at most among the ops and ONE DISTINCTCOUNTHLL /
  // DISTINCTCOUNT state.  oct = 1: the state lives in the workgroup's LDS (pg_oct_l*, the layout of pg_generic_query_l); oct = 2: pruned
  // offers (pg_oct_p*): survivors of the group floors go to the tuple stream, the partition pipeline aggregates them pass by pass.
  int32_t oct;
  int32_t oct_src;                  // index into srcs of the state's column
  int32_t oct_src_kind;             // 0 none; 1 bit-packed dictIds of an arithmetic INT dictionary (hashed from constants); 2 dictIds through
                                    // the per-dictId (index | rank << 16) table; 3 raw INT values; 4 dictIds as they are (DISTINCTCOUNT)
  int32_t oct_log2m;
  uint32_t oct_c0, oct_c1;          // kind 1: (uint32) value x m = oct_c0 + dictId x oct_c1  (m = 0x5bd1e995, MurmurHash's multiplier)
  int32_t oct_nonneg;               // kind 1: every dictionary value is >= 0 (the high word of (long) value hashes to nothing)
  int32_t oct_base, oct_step;       // kind 1 otherwise: value = oct_base + oct_step x dictId
  int32_t oct_t0, oct_t1;           // oct = 2: the wave tiles [t0, t1) of this pass
  int32_t oct_dword;                // oct = 1, HyperLogLog: the workgroup's registers are dwords in LDS (ds_max_u32), packed to bytes at the flush
  const uint32_t* oct_lut;          // kind 2
  const uint8_t* oct_floor;         // oct = 2: [n_groups] smallest register of every group so far (dword padded)
  uint32_t* oct_counts;             // oct = 2: [grid][n_groups] 32-bit COUNT partials of this pass
  uint32_t* oct_stream;             // oct = 2: survivor entries, key << (log2m + 5) | index | rank << log2m; PG_RADIX_INVALID_KEY = padding
  uint32_t* oct_cursor;             // control block: [1] overflow flag, [PG_OCT_CTRL_COUNTS + w] entries region w holds (a multiple of 256),
                                    //   [PG_OCT_CTRL_TILE_START + w] its first 2 048-entry tile in the numbering across the regions
  int64_t oct_stream_cap;           // entries the stream holds (oct_n_regions x oct_region)
  int32_t p2_stripe_cap;            // chunk ids per stripe
  int32_t p2_byte_regs;             // 1: the aggregation pass keeps HyperLogLog registers as bytes and no accumulator table (pruned-offer passes)
  int32_t oct_region;               // entries per region: the docs of a pg_oct_p workgroup in this pass + a padding block per wavefront
  int32_t oct_n_regions;            // regions = workgroups of pg_oct_p in this pass
  // pg_fast_dictrange_s family (pg_kernels_specd.hip, round 6): the loader / consumer frame over a dictionary-encoded scan and / or value
  // column.  specd = 1: srcs[pipe_src] is the one value column, scans[fast_scan] the one range scan (pipe_has_scan), dense_* the index
  // program (pipe_has_index), pipe_tail the upsert snapshot.  specd = 2: the shared-stage frame of the same plans (pg_kernels_specw.hip, PG_SPECW).
  int32_t specd;
  int32_t specd_vkind;              // 1: raw INT values; 2: value = specd_base + specd_step x dictId (arithmetic INT dictionary); 3: srcs[pipe_src].dict[dictId]
  int32_t specd_sbits;              // bits per value of the scan column's stream (32: raw INT); 0 without a scan
  int32_t specd_vbits;              // ... of the value column's
  int32_t specd_base, specd_step;
  int32_t specd_dma;                // the headline shape's LDS-DMA kernels (two column areas per strip): pg_fast_dictrange_s_*_dma
  // pg_nogroup_d (pg_kernels_scan.hip): no GROUP BY, no filter, integer accumulators over ONE dictionary-encoded INT column (<= 24-bit dictIds of a
  // sorted dictionary): 0 no; 1 value = nogroup_base + nogroup_step x dictId (arithmetic dictionary); 2 srcs[nogroup_src].dict[dictId] — from a copy in
  // the workgroup's LDS when nogroup_lds_card > 0 (the dictionary's cardinality: it fits)
  int32_t nogroup_d, nogroup_src, nogroup_bits, nogroup_lds_card;
  int64_t nogroup_base, nogroup_step;
  // pg_dictrange_fo (pg_kernels_scan.hip): the filter is [an index program AND] ONE range predicate over a dictionary-encoded column of <= 24 bits —
  // taken when nothing is aggregated (COUNT(*), docId sets, a leaf's match bitmap)
  int32_t dict_filter_only, dict_filter_pad;
  int32_t mvg_has_entries;          // pg_mv_aggr_*: an accumulator reads the entries' values (else only their number)
  int32_t mvg_dict_card;            // ... and the entries' dictionary (<= 4 096 values) is copied into LDS behind the table; 0: gathered from global memory
  int32_t p2_no_pack;               // PG_P2_NO_PACK (measurement knob): COUNT and SUM keep an LDS atomic each in pg_p2_aggregate_*s
};

#if defined(__HIPCC__)
#define PG_HD __host__ __device__
#else
#define PG_HD
#endif
PG_HD static inline int64_t pg_acc_identity(int32_t fn, int32_t is_float) {
  // MIN/MAX on floats use order-preserving int64 keys, so the integer identities serve both
  (void)is_float;
  if (fn == PG_ACC_MIN) return INT64_MAX;
  if (fn == PG_ACC_MAX) return INT64_MIN;
  return 0;
}

// ---- segment-level group trim on the device (pg_kernels_trim.hip) ----------------------------------------------------------------------
// ctrl (uint32 words): [0] groups that exist, [1] survivors below the threshold written, [2] tie-class members written, [3] overflow flag,
// [4] need_eq (places left for the tie class), [5] n_lt (survivors below the threshold), [8 + 4 p .. ] state of pass p: prefix lo, prefix hi,
// remaining, pad; [64 + 256 p ..] histogram of pass p
#define PG_TRIM_CTRL_STATE 8
#define PG_TRIM_CTRL_HIST 64
#define PG_TRIM_CTRL_WORDS (PG_TRIM_CTRL_HIST + 8 * 256)

struct PgTrimArgs {
  const int64_t* table;     // [n_ops][G]
  int64_t G;
  int32_t n_ops;
  int32_t exist_op;         // the op whose row tells whether a group exists (COUNT != 0, or MIN / MAX off its identity)
  int64_t exist_ident;
  int32_t key_op;           // >= 0: the key is that row's int64; -1: the key is digit (g / key_mult) % key_card of the raw key
  int32_t descending;
  int64_t key_mult;
  int64_t key_card;
  uint64_t* keys;           // [G] scratch
  uint32_t* ctrl;           // [PG_TRIM_CTRL_WORDS], zeroed by the caller
  int32_t k;                // trimSize
  int32_t cap;              // rows of the compact block (>= k)
  int32_t take_whole_tie_class;   // several ORDER BY expressions: the host finishes the selection
  int32_t pad;
  int64_t* out_gids;        // [cap]
  int64_t* out_table;       // [n_ops][cap]
};

