/*
 * include/spp.h -- C ABI of the MI355X-native mini-batch GNN data path
 * (libspp_hip.so), the drop-in boundary beneath SALIENT++'s `fast_sampler`
 * module (reference: fast_sampler/fast_sampler.cpp:1280-1396 is the pybind11
 * surface this replaces).
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures: device buffers are `void*`
 *     or typed pointers to HBM, streams are passed as `void*` (a hipStream_t),
 *     0 / NULL means the null stream;
 *   - every buffer is CALLER-OWNED unless stated otherwise; sampler/session
 *     objects own only their internal workspace;
 *   - every function returns SPP_OK (0) or a negative spp_status; the message
 *     is available from spp_last_error() (thread-local);
 *   - all functions are asynchronous w.r.t. the device unless documented as
 *     blocking (spp_sampler_wait, spp_session_next);
 *   - node ids are < 2^31 (the reference narrows to int32 inside sampling,
 *     fast_sampler.cpp:196-199) and tensors at the boundary are int64, as
 *     PyG / torch_sparse expect.
 */
#ifndef SPP_H
#define SPP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPP_ABI_VERSION 6
#define SPP_MAX_HOPS 8
#define SPP_MAX_PARTS 64

typedef int spp_status;
enum {
  SPP_OK = 0,
  SPP_ERR_INVALID = -1,   /* bad argument (TORCH_CHECK equivalents -> RuntimeError upstream) */
  SPP_ERR_HIP = -2,       /* HIP runtime error */
  SPP_ERR_CAPACITY = -3,  /* a batch exceeded the workspace the sampler was created for */
  SPP_ERR_STATE = -4      /* call sequence error (e.g. export before wait) */
};

int spp_abi_version(void);
const char* spp_last_error(void);
/* number of visible HIP devices, or <0 when no device / runtime (the product path fails loudly) */
int spp_device_count(void);

/* Live timing of the HBM-bound kernels with HIP events recorded on the launching stream
 * (bench.py's roofline leg).  spp_profile_enable(n) clears and starts recording: n = 1 times
 * every launch, n > 1 every n-th launch of a kind (two timing events around a ~100 us kernel cost ~7 us
 * of idle hardware queue each time: timing EVERY per-batch delivery slowed the pipeline by 5 %), 0 stops.
 * spp_profile_read is BLOCKING and returns the summed duration, launch count and processed
 * units (rows) of the TIMED launches of one kernel kind. */
#define SPP_PROF_GATHER 0     /* k_gather_rows   (serial_index)              */
#define SPP_PROF_ASSEMBLE 1   /* k_assemble      (distributed final assembly) */
#define SPP_PROF_CHAIN 2      /* one sampling chain of a group of batches, first launch to last, on its sampling stream
                                 (units = batches of the group; every chain is timed whatever n is: two events per
                                 ~0.6 ms chain; sample_adj x hops, sample_cpu.hpp:25-143 / fast_sampler.cpp:191-227) */
#define SPP_PROF_KINDS 3
void spp_profile_enable(int on);
spp_status spp_profile_read(int kind, double* total_ms, int64_t* launches, int64_t* units);

/* Run-time tuning knobs (measurement aid and the hook of embedders that pace the data path themselves).
 * "gather_wg_per_cu": workgroups per compute unit a row gather / delivery launch may put on the chip (default 16, or
 * SPP_GATHER_WG_PER_CU); value <= 0 only reads.
 * "gather_span": 1 (default; SPP_GATHER_SPAN=0) = rows of 16k + 8 bytes (200-byte rows of 100 fp16 features) out of a
 * table whose base and row stride are multiples of 16 bytes (>= the row + 8) into a 16-byte-aligned destination are moved
 * with 16-byte accesses (the loads read up to 8 bytes of the row's padding; never written); 0 = the 8-byte form;
 * value < 0 only reads.  Returns the previous value, or a negative spp_status. */
int spp_tune(const char* key, int value);

/* Asynchronously detected data errors.  The reference's CPU code does not range-check row indices
 * (fast_sampler.cpp:253-256 copies in[idx[i]] blindly); here a kernel that meets an index outside
 * its table clamps it to a valid row (no fault) and raises a bit in a per-device word in pinned host
 * memory, so a peer's bucketing bug surfaces as an error instead of wrong features.  The word is
 * read without any device synchronisation (a bit becomes visible once the offending kernel has run);
 * spp_session_next reports it as SPP_ERR_STATE, other callers poll it with spp_async_errors. */
#define SPP_AERR_GATHER_INDEX 1   /* spp_gather_rows*: idx[i] outside [0, src_rows)                       */
#define SPP_AERR_SERVE_ID 2       /* native exchange: a peer requested a row this rank does not own      */
#define SPP_AERR_ASSEMBLE 4       /* feature assembly: a source row outside its table                    */
/* returns the error mask of `device` (>= 0) or a negative spp_status; clear != 0 resets it */
int spp_async_errors(int device, int clear);

/* ------------------------------------------------------------------------- *
 * a1  std::mt19937 stream (fast_sampler/sample_cpu.hpp:11, seeding
 *     fast_sampler.cpp:994 `gen.seed(range.second*17+5)`): writes raw 32-bit
 *     outputs number skip .. skip+n-1 of mt19937(seed) to out_dev.
 * ------------------------------------------------------------------------- */
spp_status spp_mt19937_fill(uint32_t seed, int64_t skip, int64_t n, uint32_t* out_dev, void* stream);
/* seed of the batch whose idx range ends at `stop` (fast_sampler.cpp:994) */
uint32_t spp_batch_seed(int32_t stop);

/* ------------------------------------------------------------------------- *
 * a5  serial_index (fast_sampler.cpp:238-279):
 *       dst[i, :] = src[idx[i], :]  for i < min(n_idx, n_out)
 *     src is row-major [src_rows, row_bytes]; idx_elem_bytes is 8 (int64, the
 *     reference's dtype) or 4 (int32).  Rows i >= n_idx of dst are left
 *     untouched (the reference leaves them uninitialised).
 * ------------------------------------------------------------------------- */
spp_status spp_gather_rows(const void* src_dev, int64_t src_rows, int64_t row_bytes,
                           const void* idx_dev, int idx_elem_bytes, int64_t n_idx, int64_t n_out,
                           void* dst_dev, void* stream);
/* Same, for a source table whose rows are src_stride_bytes apart (>= row_bytes; 0 = dense).  The
 * resident feature table is kept with rows padded to the 128-B HBM fetch granule: a 200-B row then
 * costs 2 granules instead of 2.56 on average.  dst stays dense.
 * READABLE EXTENT: the table must hold src_rows * src_stride_bytes readable bytes (not just (src_rows - 1) * stride +
 * row_bytes): for rows of 16k + 8 bytes out of a 16-byte-aligned table with a 16-byte-multiple stride >= row_bytes + 8
 * the gather moves 16-byte pieces and the last piece of a row reads 8 bytes of that row's padding (never written;
 * spp_tune("gather_span", 0) keeps the 8-byte form, which reads row_bytes per row exactly).  The same holds for
 * spp_exchange_cfg.x_local_dev and the table of spp_session_export. */
spp_status spp_gather_rows_strided(const void* src_dev, int64_t src_rows, int64_t row_bytes,
                                   int64_t src_stride_bytes, const void* idx_dev, int idx_elem_bytes,
                                   int64_t n_idx, int64_t n_out, void* dst_dev, void* stream);

/* to_row_major (fast_sampler.cpp:281-308): column-major [rows, cols] -> row-major */
spp_status spp_to_row_major(const void* src_dev, int64_t rows, int64_t cols, int elem_bytes,
                            void* dst_dev, void* stream);

/* ------------------------------------------------------------------------- *
 * a2-a4  On-GPU multi-hop neighbour sampler: sample_adj (sample_cpu.hpp:25-143)
 *        driven by multilayer_sample (fast_sampler.cpp:191-236).
 *        Bit-exact MFG (n_id, per-hop rowptr/col) vs the CPU algorithm for the
 *        same (seeds, fanouts, graph, rng seed).
 * ------------------------------------------------------------------------- */
typedef struct spp_sampler spp_sampler;

/* Optional ownership bucketing of the batch's node list, fused into the sampling chain (worker
 * distributed branch, fast_sampler.cpp:1017-1272): with num_parts > 0 every sampled batch also
 * carries the per-owner id lists, the VIP-cache hits and perm_partition_to_mfg, and their sizes
 * arrive with the counts -- no extra launch sequence and no host synchronisation per batch. */
typedef struct spp_partition_cfg {
  int32_t num_parts;                    /* P; 0 = bucketing off                                  */
  int32_t rank;                         /* this process's partition                              */
  int64_t offsets[SPP_MAX_PARTS + 1];   /* RangePartitionBook offsets (range_partition_book.cpp:38-55) */
  int32_t use_cache;                    /* Config.use_cache (fast_sampler.cpp:1108)               */
  const int32_t* cache_map_dev;         /* int32[cache_map_len]: node -> cache row, -1 = miss
                                           (spp_cache_build_map); read while batches are in flight */
  int64_t cache_map_len;
} spp_partition_cfg;

/* Which form of the sampling chain a sampler runs.  Every form computes the SAME batches (sample_cpu.hpp:25-143,
 * fast_sampler.cpp:191-227, :994) -- the forms differ in derived tables and kernel variants only.  By default the
 * library chooses by itself (sizes, free HBM); these fields pin a choice, and spp_sampler_get_info reports what
 * was chosen.  Convention for the int32 switches: 0 = automatic (the rule stated; the environment variable named
 * is honoured, read once at spp_sampler_create), > 0 = on, < 0 = off.  A zero-filled record is "all automatic". */
typedef struct spp_sampler_opts {
  int32_t col32;            /* int32 copy of `col` (SPP_COL32; auto: on).  Off: the kernels read the int64 array        */
  int32_t deg_tags;         /* degree tags in the spare top bits of the int32 entries (SPP_DEG_TAGS; auto: on when
                               ids leave >= 3 bits spare; used by a sampler only when every later fanout < the cap)     */
  int32_t row_stubs;        /* 128-byte row-stub table (SPP_ROW_STUBS; auto: when it takes <= 1/8 of the free HBM)      */
  int32_t rng_arena;        /* mt19937 streams of a whole epoch generated once and kept (auto: when the arena fits
                               rng_arena_mb and, without an explicit budget, 1/4 of the free HBM); off: per group
                               into the slots' ping-pong buffers (k_rng_fill)                                          */
  int64_t rng_arena_mb;     /* budget of that arena in MiB (0: SPP_RNG_ARENA_MB, default 16384)                         */
  int32_t fuse_scatter;     /* bucket scatter folded into the pick kernel (SPP_FUSE_SCATTER; auto = 1: where a pick
                               workgroup's edges give bucket runs of >= 8 pairs; 2: wherever the kernel can; off: never) */
  int32_t flag_tiled;       /* flag pass over the scatter's tiles (SPP_FLAG_TILED; auto: on)                            */
  int32_t rows_coalesced;   /* rows' per-edge arrays staged through LDS for fanouts <= 28 (SPP_ROWS_COALESCED; auto: on) */
  int32_t dedup_preread;    /* candidates pre-read the LDS table before their compare-and-swap (SPP_DEDUP_PREREAD;
                               auto: off)                                                                              */
  int64_t fuse_max_edges;   /* a hop of more edges per batch is never fused (0: SPP_FUSE_MAX_EDGES, default 262144)     */
  int64_t initial_edge_cap; /* all-neighbour hops (fanout < 0): starting capacity of the per-edge scratch, grown on
                               demand (0: min(nnz, 4 Mi))                                                              */
  int64_t reserved[4];      /* zero                                                                                    */
} spp_sampler_opts;

typedef struct spp_sampler_cfg {
  const int64_t* rowptr_dev;   /* int64[num_nodes+1], HBM resident           */
  const int64_t* col_dev;      /* int64[nnz], HBM resident                   */
  int64_t num_nodes;
  int64_t nnz;
  int32_t num_hops;            /* len(sizes) <= SPP_MAX_HOPS                 */
  int64_t sizes[SPP_MAX_HOPS]; /* fanouts, e.g. {15,10,5}; <0 = all neighbours */
  int64_t max_batch;           /* max number of seeds in one batch           */
  int32_t num_slots;           /* independent batches that may be in flight  */
  int32_t device;              /* HIP device ordinal                         */
  int32_t replace;             /* 1: sample WITH replacement (sample_cpu.hpp:74-82; only the free
                                  sample_adj exposes it, multilayer_sample passes false) */
  spp_partition_cfg part;      /* part.num_parts = 0: plain (single-GPU) batches */
  int64_t graph_generation;    /* The tables derived from (rowptr_dev, col_dev) -- the degree-tagged int32 neighbour
                                  array and the row stubs -- are shared by the samplers created over the same arrays AND
                                  the same generation.  A caller that rewrites the graph in place, or frees it and may get
                                  another graph of equal size at the same address, passes a new value (0 is fine for a
                                  graph that never changes). */
  spp_sampler_opts opts;       /* chain variants (zero-filled = automatic) */
} spp_sampler_cfg;

/* What a sampler actually runs -- the outcome of spp_sampler_opts' automatic rules -- and what its one-off tables
 * cost.  The per-hop arrays are in PROCESSING order (hop 0 = the seeds' neighbours). */
typedef struct spp_sampler_info {
  int32_t col32;               /* 1: the kernels read the int32 neighbour array                                        */
  int32_t deg_tags;            /* 1: degree pass from the nodes' tags (implies col32 and row_stubs)                    */
  int32_t row_stubs;           /* 1: row-stub table in use                                                             */
  int32_t rng_arena;           /* 1: epoch arena, 0: per-group generation, -1: no Session has decided yet              */
  int32_t idbits, tag_cap;     /* bits of a node id inside an int32 entry (32: untagged), largest degree a tag holds   */
  int32_t num_hops;
  int32_t dedup_buckets_log2;  /* finest bucket count of the radix-partitioned dedup                                   */
  int32_t dedup_table_slots;   /* LDS table of k_bucket_dedup                                                          */
  int32_t generic[SPP_MAX_HOPS];        /* edge-parallel path (fanout < 0 or > 32)                                     */
  int32_t fused_pick[SPP_MAX_HOPS];     /* k_hop_pick<kFuse> (no k_bucket_scatter launch)                              */
  int32_t flag_tiled[SPP_MAX_HOPS];     /* k_hop_flag_tiled                                                            */
  int32_t rows_coalesced[SPP_MAX_HOPS]; /* k_hop_rows_coalesced                                                        */
  int32_t bucket_log2[SPP_MAX_HOPS];    /* buckets of the hop                                                          */
  /* one-off set-up work: wall-clock milliseconds THIS sampler spent building a table (0 when it found the table of
   * an earlier sampler over the same graph, or has none) and the tables' sizes in bytes */
  double col32_ms, row_stubs_ms, rng_arena_ms;
  int64_t col32_bytes, row_stubs_bytes, rng_arena_bytes;
  int64_t rng_arena_batches;   /* streams the arena holds                                                              */
} spp_sampler_info;

/* counts of one sampled batch; hops in OUTPUT order (outermost first, after the
 * std::reverse at fast_sampler.cpp:224) */
typedef struct spp_mfg_counts {
  int64_t num_nodes;              /* U = len(n_id)                           */
  int64_t num_seeds;
  int32_t num_hops;
  int64_t T[SPP_MAX_HOPS];        /* target rows of hop                      */
  int64_t S[SPP_MAX_HOPS];        /* source nodes of hop                     */
  int64_t E[SPP_MAX_HOPS];        /* sampled edges of hop                    */
  int64_t draws;                  /* RNG outputs consumed                    */
  /* with spp_partition_cfg.num_parts = P > 0: [0,P) nodes owned by partition m (cache hits
   * excluded when use_cache), [P] cache hits, [P+1] reserved (host-resident local rows: always 0,
   * every local row lives in HBM); else zeros */
  int64_t part_counts[SPP_MAX_PARTS + 2];
} spp_mfg_counts;

/* caller-owned, exact-size destination buffers for one batch's MFG */
typedef struct spp_mfg_out {
  int64_t* n_id;                  /* int64[U]                                */
  int64_t* rowptr[SPP_MAX_HOPS];  /* int64[T_h+1], output order              */
  int64_t* col[SPP_MAX_HOPS];     /* int64[E_h],   output order              */
  /* bucketing outputs (NULL = skip; ignored when the sampler has no spp_partition_cfg):          */
  int64_t* parts;                 /* int64[sum part_counts[0..P)]: global node ids grouped by owner,
                                     MFG order inside a group (fast_sampler.cpp:1063-1068)        */
  int64_t* cached;                /* int64[part_counts[P]]: cache rows of the hits (:1256)         */
  int64_t* perm;                  /* int64[U]: perm_partition_to_mfg (:1085, :1246-1252)           */
  /* Row references (opt-in; spp_session_export / spp_session_export_group with x_out_dev == NULL): instead of
   * writing the batch's feature matrix the delivery writes WHERE every row lives, for a consumer whose first layer
   * reads the rows exactly once (spp_sage_operand_forward_rows) -- x = cat(...)[perm] (transferers.py:472-486) is
   * neither written nor read back.  row_addr[j] = device address of the feature row of MFG node j: in the local
   * partition (or the one resident table of a single-GPU session), in the VIP cache, in a peer's partition (P2P
   * transport), or -- RCCL transport -- in x_remote, into which the rows received for THIS batch are copied (dense
   * rows, peer-major, request order: U - part_counts[rank] - part_counts[P] of them); the group's receive buffer is
   * free again when the launch completes, so nothing outlives the delivery but caller-owned memory and the
   * resident tables. */
  int64_t* row_addr;              /* int64[U], or NULL                                             */
  void* x_remote;                 /* [remote rows, row_bytes] dense, or NULL (nothing remote / P2P) */
} spp_mfg_out;

/* rowptr_dev / col_dev must stay valid AND unchanged for the sampler's lifetime: an int32 copy of the
 * neighbour array and a row-stub table (degree, row start and first neighbours of every node in one
 * 128-byte record; SPP_ROW_STUBS=0 disables it, it is skipped when HBM is short) are derived from them
 * once and shared by the samplers created over the same arrays and graph_generation
 * (spp_sampler_workspace_bytes includes them and the mt19937 arena).  The cache map of a spp_partition_cfg may
 * be rewritten between Sessions (its membership bits are rebuilt by spp_session_create). */
spp_status spp_sampler_create(const spp_sampler_cfg* cfg, spp_sampler** out);
void spp_sampler_destroy(spp_sampler* s);
/* bytes of HBM workspace held by the sampler (all slots) */
int64_t spp_sampler_workspace_bytes(const spp_sampler* s);
/* A persistent HIP stream owned by the sampler that the consumer may use for spp_session_export /
 * spp_sampler_export / spp_sampler_gather (as a hipStream_t).  Using it keeps the per-batch delivery
 * kernel on a hardware queue of its own; any other stream works too. */
void* spp_sampler_deliver_stream(spp_sampler* s);
/* the configuration the sampler was created with */
spp_status spp_sampler_get_cfg(const spp_sampler* s, spp_sampler_cfg* out);
/* the chain variants it runs and the cost of its one-off tables (BLOCKING when the epoch arena is still being
 * generated: rng_arena_ms is that launch's duration) */
spp_status spp_sampler_get_info(spp_sampler* s, spp_sampler_info* out);

/* Enqueue the sampling of one batch into `slot` on `stream`.
 * seeds_dev: int64[n_seeds] in HBM.  The RNG stream is mt19937(rng_seed) with
 * the first rng_skip outputs discarded (rng_skip = 0 for Session batches). */
spp_status spp_sampler_sample(spp_sampler* s, int32_t slot, const int64_t* seeds_dev,
                              int64_t n_seeds, uint32_t rng_seed, int64_t rng_skip, void* stream);
/* BLOCKING: waits until the batch in `slot` is sampled and returns its counts. */
spp_status spp_sampler_wait(spp_sampler* s, int32_t slot, spp_mfg_counts* out);
/* Write the slot's MFG into caller buffers (int64, reference layout) on `stream`. */
spp_status spp_sampler_export(spp_sampler* s, int32_t slot, const spp_mfg_out* out, void* stream);
/* Fused serial_index over the slot's node list (worker lines fast_sampler.cpp:1006-1010):
 *   dst[i,:] = src[n_id[i],:] for i < n_rows   (n_rows = U for x, batch size for y) */
spp_status spp_sampler_gather(spp_sampler* s, int32_t slot, const void* src_dev, int64_t src_rows,
                              int64_t row_bytes, int64_t src_stride_bytes /* 0 = dense */, int64_t n_rows,
                              void* dst_dev, void* stream);

/* ------------------------------------------------------------------------- *
 * a10/a11  RangePartitionBook (range_partition_book.cpp:85-112) and Cache
 *          (range_partition_book.cpp:116-195) lookups on device.
 *          offsets_host: int64[P+1] on the HOST (P <= SPP_MAX_PARTS).
 *          cache_map_dev: int32[cache_map_len] direct map, -1 = not cached.
 * ------------------------------------------------------------------------- */
spp_status spp_nid2partid(const int64_t* offsets_host, int32_t n_offsets, const int64_t* nids_dev,
                          int64_t n, int64_t* out_dev, void* stream);
spp_status spp_cache_build_map(const int64_t* cached_vertices_dev, int64_t n_cached,
                               int32_t* cache_map_dev, int64_t cache_map_len, void* stream);
spp_status spp_cache_lookup(const int32_t* cache_map_dev, int64_t cache_map_len,
                            const int64_t* nids_dev, int64_t n, uint8_t* is_cached_dev /*nullable*/,
                            int64_t* cache_nid_dev /*nullable*/, void* stream);

/* ------------------------------------------------------------------------- *
 * a8/a9  per-batch ownership bucketing of the MFG node list
 *        (worker distributed branch, fast_sampler.cpp:1017-1262).
 *   parts_out_dev  int64[U]   concat(partition_nids[0..P-1])
 *   cached_out_dev int64[U]   cache-local indices of hits (first counts[P] valid)
 *   perm_out_dev   int64[U]   perm_partition_to_mfg
 *   counts_out_dev int64[P+2] [len(parts[0..P-1]), n_cached, n_local_on_host]
 *   cpu_local_out_dev int64[U] nullable: (v-off[rank])-x_gpu_rows of local rows
 *                     living in host memory, MFG order
 *   workspace: spp_partition_workspace_bytes(U) bytes of HBM scratch.
 * ------------------------------------------------------------------------- */
int64_t spp_partition_workspace_bytes(int64_t max_nodes);
spp_status spp_partition_batch(const int64_t* n_id_dev, int64_t U, const int64_t* offsets_host,
                               int32_t P, int32_t rank, int32_t use_cache,
                               const int32_t* cache_map_dev, int64_t cache_map_len,
                               int64_t x_gpu_rows, int64_t* parts_out_dev, int64_t* cached_out_dev,
                               int64_t* perm_out_dev, int64_t* counts_out_dev,
                               int64_t* cpu_local_out_dev, void* workspace_dev,
                               int64_t workspace_bytes, void* stream);

/* a16 (transferers.py:472-486) fused final assembly, replacing zeros+scatter+cat+permute:
 *   x_out[i,:] = row perm[i] of the virtual concatenation
 *                [ parts[0] rows | ... | parts[P-1] rows | cache rows ]
 * where the rank-th segment is gathered straight from x_local (local id =
 * n_id[i]-off[rank]), the cache segment from cache_feats[cached_nids], and every
 * other segment m from recv_dev (rows received from peer m, packed in partition
 * order with the own-rank segment absent). seg_start_host: int64[P+2] prefix of
 * segment lengths in the virtual concatenation. */
spp_status spp_assemble_features(const int64_t* n_id_dev, const int64_t* perm_dev, int64_t U,
                                 const int64_t* seg_start_host, int32_t P, int32_t rank,
                                 int64_t rank_offset, const void* x_local_dev, int64_t x_local_rows,
                                 const void* recv_dev, const void* cache_feats_dev,
                                 const int64_t* cached_nids_dev, int64_t row_bytes,
                                 int64_t x_local_stride_bytes /* 0 = dense */,
                                 int64_t cache_stride_bytes /* 0 = dense */,
                                 const int64_t* recv_base_host /* int64[P] or NULL: row of recv_dev where
                                    peer m's rows for this batch start, when recv_dev holds the rows of
                                    several batches (one exchange per group) */,
                                 void* x_out_dev, void* stream);

/* ------------------------------------------------------------------------- *
 * a6/a7  Session runtime (fast_sampler.cpp:533-936 Session, :963-1016 worker,
 *        non-distributed branch).  The reference's CPU worker pool + MPMC queue
 *        + semaphore back-pressure is replaced by `max_items_in_queue` batch
 *        slots kept in flight on HIP streams; batches are delivered in index
 *        order.  Batch ranges and per-batch seeds follow fast_sampler.cpp:587-627
 *        and :994 exactly.
 * ------------------------------------------------------------------------- */
typedef struct spp_session spp_session;

typedef struct spp_session_cfg {
  const int64_t* rowptr_dev;
  const int64_t* col_dev;
  int64_t num_nodes;
  int64_t nnz;
  const int64_t* idx_dev;          /* int64[n_idx] seed nodes of this epoch, HBM */
  int64_t n_idx;
  int64_t batch_size;
  int32_t num_hops;
  int64_t sizes[SPP_MAX_HOPS];
  int32_t skip_nonfull_batch;
  int32_t force_exact_num_batches;
  int64_t exact_num_batches;
  int32_t max_items_in_queue;      /* upper bound on batches in flight (slots) */
  int32_t group_size;              /* batches sampled per launch sequence, at most 16 (0 = auto: max_items/4 capped at
                                      16 when 32 or more slots are allowed, else max_items/2 capped at 8);
                                      max_items_in_queue / group_size slot-sets (at most 8) are in flight and share
                                      the sampler's two sampling streams */
  int32_t device;
  /* Optional: borrow an existing sampler (same graph, fanouts; max_batch and num_slots at least
   * what this epoch needs) instead of allocating workspace per epoch -- the counterpart of the
   * reference's process-global worker pool that outlives Sessions (fast_sampler.cpp:512-513).
   * A borrowed sampler is not destroyed by spp_session_destroy. */
  spp_sampler* sampler;
  /* Optional ownership bucketing (NULL = off).  With a borrowed sampler it must equal the
   * sampler's own spp_partition_cfg. */
  const spp_partition_cfg* part;
  /* Optional native feature exchange (NULL = off; needs `part`).  See spp_exchange_cfg below. */
  const struct spp_exchange_cfg* exchange;
  /* Ordering of the session's own streams against the producer of its device inputs.  idx_dev (and
   * the feature / label / cache tables handed to export or spp_exchange_cfg) may still be the output
   * of work QUEUED on the caller's stream (e.g. a shuffle kernel writing this epoch's seed ids): with
   * order_after_input_stream != 0 every stream the session launches on first waits for what
   * `input_stream` (a hipStream_t, NULL = the null stream) holds at creation time.  0 = the inputs
   * are complete already. */
  void* input_stream;
  int32_t order_after_input_stream;
} spp_session_cfg;

typedef struct spp_batch_desc {
  int64_t batch_index;
  int32_t start, stop;             /* idx range, as PreparedSample's pair      */
  int32_t slot;
  spp_mfg_counts counts;
} spp_batch_desc;

spp_status spp_session_create(const spp_session_cfg* cfg, spp_session** out);
void spp_session_destroy(spp_session* s);
int64_t spp_session_num_total_batches(const spp_session* s);
int64_t spp_session_num_consumed_batches(const spp_session* s);
/* fills ranges (2*num_total_batches int32) -- the table of fast_sampler.cpp:587-627 */
spp_status spp_session_batch_ranges(const spp_session* s, int32_t* out_start_stop);
/* BLOCKING. Returns 1 and fills *out when the next batch (index order) is ready,
 * 0 at end of epoch (the reference returns None), <0 on error.
 * How far ahead the session samples (the counterpart of max_items_in_queue, fast_sampler.cpp:533-586): the slots form
 * `sets` slot-sets of group_size batches; the launcher keeps chains in flight for at most
 *     sets - refill_lag   groups beyond the groups the consumer has FINISHED
 * (refill_lag = SPP_REFILL_LAG, default 1, applied only with >= 3 sets and capped at sets - 2: the chain that refills a
 * freed set is enqueued one group later, for a set whose deliveries completed a group ago).  A group counts as finished
 * when the consumer COMES BACK FOR MORE after exporting its last batch -- at the next spp_session_next /
 * spp_session_try_next / spp_session_next_group -- not at that export: a consumer that exports a group's last batch and then
 * only polls slot events, or calls spp_session_quiesce, gets no refill until it asks for the next batch. */
int spp_session_next(spp_session* s, spp_batch_desc* out);
/* Non-blocking form (try_get_batch, fast_sampler.cpp:658-670): 2 when the next batch is not ready
 * yet, otherwise exactly what spp_session_next returns. */
int spp_session_try_next(spp_session* s, spp_batch_desc* out);
/* Write the batch returned by the last spp_session_next into caller buffers on
 * `stream`: MFG (as spp_sampler_export), optional x = x_src[n_id] and
 * y = y_src[n_id[:stop-start]]; then recycle its slot (the next pending batch
 * starts sampling into it, ordered after these copies). */
spp_status spp_session_export(spp_session* s, const spp_mfg_out* mfg,
                              const void* x_src_dev, int64_t x_rows, int64_t x_row_bytes,
                              int64_t x_src_stride_bytes /* 0 = dense */, void* x_out_dev,
                              const void* y_src_dev, int64_t y_rows, int64_t y_row_bytes, void* y_out_dev,
                              void* stream);
/* Group-at-a-time consumption.  The batches of a sampling group (spp_session_group_size of them; fewer in
 * the epoch's last group) become ready together, and delivering them with ONE launch keeps the delivery
 * stream's hardware queue busy: a launch per batch cost ~25 us of queue idle time around every ~105 us kernel
 * (completion signal, event markers, dispatch ramp), which bounded the whole pipeline.
 *   spp_session_next_group: the descriptors of ALL batches of the next group (index order).  Returns 1 and
 *     fills out[0 .. *n_out) when the group is sampled (and, with the native exchange, exchanged); 0 at the end
 *     of the epoch; 2 when block == 0 and the group is not ready yet; < 0 on error.  With consumer-issued
 *     exchanges (spp_exchange_cfg.issue_on_consumer) this is the program point at which the group's own exchange
 *     -- and, with three or more slot-sets, the next group's -- is issued.
 *   spp_session_export_group: writes the n batches returned by the last spp_session_next_group into caller
 *     buffers with one launch on `stream` (per batch as spp_session_export: MFG, x = x_src[n_id] or the rows
 *     assembled from the exchange, y = y_src[n_id[:stop-start]]), then recycles the group's slot-set.
 *   Fetch as a group, deliver one by one: instead of ONE spp_session_export_group the group returned by
 *     spp_session_next_group may be exported member by member, in index order, by n calls of spp_session_export
 *     (one launch each, on the stream given to each call; the slot-set is recycled after the last).  The caller
 *     then sizes and allocates the outputs of all n batches at once and pays one export call per batch, while the
 *     GPU sees the per-batch launches -- which measured faster than the single launch.  Once a
 *     member has been exported this way spp_session_export_group is refused for that group.
 * spp_session_next and the group calls may be mixed only at group boundaries. */
typedef struct spp_group_out {
  spp_mfg_out mfg;                 /* as spp_session_export's mfg (pointers may be NULL = skip) */
  void* x_out;                     /* [U, x_row_bytes] dense, or NULL                           */
  void* y_out;                     /* [stop - start, y_row_bytes], or NULL                      */
} spp_group_out;
int spp_session_next_group(spp_session* s, int32_t block, spp_batch_desc* out /* [group_size] */, int32_t* n_out);
spp_status spp_session_export_group(spp_session* s, int32_t n, const spp_group_out* outs /* [n] */,
                                    const void* x_src_dev, int64_t x_rows, int64_t x_row_bytes,
                                    int64_t x_src_stride_bytes /* 0 = dense */,
                                    const void* y_src_dev, int64_t y_rows, int64_t y_row_bytes, void* stream);
/* total time spp_session_next spent blocked, microseconds, and number of blocking waits
 * (fast_sampler.cpp:788-799 total_blocked_dur / total_blocked_occasions) */
int64_t spp_session_blocked_us(const spp_session* s);
int64_t spp_session_blocked_occasions(const spp_session* s);
/* batches per sampling group (they become ready together; exchanges are per group) */
int32_t spp_session_group_size(const spp_session* s);
/* the sampler owned by the session (for spp_sampler_gather on the current slot etc.) */
spp_sampler* spp_session_sampler(spp_session* s);

/* ------------------------------------------------------------------------- *
 * e1-e3  Remote-feature exchange over RCCL/xGMI, native (the DeviceDistributedPrefetcher
 *        stages of transferers.py:186-372: counts C1 :757, node ids C2 :709, feature rows C3 :521,
 *        serve K5 :645-658, combine :472-486).  One exchange per GROUP of batches (fewer, larger
 *        collectives), driven by a session-owned thread one group ahead of the consumer:
 *          all-gather of the request counts -> grouped send/recv of int32 node ids ->
 *          row gather out of the local partition -> grouped send/recv of the rows;
 *        spp_session_export then writes x in MFG order straight from {local partition,
 *        received rows, VIP cache} in the same launch that delivers the MFG and the labels.
 *        Every rank must run the same number of batches per epoch (force_exact_num_batches,
 *        as the reference's distributed mode requires).
 * ------------------------------------------------------------------------- */
typedef struct spp_comm spp_comm;

#define SPP_COMM_ID_BYTES 128
/* rank 0: make the rendezvous token (ncclGetUniqueId); ship it to the other ranks out of band
 * (e.g. torch.distributed broadcast) */
spp_status spp_comm_unique_id(void* out_id /* SPP_COMM_ID_BYTES */);
/* collective over all ranks: ncclCommInitRank on `device`.  librccl.so.1 is resolved at run time
 * (the copy PyTorch already loaded when there is one). */
spp_status spp_comm_create(const void* id, int32_t rank, int32_t world, int32_t device, spp_comm** out);
/* `world` communicators that live in ONE process and copy device-to-device (no RCCL): ranks are
 * driven by different host threads.  A TEST AID compiled into the product library (the suite rehearses
 * the exchange logic above the transport on one GPU with it): refused with SPP_ERR_INVALID unless the
 * environment says SPP_ALLOW_LOCAL_COMM=1. */
spp_status spp_comm_create_local(int32_t world, int32_t device, spp_comm** out /* [world] */);
void spp_comm_destroy(spp_comm* c);
int32_t spp_comm_rank(const spp_comm* c);
int32_t spp_comm_world(const spp_comm* c);

typedef struct spp_exchange_cfg {
  spp_comm* comm;                  /* rank / world must match spp_partition_cfg                     */
  const void* x_local_dev;         /* this rank's feature rows [offsets[rank], offsets[rank+1]), HBM */
  int64_t x_local_rows;
  int64_t row_bytes;
  const void* cache_feats_dev;     /* VIP cache rows; NULL without use_cache                         */
  int64_t cache_rows;
  int64_t x_local_stride_bytes;    /* distance between rows of x_local / of the cache (0 = dense)    */
  int64_t cache_stride_bytes;
  /* P2P transport (opt-in; SURVEY 8(e) "direct P2P loads of peer HBM inside the gather kernel"): peer_x_dev != NULL
   * names, for every rank m, the base address IN THIS PROCESS of rank m's partition (its x_local: same row_bytes and
   * stride; peer_x_dev[rank] is ignored) -- plain device pointers when the ranks share a process, otherwise
   * mappings opened with spp_ipc_open from handles the owners exported (spp_ipc_export).  The delivery then reads a
   * remote row straight out of its owner's HBM over xGMI: no id exchange, no serve gather, no send / receive
   * buffers, no host read per group, and `comm` may be NULL.  The partitions must stay allocated and unchanged while
   * any peer's Session runs. */
  const void* const* peer_x_dev;   /* host array [num_parts], or NULL = exchange over `comm`         */
  int64_t peer_x_stride_bytes;     /* distance between rows of every peer's table (0 = x_local's)    */
  int32_t issue_on_consumer;       /* 0: a session thread issues each group's exchange as soon as its
                                      sampling completes (most overlap).  1: spp_session_next issues it,
                                      at the same point of the program on every rank (own group when
                                      needed, the next one from mid-group on) -- use it when the caller
                                      interleaves collectives of its own communicator, e.g. DDP gradient
                                      all-reduces: both communicators' kernels are then queued in the
                                      same order on every rank.                                      */
} spp_exchange_cfg;

/* Cross-process mapping of a device allocation (hipIpcGetMemHandle / hipIpcOpenMemHandle), for the P2P transport:
 * the owner exports the allocation that CONTAINS ptr_dev (handle_out: SPP_IPC_HANDLE_BYTES bytes; *offset_out = ptr_dev's
 * offset inside it), ships both out of band (e.g. torch.distributed all_gather_object), and a peer process on the same
 * node opens the handle and adds the offset.  spp_ipc_close unmaps what spp_ipc_open returned. */
#define SPP_IPC_HANDLE_BYTES 64
spp_status spp_ipc_export(const void* ptr_dev, void* handle_out, int64_t* offset_out);
spp_status spp_ipc_open(const void* handle, int32_t device, void** base_out);
spp_status spp_ipc_close(void* base);

/* BLOCKING: returns when the session's threads have issued everything they can without further
 * consumption (sampling chains and exchanges of the slot-sets in flight) and that work has completed
 * on the GPU.  Call it on every rank before issuing collectives of ANOTHER communicator (e.g. a
 * torch.distributed barrier): kernels of two communicators that wait for their peers must not be
 * queued behind one another in opposite orders on different ranks.  A finished group that has not been reported yet
 * (see spp_session_next: the report is deferred to the consumer's next request) stays unreported: quiesce starts no
 * refill chain of its own. */
spp_status spp_session_quiesce(spp_session* s);
/* bytes this rank sent / received through the exchange so far (ids + rows + counts) */
spp_status spp_session_exchange_stats(const spp_session* s, int64_t* sent_bytes, int64_t* recv_bytes);

/* ------------------------------------------------------------------------- *
 * f1  VIP analytic model (driver/drivers/ddp.py:135-239 get_frequency_tensors_fast): per vertex,
 *     the probability of being touched by one mini-batch of `batch_size` seeds drawn from
 *     train_idx, propagated over `fanouts` (in the order given, as the reference iterates them).
 *     float64.  out_dev: double[num_nodes]; workspace_dev: double[3*num_nodes].
 *     create_vip_cache (ddp.py:417-570) ranks the remote vertices by it.
 * ------------------------------------------------------------------------- */
spp_status spp_vip_frequencies(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_nodes,
                               const int64_t* train_idx_dev, int64_t n_train, int64_t batch_size,
                               const int64_t* fanouts_host, int32_t num_hops, double* out_dev,
                               double* workspace_dev, void* stream);

/* ------------------------------------------------------------------------- *
 * f3  Mean aggregation of SAGEConv over one MFG hop (driver/models.py:19-56: SAGEConv(aggr='mean')
 *     on ((x, x_target), adj_t)); fp32 accumulate and output.
 *       forward : out[t,:] = sum_{e in row t} x[col[e],:] / max(deg t, 1);  x fp32 or fp16, rows
 *                 x_stride_elems apart (the batch's fp16 features can be aggregated directly);
 *                 out rows out_stride_elems apart (0 = dense), so the result can land in the left
 *                 half of the [T, 2F] operand of one fused lin_l|lin_r GEMM
 *       backward: grad_x[col[e],:] += grad_out[t,:] / deg t  (grad_x [S,F] fp32, zeroed by the
 *                 caller; hardware fp32 atomics, so the summation order is not fixed)
 * ------------------------------------------------------------------------- */
spp_status spp_csr_mean_forward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t F,
                                float* out_dev, int64_t out_stride_elems, void* stream);
spp_status spp_csr_mean_backward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                 const float* grad_out_dev, int64_t grad_out_stride_elems, int64_t F,
                                 float* grad_x_dev, void* stream);
/* The fused operand of SAGEConv, [mean_j x_j | x_target] (fp32 [T, 2F], one GEMM with [W_l | W_r] then
 * replaces lin_l(mean) + lin_r(x_target)): the targets are the first T rows of x (the MFG contract,
 * driver/models.py:44-45 `x_target = x[:size[1]]`), converted to fp32 in the same pass. */
spp_status spp_sage_operand_forward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                    const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t F,
                                    float* out_dev, int64_t out_stride_elems /* >= 2F */, void* stream);
/* The same operand straight from the RESIDENT feature table (opt-in consumer of the data path, DESIGN section 5
 * "fused first layer"): row j of the batch is table[n_id[j]] (n_id int64 [S], the batch's node ids as the Session
 * delivers them; table rows table_stride_elems apart, fp16 or fp32), so the batch's feature matrix x = table[n_id]
 * (fast_sampler.cpp:1004-1016 `x = serial_index(x_cpu, n_id)`) is never written and read back -- the rows are summed
 * in the same order as spp_sage_operand_forward sums the rows of a materialised x: the operand is bit-identical.
 * An id outside [0, table_rows) reads row 0. */
spp_status spp_sage_operand_forward_table(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                          const void* table_dev, int32_t table_is_half, int64_t table_stride_elems,
                                          int64_t table_rows, const int64_t* n_id_dev, int64_t F, float* out_dev,
                                          int64_t out_stride_elems /* >= 2F */, void* stream);
/* The same operand from ROW REFERENCES (spp_mfg_out.row_addr): row j of the batch is the F elements at device
 * address row_addr_dev[j] (fp16 or fp32; 8-byte aligned fp16 / 16-byte aligned fp32 rows when F % 4 == 0).  Same rows,
 * same summation order as spp_sage_operand_forward over the assembled x: the operand is bit-identical. */
spp_status spp_sage_operand_forward_rows(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                         const int64_t* row_addr_dev, int32_t rows_are_half, int64_t F, float* out_dev,
                                         int64_t out_stride_elems /* >= 2F */, void* stream);
/* dst[j,:] = the row_bytes bytes at row_addr_dev[j], j < n: the feature matrix itself out of row references
 * (what any consumer other than the fused first layer reads: RowRefs.materialize()). */
spp_status spp_gather_row_refs(const int64_t* row_addr_dev, int64_t n, int64_t row_bytes, void* dst_dev, void* stream);
/* Its backward: grad_x [S, F] is written completely -- rows < T start from the gradient of the x_target
 * half, the others from zero, then the mean's gradient is scattered on top (fp32 atomics). */
spp_status spp_sage_operand_backward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                     int64_t num_sources, const float* grad_out_dev, int64_t grad_out_stride_elems,
                                     int64_t F, float* grad_x_dev, void* stream);
/* The same gradient by GATHER over the transposed hop (built on the fly: count, scan, fill): no fp32
 * atomics -- 42 M of them per step at papers scale run at the chip's atomic rate.  workspace_dev:
 * spp_sage_operand_backward_workspace_bytes(T, S, E) bytes of HBM, 16-byte aligned. */
int64_t spp_sage_operand_backward_workspace_bytes(int64_t num_targets, int64_t num_sources, int64_t num_edges);
spp_status spp_sage_operand_backward_gather(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                            int64_t num_sources, int64_t num_edges, const float* grad_out_dev,
                                            int64_t grad_out_stride_elems, int64_t F, float* grad_x_dev,
                                            void* workspace_dev, int64_t workspace_bytes, void* stream);
/* x = F.relu(x); x = F.dropout(x, p, training)  (driver/models.py:47-48) in one pass; the keep / drop
 * decisions come from a counter-based generator keyed by `seed`, and the backward pass needs only y
 * (y > 0 exactly where the input was positive and kept): grad_x = grad * scale where y > 0, scale = 1/(1-p)
 * in training and 1 in eval mode. */
spp_status spp_relu_dropout_forward(const float* x_dev, int64_t n, float p, int32_t training, uint64_t seed,
                                    float* y_dev, void* stream);
spp_status spp_relu_dropout_backward(const float* grad_dev, const float* y_dev, int64_t n, float scale,
                                     float* grad_x_dev, void* stream);
/* The same two steps without the activated copy: spp_sage_operand_forward_act builds [mean | x_target] of
 * relu_dropout(x) from the PRE-activation x (fp32, dense rows, F % 4 == 0), applying the activation to every
 * row as it is loaded -- the values spp_relu_dropout_forward(x, n = S*F, p, training, seed) would have
 * written -- and spp_relu_dropout_backward_pre recomputes the mask from x and the same (p, training, seed):
 * grad_x = grad / (1-p) where x > 0 and the element was kept (n % 4 == 0). */
spp_status spp_sage_operand_forward_act(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                        const float* x_dev, int64_t F, float* out_dev, int64_t out_stride_elems,
                                        float p, int32_t training, uint64_t seed, void* stream);
spp_status spp_relu_dropout_backward_pre(const float* grad_dev, const float* z_dev, int64_t n, float p,
                                         int32_t training, uint64_t seed, float* grad_x_dev, void* stream);
/* spp_sage_operand_backward_gather with spp_relu_dropout_backward_pre applied to every row before it is
 * stored: grad_x_dev becomes the gradient w.r.t. the pre-activation z_pre_dev (dense fp32 [S, F]). */
spp_status spp_sage_operand_backward_gather_act(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                                int64_t num_sources, int64_t num_edges, const float* grad_out_dev,
                                                int64_t grad_out_stride_elems, int64_t F, float* grad_x_dev,
                                                void* workspace_dev, int64_t workspace_bytes, const float* z_pre_dev,
                                                float p, int32_t training, uint64_t seed, void* stream);

/* GATConv(heads=1) message passing over one MFG hop (driver/models.py:195-231):
 *   e_ij = leaky_relu(a_src[j] + a_dst[i], negative_slope) over row i without its diagonal entry plus
 *   the self loop (i, i) that GATConv adds (set_diag);  out[i,:] = sum_j softmax_j(e_ij) h[j,:].
 *   h fp32 [S,F] dense (targets are its first rows), a_src fp32[S], a_dst fp32[T]; row_max/row_sum
 *   fp32[T] keep the softmax statistics for the backward pass.  Backward: grad_h [S,F] and
 *   grad_a_src [S] zeroed by the caller (fp32 atomics), grad_a_dst [T] written. */
spp_status spp_gat_forward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                           const float* h_dev, int64_t F, const float* a_src_dev, const float* a_dst_dev,
                           float negative_slope, float* out_dev, float* row_max_dev, float* row_sum_dev,
                           void* stream);
spp_status spp_gat_backward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                            const float* h_dev, int64_t F, const float* a_src_dev, const float* a_dst_dev,
                            float negative_slope, const float* out_dev, const float* row_max_dev,
                            const float* row_sum_dev, const float* grad_out_dev, float* grad_h_dev,
                            float* grad_a_src_dev, float* grad_a_dst_dev, void* stream);

/* GATConv in aggregate-then-project form (same function, far less work: see csrc/aggregate.hip).
 * With v_src = W^T att_src, v_dst = W^T att_dst (K-vectors):
 *   spp_gat_logits:            a_src[j] = x_j . v_src (all S rows), a_dst[i] = x_i . v_dst (the first T rows)
 *   spp_gat_aggregate_forward: z_i = sum_j softmax_j(leaky_relu(a_src[j] + a_dst[i])) x_j  over row i of the hop
 *                              (diagonal entry dropped, self loop added, as GATConv's set_diag); the layer's
 *                              output is then z @ W^T.  x rows are fp16 or fp32, K % 4 == 0.
 *   spp_gat_aggregate_backward: from grad_z: grad_a_src [S] (caller zeroes it), grad_a_dst [T], and -- when
 *                              grad_x_dev != NULL (caller zeroes it) -- grad_x[j,:] += alpha_ij grad_z_i.
 *   spp_gat_logits_backward:   grad_v_src[c] = sum_j grad_a_src[j] x[j,c], grad_v_dst[c] = sum_{i<T} grad_a_dst[i] x[i,c]. */
spp_status spp_gat_logits(const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t num_sources,
                          int64_t num_targets, int64_t K, const float* v_src_dev, const float* v_dst_dev,
                          float* a_src_dev, float* a_dst_dev, void* stream);
spp_status spp_gat_logits_backward(const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t num_sources,
                                   int64_t num_targets, int64_t K, const float* grad_a_src_dev,
                                   const float* grad_a_dst_dev, float* grad_v_src_dev, float* grad_v_dst_dev,
                                   void* stream);
spp_status spp_gat_aggregate_forward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                     const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t K,
                                     const float* a_src_dev, const float* a_dst_dev, float negative_slope,
                                     float* z_dev, float* row_max_dev, float* row_sum_dev, void* stream);
spp_status spp_gat_aggregate_backward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                      const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t K,
                                      const float* a_src_dev, const float* a_dst_dev, float negative_slope,
                                      const float* z_dev, const float* row_max_dev, const float* row_sum_dev,
                                      const float* grad_z_dev, float* grad_x_dev, float* grad_a_src_dev,
                                      float* grad_a_dst_dev, void* stream);

/* spp_gat_aggregate_backward with the input gradient by GATHER over the transposed hop (built on the fly: count,
 * scan, fill) instead of E x K fp32 atomics: grad_x_dev [S, K] fp32 is written COMPLETELY, including the rank-1
 * terms of the logits -- grad_a_src[s] v_src for every source and grad_a_dst[s] v_dst for the first T -- that
 * the atomic form leaves to the caller (a_src = x v_src, a_dst = x[:T] v_dst).  grad_a_src_dev [S] is zeroed by the
 * caller as before.  workspace_dev: spp_gat_aggregate_backward_gather_workspace_bytes(T, S, E) bytes, 16-byte aligned. */
int64_t spp_gat_aggregate_backward_gather_workspace_bytes(int64_t num_targets, int64_t num_sources, int64_t num_edges);
spp_status spp_gat_aggregate_backward_gather(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                             int64_t num_sources, int64_t num_edges, const void* x_dev,
                                             int32_t x_is_half, int64_t x_stride_elems, int64_t K,
                                             const float* a_src_dev, const float* a_dst_dev, float negative_slope,
                                             const float* z_dev, const float* row_max_dev, const float* row_sum_dev,
                                             const float* grad_z_dev, const float* v_src_dev, const float* v_dst_dev,
                                             float* grad_x_dev, float* grad_a_src_dev, float* grad_a_dst_dev,
                                             void* workspace_dev, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPP_H */
