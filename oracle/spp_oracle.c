/*
 * oracle/spp_oracle.c -- TEST INFRASTRUCTURE ONLY.  See spp_oracle.h.
 *
 * CPU restatement (plain C99 + pthreads) of the SALIENT++ fast_sampler hot
 * path.  Pinned bit-for-bit against the compiled, unmodified reference via
 * tests/golden (tests/test_oracle_golden.py).  Never linked by the product.
 */
#define _POSIX_C_SOURCE 200809L
#include "spp_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ */
/* a1: std::mt19937                                                    */
/* ------------------------------------------------------------------ */

/* std::mersenne_twister_engine<uint_fast32_t,32,624,397,31,0x9908b0df,11,0xffffffff,
 * 7,0x9d2c5680,15,0xefc60000,18,1812433253>::seed(value)  (sample_cpu.hpp:11) */
void orc_mt_seed(orc_mt* s, uint32_t seed) {
  s->mt[0] = seed;
  for (int i = 1; i < 624; ++i) {
    uint32_t p = s->mt[i - 1];
    s->mt[i] = 1812433253u * (p ^ (p >> 30)) + (uint32_t)i;
  }
  s->idx = 624;
}

static void mt_twist(orc_mt* s) {
  uint32_t* mt = s->mt;
  for (int k = 0; k < 624; ++k) {
    uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
    uint32_t v = mt[(k + 397) % 624] ^ (y >> 1);
    if (y & 1u) v ^= 0x9908b0dfu;
    mt[k] = v;
  }
  s->idx = 0;
}

uint32_t orc_mt_next(orc_mt* s) {
  if (s->idx >= 624) mt_twist(s);
  uint32_t y = s->mt[s->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

void orc_mt_fill(uint32_t seed, int64_t skip, int64_t n, uint32_t* out) {
  orc_mt s;
  orc_mt_seed(&s, seed);
  for (int64_t i = 0; i < skip; ++i) (void)orc_mt_next(&s);
  for (int64_t i = 0; i < n; ++i) out[i] = orc_mt_next(&s);
}

/* gen.seed(pair.second * 17 + 5)  (fast_sampler.cpp:994); pair.second is int32 */
uint32_t orc_batch_seed(int32_t stop) { return (uint32_t)(stop * 17 + 5); }

/* ------------------------------------------------------------------ */
/* a6: batch ranges (fast_sampler.cpp:587-627)                         */
/* ------------------------------------------------------------------ */
int64_t orc_batch_ranges(int64_t n, int64_t batch_size, int skip_nonfull_batch,
                         int force_exact_num_batches, int64_t exact_num_batches,
                         int32_t* out) {
  int64_t nb = 0;
  if (force_exact_num_batches) {
    int64_t k = exact_num_batches;
    if (k <= 0) return 0;
    /* :593-608 -- avg = n/k - 1, remainder dealt round-robin from index 0 */
    uint64_t avg = (uint64_t)(n / k) - 1;
    int64_t rem = n - (int64_t)avg * k;
    uint64_t* bs = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)k);
    for (int64_t i = 0; i < k; ++i) bs[i] = avg;
    while (rem > 0) {
      for (int64_t i = 0; i < k; ++i) {
        if (rem <= 0) break;
        bs[i]++;
        rem--;
      }
    }
    uint64_t sum = 0;
    for (int64_t i = 0; i < k; ++i) {
      if (out) {
        out[2 * nb] = (int32_t)sum;
        out[2 * nb + 1] = (int32_t)(sum + bs[i]);
      }
      sum += bs[i];
      nb++;
    }
    free(bs);
  } else {
    /* :618-626 */
    for (int64_t i = 0; i < n; i += batch_size) {
      int64_t e = i + batch_size;
      if (e > n) e = n;
      int64_t this_bs = e - i;
      if (skip_nonfull_batch && this_bs < batch_size) continue;
      if (out) {
        out[2 * nb] = (int32_t)i;
        out[2 * nb + 1] = (int32_t)(i + this_bs);
      }
      nb++;
    }
  }
  return nb;
}

/* ------------------------------------------------------------------ */
/* int32 -> int32 open-addressing map (stands in for phmap::flat_hash_map;
 * only insert-if-absent / overwrite semantics are observable)          */
/* ------------------------------------------------------------------ */
typedef struct {
  int32_t* keys;
  int32_t* vals; /* -1 == empty */
  uint64_t cap;  /* power of two */
  uint64_t n;
} i32map;

static void map_init(i32map* m, uint64_t want) {
  uint64_t cap = 64;
  while (cap < want * 2) cap <<= 1;
  m->cap = cap;
  m->n = 0;
  m->keys = (int32_t*)malloc(sizeof(int32_t) * cap);
  m->vals = (int32_t*)malloc(sizeof(int32_t) * cap);
  memset(m->vals, 0xff, sizeof(int32_t) * cap);
}
static void map_free(i32map* m) {
  free(m->keys);
  free(m->vals);
}
static inline uint64_t map_hash(int32_t k) {
  uint64_t x = (uint32_t)k;
  x *= 0x9E3779B97F4A7C15ull;
  return x >> 20;
}
static void map_grow(i32map* m);
/* returns pointer to value slot; *inserted set if the key was absent (value initialised to v) */
static inline int32_t* map_insert(i32map* m, int32_t k, int32_t v, int* inserted) {
  if ((m->n + 1) * 2 > m->cap) map_grow(m);
  uint64_t mask = m->cap - 1;
  uint64_t h = map_hash(k) & mask;
  for (;;) {
    if (m->vals[h] < 0) {
      m->keys[h] = k;
      m->vals[h] = v;
      m->n++;
      *inserted = 1;
      return &m->vals[h];
    }
    if (m->keys[h] == k) {
      *inserted = 0;
      return &m->vals[h];
    }
    h = (h + 1) & mask;
  }
}
static void map_grow(i32map* m) {
  i32map nm;
  map_init(&nm, m->cap);
  for (uint64_t i = 0; i < m->cap; ++i) {
    if (m->vals[i] >= 0) {
      int ins;
      map_insert(&nm, m->keys[i], m->vals[i], &ins);
    }
  }
  map_free(m);
  *m = nm;
}

/* ------------------------------------------------------------------ */
/* growable int32 vector                                               */
/* ------------------------------------------------------------------ */
typedef struct {
  int32_t* d;
  int64_t n, cap;
} i32vec;
static void vec_init(i32vec* v, int64_t cap) {
  v->cap = cap < 16 ? 16 : cap;
  v->n = 0;
  v->d = (int32_t*)malloc(sizeof(int32_t) * (size_t)v->cap);
}
static inline void vec_push(i32vec* v, int32_t x) {
  if (v->n == v->cap) {
    v->cap *= 2;
    v->d = (int32_t*)realloc(v->d, sizeof(int32_t) * (size_t)v->cap);
  }
  v->d[v->n++] = x;
}

/* ------------------------------------------------------------------ */
/* a2-a4: sample_adj / multilayer_sample                               */
/* ------------------------------------------------------------------ */
typedef struct {
  int64_t T, S, E;
  int64_t* rowptr; /* T+1 */
  int64_t* col;    /* E */
} orc_hop;

struct orc_mfg {
  int64_t U;
  int64_t* n_id;
  int n_hops;
  orc_hop* hops; /* output order */
  int64_t draws;
};

static int cmp_i32(const void* a, const void* b) {
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}

/* One hop: the map-taking sample_adj overload (sample_cpu.hpp:25-143).
 * n_ids grows in place; emits out_rowptr / out_col (local ids, each row sorted). */
static void sample_adj_hop(const int64_t* rowptr, const int64_t* col, i32vec* n_ids, i32map* map,
                           int32_t num_neighbors, int replace, orc_mt* rng, orc_hop* out,
                           int64_t* draws) {
  const int64_t T = n_ids->n; /* idx_size: targets = all nodes collected so far (:30) */
  int64_t* out_rowptr = (int64_t*)malloc(sizeof(int64_t) * (size_t)(T + 1));
  out_rowptr[0] = 0;
  i32vec cols; /* concatenation of cols[i]; rows delimited by out_rowptr */
  vec_init(&cols, T * (num_neighbors > 0 ? num_neighbors : 8));
  int32_t* perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)(num_neighbors > 0 ? num_neighbors : 1));

  for (int64_t i = 0; i < T; ++i) { /* expand_neighborhood (:43-65): strictly sequential */
    const int32_t n = n_ids->d[i];
    const int64_t row_start = rowptr[n];
    const int64_t row_end = rowptr[n + 1];
    const int32_t neighbor_count = (int32_t)(row_end - row_start); /* narrowed at the lambda call */
    const int64_t before = cols.n;

#define ADD_NEIGHBOR(P)                                                           \
  do {                                                                            \
    const int64_t e_ = row_start + (int64_t)(P);                                  \
    const int32_t c_ = (int32_t)col[e_];                                          \
    int ins_;                                                                     \
    int32_t* slot_ = map_insert(map, c_, (int32_t)n_ids->n, &ins_); /* (:54) */   \
    if (ins_) vec_push(n_ids, c_);                                   /* (:55-57)*/\
    vec_push(&cols, *slot_);                                         /* (:59) */  \
  } while (0)

    if (num_neighbors < 0) { /* no sampling (:67-73) */
      for (int32_t j = 0; j < neighbor_count; ++j) ADD_NEIGHBOR(j);
    } else if (replace) { /* with replacement (:74-82) */
      if (neighbor_count > 0) {
        for (int32_t j = 0; j < num_neighbors; ++j) {
          uint32_t r = orc_mt_next(rng);
          (*draws)++;
          ADD_NEIGHBOR((int32_t)(r % (uint32_t)neighbor_count));
        }
      }
    } else { /* Robert Floyd without replacement (:83-112) */
      if (neighbor_count <= num_neighbors) {
        for (int32_t j = 0; j < neighbor_count; ++j) ADD_NEIGHBOR(j);
      } else {
        int np = 0;
        for (int32_t j = neighbor_count - num_neighbors; j < neighbor_count; ++j) {
          const int32_t option = (int32_t)(orc_mt_next(rng) % (uint32_t)j); /* (:99) */
          (*draws)++;
          int found = 0;
          for (int q = 0; q < np; ++q) /* std::find over perm (:101) */
            if (perm[q] == option) {
              found = 1;
              break;
            }
          const int32_t winner = found ? j : option;
          perm[np++] = winner;
          ADD_NEIGHBOR(winner);
        }
      }
    }
#undef ADD_NEIGHBOR
    out_rowptr[i + 1] = out_rowptr[i] + (cols.n - before); /* (:63) */
  }

  const int64_t E = out_rowptr[T];
  int64_t* out_col = (int64_t*)malloc(sizeof(int64_t) * (size_t)(E > 0 ? E : 1));
  for (int64_t i = 0; i < T; ++i) { /* per-row std::sort + flatten (:123-139) */
    int64_t a = out_rowptr[i], b = out_rowptr[i + 1];
    qsort(cols.d + a, (size_t)(b - a), sizeof(int32_t), cmp_i32);
    for (int64_t k = a; k < b; ++k) out_col[k] = cols.d[k];
  }
  free(cols.d);
  free(perm);
  out->T = T;
  out->S = n_ids->n;
  out->E = E;
  out->rowptr = out_rowptr;
  out->col = out_col;
}

orc_mfg* orc_multilayer_sample(const int64_t* rowptr, const int64_t* col, const int64_t* seeds,
                               int64_t n_seeds, const int64_t* sizes, int n_sizes, orc_mt* rng) {
  orc_mfg* m = (orc_mfg*)calloc(1, sizeof(orc_mfg));
  i32vec n_ids;
  vec_init(&n_ids, n_seeds * 4);
  for (int64_t i = 0; i < n_seeds; ++i) vec_push(&n_ids, (int32_t)seeds[i]); /* :196-199 */
  /* get_initial_sample_adj_hash_map (sample_cpu.hpp:13-19): n_id_map[n_ids[i]] = i,
   * i.e. a duplicated seed keeps its LAST position. */
  i32map map;
  map_init(&map, (uint64_t)n_seeds * 4 + 64);
  for (int64_t i = 0; i < n_seeds; ++i) {
    int ins;
    int32_t* slot = map_insert(&map, n_ids.d[i], (int32_t)i, &ins);
    *slot = (int32_t)i;
  }
  m->n_hops = n_sizes;
  m->hops = (orc_hop*)calloc((size_t)(n_sizes > 0 ? n_sizes : 1), sizeof(orc_hop));
  for (int h = 0; h < n_sizes; ++h) { /* :207-215 */
    /* std::reverse (:224): hop h lands at output index n_sizes-1-h */
    sample_adj_hop(rowptr, col, &n_ids, &map, (int32_t)sizes[h], 0, rng,
                   &m->hops[n_sizes - 1 - h], &m->draws);
  }
  m->U = n_ids.n;
  m->n_id = (int64_t*)malloc(sizeof(int64_t) * (size_t)(m->U > 0 ? m->U : 1));
  for (int64_t i = 0; i < m->U; ++i) m->n_id[i] = (int64_t)n_ids.d[i]; /* :219-222 */
  free(n_ids.d);
  map_free(&map);
  return m;
}

orc_mfg* orc_sample_adj(const int64_t* rowptr, const int64_t* col, const int64_t* idx,
                        int64_t n_idx, int32_t num_neighbors, int replace, orc_mt* rng) {
  orc_mfg* m = (orc_mfg*)calloc(1, sizeof(orc_mfg));
  i32vec n_ids;
  vec_init(&n_ids, n_idx * 4);
  for (int64_t i = 0; i < n_idx; ++i) vec_push(&n_ids, (int32_t)idx[i]); /* sample_cpu.hpp:157-159 */
  i32map map;
  map_init(&map, (uint64_t)n_idx * 4 + 64);
  for (int64_t i = 0; i < n_idx; ++i) {
    int ins;
    int32_t* slot = map_insert(&map, n_ids.d[i], (int32_t)i, &ins);
    *slot = (int32_t)i;
  }
  m->n_hops = 1;
  m->hops = (orc_hop*)calloc(1, sizeof(orc_hop));
  sample_adj_hop(rowptr, col, &n_ids, &map, num_neighbors, replace, rng, &m->hops[0], &m->draws);
  m->U = n_ids.n;
  m->n_id = (int64_t*)malloc(sizeof(int64_t) * (size_t)(m->U > 0 ? m->U : 1));
  for (int64_t i = 0; i < m->U; ++i) m->n_id[i] = (int64_t)n_ids.d[i];
  free(n_ids.d);
  map_free(&map);
  return m;
}

void orc_mfg_free(orc_mfg* m) {
  if (!m) return;
  for (int h = 0; h < m->n_hops; ++h) {
    free(m->hops[h].rowptr);
    free(m->hops[h].col);
  }
  free(m->hops);
  free(m->n_id);
  free(m);
}
int64_t orc_mfg_num_nodes(const orc_mfg* m) { return m->U; }
const int64_t* orc_mfg_n_id(const orc_mfg* m) { return m->n_id; }
int orc_mfg_num_hops(const orc_mfg* m) { return m->n_hops; }
int64_t orc_mfg_num_draws(const orc_mfg* m) { return m->draws; }
int64_t orc_mfg_hop_T(const orc_mfg* m, int h) { return m->hops[h].T; }
int64_t orc_mfg_hop_S(const orc_mfg* m, int h) { return m->hops[h].S; }
int64_t orc_mfg_hop_E(const orc_mfg* m, int h) { return m->hops[h].E; }
const int64_t* orc_mfg_hop_rowptr(const orc_mfg* m, int h) { return m->hops[h].rowptr; }
const int64_t* orc_mfg_hop_col(const orc_mfg* m, int h) { return m->hops[h].col; }
int64_t orc_mfg_total_edges(const orc_mfg* m) {
  int64_t e = 0;
  for (int h = 0; h < m->n_hops; ++h) e += m->hops[h].E;
  return e;
}

/* ------------------------------------------------------------------ */
/* a5: serial_index (fast_sampler.cpp:238-279)                         */
/* ------------------------------------------------------------------ */
void orc_serial_index(const void* in, int64_t row_bytes, const int64_t* idx, int64_t n_idx,
                      int64_t n, void* out) {
  const char* src = (const char*)in;
  char* dst = (char*)out;
  int64_t m = n_idx < n ? n_idx : n; /* :253 */
  for (int64_t i = 0; i < m; ++i) memcpy(dst + i * row_bytes, src + idx[i] * row_bytes, (size_t)row_bytes);
}

/* to_row_major (fast_sampler.cpp:281-308): outptr[r*tc + c] = inptr[c*tr + r] */
void orc_to_row_major(const void* in, int64_t tr, int64_t tc, int64_t eb, void* out) {
  const char* src = (const char*)in;
  char* dst = (char*)out;
  for (int64_t r = 0; r < tr; ++r)
    for (int64_t c = 0; c < tc; ++c) memcpy(dst + (r * tc + c) * eb, src + (c * tr + r) * eb, (size_t)eb);
}

/* ------------------------------------------------------------------ */
/* a10: RangePartitionBook (range_partition_book.cpp:85-112)           */
/* ------------------------------------------------------------------ */
/* searchsorted(offsets, nid, right=True) - 1  (:98-100) */
static inline int64_t owner_of(const int64_t* offsets, int n_offsets, int64_t v) {
  int64_t cnt = 0;
  for (int k = 0; k < n_offsets; ++k) cnt += (offsets[k] <= v);
  return cnt - 1;
}
void orc_nid2partid(const int64_t* offsets, int n_offsets, const int64_t* nids, int64_t n,
                    int64_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = owner_of(offsets, n_offsets, nids[i]);
}
void orc_nid2localnid(const int64_t* offsets, int p, const int64_t* nids, int64_t n, int64_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = nids[i] - offsets[p]; /* :95-96 */
}
void orc_nid_is_local(const int64_t* offsets, int rank, const int64_t* nids, int64_t n,
                      uint8_t* out) {
  for (int64_t i = 0; i < n; ++i)
    out[i] = (nids[i] >= offsets[rank]) && (nids[i] < offsets[rank + 1]); /* :105-107 */
}

/* ------------------------------------------------------------------ */
/* a11: Cache (range_partition_book.cpp:116-195)                       */
/* ------------------------------------------------------------------ */
struct orc_cache {
  int32_t* map;   /* fast_cached_vertices_map   (:152) */
  uint8_t* inmap; /* fast_cached_vertices_isinmap (:153) */
  int64_t len;    /* the reference hard-codes 200000000 */
};
orc_cache* orc_cache_create(const int64_t* cached_vertices, int64_t n_cached, int64_t table_len) {
  orc_cache* c = (orc_cache*)calloc(1, sizeof(orc_cache));
  c->len = table_len;
  c->map = (int32_t*)calloc((size_t)table_len, sizeof(int32_t));
  c->inmap = (uint8_t*)calloc((size_t)table_len, 1);
  for (int64_t i = 0; i < n_cached; ++i) { /* :154-158: duplicates keep the last index */
    c->map[cached_vertices[i]] = (int32_t)i;
    c->inmap[cached_vertices[i]] = 1;
  }
  return c;
}
void orc_cache_free(orc_cache* c) {
  if (!c) return;
  free(c->map);
  free(c->inmap);
  free(c);
}
void orc_cache_nid_is_cached(const orc_cache* c, const int64_t* nids, int64_t n, uint8_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = c->inmap[nids[i]]; /* :178-181 */
}
void orc_cache_nid2cachenid(const orc_cache* c, const int64_t* nids, int64_t n, int64_t* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = c->map[nids[i]]; /* :190-193 */
}

/* ------------------------------------------------------------------ */
/* a8/a9: distributed worker branch (fast_sampler.cpp:1017-1262)       */
/* ------------------------------------------------------------------ */
int orc_partition_batch(const int64_t* n_id, int64_t U, const int64_t* offsets, int P, int rank,
                        int use_cache, const orc_cache* cache, int64_t x_gpu_rows,
                        int64_t* parts_out, int64_t* part_counts, int64_t* cached_out,
                        int64_t* n_cached, int64_t* perm_out, int64_t* cpu_local_out,
                        int64_t* n_cpu_local) {
  const int n_off = P + 1;
  /* local rows that live in host memory, MFG order (:1041-1052 / :1142-1155) */
  int64_t ncl = 0;
  for (int64_t i = 0; i < U; ++i) {
    int64_t v = n_id[i];
    if (v >= offsets[rank] && v < offsets[rank + 1]) {
      int64_t l = v - offsets[rank];
      if (l >= x_gpu_rows) cpu_local_out[ncl++] = l - x_gpu_rows;
    }
  }
  *n_cpu_local = ncl;

  int64_t* base = (int64_t*)calloc((size_t)P + 2, sizeof(int64_t));
  int64_t* cnt = (int64_t*)calloc((size_t)P + 1, sizeof(int64_t));
  if (!use_cache) {
    /* :1063-1087: machine_id = nid2partid; bincount; stable scatter; perm[i] = off[m] + count[m]++ */
    for (int64_t i = 0; i < U; ++i) cnt[owner_of(offsets, n_off, n_id[i])]++;
    for (int m = 0; m < P; ++m) {
      part_counts[m] = cnt[m];
      base[m + 1] = base[m] + cnt[m];
      cnt[m] = 0;
    }
    for (int64_t i = 0; i < U; ++i) {
      int64_t m = owner_of(offsets, n_off, n_id[i]);
      parts_out[base[m] + cnt[m]] = n_id[i];
      perm_out[i] = base[m] + cnt[m];
      cnt[m]++;
    }
    *n_cached = 0; /* :1106 */
  } else {
    /* :1128-1252.  bucket P == cache hits (concatenated last). */
    for (int64_t i = 0; i < U; ++i) {
      int64_t v = n_id[i];
      int b;
      if (v >= offsets[rank] && v < offsets[rank + 1]) b = rank;
      else if (cache->inmap[v]) b = P;
      else b = (int)owner_of(offsets, n_off, v);
      cnt[b]++;
    }
    for (int m = 0; m <= P; ++m) {
      if (m < P) part_counts[m] = cnt[m];
      base[m + 1] = base[m] + cnt[m];
    }
    *n_cached = cnt[P];
    for (int m = 0; m <= P; ++m) cnt[m] = 0;
    for (int64_t i = 0; i < U; ++i) {
      int64_t v = n_id[i];
      int b;
      if (v >= offsets[rank] && v < offsets[rank + 1]) b = rank;
      else if (cache->inmap[v]) b = P;
      else b = (int)owner_of(offsets, n_off, v);
      if (b < P) parts_out[base[b] + cnt[b]] = v;
      else cached_out[cnt[b]] = cache->map[v]; /* nid2cachenid (:1256) */
      perm_out[i] = base[b] + cnt[b]; /* inverse of the flipped permutation (:1246-1252) */
      cnt[b]++;
    }
  }
  free(base);
  free(cnt);
  return 0;
}

/* ------------------------------------------------------------------ */
/* CPU baseline: one epoch of the non-distributed worker loop          */
/* (fast_sampler.cpp:963-1016) on pthreads                             */
/* ------------------------------------------------------------------ */
typedef struct {
  const int64_t *rowptr, *col, *y, *idx, *sizes;
  const void* x;
  int64_t x_row_bytes;
  const int32_t* ranges;
  int64_t n_batches;
  int n_sizes;
  int64_t next; /* atomic batch cursor */
  pthread_mutex_t mu;
  orc_epoch_stats st;
} epoch_ctx;

static uint64_t fnv64(const void* p, size_t n, uint64_t h) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) {
    h ^= b[i];
    h *= 1099511628211ull;
  }
  return h;
}

static void* epoch_worker(void* arg) {
  epoch_ctx* c = (epoch_ctx*)arg;
  orc_mt rng;
  char* xbuf = NULL;
  size_t xcap = 0;
  int64_t* ybuf = NULL;
  size_t ycap = 0;
  int64_t edges = 0, nodes = 0, nb = 0;
  uint64_t cs = 0;
  for (;;) {
    int64_t b = __atomic_fetch_add(&c->next, 1, __ATOMIC_RELAXED);
    if (b >= c->n_batches) break;
    int32_t start = c->ranges[2 * b], stop = c->ranges[2 * b + 1];
    orc_mt_seed(&rng, orc_batch_seed(stop)); /* :994 */
    orc_mfg* m = orc_multilayer_sample(c->rowptr, c->col, c->idx + start, stop - start, c->sizes,
                                       c->n_sizes, &rng); /* :998 */
    if (c->x) { /* x_s = serial_index(x_cpu, n_id) (:1006) */
      size_t need = (size_t)m->U * (size_t)c->x_row_bytes;
      if (need > xcap) {
        free(xbuf);
        xbuf = (char*)malloc(need);
        xcap = need;
      }
      orc_serial_index(c->x, c->x_row_bytes, m->n_id, m->U, m->U, xbuf);
    }
    if (c->y) { /* y_s = serial_index(y, n_id, this_batch_size) (:1009) */
      size_t need = (size_t)(stop - start);
      if (need > ycap) {
        free(ybuf);
        ybuf = (int64_t*)malloc(need * sizeof(int64_t));
        ycap = need;
      }
      orc_serial_index(c->y, 8, m->n_id, m->U, stop - start, ybuf);
    }
    uint64_t h = 1469598103934665603ull;
    h = fnv64(m->n_id, (size_t)m->U * 8, h);
    for (int k = 0; k < m->n_hops; ++k) h = fnv64(m->hops[k].col, (size_t)m->hops[k].E * 8, h);
    cs += h;
    edges += orc_mfg_total_edges(m);
    nodes += m->U;
    nb++;
    orc_mfg_free(m);
  }
  free(xbuf);
  free(ybuf);
  pthread_mutex_lock(&c->mu);
  c->st.batches += nb;
  c->st.sampled_edges += edges;
  c->st.mfg_nodes += nodes;
  c->st.checksum += cs;
  pthread_mutex_unlock(&c->mu);
  return NULL;
}

int orc_epoch_run(const int64_t* rowptr, const int64_t* col, const void* x, int64_t x_row_bytes,
                  const int64_t* y, const int64_t* idx, const int32_t* ranges, int64_t n_batches,
                  const int64_t* sizes, int n_sizes, int num_threads, orc_epoch_stats* out) {
  epoch_ctx c;
  memset(&c, 0, sizeof(c));
  c.rowptr = rowptr;
  c.col = col;
  c.x = x;
  c.x_row_bytes = x_row_bytes;
  c.y = y;
  c.idx = idx;
  c.ranges = ranges;
  c.n_batches = n_batches;
  c.sizes = sizes;
  c.n_sizes = n_sizes;
  pthread_mutex_init(&c.mu, NULL);
  if (num_threads < 1) num_threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)num_threads);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < num_threads; ++i) pthread_create(&th[i], NULL, epoch_worker, &c);
  for (int i = 0; i < num_threads; ++i) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(th);
  pthread_mutex_destroy(&c.mu);
  c.st.seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  *out = c.st;
  return 0;
}
