// Host-side entry points: library info, edge-list/CSR construction, SpMM row
// binning, and the MT19937 negative sampler.  No device work in this file.
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace rk {

char *err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace rk

extern "C" {

int rk_abi_version(void) { return RK_ABI_VERSION; }

const char *rk_last_error(void) { return rk::err_buf(); }

int rk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// ---------------------------------------------------------------- G1 ------
int rk_build_coo_host(const int64_t *users, const int64_t *items,
                      const float *ratings, int64_t n_inter, int32_t n_users,
                      int32_t n_items, int32_t add_self_loops, int32_t *src,
                      int32_t *dst, float *w) {
  RK_CHECK_ARG(n_inter >= 0 && n_users >= 0 && n_items >= 0, "negative size");
  RK_CHECK_ARG(n_inter == 0 || (users && items), "null interaction arrays");
  RK_CHECK_ARG(src && dst && w, "null output arrays");
  const int64_t n_nodes = (int64_t)n_users + n_items;
  const int64_t nnz = 2 * n_inter + (add_self_loops ? n_nodes : 0);
  RK_CHECK_ARG(nnz < INT32_MAX && n_nodes < INT32_MAX, "graph too large for int32 indices");
  // the reference starts max_rating at the int 0 and takes max(rating, max_rating)
  float max_rating = 0.0f;
  for (int64_t j = 0; j < n_inter; ++j) {
    const int64_t u = users[j], i = items[j];
    RK_CHECK_ARG(u >= 0 && u < n_users, "user id %lld out of range at interaction %lld",
                 (long long)u, (long long)j);
    RK_CHECK_ARG(i >= 0 && i < n_items, "item id %lld out of range at interaction %lld",
                 (long long)i, (long long)j);
    const float r = ratings ? ratings[j] : 1.0f;
    src[2 * j] = (int32_t)u;
    dst[2 * j] = (int32_t)(i + n_users);
    w[2 * j] = r;
    src[2 * j + 1] = (int32_t)(i + n_users);
    dst[2 * j + 1] = (int32_t)u;
    w[2 * j + 1] = r;
    if (r > max_rating) max_rating = r;
  }
  if (add_self_loops) {
    const int64_t base = 2 * n_inter;
    for (int64_t n = 0; n < n_nodes; ++n) {
      src[base + n] = (int32_t)n;
      dst[base + n] = (int32_t)n;
      w[base + n] = max_rating;
    }
  }
  return RK_OK;
}

int rk_csr_from_coo_host(const int32_t *key, const int32_t *other, int64_t nnz,
                         int32_t n_nodes, int32_t *ptr, int32_t *idx,
                         int32_t *eid) {
  RK_CHECK_ARG(nnz >= 0 && nnz < INT32_MAX && n_nodes >= 0, "bad sizes");
  RK_CHECK_ARG(ptr && (nnz == 0 || (key && other && idx && eid)), "null arrays");
  std::fill(ptr, ptr + (size_t)n_nodes + 1, 0);
  for (int64_t e = 0; e < nnz; ++e) {
    const int32_t k = key[e];
    RK_CHECK_ARG(k >= 0 && k < n_nodes, "node id %d out of range at edge %lld", k, (long long)e);
    RK_CHECK_ARG(other[e] >= 0 && other[e] < n_nodes, "node id %d out of range at edge %lld",
                 other[e], (long long)e);
    ptr[k + 1]++;
  }
  for (int32_t v = 0; v < n_nodes; ++v) ptr[v + 1] += ptr[v];
  std::vector<int32_t> cur(ptr, ptr + n_nodes);
  for (int64_t e = 0; e < nnz; ++e) {  // ascending edge id => stable
    const int32_t p = cur[key[e]]++;
    idx[p] = other[e];
    eid[p] = (int32_t)e;
  }
  return RK_OK;
}

// ---------------------------------------------------------------- N -------
// MT19937 exactly as CPython's _randommodule.c uses it.
namespace {

constexpr int MT_N = 624, MT_M = 397;

inline void mt_init_genrand(uint32_t *mt, uint32_t s) {
  mt[0] = s;
  for (int i = 1; i < MT_N; ++i)
    mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  mt[MT_N] = MT_N;
}

inline void mt_init_by_array(uint32_t *mt, const uint32_t *key, size_t klen) {
  mt_init_genrand(mt, 19650218u);
  size_t i = 1, j = 0;
  size_t k = (MT_N > klen ? MT_N : klen);
  for (; k; --k) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    ++i;
    ++j;
    if (i >= MT_N) {
      mt[0] = mt[MT_N - 1];
      i = 1;
    }
    if (j >= klen) j = 0;
  }
  for (k = MT_N - 1; k; --k) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    ++i;
    if (i >= MT_N) {
      mt[0] = mt[MT_N - 1];
      i = 1;
    }
  }
  mt[0] = 0x80000000u;
}

inline uint32_t mt_next(uint32_t *mt) {
  uint32_t &mti = mt[MT_N];
  if (mti >= MT_N) {
    static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
    int kk;
    uint32_t y;
    for (kk = 0; kk < MT_N - MT_M; ++kk) {
      y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
    }
    for (; kk < MT_N - 1; ++kk) {
      y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
    }
    y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
    mti = 0;
  }
  uint32_t y = mt[mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

inline int bit_length(uint32_t n) { return n ? 32 - __builtin_clz(n) : 0; }

// Random._randbelow_with_getrandbits for 0 < n < 2^31
inline uint32_t mt_randbelow(uint32_t *mt, uint32_t n) {
  const int k = bit_length(n);
  uint32_t r = mt_next(mt) >> (32 - k);
  while (r >= n) r = mt_next(mt) >> (32 - k);
  return r;
}

// random.random(): 53-bit double from two outputs
inline double mt_random(uint32_t *mt) {
  const uint32_t a = mt_next(mt) >> 5, b = mt_next(mt) >> 6;
  return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

// rank-th smallest id in [0, n_items) that is not in the sorted list pos[0..len)
inline int64_t kth_missing(const int32_t *pos, int32_t len, int64_t rank) {
  int32_t lo = 0, hi = len;
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if ((int64_t)pos[mid] - mid <= rank)
      lo = mid + 1;
    else
      hi = mid;
  }
  return rank + lo;
}

}  // namespace

}  // extern "C"

namespace {
// CPython set emulation.  `list(all_items - pos)` iterates the result set in
// hash-table slot order.  When len(pos) < len(all_items) >> 2 CPython copies
// all_items (table larger than twice the item count, so a dense id sits in slot
// id) and removes pos: ascending ids.  Otherwise it inserts the surviving ids one
// by one into a fresh table that grows as it fills; ids at or above the table
// size wrap around and the order is no longer ascending.  CandidateOrder
// rebuilds that table (Objects/setobject.c: set_add_entry, set_table_resize,
// set_insert_clean; LINEAR_PROBES 9, PERTURB_SHIFT 5, hash(int) = int).
struct CandidateOrder {
  std::vector<int32_t> key;
  std::vector<uint8_t> used;
  size_t mask = 7, fill = 0;

  void insert_clean(std::vector<int32_t> &k, std::vector<uint8_t> &u, size_t m, int32_t id) {
    size_t perturb = (size_t)id, i = (size_t)id & m;
    while (true) {
      if (!u[i]) { u[i] = 1; k[i] = id; return; }
      if (i + 9 <= m) {
        for (size_t j = 1; j <= 9; ++j)
          if (!u[i + j]) { u[i + j] = 1; k[i + j] = id; return; }
      }
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & m;
    }
  }
  void resize(size_t minused) {
    size_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    std::vector<int32_t> nk(newsize);
    std::vector<uint8_t> nu(newsize, 0);
    for (size_t e = 0; e <= mask; ++e)
      if (used[e]) insert_clean(nk, nu, newsize - 1, key[e]);
    key.swap(nk);
    used.swap(nu);
    mask = newsize - 1;
  }
  void add(int32_t id) {
    size_t perturb = (size_t)id, i = (size_t)id & mask;
    while (true) {
      const size_t probes = (i + 9 <= mask) ? 9 : 0;
      for (size_t j = 0; j <= probes; ++j)
        if (!used[i + j]) {
          used[i + j] = 1;
          key[i + j] = id;
          ++fill;
          if (fill * 5 >= mask * 3) resize(fill > 50000 ? fill * 2 : fill * 4);
          return;
        }
      perturb >>= 5;
      i = (i * 5 + 1 + perturb) & mask;
    }
  }
  // candidate ids of one user in the order list(all_items - pos) yields them
  void build(const int32_t *pos, int32_t len, int32_t n_items, std::vector<int32_t> &out) {
    key.assign(8, 0);
    used.assign(8, 0);
    mask = 7;
    fill = 0;
    int32_t p = 0;
    for (int32_t id = 0; id < n_items; ++id) {
      if (p < len && pos[p] == id) { ++p; continue; }
      add(id);
    }
    out.clear();
    for (size_t e = 0; e <= mask; ++e)
      if (used[e]) out.push_back(key[e]);
  }
};

}  // namespace

extern "C" {

int rk_mt_seed_host(uint32_t *state, uint64_t seed) {
  RK_CHECK_ARG(state, "null state");
  // random.seed(int): key = little-endian 32-bit words of abs(seed), at least one word
  uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  mt_init_by_array(state, key, key[1] ? 2 : 1);
  state[MT_N] = MT_N;
  return RK_OK;
}

int rk_neg_order_host(const int32_t *pos, int32_t len, int32_t n_items, int32_t *order_out) {
  RK_CHECK_ARG(len >= 0 && n_items > 0 && len <= n_items && (pos || len == 0) && (order_out || len == n_items),
               "bad candidate-order arguments");
  std::vector<int32_t> order;
  CandidateOrder emu;
  emu.build(pos, len, n_items, order);
  if (!order.empty()) memcpy(order_out, order.data(), order.size() * sizeof(int32_t));
  return RK_OK;
}

int rk_neg_sample_host(uint32_t *state, const int64_t *users, int64_t B,
                       int32_t R, const int32_t *pos_ptr,
                       const int32_t *pos_items, int32_t n_items,
                       int32_t *rank_out, int64_t *neg_out) {
  return rk_neg_sample_cached_host(state, users, B, R, pos_ptr, pos_items, n_items, nullptr, nullptr, nullptr, 0,
                                   rank_out, neg_out);
}

int rk_neg_sample_cached_host(uint32_t *state, const int64_t *users, int64_t B,
                              int32_t R, const int32_t *pos_ptr,
                              const int32_t *pos_items, int32_t n_items,
                              const int64_t *cached_users, const int64_t *cached_ptr,
                              const int32_t *cached_order, int32_t n_cached,
                              int32_t *rank_out, int64_t *neg_out) {
  RK_CHECK_ARG(state && users && pos_ptr && pos_items, "null input");
  RK_CHECK_ARG(B >= 0 && n_items > 0, "bad sizes");
  RK_CHECK_ARG(n_cached >= 0 && (n_cached == 0 || (cached_users && cached_ptr && cached_order)),
               "incomplete candidate-order cache");
  // graph_data_loader.py:365-366
  RK_CHECK_ARG(R > 0, "Negative ratio must be greater than zero");
  // random.sample: setsize = 21; if k > 5: setsize += 4 ** ceil(log(3k, 4))
  int64_t setsize = 21;
  if (R > 5) {
    int64_t p = 1;
    while (p < 3 * (int64_t)R) p *= 4;
    setsize += p;
  }
  std::vector<int32_t> pool, sel, order;
  CandidateOrder emu;
  for (int64_t s = 0; s < B; ++s) {
    const int64_t u = users[s];
    const int32_t *pos = pos_items + pos_ptr[u];
    const int32_t len = pos_ptr[u + 1] - pos_ptr[u];
    const int64_t n = (int64_t)n_items - len;
    int32_t *rk = rank_out ? rank_out + s * R : nullptr;
    int64_t *ng = neg_out ? neg_out + s * R : nullptr;
    if (n <= 0) {
      // graph_data_loader.py:380-383: candidates = [0]; sample (R == 1) or choices
      for (int32_t r = 0; r < R; ++r) {
        if (R == 1)
          (void)mt_randbelow(state, 1);
        else
          (void)mt_random(state);
        if (rk) rk[r] = -1;  // rank -1: dummy item 0
        if (ng) ng[r] = 0;
      }
      continue;
    }
    // set_difference takes the copy-and-remove path iff (len(all_items) >> 2) > len(pos)
    const bool ascending = (n_items >> 2) > len;
    const int32_t *ord = nullptr;
    if (!ascending) {
      // the emulated table costs a pass over all item ids: taken from the caller's cache when
      // the user is in it (the few users this concerns recur in batch after batch)
      const int64_t *hit = std::lower_bound(cached_users, cached_users + n_cached, u);
      if (hit != cached_users + n_cached && *hit == u) {
        const int64_t at = hit - cached_users;
        RK_CHECK_ARG(cached_ptr[at + 1] - cached_ptr[at] == n, "cached candidate order has the wrong length");
        ord = cached_order + cached_ptr[at];
      } else {
        emu.build(pos, len, n_items, order);
        ord = order.data();
      }
    }
    // j-th candidate of the reference's list -> item id and its ascending rank
    auto emit = [&](int32_t r, int64_t j) {
      int64_t item, rank;
      if (ascending) {
        rank = j;
        item = kth_missing(pos, len, j);
      } else {
        item = ord[(size_t)j];
        rank = item - (std::lower_bound(pos, pos + len, (int32_t)item) - pos);
      }
      if (rk) rk[r] = (int32_t)rank;
      if (ng) ng[r] = item;
    };
    if (n < R) {
      // random.choices(cands, k=R): floor(random() * n)
      for (int32_t r = 0; r < R; ++r) emit(r, (int64_t)(mt_random(state) * (double)n));
    } else if (n <= setsize) {
      pool.resize((size_t)n);
      for (int64_t q = 0; q < n; ++q) pool[(size_t)q] = (int32_t)q;
      for (int32_t r = 0; r < R; ++r) {
        const uint32_t j = mt_randbelow(state, (uint32_t)(n - r));
        const int32_t pick = pool[j];
        pool[j] = pool[(size_t)(n - r - 1)];
        emit(r, pick);
      }
    } else {
      sel.clear();
      for (int32_t r = 0; r < R; ++r) {
        uint32_t j = mt_randbelow(state, (uint32_t)n);
        while (std::find(sel.begin(), sel.end(), (int32_t)j) != sel.end())
          j = mt_randbelow(state, (uint32_t)n);
        sel.push_back((int32_t)j);
        emit(r, j);
      }
    }
  }
  return RK_OK;
}

}  // extern "C"
