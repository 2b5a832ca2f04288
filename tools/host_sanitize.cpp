// Host entry points of librecommend_amd under AddressSanitizer + UBSan (host code only):
// edge list, CSR construction and the negative sampler on random inputs with exact-size
// buffers, including the empty and the rejected cases.  Built and run by
// tests/test_host_sanitize.py:
//   hipcc -x hip --offload-arch=gfx950 -O1 -g -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
//         tools/host_sanitize.cpp recommend_amd/csrc/host.cpp -o /tmp/host_sanitize
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#include "../include/recommend_amd.h"

#define EXPECT(c)                                                     \
  do {                                                                \
    if (!(c)) {                                                       \
      printf("FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, rk_last_error()); \
      return 1;                                                       \
    }                                                                 \
  } while (0)

int main() {
  std::mt19937_64 rng(7);
  for (int round = 0; round < 20; ++round) {
    const int32_t U = 1 + (int32_t)(rng() % 300), I = 1 + (int32_t)(rng() % 200);
    const int64_t n = round == 0 ? 0 : (int64_t)(rng() % 4000);
    std::vector<int64_t> users(n), items(n);
    std::vector<float> ratings(n);
    for (int64_t j = 0; j < n; ++j) {
      users[j] = (int64_t)(rng() % U);
      items[j] = (int64_t)(rng() % I);
      ratings[j] = (float)(1 + rng() % 5);
    }
    for (int loops = 0; loops < 2; ++loops) {
      const int64_t nnz = 2 * n + (loops ? U + I : 0);
      // (exact sizes, so that one element too many is caught; an empty graph still needs non-NULL
      // output arrays)
      const size_t cap = nnz ? (size_t)nnz : 1;
      std::vector<int32_t> src(cap), dst(cap);
      std::vector<float> w(cap);
      EXPECT(rk_build_coo_host(users.data(), items.data(), ratings.data(), n, U, I, loops, src.data(), dst.data(),
                               w.data()) == RK_OK);
      const int32_t N = U + I;
      std::vector<int32_t> ptr(N + 1), idx(cap), eid(cap);
      EXPECT(rk_csr_from_coo_host(dst.data(), src.data(), nnz, N, ptr.data(), idx.data(), eid.data()) == RK_OK);
      EXPECT(ptr[0] == 0 && ptr[N] == nnz);
      for (int32_t v = 0; v < N; ++v)
        for (int32_t p = ptr[v]; p < ptr[v + 1]; ++p) {
          EXPECT(dst[eid[p]] == v && src[eid[p]] == idx[p]);
          EXPECT(p == ptr[v] || eid[p - 1] < eid[p]);   // stable
        }
    }
    if (n > 0) {   // out-of-range ids are refused, nothing is written past the buffers
      std::vector<int32_t> src(2 * n), dst(2 * n);
      std::vector<float> w(2 * n);
      users[n / 2] = U;
      EXPECT(rk_build_coo_host(users.data(), items.data(), ratings.data(), n, U, I, 0, src.data(), dst.data(),
                               w.data()) == RK_ERR_ARG);
      users[n / 2] = 0;
    }
    // negative sampler: positives of each user as sorted CSR
    std::vector<std::vector<int32_t>> pos(U);
    for (int64_t j = 0; j < n; ++j) pos[users[j]].push_back((int32_t)items[j]);
    std::vector<int32_t> pptr(U + 1, 0), pit;
    for (int32_t u = 0; u < U; ++u) {
      std::sort(pos[u].begin(), pos[u].end());
      pos[u].erase(std::unique(pos[u].begin(), pos[u].end()), pos[u].end());
      if ((int32_t)pos[u].size() == I) pos[u].pop_back();   // keep one negative available
      pit.insert(pit.end(), pos[u].begin(), pos[u].end());
      pptr[u + 1] = (int32_t)pit.size();
    }
    if (pit.empty()) pit.push_back(0);
    std::vector<uint32_t> state(625);
    EXPECT(rk_mt_seed_host(state.data(), 42 + round) == RK_OK);
    const int64_t B = 1 + (int64_t)(rng() % 64);
    std::vector<int64_t> bu(B);
    for (auto &u : bu) u = (int64_t)(rng() % U);
    for (int32_t R : {1, 3, 8}) {
      std::vector<int32_t> rank(B * R);
      std::vector<int64_t> neg(B * R);
      EXPECT(rk_neg_sample_host(state.data(), bu.data(), B, R, pptr.data(), pit.data(), I, rank.data(), neg.data()) == RK_OK);
      for (int64_t k = 0; k < B * R; ++k) EXPECT(neg[k] >= 0 && neg[k] < I);
    }
    EXPECT(rk_neg_sample_host(state.data(), bu.data(), B, 0, pptr.data(), pit.data(), I, nullptr, nullptr) == RK_ERR_ARG);
  }
  printf("host entry points clean under ASan/UBSan\n");
  return 0;
}
