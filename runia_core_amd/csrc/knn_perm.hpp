// Row order of the kNN bank's bf16 pieces (knn_bf16.hip) and of everything indexed like them.
//
// The candidate filter takes its per-query threshold from the distances to a SAMPLE of the bank, and any prefix of the
// piece rows has to be such a sample whatever k asks for: the bank's runs of 16 consecutive rows are laid out in the order
// t -> (t * g) mod R (R full runs, g coprime to R and next to R / golden ratio), a Kronecker sequence - the first n
// runs of it are spread over the bank with gaps of at most three different lengths (three-distance theorem), for every
// n.  A bank sorted by class is therefore sampled class by class in proportion.  The ragged tail (M mod 16 rows) keeps
// its place, so piece rows >= M are padding and nothing else is.
#pragma once
#include <stdint.h>

struct KnnPerm {
  int64_t full;  // 16 * R: piece rows below it are permuted
  int64_t runs;  // R
  int64_t g;     // 0 = identity
};

__host__ __device__ inline int64_t knn_perm_row(int64_t p, const KnnPerm pm) {
  if (pm.g == 0 || p >= pm.full) return p;
  return (((p >> 4) * pm.g) % pm.runs) * 16 + (p & 15);
}

static inline KnnPerm knn_perm_identity() { return KnnPerm{0, 0, 0}; }
static inline KnnPerm knn_perm_for(int64_t M) {
  const int64_t runs = M / 16;
  if (runs < 2) return knn_perm_identity();
  auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; };
  int64_t g = (int64_t)((double)runs * 0.6180339887498949);
  if (g < 1) g = 1;
  while (gcd(g, runs) != 1) ++g;
  return KnnPerm{16 * runs, runs, g};
}
