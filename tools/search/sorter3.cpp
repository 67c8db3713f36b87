// Search for a cheap 16-input sorting network built from 2-sorters (v_min_f32 + v_max_f32: 2 instructions) and
// 3-sorters (v_min3_f32 + v_med3_f32 + v_max3_f32: 3 instructions), for K1's register sort.
// Evolutionary search in the style of the published sorting-network hunters: mutate a valid network, keep the
// mutant when it still sorts all 2^N 0/1 inputs (0-1 principle) and does not cost more.
//   g++ -O3 -march=native -o /tmp/sorter3 tools/search/sorter3.cpp && /tmp/sorter3 <seed> <seconds> [N] ["start network"]
// N < 0: search for a MERGING network of two sorted halves of |N| / 2 inputs instead (checked on every pair of sorted
// 0/1 halves, (|N|/2 + 1)^2 vectors; start: the last stage of Batcher's odd-even merge sort).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <random>
#include <vector>

static int N = 16;
struct El { int8_t a, b, c; };  // c < 0: 2-sorter (a < b); else 3-sorter (a < b < c)
static inline int cost(const std::vector<El>& n) { int s = 0; for (auto& e : n) s += e.c < 0 ? 2 : 3; return s; }

static std::vector<uint64_t> init_words;  // N wires x W words: wire i of input vector v = bit i of v
static int W;

static bool sorts(const std::vector<El>& net) {
  static thread_local std::vector<uint64_t> w;
  w = init_words;
  for (auto& e : net) {
    uint64_t* A = &w[(size_t)e.a * W]; uint64_t* B = &w[(size_t)e.b * W];
    if (e.c < 0) {
      for (int i = 0; i < W; ++i) { const uint64_t x = A[i], y = B[i]; A[i] = x & y; B[i] = x | y; }
    } else {
      uint64_t* C = &w[(size_t)e.c * W];
      for (int i = 0; i < W; ++i) {
        const uint64_t x = A[i], y = B[i], z = C[i];
        A[i] = x & y & z; C[i] = x | y | z; B[i] = (x & y) | (y & z) | (x & z);
      }
    }
  }
  // sorted ascending: wire i <= wire i+1 for every vector  <=>  (w[i] & ~w[i+1]) == 0
  for (int i = 0; i + 1 < N; ++i) {
    const uint64_t* A = &w[(size_t)i * W]; const uint64_t* B = &w[(size_t)(i + 1) * W];
    for (int k = 0; k < W; ++k) if (A[k] & ~B[k]) return false;
  }
  return true;
}

int main(int argc, char** argv) {
  const unsigned seed = argc > 1 ? atoi(argv[1]) : 1;
  const double seconds = argc > 2 ? atof(argv[2]) : 60;
  if (argc > 3) N = atoi(argv[3]);
  const bool merge_mode = N < 0;
  if (merge_mode) N = -N;
  std::vector<uint64_t> vecs;  // the 0/1 test vectors (bit i = wire i)
  if (merge_mode) {
    const int H = N / 2;
    for (int a = 0; a <= H; ++a)
      for (int b = 0; b <= H; ++b) {  // halves sorted ascending: a (b) ones at the top of the lower (upper) half
        uint64_t v = 0;
        for (int i = H - a; i < H; ++i) v |= 1ull << i;
        for (int i = N - b; i < N; ++i) v |= 1ull << i;
        vecs.push_back(v);
      }
  } else {
    for (uint64_t v = 0; v < ((uint64_t)1 << N); ++v) vecs.push_back(v);
  }
  const size_t V = vecs.size();
  W = (int)((V + 63) / 64);
  init_words.assign((size_t)N * W, 0);
  for (size_t v = 0; v < V; ++v)
    for (int i = 0; i < N; ++i) if ((vecs[v] >> i) & 1) init_words[(size_t)i * W + (v >> 6)] |= 1ull << (v & 63);
  std::mt19937_64 rng(seed * 7919u + 13u);
  auto rnd = [&](int n) { return (int)(rng() % (uint64_t)n); };
  // start: odd-even transposition sort is always valid; bubble-ish start keeps the search unbiased but slow, so use
  // Batcher's odd-even merge sort as the seed network
  std::vector<El> cur;
  for (int p = merge_mode ? N / 2 : 1; p < N; p <<= 1)
    for (int k = p; k >= 1; k >>= 1)
      for (int j = k % p; j + k < N; j += 2 * k)
        for (int i = 0; i < k; ++i)
          if (i + j + k < N && (i + j) / (2 * p) == (i + j + k) / (2 * p)) cur.push_back({(int8_t)(i + j), (int8_t)(i + j + k), -1});
  if (argc > 4) {  // continue from a network given as "(a,b) (a,b,c) ..."
    cur.clear();
    for (const char* p = argv[4]; *p; ++p) {
      if (*p != '(') continue;
      int a = -1, b = -1, c = -1;
      const int n = sscanf(p, "(%d,%d,%d)", &a, &b, &c);
      if (n == 2 || (n == 3 && c < 0)) cur.push_back({(int8_t)a, (int8_t)b, -1});
      else cur.push_back({(int8_t)a, (int8_t)b, (int8_t)c});
    }
  }
  if (!sorts(cur)) { fprintf(stderr, "seed network invalid\n"); return 1; }
  int cc = cost(cur);
  std::vector<El> best = cur; int bc = cc;
  auto norm = [&](El& e) {
    if (e.c < 0) { if (e.a > e.b) std::swap(e.a, e.b); }
    else { int8_t t[3] = {e.a, e.b, e.c}; if (t[0] > t[1]) std::swap(t[0], t[1]); if (t[1] > t[2]) std::swap(t[1], t[2]); if (t[0] > t[1]) std::swap(t[0], t[1]); e.a = t[0]; e.b = t[1]; e.c = t[2]; }
  };
  auto okel = [&](const El& e) { return e.a != e.b && (e.c < 0 || (e.c != e.a && e.c != e.b)); };
  const clock_t t0 = clock();
  uint64_t iters = 0, acc = 0;
  while ((double)(clock() - t0) / CLOCKS_PER_SEC < seconds) {
    ++iters;
    std::vector<El> m = cur;
    const int nm = 1 + (rnd(4) == 0) + (rnd(16) == 0);
    for (int q = 0; q < nm; ++q) {
      const int kind = rnd(8);
      if (m.empty()) break;
      const int i = rnd((int)m.size());
      if (kind == 0) { m.erase(m.begin() + i); }
      else if (kind == 1) {  // 2 -> 3 (add a wire) or 3 -> 2 (drop a wire)
        El e = m[i];
        if (e.c < 0) { e.c = (int8_t)rnd(N); } else { const int d = rnd(3); if (d == 0) e.a = e.c; else if (d == 1) e.b = e.c; e.c = -1; }
        if (!okel(e)) continue; norm(e); m[i] = e;
      } else if (kind == 2) {  // change one wire
        El e = m[i]; const int d = rnd(e.c < 0 ? 2 : 3); const int8_t nw = (int8_t)rnd(N);
        if (d == 0) e.a = nw; else if (d == 1) e.b = nw; else e.c = nw;
        if (!okel(e)) continue; norm(e); m[i] = e;
      } else if (kind == 3) {  // swap with a neighbour
        if (i + 1 < (int)m.size()) std::swap(m[i], m[i + 1]);
      } else if (kind == 4) {  // move an element
        El e = m[i]; m.erase(m.begin() + i); m.insert(m.begin() + rnd((int)m.size() + 1), e);
      } else if (kind == 5) {  // merge two elements sharing a wire into a 3-sorter at the later position
        if (i + 1 >= (int)m.size()) continue;
        const int j = i + 1 + rnd(std::min<int>(8, (int)m.size() - i - 1));
        if (m[i].c >= 0 || m[j].c >= 0) continue;
        int8_t s[4] = {m[i].a, m[i].b, m[j].a, m[j].b}; int8_t u[4]; int nu = 0;
        for (int t = 0; t < 4; ++t) { bool dup = false; for (int r = 0; r < nu; ++r) dup |= u[r] == s[t]; if (!dup) u[nu++] = s[t]; }
        if (nu != 3) continue;
        El e{u[0], u[1], u[2]}; norm(e); m[j] = e; m.erase(m.begin() + i);
      } else if (kind == 6) {  // insert a random 2-sorter (cost up: only survives with a removal in the same step)
        El e{(int8_t)rnd(N), (int8_t)rnd(N), -1}; if (!okel(e)) continue; norm(e); m.insert(m.begin() + rnd((int)m.size() + 1), e);
      } else {  // replace by a random 3-sorter
        El e{(int8_t)rnd(N), (int8_t)rnd(N), (int8_t)rnd(N)}; if (!okel(e)) continue; norm(e); m[i] = e;
      }
    }
    const int mc = cost(m);
    if (mc > cc + 0) continue;
    if (!sorts(m)) continue;
    cur.swap(m); cc = mc; ++acc;
    if (cc < bc) {
      best = cur; bc = cc;
      fprintf(stderr, "[seed %u] %.0fs iters %llu cost %d (%zu elements)\n", seed, (double)(clock() - t0) / CLOCKS_PER_SEC, (unsigned long long)iters, bc, best.size());
    }
  }
  printf("cost %d :", bc);
  for (auto& e : best) { if (e.c < 0) printf(" (%d,%d)", e.a, e.b); else printf(" (%d,%d,%d)", e.a, e.b, e.c); }
  printf("\n");
  return 0;
}
