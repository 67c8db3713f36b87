// Counter-based uniform draws for the DropBlock layers (throughput mode of the sampler).
// Upstream draws torch.rand(1, H, W) per drop layer on the CPU generator (dropblock==0.3.0, called from
// feature_extraction/abstract_classes.py:93) - a sequential stream no GPU can replay.  Here every draw is a pure
// function of (seed, image, draw index): Philox4x32-10 (Salmon et al., SC'11; known-answer vectors checked in
// tests/test_oracle_goldens.py against the NumPy restatement and on the GPU against runia_mc_draws_f32).
//   draw i of image g (i = layer*H*W + position):  word t = i / 64, lane L = i % 64
//     -> component (t & 3) of philox(counter = (g.lo, g.hi, L + 64*(t >> 2), 0), key = (seed.lo, seed.hi))
//     -> u = (bits >> 8) * 2^-24  in [0, 1)
// so a wave whose lane L owns draws L, 64+L, 128+L, ... needs one Philox block per four of them.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace runia_philox {

struct u4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                     uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u4{c0, c1, c2, c3};
}

__host__ __device__ __forceinline__ float to_uniform(uint32_t bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }

// the Philox block that holds words 4*q .. 4*q+3 of lane `lane` of image `img`; `attempt` (fourth counter word) is 0
// for the draws proper and 1, 2, ... for the redraws of drop layers that removed a whole map (opt-in, K0)
__device__ __forceinline__ u4 lane_block(uint64_t seed, uint64_t img, int lane, int q, uint32_t attempt = 0u) {
  return philox4x32_10((uint32_t)img, (uint32_t)(img >> 32), (uint32_t)(lane + 64 * q), attempt, (uint32_t)seed,
                       (uint32_t)(seed >> 32));
}

__device__ __forceinline__ float component(const u4& b, int j) {
  return to_uniform(j == 0 ? b.x : (j == 1 ? b.y : (j == 2 ? b.z : b.w)));
}

// draw i of image img (generic form: one block per call)
__device__ __forceinline__ float draw(uint64_t seed, uint64_t img, int i) {
  const int t = i >> 6;
  return component(lane_block(seed, img, i & 63, t >> 2), t & 3);
}

}  // namespace runia_philox
