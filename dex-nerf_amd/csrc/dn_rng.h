// Counter-based random numbers for the training loop's draws (reference: np.random.choice of the pixels,
// train_dexnerf_rgb.py:229-236; torch.rand / torch.randn for the stratified jitter, the resampling u and the density noise,
// nerf/train_utils.py:126-133, nerf/nerf_helpers.py:283-287, nerf/volume_rendering_utils.py:32-38).
//
// The draws are INPUTS of every entry point (parity tests inject the reference's); these generators are what an entry point
// uses when the caller passes NULL for a draw together with an RNG state - a caller-owned device record of four words
//   { seed_lo, seed_hi, cur, nxt }
// `cur` is the iteration counter every kernel of an iteration reads, `nxt` the next one: dn_select_rays_draw (first kernel of
// an iteration) copies nxt to cur, dn_mse2_loss (after the forward) writes nxt = cur + 1 - no kernel reads a word another
// thread of the same launch writes, and a replayed HIP graph advances by itself.
// A value is a pure function of (seed, iteration, stream, element index): the forward and the backward of an iteration see the
// same density noise without it ever being stored.  Philox-4x32-10 (Salmon et al., SC'11), the generator torch uses on devices.
#pragma once
#include <cstdint>

namespace dn {

enum : uint32_t { kRngStreamJitter = 0, kRngStreamNoiseCoarse = 1, kRngStreamU = 2, kRngStreamNoiseFine = 3, kRngStreamPixels = 4, kRngStreamView = 5 };

struct RngRef {
  const uint32_t* state;   // device record {seed_lo, seed_hi, cur, nxt}, or NULL: no in-kernel draws
  uint32_t stream;
};

__host__ __device__ inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c[0];
  const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c[2];
  const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c[3] ^ k1;
  c[1] = static_cast<uint32_t>(p1);
  c[3] = static_cast<uint32_t>(p0);
  c[0] = n0;
  c[2] = n2;
}

// four 32-bit words for (key, counter)
__host__ __device__ inline void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t (&c)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__host__ __device__ inline void rng_words(uint32_t seed_lo, uint32_t seed_hi, uint32_t iteration, uint32_t stream, uint64_t index,
                                          uint32_t (&out)[4]) {
  out[0] = static_cast<uint32_t>(index);
  out[1] = static_cast<uint32_t>(index >> 32);
  out[2] = stream;
  out[3] = iteration;
  philox4x32_10(seed_lo, seed_hi, out);
}

// uniform in [0, 1) with 24 random bits (torch.rand's fp32 resolution)
__host__ __device__ inline float rng_to_uniform(uint32_t w) { return static_cast<float>(w >> 8) * (1.0f / 16777216.0f); }

#ifdef __HIPCC__
__device__ inline float rng_uniform(RngRef r, uint64_t index) {
  uint32_t w[4];
  rng_words(r.state[0], r.state[1], r.state[2], r.stream, index, w);
  return rng_to_uniform(w[0]);
}

// standard normal (Box-Muller on two of the four words; u1 in (0, 1])
__device__ inline float rng_normal(RngRef r, uint64_t index) {
  uint32_t w[4];
  rng_words(r.state[0], r.state[1], r.state[2], r.stream, index, w);
  const float u1 = static_cast<float>((w[0] >> 8) + 1u) * (1.0f / 16777216.0f);
  const float u2 = rng_to_uniform(w[1]);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}
#endif

// A permutation of [0, n) by cycle walking over a 4-round Feistel network on the smallest even number of bits covering n
// (format-preserving: every value in [0, n) is hit exactly once): element i of the iteration's pixel draw = perm(i), so the first
// k elements are k distinct pixels - np.random.choice(n, k, replace=False) without the n-element shuffle.
__host__ __device__ inline uint32_t feistel_permute(uint32_t i, uint32_t n, uint32_t seed_lo, uint32_t seed_hi, uint32_t iteration) {
  uint32_t bits = 2;
  while ((1ull << bits) < n) bits += 2;
  const uint32_t half = bits / 2, mask = (1u << half) - 1u;
  uint32_t x = i;
  do {
    uint32_t l = x >> half, r = x & mask;
#pragma unroll
    for (uint32_t round = 0; round < 4; ++round) {
      // round function: the full generator on (half, round, stream, iteration) - with two Philox rounds per Feistel round the
      // marginals over 4,000 iterations were measurably non-uniform (chi-square twice its expectation); a draw is a few thousand
      // pixels per iteration, so the ~100 multiplies per pixel do not show
      uint32_t c[4] = {r, round, kRngStreamPixels, iteration};
      philox4x32_10(seed_lo, seed_hi, c);
      const uint32_t f = c[0];
      const uint32_t nl = r;
      r = l ^ (f & mask);
      l = nl;
    }
    x = (l << half) | r;
  } while (x >= n);   // cycle walking: re-encrypt until the value falls inside [0, n); expected < 4 rounds (the domain is < 4 n)
  return x;
}

}  // namespace dn
