// Dropout mask shared by the training kernels (csrc/train.hip) and the static-weight GEMM (csrc/pw_gemm.hip): a counter
// hash of (seed, element index) - forward and backward regenerate it, nothing is stored, and the CPU oracle reproduces it
// bit for bit (oracle/train_path.py).
#pragma once
#include "ac_common.h"

namespace {

// ---- dropout mask: splitmix64 of (seed, index); keep iff top 32 bits >= p * 2^32 ------------------------
__device__ __forceinline__ uint32_t drop_hash(uint64_t seed, uint64_t idx) {
  uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}
struct Drop {
  uint64_t seed;
  const unsigned long long* base;  // optional device word: effective seed = seed + (*base << 16), so that a
                                   // captured HIP graph draws fresh masks on every replay
  uint32_t thresh;  // p * 2^32 (0: dropout off)
  float scale;      // 1 / (1 - p)
  __device__ __forceinline__ float mask(uint64_t idx) const {
    if (thresh == 0) return 1.0f;
    const uint64_t s = base ? seed + ((uint64_t)*base << 16) : seed;
    return drop_hash(s, idx) >= thresh ? scale : 0.0f;
  }
};
static Drop make_drop(float p, uint64_t seed, const unsigned long long* base) {
  Drop d;
  d.seed = seed;
  d.base = base;
  if (p <= 0.f) { d.thresh = 0; d.scale = 1.f; }
  else {
    double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    d.scale = 1.0f / (1.0f - p);
  }
  return d;
}

}  // namespace
