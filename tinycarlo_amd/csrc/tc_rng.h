// Counter-based spawn sampling for device-side auto-reset ("throughput mode").
//
// Not a restatement of anything in the reference -- its resets draw from the env's numpy PCG64 generator
// (map.py:61), which stays available as the seed-parity mode (host-drawn spawn_queue).  Here the k-th re-spawn of env
// e takes output number n = (e << 32 | k) of the SplitMix64 sequence started at `seed` (Steele, Lea, Flood 2014:
// z = seed + (n+1)*0x9E3779B97F4A7C15, then the two xor-shift-multiply rounds), i.e. random access into one
// stream, no per-env generator state besides the reset counter that already exists (spawn_cursor).
// The table holds the candidates of map.py:61 that have an out-edge (duplicates kept), so a uniform draw from it has
// the distribution of the reference's draw-again-on-sinks loop (map.py:62-64).
// Plain C / HIP like tc_trig.h: the CPU test oracle includes this header too (the package never includes the oracle).
#ifndef TC_RNG_H
#define TC_RNG_H
#include <stdint.h>

#include "tc_trig.h" /* TC_HD */

TC_HD uint64_t tc_splitmix64_at(uint64_t seed, uint64_t n) {
  uint64_t z = seed + (n + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// index into a table of `count` (> 0) entries for re-spawn number `cursor` of env `env`:
// high 32 bits scaled by count (Lemire's multiply-shift; bias <= count / 2^32)
TC_HD uint32_t tc_spawn_index(uint64_t seed, uint32_t env, uint32_t cursor, uint32_t count) {
  uint64_t z = tc_splitmix64_at(seed, ((uint64_t)env << 32) | cursor);
  return (uint32_t)(((z >> 32) * (uint64_t)count) >> 32);
}

// One blob of NoiseObservationWrapper (wrapper/observation.py:18-20): centre (x, y) inside the frame, radius in
// [1, max_radius), mode 1 = "copy in" with probability 0.3 (else erase), src = the plane copied from.  Blob k of
// (env, step) takes two outputs of the sub-stream SplitMix64(seed)[env << 32 | step].  The reference draws these
// from the global numpy generator; this is the device-side stand-in with the same distributions (src and mode from
// 8 and 24 bits).
typedef struct {
  int32_t x, y, r, mode, src;
} tc_blob;

TC_HD tc_blob tc_noise_blob(uint64_t seed, uint32_t env, uint32_t step, uint32_t k, int W, int H, int max_radius,
                            int C) {
  const uint64_t base = tc_splitmix64_at(seed, ((uint64_t)env << 32) | step);
  const uint64_t a = tc_splitmix64_at(base, 2ull * k), b = tc_splitmix64_at(base, 2ull * k + 1);
  tc_blob o;
  o.x = (int32_t)(((a >> 32) * (uint64_t)W) >> 32);
  o.y = (int32_t)(((a & 0xffffffffull) * (uint64_t)H) >> 32);
  o.r = 1 + (int32_t)(((b >> 32) * (uint64_t)(max_radius - 1)) >> 32);
  o.mode = ((b >> 8) & 0xffffffull) < 5033165ull ? 1 : 0; /* 0.3 * 2^24 */
  o.src = (int32_t)(((b & 0xffull) * (uint64_t)C) >> 8);
  return o;
}
#endif
