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
#endif
