/*
 * ORACLE (test infrastructure only — never linked into libmaro_amd.so).
 *
 * CPython's `random.Random` restated in C: MT19937 with init_by_array seeding, random(),
 * uniform(), getrandbits(k<=32), randint via _randbelow_with_getrandbits.
 * Source of truth: CPython 3.10 Modules/_randommodule.c (genrand_uint32, init_genrand,
 * init_by_array, random_random, random_seed) and Lib/random.py (uniform, randrange, _randbelow).
 * The reference reaches it through maro/simulator/utils/sim_random.py:56-63 (`Random().seed(s)`)
 * and maro/data_lib/cim/utils.py:30-41 (`rand.uniform(-noise, noise)`).
 * Pinned against the interpreter's own `random` module by tests/test_oracle_mt.py.
 */
#ifndef ORACLE_MT19937_H_
#define ORACLE_MT19937_H_
#include <stdint.h>

#define MT_N 624
#define MT_M 397

typedef struct {
  uint32_t mt[MT_N];
  int idx;
} mt_state;

static void mt_init_genrand(mt_state* s, uint32_t seed) {
  s->mt[0] = seed;
  for (int i = 1; i < MT_N; i++) s->mt[i] = 1812433253U * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
  s->idx = MT_N;
}

static void mt_init_by_array(mt_state* s, const uint32_t* key, int key_length) {
  mt_init_genrand(s, 19650218U);
  uint32_t* mt = s->mt;
  int i = 1, j = 0;
  int k = MT_N > key_length ? MT_N : key_length;
  for (; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
    i++; j++;
    if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    if (j >= key_length) j = 0;
  }
  for (k = MT_N - 1; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
    i++;
    if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
  }
  mt[0] = 0x80000000U;
}

/* random.seed(int): key = 32-bit little-endian digits of abs(seed), at least one digit. */
static void mt_seed_int(mt_state* s, int64_t seed) {
  uint64_t a = seed < 0 ? (uint64_t)(-(seed + 1)) + 1u : (uint64_t)seed;
  uint32_t key[2] = {(uint32_t)(a & 0xffffffffu), (uint32_t)(a >> 32)};
  mt_init_by_array(s, key, key[1] ? 2 : 1);
}

static uint32_t mt_genrand_uint32(mt_state* s) {
  static const uint32_t mag01[2] = {0x0U, 0x9908b0dfU};
  uint32_t y;
  uint32_t* mt = s->mt;
  if (s->idx >= MT_N) {
    int kk;
    for (kk = 0; kk < MT_N - MT_M; kk++) {
      y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
      mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 0x1U];
    }
    for (; kk < MT_N - 1; kk++) {
      y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
      mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 0x1U];
    }
    y = (mt[MT_N - 1] & 0x80000000U) | (mt[0] & 0x7fffffffU);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 0x1U];
    s->idx = 0;
  }
  y = mt[s->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680U;
  y ^= (y << 15) & 0xefc60000U;
  y ^= (y >> 18);
  return y;
}

/* random.random(): 53-bit double in [0,1). */
static double mt_random(mt_state* s) {
  uint32_t a = mt_genrand_uint32(s) >> 5, b = mt_genrand_uint32(s) >> 6;
  return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

/* random.uniform(a, b) = a + (b - a) * random()   (Lib/random.py) */
static double mt_uniform(mt_state* s, double a, double b) { return a + (b - a) * mt_random(s); }

/* random.randint(0, n-1) for 0 < n <= 2**31: k = n.bit_length(); r = getrandbits(k) until r < n. */
static uint32_t mt_randbelow(mt_state* s, uint32_t n) {
  int k = 0;
  for (uint32_t t = n; t; t >>= 1) k++;
  uint32_t r = mt_genrand_uint32(s) >> (32 - k);
  while (r >= n) r = mt_genrand_uint32(s) >> (32 - k);
  return r;
}

#endif
