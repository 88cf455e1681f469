/*
 * chem_philox.h -- counter-based random streams of the reactive-MD path.
 *
 * Philox4x32-10 (Salmon et al., SC'11; Random123) restated from the paper; pinned by the
 * published known-answer vectors in tests/test_philox.py.  The reference draws from a
 * per-rank esutil.RNG(seed) (start_simulation.py:149) whose streams can never be
 * reproduced; these keyed streams replace it so that results do not
 * depend on the domain decomposition or on the particle order in memory.
 *
 * Header-only; compiles as host code (g++) and as HIP device code.
 */
#ifndef CHEM_PHILOX_H
#define CHEM_PHILOX_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CHEM_HD __host__ __device__ inline
#else
#define CHEM_HD inline
#endif

namespace chem_philox {

CHEM_HD void mulhilo(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo) {
  uint64_t p = (uint64_t)a * (uint64_t)b;
  *hi = (uint32_t)(p >> 32);
  *lo = (uint32_t)p;
}

CHEM_HD void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    mulhilo(0xD2511F53u, c0, &hi0, &lo0);
    mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* uniform in (0,1): (x + 0.5) / 2^32, exact in fp64 */
CHEM_HD double u01(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

/* Langevin noise of particle `tag` for the force evaluation (step, phase).
 * phase 0: evaluation at run() start; phase 1: in-loop evaluation of step index `step`. */
CHEM_HD void langevin_draw(uint64_t seed, uint64_t step, uint32_t phase, uint32_t tag, uint32_t out[4]) {
  uint32_t ctr[4] = {tag, (uint32_t)step, (uint32_t)(step >> 32), 0x4C414E47u ^ phase};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  philox4x32_10(ctr, key, out);
}

/* Reaction acceptance draw for the unordered particle pair (tag_lo < tag_hi), reaction r,
 * at integrator step `step`.  out[0]: acceptance uniform; out[1]: partner-choice hash
 * (random, i.e. non-nearest, mode). */
CHEM_HD void reaction_draw(uint64_t seed, uint64_t step, uint32_t tag_lo, uint32_t tag_hi, uint32_t r, uint32_t out[4]) {
  uint32_t ctr[4] = {tag_lo, tag_hi, (uint32_t)step, (r << 24) ^ (uint32_t)(step >> 32)};
  uint32_t key[2] = {(uint32_t)seed ^ 0x52454143u, (uint32_t)(seed >> 32)};
  philox4x32_10(ctr, key, out);
}

}  /* namespace chem_philox */
#endif
