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
#include <math.h>

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

/* ATRPActivator (chem_atrp_init): selection key out[0] and acceptance uniform out[1] of particle `tag` at step `step`. */
CHEM_HD void atrp_draw(uint64_t seed, uint64_t step, uint32_t tag, uint32_t out[4]) {
  uint32_t ctr[4] = {tag, (uint32_t)step, (uint32_t)(step >> 32), 0x41545250u};
  uint32_t key[2] = {(uint32_t)seed ^ 0x41435456u, (uint32_t)(seed >> 32)};
  philox4x32_10(ctr, key, out);
}

/* ---- StochasticVelocityRescaling (Bussi, Donadio, Parrinello, J. Chem. Phys. 126, 014101 (2007)) ----
 * One scalar per step: the new kinetic energy drawn from the canonical distribution relaxing with time
 * constant taut (in steps).  The stream is keyed (seed, step); oracle and device run this same code on
 * their own kinetic energy, so the factor agrees to rounding. */
struct SvrStream {
  uint64_t seed, step; uint32_t n; uint32_t buf[4]; int have;
  CHEM_HD SvrStream(uint64_t s, uint64_t st) : seed(s), step(st), n(0), have(0) { buf[0] = buf[1] = buf[2] = buf[3] = 0; }
  CHEM_HD double uniform() {
    if (!have) {
      uint32_t ctr[4] = {n++, (uint32_t)step, (uint32_t)(step >> 32), 0x53565253u};
      uint32_t key[2] = {(uint32_t)seed ^ 0x42555353u, (uint32_t)(seed >> 32)};
      philox4x32_10(ctr, key, buf);
      have = 4;
    }
    return u01(buf[4 - have--]);
  }
  CHEM_HD double gauss() {   /* Box-Muller, one value per two uniforms */
    const double u1 = uniform(), u2 = uniform();
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
  }
  CHEM_HD double gamma(double a) {   /* Gamma(shape a >= 1, scale 1): Marsaglia & Tsang, ACM TOMS 26 (2000) */
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (int it = 0; it < 1000; ++it) {
      const double x = gauss(), t = 1.0 + c * x;
      if (t <= 0.0) continue;
      const double v = t * t * t, u = uniform();
      if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return d * v;
    }
    return d;   /* never reached in practice (acceptance > 95 %) */
  }
  CHEM_HD double sum_noises(int64_t nn) {   /* sum of nn squared unit gaussians = chi^2(nn) = 2 Gamma(nn/2) */
    if (nn <= 0) return 0.0;
    if (nn == 1) { const double g = gauss(); return g * g; }
    return 2.0 * gamma(0.5 * (double)nn);
  }
};

/* velocity scale factor sqrt(K_new / K): K current kinetic energy, K_ref = ndeg kT / 2, taut = coupling / dt */
CHEM_HD double svr_lambda(uint64_t seed, uint64_t step, double K, double K_ref, int64_t ndeg, double taut) {
  SvrStream s(seed, step);
  const double factor = taut > 0.1 ? exp(-1.0 / taut) : 0.0;
  const double rr = s.gauss();
  const double Knew = K + (1.0 - factor) * (K_ref * (s.sum_noises(ndeg - 1) + rr * rr) / (double)ndeg - K)
                    + 2.0 * rr * sqrt(K * K_ref / (double)ndeg * (1.0 - factor) * factor);
  return Knew > 0.0 && K > 0.0 ? sqrt(Knew / K) : 1.0;
}

}  /* namespace chem_philox */
#endif
