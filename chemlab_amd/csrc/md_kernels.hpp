// md_kernels.hpp -- hand-written CDNA4 (gfx950, wave64) kernels of the reactive-MD inner loop.
//
// Everything is templated on the storage/compute real type R (float: production,
// double: parity mode).  Layout in HBM (SoA of 16/32-byte vectors, cell-sorted order):
//   x4[i] = (x, y, z, (R)type)       v4[i] = (vx, vy, vz, mass)      f4[i] = (fx, fy, fz, 0)
//   tag[i]   : dense particle tag (rank of the external id)          img4[i]: image counters
//   by-tag   : rtag[tag] -> i, state[tag], res_id[tag], mol_id[tag]
//   nlist    : full neighbour list, row-major, row stride S ints, nn[i] entries used
// No MFMA anywhere: there is no dense contraction on this path (pair sums are gathers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chem_mi355.h"
#include "../../include/chem_philox.h"

namespace chem {

template <typename R> struct Vec4T;
template <> struct Vec4T<float> { using type = float4; };
template <> struct Vec4T<double> { using type = double4; };
template <typename R> using Vec4 = typename Vec4T<R>::type;

template <typename R> __device__ __forceinline__ Vec4<R> mk4(R a, R b, R c, R d);
template <> __device__ __forceinline__ float4 mk4<float>(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
template <> __device__ __forceinline__ double4 mk4<double>(double a, double b, double c, double d) { return make_double4(a, b, c, d); }

__device__ __forceinline__ float rint_r(float x) { return rintf(x); }
__device__ __forceinline__ double rint_r(double x) { return rint(x); }
__device__ __forceinline__ float sqrt_r(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_r(double x) { return sqrt(x); }
__device__ __forceinline__ float floor_r(float x) { return floorf(x); }
__device__ __forceinline__ float rcp_r(float x) { return __builtin_amdgcn_rcpf(x); }   // 1 ulp, fp32 production mode
__device__ __forceinline__ double rcp_r(double x) { return 1.0 / x; }
__device__ __forceinline__ double floor_r(double x) { return floor(x); }

constexpr int kMaxTypes = CHEM_MAX_TYPES;
#ifndef CHEM_INT_PER_BLOCK
#define CHEM_INT_PER_BLOCK 512
#endif
constexpr int kIntPerBlock = CHEM_INT_PER_BLOCK;   // particles per workgroup of k_integrate (= entries of the blockmax array per 512 particles)
constexpr int kWave = 64;

// ---- device-resident control block (one per context) ---------------------------------
struct DevCtl {
  unsigned long long step_max2_bits;  // max |dx|^2 of the current step (bits of a non-negative real)
  double acc_maxdist;                 // accumulated sqrt(max|dx|^2) since the last rebuild
  int need_rebuild;                   // decision of the current step (read by the rebuild chain)
  int force_rebuild;                  // host request (topology/exclusions changed)
  int nl_overflow;                    // max neighbours seen when a row overflowed (0 = fine)
  int stage_overflow;                 // stencil tile exceeded the LDS capacity
  int cand_count;                     // reaction candidates appended
  int cand_overflow;
  int rebuild_count;
  int skin_violation;                 // list used with acc_maxdist > skin/2 (must never happen)
  int alive;                          // reaction resolve: pairs still undecided
  int accepted;                       // reaction resolve: accepted events
  long long nlist_entries;
  double step_m2;                     // max |dx|^2 of the step (after the cross-rank max in DD mode)
  int mig_error;                      // a particle left its slab by more than one layer / migration buffer overflow
  int bonded_missing;                 // a bonded partner is neither owned nor a ghost on this rank
  int barrier_timeout;                // fused rebuild: a grid barrier gave up waiting (fatal)
  int pad0;
  double acc_pp[2];                   // fused rebuild: accumulated distance, double-buffered by launch parity
  unsigned long long bw64;            // bonded work list: owners (low word) and entries (high word), one atomic per chunk
  int excl_slot_error;                // list build: an excluded partner was not found in the cell computed from its position (internal)
  int bucket_overflow;                // fused rebuild: a cell holds more particles than a bucket row (value = needed capacity)
  // Rebuild count of the REFERENCE rule (accumulated per-step maxima against the workload's skin/2, reset by every forced
  // rebuild): what chem_get_timers reports.  With an internal list skin wider than the workload's (option list_skin) the
  // lists are rebuilt less often than that; forces do not depend on it (the force kernel applies the exact cutoff).
  int ref_rebuilds;
  double acc_ref;
  // Recoverable stop of an asynchronous run: the fused rebuild met a cell fuller than a bucket row at step `halt_step`.  Every
  // later launch of the run leaves at once (integrate, forces, bonded, rebuild), so the state stays "drifted, forces of
  // halt_step not evaluated"; the host finds the flag at its next synchronisation, rebuilds with wider rows (or the unfused
  // chain) and resumes at that step.
  int halt;
  int bond_slot_miss;                 // list build: a bonded partner was not among the located excluded partners (internal: the host only enables inline bonds when it cannot happen)
  long long halt_step;
};

// bonded tables (per-tag CSR, see K4-K6 below; declared here because the list build records the LDS slots of bonded partners)
constexpr int kBondSlots = 8;   // inline bonds: recorded partner slots per home particle (two 16-byte quads)
struct BondedEntry { int t0, t1, t2, meta; };   // tuple tags in order (self included); meta = slot | mypos<<28; quadruples use a 2nd entry for t3
struct BondedParam { int kind, list, arity, pad; double p[CHEM_MAX_POT_PARAMS]; };

template <typename R> struct Box {
  R L[3], invL[3];
  int nc[3];       // cells per axis (0 => brute-force list build); in z-ghost mode nc[2] = own layers + 2
  int ncell;
  R cell_inv[3];   // cells per unit length (global grid)
  // slab domain decomposition along z (chem_comm_init): layer 0 and nc[2]-1 are ghost layers that
  // hold copies of the neighbour ranks' boundary layers; x and y stay periodic in-kernel
  int zghost;      // 0: z periodic in-kernel, 1: ghost layers
  int z0g, nzg;    // first own global layer, global layer count
  R shz_lo, shz_hi;  // shift to apply to the lower / upper ghost layer (+-Lz across the periodic boundary)
  // Fixed-point positions (R = float only, see "position codec" below): q = rint((x - L/2) / qs), qs = L / 2^31
  double qs[3], qh[3];     // scale and half box edge (fp64: exact decoding)
  R qsf[3], qinvf[3];      // scale and 1 / scale in R (staging, integrator)
  // Tile widths along x (see tile_x0): the first xs_nb tiles of a row are HX cells wide, the others xs_w (1 or 2) cells.
  // xs_nb * HX >= nc[0]: all tiles HX wide.  Narrow tiles are the small change that fills the last round of a force launch.
  int xs_nb, xs_w;
};

// ---- position codec ------------------------------------------------------------------------------------------------
// fp64 build: x4 = (x, y, z, type) in real coordinates.
// fp32 build: x4.xyz carry the BIT PATTERNS of int32 fixed-point coordinates  q = rint((x - L/2) / qs),  qs = L / 2^31
//   (resolution 5e-8 at L = 108, uniform over the box; the box is q in [-2^30, 2^30), int32 leaves half a box of headroom
//   on either side for the drift between two rebuilds; one box length is 2^31, i.e. the periodic shift is a wrap of the
//   32-bit arithmetic).  x4.w stays the type as a float.  Why: an absolute fp32 coordinate at x ~ 100 has an ulp of 7.6e-6,
//   and that -- not the fp32 pair arithmetic -- set the force error at the headline size (2e-4 of the largest force at
//   L = 107.7; SURVEY App. E asks 1e-5).  The kernels that matter stage TILE-LOCAL coordinates: (q + shift - q_ref) is
//   exact in integers and only the small result (|u| <= 7.1) is rounded to fp32.  The integrator adds rint(dt v / qs).
constexpr int kQHalf = 1 << 30;
__device__ __forceinline__ int qbits(float v) { return __float_as_int(v); }
__device__ __forceinline__ float qmake(int q) { return __int_as_float(q); }
struct D3 { double x, y, z; };
// exact real coordinates
__device__ __forceinline__ D3 pos_real(const float4& p, const double* qs, const double* qh) {
  return {(double)qbits(p.x) * qs[0] + qh[0], (double)qbits(p.y) * qs[1] + qh[1], (double)qbits(p.z) * qs[2] + qh[2]};
}
__device__ __forceinline__ D3 pos_real(const double4& p, const double*, const double*) { return {p.x, p.y, p.z}; }
// real coordinates in R (fp32 build: absolute fp32, for the paths where that is enough: fall-back kernels, sub-bin keys)
__device__ __forceinline__ float4 pos_abs(const float4& p, const Box<float>& b) {
  return make_float4((float)((double)qbits(p.x) * b.qs[0] + b.qh[0]), (float)((double)qbits(p.y) * b.qs[1] + b.qh[1]),
                     (float)((double)qbits(p.z) * b.qs[2] + b.qh[2]), p.w);
}
__device__ __forceinline__ double4 pos_abs(const double4& p, const Box<double>&) { return p; }
// cell index along axis d (nc cells), the box being [0, L): exact integer arithmetic in the fp32 build
__device__ __forceinline__ int pos_cell(float c, int d, int nc, const Box<float>&) {
  const long long o = (long long)qbits(c) + kQHalf;
  int cc = (int)((o * nc) >> 31);
  return cc >= nc ? nc - 1 : (cc < 0 ? 0 : cc);
}
__device__ __forceinline__ int pos_cell(double c, int d, int nc, const Box<double>& b) {
  int cc = (int)(c * ((double)nc * b.invL[d]));
  return cc >= nc ? nc - 1 : (cc < 0 ? 0 : cc);
}
// fold coordinate d into the box, counting the images; returns true when the value changed
__device__ __forceinline__ bool pos_fold(float& c, int& img, int d, const Box<float>&) {
  int q = qbits(c);
  if (q >= kQHalf) { q = (int)((unsigned)q + 0x80000000u); img += 1; c = qmake(q); return true; }
  if (q < -kQHalf) { q = (int)((unsigned)q + 0x80000000u); img -= 1; c = qmake(q); return true; }
  return false;
}
__device__ __forceinline__ bool pos_fold(double& c, int& img, int d, const Box<double>& box) {
  bool moved = false;
  double s_ = floor(c * box.invL[d]);
  if (s_ != 0.0) { c -= s_ * box.L[d]; img += (int)s_; moved = true; }
  if (c >= box.L[d]) { c -= box.L[d]; img += 1; moved = true; }
  if (c < 0.0) { c += box.L[d]; img -= 1; moved = true; }
  return moved;
}
// x + dx (integrator drift)
__device__ __forceinline__ float pos_add(float c, float dx, float qinv) { return qmake(qbits(c) + __float2int_rn(dx * qinv)); }
__device__ __forceinline__ double pos_add(double c, double dx, double) { return c + dx; }
// a - b in real units (displacement since a snapshot)
__device__ __forceinline__ float pos_diff(float a, float b, float qs) { return (float)(qbits(a) - qbits(b)) * qs; }
__device__ __forceinline__ double pos_diff(double a, double b, double) { return a - b; }
// encoding of a real coordinate (descriptor tables: reference point of a tile) and of a periodic shift of k box lengths
__device__ __forceinline__ float pos_enc(double x, int d, const Box<float>& b) { return qmake((int)(long long)rint((x - b.qh[d]) / b.qs[d])); }
__device__ __forceinline__ double pos_enc(double x, int d, const Box<double>&) { return x; }
__device__ __forceinline__ float pos_shift(int k, int d, const Box<float>&) { return qmake((int)((unsigned)k << 31)); }     // +-1 box = 2^31: the same wrap
__device__ __forceinline__ double pos_shift(int k, int d, const Box<double>& b) { return (double)k * b.L[d]; }
// tile-local coordinate (p + shift) - ref as a float / double: integers in the fp32 build, only the small result is rounded
__device__ __forceinline__ float pos_local(float p, float shift, float ref, float qs) {
  return (float)(int)((unsigned)qbits(p) + (unsigned)qbits(shift) - (unsigned)qbits(ref)) * qs;
}
__device__ __forceinline__ double pos_local(double p, double shift, double ref, double) { return (p + shift) - ref; }
// minimum-image difference a - b along axis d (fall-back kernels that gather positions through global memory)
__device__ __forceinline__ float pos_delta(float a, float b, int d, const Box<float>& box) {
  int dq = (int)((unsigned)qbits(a) - (unsigned)qbits(b));
  if (dq >= kQHalf) dq = (int)((unsigned)dq + 0x80000000u); else if (dq < -kQHalf) dq = (int)((unsigned)dq + 0x80000000u);
  return (float)dq * box.qsf[d];
}
__device__ __forceinline__ double pos_delta(double a, double b, int d, const Box<double>& box) {
  const double x = a - b;
  return x - box.L[d] * rint(x * box.invL[d]);
}

// Non-bonded parameters of one type pair, pre-multiplied (gromacs_topology.py:715-721,
// doc/topology.rst:12-14).  PairCore is the only thing the LJ fast path touches (one 16-byte
// LDS read per pair in fp32); PairExt holds energy and table terms.
//   kind 1: ff = r^-6 (lj1 r^-6 - lj2) r^-2,  e = r^-6 (e1 r^-6 - e2) + shift
//   kind 2: linear table interpolation, rows (f_k, f_k+1 - f_k, e_k, e_k+1 - e_k)
template <typename R> struct PairCore { R rc2, lj1, lj2, kind; };   // kind as R: 0 none (rc2<0), 1 LJ, 2 table
template <typename R> struct PairExt { R e1, e2, shift, r0, inv_dr; int toff, nrow, pad; };

// ---- helpers -------------------------------------------------------------------------
template <typename R> __device__ __forceinline__ R minimg1(R d, R L, R invL) { return d - L * rint_r(d * invL); }

__device__ __forceinline__ unsigned long long real_bits(float x) { return (unsigned long long)__float_as_uint(x); }
__device__ __forceinline__ unsigned long long real_bits(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ double bits_real_f(unsigned long long b) { return (double)__uint_as_float((unsigned)b); }
__device__ __forceinline__ double bits_real_d(unsigned long long b) { return __longlong_as_double((long long)b); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ unsigned long long lanemask_lt() {
  int l = lane_id();
  return l ? (~0ull >> (64 - l)) : 0ull;
}

// =======================================================================================
// K7/K8  velocity-Verlet halves (start_simulation.py:165-167,780)
// =======================================================================================
constexpr int kFoldSlots = 64;   // words of the atomic displacement fold of the decomposed path (k_integrate, k_pair_tiles guard 2)
template <typename R> struct LangevinP { int on; double kT, gamma, dt; uint64_t seed; uint64_t step; uint32_t phase;
  uint32_t tmask; };   // thermal groups (LangevinThermostat.add_valid_types, start_simulation.py:312-336): bit t set = type t is thermalised; 0 = every type

template <typename R>
__device__ __forceinline__ void langevin_force(const LangevinP<R>& lp, int tag, R mass, R vx, R vy, R vz, R& fx, R& fy, R& fz) {
  uint32_t r[4];
  chem_philox::langevin_draw(lp.seed, lp.step, lp.phase, (uint32_t)tag, r);
  double m = (double)mass;
  double pref = sqrt(24.0 * lp.kT * lp.gamma * m / lp.dt);
  fx += (R)(-lp.gamma * m * (double)vx + pref * (chem_philox::u01(r[0]) - 0.5));
  fy += (R)(-lp.gamma * m * (double)vy + pref * (chem_philox::u01(r[1]) - 0.5));
  fz += (R)(-lp.gamma * m * (double)vz + pref * (chem_philox::u01(r[2]) - 0.5));
}

// mode bits: 1 = second half-kick (integrate2), 2 = first half-kick + drift (integrate1),
// 3 = fused integrate2(step s) + integrate1(step s+1); 4 = add Langevin force to f first
// (thermalize at aftCalcF) and, when mode has no drift..., store f_total back.
// x0 != nullptr: rebuild criterion "displacement" (max |x - x_at_last_build|^2); nullptr: the
// reference's accumulated per-step maxima (max |dt v|^2 of this step)
template <typename R> struct PosScale { R s[3], inv[3]; };   // fixed-point scale of the positions (fp32 build; unused in fp64)
template <typename R, int MODE, bool LANG, bool STOREF>
__global__ __launch_bounds__(256) void k_integrate(int n, Vec4<R>* __restrict__ x4, Vec4<R>* __restrict__ v4,
                                                    Vec4<R>* __restrict__ f4, const int* __restrict__ tag,
                                                    R dt, LangevinP<R> lp, unsigned long long* __restrict__ blockmax,
                                                    const Vec4<R>* __restrict__ x0, R cap, PosScale<R> ps, const DevCtl* __restrict__ ctl,
                                                    unsigned long long* __restrict__ foldmax = nullptr) {
  // foldmax (decomposed path): kFoldSlots words that collect the block maxima with one atomicMax per block (block b -> word
  // b % kFoldSlots: a few dozen atomics per address, spread over the launch) -- the one-block fold launch between this kernel
  // and the halo exchange is gone; the words travel with the halo, the force kernel's prologue takes the maximum and clears them
  if (ctl->halt) return;   // (uniform scalar load; see DevCtl::halt)
  // kIntPerBlock particles per 256-thread block: every thread owns kIntPerBlock/256 particles and issues
  // all their loads before the first dependent instruction (more bytes in flight per wave for this
  // purely HBM-bound kernel); the arithmetic per particle is unchanged
  constexpr int PPT = kIntPerBlock / 256;
  const int base = blockIdx.x * kIntPerBlock + threadIdx.x;
  Vec4<R> vv[PPT], ff[PPT], xx[PPT];
  int tg[PPT];
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int i = base + u * 256;
    tg[u] = 0;
    if (i < n) {
      vv[u] = v4[i]; ff[u] = f4[i];
      if (LANG) tg[u] = tag[i];
      if (MODE & 2) xx[u] = x4[i];
    }
  }
  R d2 = 0;
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int i = base + u * 256;
    if (i >= n) continue;
    Vec4<R> v = vv[u], f = ff[u];
    if (cap > (R)0) {   // CapForce: f4 holds the conservative force of this step (the host passes cap only then)
      const R f2 = f.x * f.x + f.y * f.y + f.z * f.z;
      if (f2 > cap * cap) { const R s = cap / sqrt_r(f2); f.x *= s; f.y *= s; f.z *= s; }
    }
    if (LANG) {
      bool on = true;
      if (lp.tmask) {   // (uniform branch; the second half kick alone does not load the position: the type comes from x4 then)
        const int ty = (int)((MODE & 2) ? xx[u].w : x4[i].w);
        on = (lp.tmask >> (ty & 31)) & 1u;
      }
      if (on) langevin_force<R>(lp, tg[u], v.w, v.x, v.y, v.z, f.x, f.y, f.z);
      if (STOREF) f4[i] = f;
    }
    R hm = (R)0.5 * dt / v.w;
    if (MODE & 1) { v.x += hm * f.x; v.y += hm * f.y; v.z += hm * f.z; }
    if (MODE & 2) {
      v.x += hm * f.x; v.y += hm * f.y; v.z += hm * f.z;
      Vec4<R> x = xx[u];
      R dx = dt * v.x, dy = dt * v.y, dz = dt * v.z;
      x.x = pos_add(x.x, dx, ps.inv[0]); x.y = pos_add(x.y, dy, ps.inv[1]); x.z = pos_add(x.z, dz, ps.inv[2]);
      x4[i] = x;
      if (x0) { const Vec4<R> o = x0[i]; dx = pos_diff(x.x, o.x, ps.s[0]); dy = pos_diff(x.y, o.y, ps.s[1]); dz = pos_diff(x.z, o.z, ps.s[2]); }
      const R dd = dx * dx + dy * dy + dz * dz;
      d2 = dd > d2 ? dd : d2;
    }
    v4[i] = v;
  }
  if (MODE & 2) {
    // max |dx|^2 of the block -> blockmax[blockIdx.x]; folded by k_rebuild_decide (no contended atomics)
    __shared__ unsigned long long wm[4];
    for (int o = 32; o > 0; o >>= 1) { R t = __shfl_xor(d2, o); d2 = t > d2 ? t : d2; }
    if (lane_id() == 0) wm[threadIdx.x >> 6] = real_bits(d2);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long m = wm[0];
      for (int k = 1; k < 4; ++k) m = wm[k] > m ? wm[k] : m;
      blockmax[blockIdx.x] = m;
      if (foldmax) atomicMax(&foldmax[blockIdx.x % kFoldSlots], m);
    }
  }
}

// one block: fold the step's per-block max displacement into the accumulated distance and
// decide whether the Verlet list must be rebuilt (VelocityVerlet::run).
// phase 3: both (single GPU); phase 1: fold only -> ctl->step_m2 (then max over ranks);
// phase 2: decide from ctl->step_m2.
template <typename R>
__global__ __launch_bounds__(1024) void k_rebuild_decide(DevCtl* ctl, unsigned long long* __restrict__ blockmax, int nblk, double half_skin, int criterion, int phase,
                                                         const double* __restrict__ gathered, int ngathered, volatile int* __restrict__ host_flag, int ticket,
                                                         double half_skin_ref) {      // (half_skin: the skin the lists were built for; _ref: the workload's, for the reported count)
  unsigned long long m = 0;
  if (phase & 1) {
    for (int k = threadIdx.x; k < nblk; k += 1024) { unsigned long long b = blockmax[k]; blockmax[k] = 0ull; m = b > m ? b : m; }
    for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_xor(m, o); m = t > m ? t : m; }
  }
  __shared__ unsigned long long wm[16];
  if (lane_id() == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m2;
    if (phase & 1) {
      for (int k = 1; k < 16; ++k) m = wm[k] > m ? wm[k] : m;
      m2 = sizeof(R) == 4 ? bits_real_f(m) : bits_real_d(m);
      ctl->step_m2 = m2;
    } else if (gathered) { m2 = gathered[0]; for (int q = 1; q < ngathered; ++q) m2 = gathered[q] > m2 ? gathered[q] : m2; }
    else m2 = ctl->step_m2;
    if (phase & 2) {
      const int forced = ctl->force_rebuild;
      double acc = criterion ? sqrt(m2) : ctl->acc_maxdist + sqrt(m2);
      double accr = criterion ? sqrt(m2) : ctl->acc_ref + sqrt(m2);
      const int need = (acc > half_skin) || forced;
      if (need) { acc = 0.0; ctl->force_rebuild = 0; ctl->rebuild_count++; }
      if ((accr > half_skin_ref) || forced) { accr = 0.0; ctl->ref_rebuilds++; }      // the reference rule on the workload's skin
      ctl->acc_maxdist = acc; ctl->acc_ref = accr;
      ctl->need_rebuild = need;
      if (host_flag) {   // pinned, host-visible: [0] decision, [1] ticket (written last); the host spins on the ticket
        host_flag[0] = need;
        __threadfence_system();
        host_flag[1] = ticket;
      }
    }
  }
}

// device -> pinned host memory by a kernel (16-byte stores over PCIe): the copy engine's first
// transfer after a large allocation burst took 7 ms for 8 MB here
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n16; k += (size_t)gridDim.x * blockDim.x) dst[k] = src[k];
}

// a handful of device integers to pinned host memory with ONE poll on the host (the slab rebuild
// used to read them with one blocking copy each): host[0] = ticket (written last), host[1..n] = values
struct CollectArgs { const int* src[8]; int n; };
__global__ void k_collect_ints(CollectArgs a, volatile int* host, int ticket) {
  if ((int)threadIdx.x < a.n) host[1 + threadIdx.x] = *a.src[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) host[0] = ticket;
}

// =======================================================================================
// K1  cell binning + canonical sort (storage.decompose(), start_simulation.py:158-171)
// =======================================================================================
// migration buffer of one direction (SoA, fixed capacity, count in front): particles that left the
// slab travel to the neighbour rank with their full state
template <typename R> struct MigBuf { int* count; Vec4<R>* x; Vec4<R>* v; int4* img; int* tag; int cap; };

// bucket != nullptr (fused rebuild): member `slot` of cell `cid` is recorded at bucket[cid * bcap + slot] right away
// (fixed-capacity rows), so that the sort phase needs no separate placement pass; cell_of / slot_of are not written.
template <typename R>
__device__ __forceinline__ void dev_bin(int i0, int n, Vec4<R>* x4, const Vec4<R>* v4, const int* tag, int4* img4, const Box<R>& box,
                                        int* cell_cnt, int* cell_of, int* slot_of, const MigBuf<R>& mdn, const MigBuf<R>& mup, DevCtl* ctl,
                                        int* bucket = nullptr, int bcap = 0, int* seg_tot = nullptr, int seg_shift = 0) {
  // Four particles per thread and trip, in three stages -- all position loads, then all slot atomics, then the stores --
  // so that a thread waits for ONE load and ONE returning atomic round trip per four particles instead of one each per
  // particle (the phase is a chain of dependent device-scope round trips, not bandwidth).
  // The trip bound is wave-uniform (rounded up) because the slot assignment uses cross-lane ops.
  constexpr int U = 4;
  const int iend = i0 + n;
  const int G = gridDim.x * blockDim.x;
  const int lane = lane_id();
  for (int ib0 = i0 + blockIdx.x * blockDim.x; ib0 < iend; ib0 += U * G) {
    Vec4<R> xs[U]; int4 ims[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = ib0 + u * G + (int)threadIdx.x;
      if (i < iend) { xs[u] = x4[i]; ims[u] = img4[i]; }
    }
    int cids[U], bases[U], starts[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = ib0 + u * G + (int)threadIdx.x;
      int cid = -1;                       // -1: lane idle or particle migrates away
      if (i < iend) {
        Vec4<R> x = xs[u];
        int4 im = ims[u];
        R* p = &x.x; int* ip = &im.x;
        int c[3];
        bool moved = false;   // folded back into the box: only then position and image counters are written back
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          moved |= pos_fold(p[d], ip[d], d, box);
          const int ncd = (d == 2 && box.zghost) ? box.nzg : (box.nc[d] > 0 ? box.nc[d] : 1);
          c[d] = pos_cell(p[d], d, ncd, box);
        }
        int dir = 0;   // 0 stays, -1 leaves downwards, +1 upwards
        if (box.zghost) {
          // c[2] is the global layer of the folded z; the slab owns layers [z0g, z0g + own)
          const int own = box.nc[2] - 2, gz = c[2];
          const int gu = (box.z0g + own) % box.nzg, gd = (box.z0g - 1 + box.nzg) % box.nzg;
          if (gz >= box.z0g && gz < box.z0g + own) c[2] = gz - box.z0g + 1;
          else if (gz == gu) dir = 1;
          else if (gz == gd) dir = -1;
          else { ctl->mig_error = 1; c[2] = 1; }   // moved by more than one layer: impossible within skin/2
        }
        if (moved) { x4[i] = x; img4[i] = im; }
        if (dir) {
          const MigBuf<R>& mb = dir < 0 ? mdn : mup;
          const int k = atomicAdd(mb.count, 1);
          if (k < mb.cap) { mb.x[k] = x; mb.v[k] = v4[i]; mb.img[k] = im; mb.tag[k] = tag[i]; } else ctl->mig_error = 2;
          cell_of[i] = -1;
        } else {
          cid = box.nc[0] > 0 ? (c[2] * box.nc[1] + c[1]) * box.nc[0] + c[0] : 0;
          if (!bucket) cell_of[i] = cid;
        }
      }
      cids[u] = cid;
    }
    // Slot inside the cell.  The arrays are still in the cell order of the previous rebuild, so
    // neighbouring lanes mostly share a cell: one atomic per run of equal cells in the wave
    // (run head adds the run length, the others take base + rank) instead of one per particle --
    // ~18 same-address atomics per cell were the whole cost of this kernel.
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int cid = cids[u];
      const int prev = __shfl_up(cid, 1);
      const bool head = lane == 0 || prev != cid;
      const unsigned long long hm = __ballot(head);
      const unsigned long long upto = hm & (~0ull >> (63 - lane));          // heads at lanes <= lane
      const int start = 63 - __clzll((long long)upto);
      const unsigned long long above = lane == 63 ? 0ull : (hm >> (lane + 1));
      const int end = above ? lane + __ffsll((long long)above) : 64;        // first head after this lane
      int base = 0;
      if (head && cid >= 0) base = atomicAdd(&cell_cnt[cid], end - start);
      bases[u] = base; starts[u] = start;
      if (seg_tot) {
        // particles per segment of 2^seg_shift cells (fused rebuild): one atomic per run of equal SEGMENTS in the wave
        // (a wave usually sits inside one segment; per-cell-run atomics on the same word serialise in the L2)
        const int sg = cid >= 0 ? cid >> seg_shift : -1;
        const int sprev = __shfl_up(sg, 1);
        const bool shead = lane == 0 || sprev != sg;
        const unsigned long long shm = __ballot(shead);
        const unsigned long long sabove = lane == 63 ? 0ull : (shm >> (lane + 1));
        const int send = sabove ? lane + __ffsll((long long)sabove) : 64;
        if (shead && sg >= 0) atomicAdd(&seg_tot[sg], send - lane);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = ib0 + u * G + (int)threadIdx.x;
      const int cid = cids[u];
      const int base = __shfl(bases[u], starts[u]);
      if (cid >= 0) {
        const int slot = base + (lane - starts[u]);
        if (!bucket) slot_of[i] = slot;
        else if (slot < bcap) bucket[(size_t)cid * bcap + slot] = i;
        else atomicMax(&ctl->bucket_overflow, slot + 1);
      }
    }
  }
}

template <typename R>
__global__ __launch_bounds__(256) void k_bin(int i0, int n, Vec4<R>* __restrict__ x4, const Vec4<R>* __restrict__ v4, const int* __restrict__ tag,
                                             int4* __restrict__ img4, Box<R> box,
                                             int* __restrict__ cell_cnt, int* __restrict__ cell_of,
                                             int* __restrict__ slot_of, MigBuf<R> mdn, MigBuf<R> mup, DevCtl* ctl) {
  if (!ctl->need_rebuild) return;
  dev_bin<R>(i0, n, x4, v4, tag, img4, box, cell_cnt, cell_of, slot_of, mdn, mup, ctl);
}

// arrivals of a migration exchange are appended behind the current particles
template <typename R>
__global__ __launch_bounds__(256) void k_append_arrivals(MigBuf<R> in, int count, int dst0, Vec4<R>* __restrict__ x4, Vec4<R>* __restrict__ v4,
                                                        int* __restrict__ tag, int4* __restrict__ img4) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  x4[dst0 + k] = in.x[k]; v4[dst0 + k] = in.v[k]; tag[dst0 + k] = in.tag[k]; img4[dst0 + k] = in.img[k];
}

// single-block exclusive scan of cell counts -> cell_start[0..ncell]; zeroes cell_cnt for next time.
// Each of the 16 waves scans one contiguous 1/16 of the cells with wave shuffles only (running
// carry, four 256-cell groups in flight, no block barrier in the loop), then one barrier publishes
// the wave totals and a second sweep adds each wave's offset to what it wrote.
__global__ __launch_bounds__(1024) void k_scan_cells(int ncell, int base0, int* __restrict__ cell_cnt, int* __restrict__ cell_start,
                                                     const DevCtl* ctl) {
  if (!ctl->need_rebuild) return;
  __shared__ int wsum[16];
  const int w = threadIdx.x >> 6, lane = lane_id();
  const int per = (((ncell + 15) >> 4) + 255) & ~255;
  const int lo = w * per, hi = min(ncell, lo + per);
  constexpr int U = 4;
  int carry = 0;
  for (int b = lo; b < hi; b += 256 * U) {
    int v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i0 = b + u * 256 + lane * 4;
      if (i0 + 3 < hi) {
        const int4 t = *reinterpret_cast<const int4*>(cell_cnt + i0);
        v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
        *reinterpret_cast<int4*>(cell_cnt + i0) = make_int4(0, 0, 0, 0);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[u][k] = 0; if (i0 + k < hi) { v[u][k] = cell_cnt[i0 + k]; cell_cnt[i0 + k] = 0; } }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i0 = b + u * 256 + lane * 4;
      const int sum = v[u][0] + v[u][1] + v[u][2] + v[u][3];
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
      int run = carry + incl - sum;
      if (i0 + 3 < hi) {
        int4 o4; o4.x = run; o4.y = run + v[u][0]; o4.z = o4.y + v[u][1]; o4.w = o4.z + v[u][2];
        *reinterpret_cast<int4*>(cell_start + i0) = o4;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (i0 + k < hi) cell_start[i0 + k] = run; run += v[u][k]; }
      }
      carry += __shfl(incl, 63);
    }
  }
  if (lane == 0) wsum[w] = carry;
  __syncthreads();
  int off = base0, tot = base0;
  for (int k = 0; k < 16; ++k) { if (k < w) off += wsum[k]; tot += wsum[k]; }
  if (off != 0) {
    for (int i0 = lo + lane * 4; i0 < hi; i0 += 256) {
      if (i0 + 3 < hi) {
        int4 t = *reinterpret_cast<int4*>(cell_start + i0);
        t.x += off; t.y += off; t.z += off; t.w += off;
        *reinterpret_cast<int4*>(cell_start + i0) = t;
      } else {
        for (int k = 0; k < 4; ++k) if (i0 + k < hi) cell_start[i0 + k] += off;
      }
    }
  }
  if (threadIdx.x == 0) cell_start[ncell] = tot;
}

__global__ __launch_bounds__(256) void k_place(int i0, int n, const int* __restrict__ cell_of, const int* __restrict__ slot_of,
                                               const int* __restrict__ cell_start, int* __restrict__ perm, const DevCtl* ctl) {
  if (!ctl->need_rebuild) return;
  for (int i = i0 + blockIdx.x * blockDim.x + threadIdx.x; i < i0 + n; i += gridDim.x * blockDim.x) {
    const int c = cell_of[i];
    if (c >= 0) perm[cell_start[c] + slot_of[i]] = i;   // c < 0: migrated away
  }
}

// One wave per cell: rank the members by (x sub-bin, tag) -- a canonical order inside a cell, so the whole
// pipeline is run-to-run deterministic -- and gather the particle arrays into cell-sorted order.
// The cell is cut into NSUB slices along x; cell_sub[c] packs the number of members in front of slice
// 1, 2, 3 (8 bits each): the list build uses it to test only the x-window of a stencil row that can hold
// neighbours of a given home particle (dev_nlist_tile).  Crowded cells (> 64 members) keep plain tag
// order and store -1 ("no sub-bin information": the list build then takes the whole cell).
constexpr int NSUB = 4;
// key = x slice (2 bits) | y half | z half | tag: x-major so that cell_sub stays a prefix over the x slices; the
// y/z halves make neighbouring lanes of the force kernel (consecutive home particles) spatial neighbours, whose
// slot-sorted lists then read nearby LDS slots in the same instruction (fewer bank conflicts, more broadcasts)
template <typename R>
__device__ __forceinline__ int sort_key(const Vec4<R>& xq, int c, const Box<R>& box, int tg) {
  const Vec4<R> x = pos_abs(xq, box);   // (slices and halves only order the cell and prune windows: absolute fp32 is enough)
  const int nx = box.nc[0] > 0 ? box.nc[0] : 1, ny = box.nc[1] > 0 ? box.nc[1] : 1, nz = box.nc[2] > 0 ? box.nc[2] : 1;
  const int cx = c % nx, cy = (c / nx) % ny;
  const R clx = box.L[0] / (R)nx, cly = box.L[1] / (R)ny;
  int b = (int)((x.x - (R)cx * clx) * ((R)NSUB / clx));
  b = b < 0 ? 0 : (b > NSUB - 1 ? NSUB - 1 : b);
  const int hy = (x.y - (R)cy * cly) * (R)2 >= cly ? 1 : 0;
  // z: the half is taken from the fractional cell coordinate (slab mode shifts the layer index, not the split)
  const R zc = x.z * box.cell_inv[2];
  const int hz = (zc - floor_r(zc)) >= (R)0.5 ? 1 : 0;
  (void)nz;
  return (b << 29) | (hy << 28) | (hz << 27) | tg;     // tags < 2^27 (chem_set_particles)
}
template <typename R>
__device__ __forceinline__ void dev_sort_gather(int ncell, const int* cell_start, const int* perm, const Vec4<R>* x4, const Vec4<R>* v4,
                                                const int* tag, const int4* img4, Vec4<R>* x4o, Vec4<R>* v4o, int* tago, int4* img4o,
                                                const Box<R>& box, int* cell_sub) {
  // Two cells per wave (one per half-wave, width-32 shuffles) while both have <= 32 members -- the mean
  // occupancy is ~18 -- otherwise one cell per wave / the global-memory ranking for crowded cells.
  const int l = lane_id(), hl = l & 31, half = l >> 5;
  const int nw = gridDim.x * (blockDim.x >> 6);
  for (int c0 = 2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); c0 < ncell; c0 += 2 * nw) {
    const int s0 = cell_start[c0], s1 = cell_start[c0 + 1], s2 = c0 + 1 < ncell ? cell_start[c0 + 2] : s1;
    const int cnt0 = s1 - s0, cnt1 = s2 - s1;
    if (cnt0 <= 32 && cnt1 <= 32) {
      const int s = half ? s1 : s0, cnt = half ? cnt1 : cnt0, cmax = cnt0 > cnt1 ? cnt0 : cnt1;
      const int cc = c0 + half;
      int pi = 0, tg = 0, key = 0x7fffffff;
      Vec4<R> xp = mk4<R>(0, 0, 0, 0);
      if (hl < cnt) { pi = perm[s + hl]; tg = tag[pi]; xp = x4[pi]; key = sort_key<R>(xp, cc, box, tg); }
      int rank = 0;
      for (int k = 0; k < cmax; ++k) { const int tk = __shfl(key, k, 32); rank += (k < cnt && tk < key) ? 1 : 0; }
      // members in front of slice 1, 2, 3 (per half-wave)
      const int bin = key >> 29;
      unsigned int packed = 0;
#pragma unroll
      for (int b = 1; b < NSUB; ++b) {
        const unsigned long long m = __ballot(hl < cnt && bin < b);
        packed |= (unsigned int)__popc((unsigned int)(half ? (m >> 32) : m)) << (8 * (b - 1));
      }
      if (hl == 0 && cc < ncell && cell_sub) cell_sub[cc] = (int)packed;
      if (hl < cnt) {
        const int dst = s + rank;
        x4o[dst] = xp; v4o[dst] = v4[pi]; tago[dst] = tg; img4o[dst] = img4[pi];
      }
      continue;
    }
    for (int cc = c0; cc < c0 + 2 && cc < ncell; ++cc) {
      const int s = cell_start[cc], cnt = cell_start[cc + 1] - s;
      if (cnt <= 64) {
        int pi = 0, tg = 0, key = 0x7fffffff;
        Vec4<R> xp = mk4<R>(0, 0, 0, 0);
        if (l < cnt) { pi = perm[s + l]; tg = tag[pi]; xp = x4[pi]; key = sort_key<R>(xp, cc, box, tg); }
        int rank = 0;
        for (int k = 0; k < cnt; ++k) { const int tk = __shfl(key, k); rank += (tk < key) ? 1 : 0; }
        const int bin = key >> 29;
        unsigned int packed = 0;
#pragma unroll
        for (int b = 1; b < NSUB; ++b) packed |= (unsigned int)__popcll(__ballot(l < cnt && bin < b)) << (8 * (b - 1));
        if (l == 0 && cell_sub) cell_sub[cc] = (int)packed;
        if (l < cnt) {
          const int dst = s + rank;
          x4o[dst] = xp; v4o[dst] = v4[pi]; tago[dst] = tg; img4o[dst] = img4[pi];
        }
      } else {
        // crowded cell: each lane ranks its members against all others through global memory (tag order)
        if (l == 0 && cell_sub) cell_sub[cc] = -1;
        for (int a = l; a < cnt; a += 64) {
          const int pi = perm[s + a], tg = tag[pi];
          int rank = 0;
          for (int k = 0; k < cnt; ++k) rank += (tag[perm[s + k]] < tg) ? 1 : 0;
          const int dst = s + rank;
          x4o[dst] = x4[pi]; v4o[dst] = v4[pi]; tago[dst] = tg; img4o[dst] = img4[pi];
        }
      }
    }
  }
}

// Fused-rebuild variant: members come from the fixed-capacity bucket rows filled while binning.  A cell's first index
// = offset of its segment (s_off, from the per-segment totals of the binning pass) + the counts of the cells in front
// of it inside the segment (<= 63 values: one load per lane and a wave reduction) -- no separate scan phase; it is
// published as cell_start, and the tag -> index map is written with the gather.  <= 64 members per cell (checked at
// binning time).
template <typename R>
__device__ __forceinline__ void dev_sort_gather_bucket(int ncell, const int* cell_cnt, const int* s_off, int seg_shift,
                                                       const int* bucket, int bcap, int* cell_start, const Vec4<R>* x4, const Vec4<R>* v4,
                                                       const int* tag, const int4* img4, Vec4<R>* x4o, Vec4<R>* v4o, int* tago, int4* img4o,
                                                       int* rtag, const Box<R>& box, int* cell_sub) {
  const int l = lane_id(), hl = l & 31, half = l >> 5;
  const int nw = gridDim.x * (blockDim.x >> 6);
  const int per = 1 << seg_shift;
  for (int c0 = 2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); c0 < ncell; c0 += 2 * nw) {
    const bool two = c0 + 1 < ncell;
    // c0 is even and segments hold an even number (>= 2) of cells: c0 and c0 + 1 share a segment
    const int seg = c0 >> seg_shift, lo = seg << seg_shift;
    int before = 0;
    for (int k = lo + l; k < c0; k += 64) before += cell_cnt[k];     // per <= 64: one trip
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    (void)per;
    const int cnt0 = cell_cnt[c0], cnt1 = two ? cell_cnt[c0 + 1] : 0;
    const int s0 = s_off[seg] + before, s1 = s0 + cnt0;
    if (l == 0) { cell_start[c0] = s0; if (two) cell_start[c0 + 1] = s1; }
    if (cnt0 <= 32 && cnt1 <= 32) {
      const int s = half ? s1 : s0, cnt = half ? cnt1 : cnt0, cmax = cnt0 > cnt1 ? cnt0 : cnt1;
      const int cc = c0 + half;
      int pi = 0, tg = 0, key = 0x7fffffff;
      Vec4<R> xp = mk4<R>(0, 0, 0, 0);
      if (hl < cnt) { pi = bucket[(size_t)cc * bcap + hl]; tg = tag[pi]; xp = x4[pi]; key = sort_key<R>(xp, cc, box, tg); }
      int rank = 0;
      for (int k = 0; k < cmax; ++k) { const int tk = __shfl(key, k, 32); rank += (k < cnt && tk < key) ? 1 : 0; }
      const int bin = key >> 29;
      unsigned int packed = 0;
#pragma unroll
      for (int b = 1; b < NSUB; ++b) {
        const unsigned long long m = __ballot(hl < cnt && bin < b);
        packed |= (unsigned int)__popc((unsigned int)(half ? (m >> 32) : m)) << (8 * (b - 1));
      }
      if (hl == 0 && cc < ncell) cell_sub[cc] = (int)packed;
      if (hl < cnt) {
        const int dst = s + rank;
        x4o[dst] = xp; v4o[dst] = v4[pi]; tago[dst] = tg; img4o[dst] = img4[pi]; rtag[tg] = dst;
      }
      continue;
    }
    for (int cc = c0; cc < c0 + 2 && cc < ncell; ++cc) {
      const int s = cc == c0 ? s0 : s1, cnt = min(cc == c0 ? cnt0 : cnt1, 64);
      int pi = 0, tg = 0, key = 0x7fffffff;
      Vec4<R> xp = mk4<R>(0, 0, 0, 0);
      if (l < cnt) { pi = bucket[(size_t)cc * bcap + l]; tg = tag[pi]; xp = x4[pi]; key = sort_key<R>(xp, cc, box, tg); }
      int rank = 0;
      for (int k = 0; k < cnt; ++k) { const int tk = __shfl(key, k); rank += (tk < key) ? 1 : 0; }
      const int bin = key >> 29;
      unsigned int packed = 0;
#pragma unroll
      for (int b = 1; b < NSUB; ++b) packed |= (unsigned int)__popcll(__ballot(l < cnt && bin < b)) << (8 * (b - 1));
      if (l == 0) cell_sub[cc] = (int)packed;
      if (l < cnt) {
        const int dst = s + rank;
        x4o[dst] = xp; v4o[dst] = v4[pi]; tago[dst] = tg; img4o[dst] = img4[pi]; rtag[tg] = dst;
      }
    }
  }
}

template <typename R>
__global__ __launch_bounds__(256) void k_sort_gather(int ncell, const int* __restrict__ cell_start, const int* __restrict__ perm,
                                                     const Vec4<R>* __restrict__ x4, const Vec4<R>* __restrict__ v4,
                                                     const int* __restrict__ tag, const int4* __restrict__ img4,
                                                     Vec4<R>* __restrict__ x4o, Vec4<R>* __restrict__ v4o,
                                                     int* __restrict__ tago, int4* __restrict__ img4o, const DevCtl* ctl, Box<R> box, int* __restrict__ cell_sub) {
  if (!ctl->need_rebuild) return;
  dev_sort_gather<R>(ncell, cell_start, perm, x4, v4, tag, img4, x4o, v4o, tago, img4o, box, cell_sub);
}

template <typename R>
__global__ __launch_bounds__(256) void k_copyback(const int* __restrict__ cell_start, int c_first, int c_last, const Vec4<R>* __restrict__ x4o, const Vec4<R>* __restrict__ v4o,
                                                  const int* __restrict__ tago,
                                                  const int4* __restrict__ img4o, Vec4<R>* __restrict__ x4, Vec4<R>* __restrict__ v4,
                                                  int* __restrict__ tag, int4* __restrict__ img4,
                                                  int* __restrict__ rtag, Vec4<R>* __restrict__ x0, const DevCtl* ctl) {
  if (!ctl->need_rebuild) return;
  const int kbeg = cell_start[c_first], kend = cell_start[c_last];   // the own (non-ghost) cells
  for (int k = kbeg + blockIdx.x * blockDim.x + threadIdx.x; k < kend; k += gridDim.x * blockDim.x) {
    const Vec4<R> xk = x4o[k];
    x4[k] = xk; if (x0) x0[k] = xk;
    v4[k] = v4o[k]; img4[k] = img4o[k];
    int t = tago[k]; tag[k] = t; rtag[t] = k;
  }
}

// ---- slab decomposition: ghost layers --------------------------------------------------
// per-cell counts of the two boundary layers (what the neighbours need to lay out their ghosts)
__global__ __launch_bounds__(256) void k_layer_counts(int nxy, int own_layers, const int* __restrict__ cell_start,
                                                      int* __restrict__ cnt_dn, int* __restrict__ cnt_up) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > nxy) return;
  const int c1 = nxy, cn = own_layers * nxy;   // first cell of own layer 1 / of the last own layer
  if (k < nxy) { cnt_dn[k] = cell_start[c1 + k + 1] - cell_start[c1 + k]; cnt_up[k] = cell_start[cn + k + 1] - cell_start[cn + k]; }
  else { cnt_dn[nxy] = cell_start[c1 + nxy] - cell_start[c1]; cnt_up[nxy] = cell_start[cn + nxy] - cell_start[cn]; }
}

// cell_start of the two ghost layers from the neighbours' per-cell counts (single block):
// lower ghosts are right-aligned in front of the reals, upper ghosts follow them
__global__ __launch_bounds__(1024) void k_ghost_cells(int nxy, int own_layers, const int* __restrict__ gcnt_lo, const int* __restrict__ gcnt_up,
                                                      int* __restrict__ cell_start) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int c_up = (own_layers + 1) * nxy;
  for (int pass = 0; pass < 2; ++pass) {
    const int* cnt = pass ? gcnt_up : gcnt_lo;
    const int base_cell = pass ? c_up : 0;
    const int start = pass ? cell_start[c_up] : cell_start[nxy] - gcnt_lo[nxy];   // upper ghosts begin where the reals end
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    for (int b = 0; b < nxy; b += 1024) {
      const int i = b + threadIdx.x;
      const int v = i < nxy ? cnt[i] : 0;
      int incl = v;
      for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane_id() >= o) incl += t; }
      if (lane_id() == 63) wsum[w] = incl;
      __syncthreads();
      int woff = 0, tot = 0;
      for (int k = 0; k < 16; ++k) { if (k < w) woff += wsum[k]; tot += wsum[k]; }
      if (i < nxy) cell_start[base_cell + i] = start + carry_s + woff + incl - v;
      __syncthreads();
      if (threadIdx.x == 0) carry_s += tot;
      __syncthreads();
    }
    if (pass && threadIdx.x == 0) cell_start[c_up + nxy] = start + carry_s;
    __syncthreads();
  }
}

// rtag for ghost copies (only where the particle is not owned here)
// gtag: the ghost copy of every tag that has one here, owned or not (a one-rank slab holds a real particle AND its ghost).
// One launch for both ghost ranges: [g0a, g0a + nga) in front of the reals, [g0b, g0b + ngb) behind them.
__global__ __launch_bounds__(256) void k_ghost_rtag(int g0a, int nga, int g0b, int ngb, const int* __restrict__ tag, int* __restrict__ rtag, int* __restrict__ gtag) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nga + ngb) return;
  const int g = k < nga ? g0a + k : g0b + (k - nga);
  const int t = tag[g];
  atomicCAS(&rtag[t], -1, g);
  gtag[t] = g;
}
// first launch of a slab rebuild: the request is being served (need_rebuild on for the chain's kernels, force_rebuild off -- the
// force kernel's decision only reads it), headers of the two migration buffers cleared -- one launch instead of four memsets
__global__ void k_dd_begin(DevCtl* ctl, int* __restrict__ mig_dn, int* __restrict__ mig_up) {
  const int t = threadIdx.x;
  if (t == 0) { ctl->need_rebuild = 1; ctl->force_rebuild = 0; }
  if (t < 4) { mig_dn[t] = 0; mig_up[t] = 0; }
}
// what a slab rebuild clears before the sorted copies come back: the tag maps, and the x sub-bin words of the two ghost
// layers (filled by the neighbours' particles afterwards: no sub-bin information for their cells) -- one launch
__global__ __launch_bounds__(256) void k_dd_clear(int* __restrict__ rtag, int* __restrict__ gtag, int nglob, int* __restrict__ sub_lo, int* __restrict__ sub_hi, int nxy) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nglob) { rtag[k] = -1; gtag[k] = -1; }
  if (k < nxy) { sub_lo[k] = -1; sub_hi[k] = -1; }
}

// =======================================================================================
// K2  Verlet neighbour-list build (VerletList(system, cutoff, exclusionlist),
//     start_simulation.py:193-197).  One workgroup per home cell: the 27-cell stencil is
//     staged in LDS with the periodic shift applied, each wave then owns home particles and
//     tests 64 staged candidates per instruction; __ballot + popcount compacts the hits into
//     the particle's row with coalesced stores.
// =======================================================================================
__device__ __forceinline__ float idx_as_real(int i, float) { return __int_as_float(i); }
__device__ __forceinline__ double idx_as_real(int i, double) { return __longlong_as_double((long long)i); }
__device__ __forceinline__ int real_as_idx(float w) { return __float_as_int(w); }
__device__ __forceinline__ int real_as_idx(double w) { return (int)__double_as_longlong(w); }

// rows are padded with the particle's own index up to a multiple of 4 entries so that the
// force kernel can read them as int4 (the self entry has r = 0 and is masked there)
__device__ __forceinline__ void finish_row(int* row, int* nn, int p, int cnt, int S, int l, DevCtl* ctl) {
  const int c = cnt < S ? cnt : S;
  const int pad = (4 - (c & 3)) & 3;
  if (l < pad) row[c + l] = p;
  if (l == 0) { nn[p] = c; if (cnt > S) atomicMax(&ctl->nl_overflow, cnt); }
}

template <typename R, int CAP>
__global__ __launch_bounds__(256) void k_nlist_cells(int n, const Vec4<R>* __restrict__ x4, const int* __restrict__ tag,
                                                     const int* __restrict__ cell_start, Box<R> box, R rl2,
                                                     const int* __restrict__ excl_start, const int* __restrict__ excl_list,
                                                     int has_excl, int* __restrict__ nlist, int* __restrict__ nn, int S,
                                                     DevCtl* ctl) {
  if (!ctl->need_rebuild) return;
  __shared__ Vec4<R> sx[CAP];          // (x, y, z, bits of the global index), periodic shift applied
  __shared__ int seg_start[27], seg_cnt[27], seg_off[28];
  __shared__ R seg_shift[27][3];
  const int nx = box.nc[0], ny = box.nc[1], nz = box.nc[2];
  const int w = threadIdx.x >> 6, l = lane_id();
  for (int c = blockIdx.x; c < box.ncell; c += gridDim.x) {
    const int cx = c % nx, cy = (c / nx) % ny, cz = c / (nx * ny);
    __syncthreads();   // previous cell's tile fully consumed
    if (threadIdx.x < 27) {
      int k = threadIdx.x;
      int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
      int ox = cx + dx, oy = cy + dy, oz = cz + dz;
      R sh[3] = {0, 0, 0};
      if (ox < 0) { ox += nx; sh[0] = -box.L[0]; } else if (ox >= nx) { ox -= nx; sh[0] = box.L[0]; }
      if (oy < 0) { oy += ny; sh[1] = -box.L[1]; } else if (oy >= ny) { oy -= ny; sh[1] = box.L[1]; }
      if (oz < 0) { oz += nz; sh[2] = -box.L[2]; } else if (oz >= nz) { oz -= nz; sh[2] = box.L[2]; }
      int oc = (oz * ny + oy) * nx + ox;
      seg_start[k] = cell_start[oc];
      seg_cnt[k] = cell_start[oc + 1] - cell_start[oc];
      seg_shift[k][0] = sh[0]; seg_shift[k][1] = sh[1]; seg_shift[k][2] = sh[2];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int o = 0;
      for (int k = 0; k < 27; ++k) { seg_off[k] = o; o += seg_cnt[k]; }
      seg_off[27] = o;
    }
    __syncthreads();
    const int total_all = seg_off[27];
    const int hs = cell_start[c], he = cell_start[c + 1];
    // A stencil that holds more than CAP particles (dense systems that have left the LDS tiles) is walked in windows of
    // CAP slots; a home particle's running count waits in nn[] between two windows (written and read back by lane 0 of
    // the wave that owns the particle).
    for (int base = 0; base < total_all || base == 0; base += CAP) {
      if (base) __syncthreads();
      const int total = min(total_all - base, CAP);
      // stage: one wave per stencil cell round-robin, coalesced 16/32-byte loads
      for (int k = w; k < 27; k += 4) {
        const int s0 = seg_start[k], cnt = seg_cnt[k], o0 = seg_off[k] - base;
        const R sh0 = seg_shift[k][0], sh1 = seg_shift[k][1], sh2 = seg_shift[k][2];
        for (int t = l; t < cnt; t += 64) {
          const int dst = o0 + t;
          if (dst >= 0 && dst < CAP) {
            Vec4<R> p = pos_abs(x4[s0 + t], box);
            p.x += sh0; p.y += sh1; p.z += sh2; p.w = idx_as_real(s0 + t, (R)0);
            sx[dst] = p;
          }
        }
      }
      __syncthreads();
      const bool last = base + CAP >= total_all;
      for (int p = hs + w; p < he; p += 4) {
        const Vec4<R> xi = pos_abs(x4[p], box);
        int e0 = 0, e1 = 0;
        if (has_excl) { int tg = tag[p]; e0 = excl_start[tg]; e1 = excl_start[tg + 1]; }
        int cnt = 0;
        if (base) { if (l == 0) cnt = nn[p]; cnt = __shfl(cnt, 0); }
        int* row = nlist + (size_t)p * S;
        for (int s0 = 0; s0 < total; s0 += 64) {
          const int s = s0 + l;
          bool ok = s < total;
          int j = -1;
          if (ok) {
            const Vec4<R> xj = sx[s];
            j = real_as_idx(xj.w);
            const R dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
            const R r2 = dx * dx + dy * dy + dz * dz;
            ok = (r2 <= rl2) && (j != p);
            if (ok && e1 > e0) {
              const int tj = tag[j];
              for (int e = e0; e < e1; ++e) if (excl_list[e] == tj) { ok = false; break; }
            }
          }
          const unsigned long long m = __ballot(ok);
          if (ok) {
            const int pos = cnt + __popcll(m & lanemask_lt());
            if (pos < S) row[pos] = j;
          }
          cnt += __popcll(m);
        }
        if (last) finish_row(row, nn, p, cnt, S, l, ctl);
        else if (l == 0) nn[p] = cnt;
      }
    }
  }
}

// brute-force variant for boxes with fewer than 3 cells per axis (tiny test systems):
// one wave per particle, all other particles are candidates, minimum image.
template <typename R>
__global__ __launch_bounds__(256) void k_nlist_brute(int n, const Vec4<R>* __restrict__ x4, const int* __restrict__ tag,
                                                     Box<R> box, R rl2, const int* __restrict__ excl_start,
                                                     const int* __restrict__ excl_list, int has_excl,
                                                     int* __restrict__ nlist, int* __restrict__ nn, int S, DevCtl* ctl) {
  if (!ctl->need_rebuild) return;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  const int l = lane_id();
  const Vec4<R> xi = x4[p];
  int e0 = 0, e1 = 0;
  if (has_excl) { int tg = tag[p]; e0 = excl_start[tg]; e1 = excl_start[tg + 1]; }
  int cnt = 0;
  int* row = nlist + (size_t)p * S;
  for (int s0 = 0; s0 < n; s0 += 64) {
    const int j = s0 + l;
    bool ok = j < n && j != p;
    if (ok) {
      const Vec4<R> xj = x4[j];
      const R dx = pos_delta(xi.x, xj.x, 0, box), dy = pos_delta(xi.y, xj.y, 1, box), dz = pos_delta(xi.z, xj.z, 2, box);
      ok = dx * dx + dy * dy + dz * dz <= rl2;
      if (ok && e1 > e0) {
        const int tj = tag[j];
        for (int e = e0; e < e1; ++e) if (excl_list[e] == tj) { ok = false; break; }
      }
    }
    const unsigned long long m = __ballot(ok);
    if (ok) { const int pos = cnt + __popcll(m & lanemask_lt()); if (pos < S) row[pos] = j; }
    cnt += __popcll(m);
  }
  finish_row(row, nn, p, cnt, S, l, ctl);
}

// =======================================================================================
// K3  non-bonded pair forces over the Verlet list (VerletListLennardJones /
//     VerletListTabulated, gromacs_topology.py:511-512,696-721).  TPP lanes cooperate on one
//     particle: lane t handles neighbours t, t+TPP, ...; partial forces are combined with a
//     segmented xor-shuffle reduction.  Full list => no atomics, deterministic sums.
// =======================================================================================
template <typename R, bool ENERGY>
__device__ __forceinline__ void pair_term(const PairCore<R> pc, const PairExt<R>* __restrict__ pext, int pidx,
                                          const Vec4<R>* __restrict__ tab, R r2, R dx, R dy, R dz,
                                          R& fx, R& fy, R& fz, double& e_lj, double& e_tab, double& vir) {
  if (r2 <= pc.rc2) {
    if (pc.kind == (R)1) {
      const R r2i = rcp_r(r2), r6i = r2i * r2i * r2i;
      const R ff = r6i * (pc.lj1 * r6i - pc.lj2) * r2i;
      fx += ff * dx; fy += ff * dy; fz += ff * dz;
      if (ENERGY) { const PairExt<R> px = pext[pidx]; e_lj += (double)(r6i * (px.e1 * r6i - px.e2) + px.shift); vir += (double)(ff * r2); }
    } else {
      const PairExt<R> px = pext[pidx];
      const R r = sqrt_r(r2);
      R t = (r - px.r0) * px.inv_dr;
      const R tmax = (R)(px.nrow - 1);
      t = t < (R)0 ? (R)0 : (t > tmax ? tmax : t);
      int k = (int)t;
      if (k > px.nrow - 2) k = px.nrow - 2;
      const R wgt = t - (R)k;
      const Vec4<R> row = tab[px.toff + k];  // (f_k, df_k, e_k, de_k)
      const R ff = (row.x + wgt * row.y) / r;
      fx += ff * dx; fy += ff * dy; fz += ff * dz;
      if (ENERGY) { e_tab += (double)(row.z + wgt * row.w); vir += (double)(ff * r2); }
    }
  }
}

template <typename R, int TPP, bool ENERGY>
__global__ __launch_bounds__(256) void k_pair_force(int n, const Vec4<R>* __restrict__ x4, Vec4<R>* __restrict__ f4,
                                                    const int* __restrict__ nlist, const int* __restrict__ nn, int S,
                                                    Box<R> box, const PairCore<R>* __restrict__ pcore,
                                                    const PairExt<R>* __restrict__ pext, int ntypes,
                                                    const Vec4<R>* __restrict__ tab, double* __restrict__ eout,
                                                    double half_skin, DevCtl* ctl) {
  __shared__ PairCore<R> spc[kMaxTypes * kMaxTypes];
  for (int k = threadIdx.x; k < ntypes * ntypes; k += blockDim.x) spc[k] = pcore[k];
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0 && ctl->acc_maxdist > half_skin) ctl->skin_violation = 1;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = gid / TPP, sub = gid % TPP;
  R fx = 0, fy = 0, fz = 0;
  double e_lj = 0, e_tab = 0, vir = 0;
  if (i < n) {
    const Vec4<R> xi = x4[i];
    const int ti = (int)xi.w;
    const int cnt = nn[i];
    const int4* row = reinterpret_cast<const int4*>(nlist + (size_t)i * S);
    const int pbase = ti * ntypes;
    // lane `sub` takes 4 consecutive neighbours per trip: one 16-byte index load, then four
    // independent position gathers in flight (rows are padded with the self index)
    for (int k = sub * 4; k < cnt; k += TPP * 4) {
      const int4 jj = row[k >> 2];
      const Vec4<R> xa = x4[jj.x], xb = x4[jj.y], xc = x4[jj.z], xd = x4[jj.w];
      const Vec4<R> xs[4] = {xa, xb, xc, xd};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const Vec4<R> xj = xs[u];
        const R dx = pos_delta(xi.x, xj.x, 0, box), dy = pos_delta(xi.y, xj.y, 1, box), dz = pos_delta(xi.z, xj.z, 2, box);
        R r2 = dx * dx + dy * dy + dz * dz;
        r2 = (k + u < cnt) ? r2 : (R)1e30;     // padding entries (self index) never interact
        const int pidx = pbase + (int)xj.w;
        pair_term<R, ENERGY>(spc[pidx], pext, pidx, tab, r2, dx, dy, dz, fx, fy, fz, e_lj, e_tab, vir);
      }
    }
  }
  if (TPP > 1) {
#pragma unroll
    for (int o = TPP / 2; o > 0; o >>= 1) {
      fx += __shfl_xor(fx, o); fy += __shfl_xor(fy, o); fz += __shfl_xor(fz, o);
    }
  }
  if (i < n && sub == 0) f4[i] = mk4<R>(fx, fy, fz, (R)0);
  if (ENERGY) {
    // full list: every pair is visited from both ends -> half weights
    __shared__ double red[3][4];
    for (int o = 32; o > 0; o >>= 1) { e_lj += __shfl_xor(e_lj, o); e_tab += __shfl_xor(e_tab, o); vir += __shfl_xor(vir, o); }
    if (lane_id() == 0) { red[0][threadIdx.x >> 6] = e_lj; red[1][threadIdx.x >> 6] = e_tab; red[2][threadIdx.x >> 6] = vir; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0, b = 0, c = 0;
      for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { a += red[0][k]; b += red[1][k]; c += red[2][k]; }
      eout[3 * blockIdx.x + 0] = 0.5 * a; eout[3 * blockIdx.x + 1] = 0.5 * b; eout[3 * blockIdx.x + 2] = 0.5 * c;
    }
  }
}

// =======================================================================================
// Tiled path (production for boxes with >= HX+2, HY+2, HZ+2 cells per axis).
//
// A tile = a block of HX x HY x HZ home cells.  Its stencil, (HX+2) x (HY+2) x (HZ+2) cells,
// is staged ONCE per workgroup into LDS with the periodic shift applied (coalesced 16-byte
// loads, each cache line fetched once), instead of 74 divergent 16-byte gathers per particle
// that cost one L1 line lookup each.  The neighbour list stores 16-bit slots of that LDS
// image (tile-local indices): half the HBM bytes of an int32 list, no minimum-image
// arithmetic in the pair loop, and the position gathers become ds_read_b128.
// The staged order is a pure function of cell_start, which only changes at a rebuild, so the
// build kernel and every later force launch see the same slot numbering.
// =======================================================================================
#ifndef CHEM_HX
#define CHEM_HX 3
#endif
#ifndef CHEM_HY
#define CHEM_HY 3
#endif
#ifndef CHEM_HZ
#define CHEM_HZ 3
#endif
constexpr int HX = CHEM_HX, HY = CHEM_HY, HZ = CHEM_HZ;   // ~490 home particles at 18/cell: one pass of a 512-thread block; stencil 5^3 cells (x4.6)
constexpr int SX = HX + 2, SY = HY + 2, SZ = HZ + 2;
constexpr int NROW = SY * SZ;          // x-rows of the stencil
constexpr int NHSEG = HY * HZ;         // home x-runs (contiguous in memory)

// Tiles along x: xs_nb tiles of HX cells, then tiles of xs_w cells (Box::xs_nb / xs_w).  A force launch of one-shot
// workgroups ends with a partly filled last round (1728 equal tiles on 768 resident slots: 2.25 rounds take the time of 3);
// a share of narrow tiles -- shorter jobs, scheduled last by the largest-first order -- fills it.  Splitting along x (the
// fastest tile index) gives every XCD's contiguous tile range the same mix.
__host__ __device__ inline int tile_nbx(int nx, int xs_nb) { return xs_nb * HX >= nx ? (nx + HX - 1) / HX : xs_nb; }      // wide tiles in a row (the last may be cut)
__host__ __device__ inline int tile_ntx(int nx, int xs_nb, int xs_w) {
  const int nb = tile_nbx(nx, xs_nb);
  return nb * HX >= nx ? nb : nb + (nx - nb * HX + xs_w - 1) / xs_w;
}
__host__ __device__ inline void tile_xrange(int tx, int nx, int xs_nb, int xs_w, int& cx0, int& hx) {
  const int nb = tile_nbx(nx, xs_nb);
  if (tx < nb) { cx0 = tx * HX; hx = HX < nx - cx0 ? HX : nx - cx0; }
  else { cx0 = nb * HX + (tx - nb) * xs_w; hx = xs_w < nx - cx0 ? xs_w : nx - cx0; }
}

// positions live in dynamic LDS (capacity chosen at run time from the cell occupancy):
// sx[0..cap], slot `total` is a far-away dummy used as row padding
template <typename R> struct TileLDS {
  int rowoff[NROW + 1];                // first slot of each stencil row
  int celloff[NROW][SX + 1];           // slot offset of every cell inside its row
  int cellg[NROW][SX];                 // global index of the first particle of the cell
  R cellshx[NROW][SX];                 // periodic shift in x (per cell), y/z (per row)
  R rowshy[NROW], rowshz[NROW];
  int hstart[NHSEG], hoff[NHSEG + 1];  // home x-runs: global start, prefix of counts
  int geom[8];                         // hx, hy, hz, total, nhome, hbase, origin cell (10 bits per axis)
  // list build only: x sub-bin prefixes of every stencil cell (cell_sub, see dev_sort_gather; -1 = unknown),
  // staged coordinates of the lower corner of stencil cell (0,0,0), cell edges, sub-bins per unit length
  int cellsub[NROW][SX];
  // org: reference point of the staged coordinates = lower corner of stencil cell (0,0,0) + kRefCells cell edges (the centre of
  // a full 5x5x5 stencil: |staged coordinate| <= 2.5 edges + skin/2, the smallest magnitudes fp32 can be given), in the
  // encoding of the position arrays (pos_enc); clen: cell edges; subinv: sub-bins per unit length; qs: fixed-point scale
  R org[3], clen[3], subinv, pad_;
  R qs[3], pad2_;
};
constexpr float kRefCells = 2.5f;

// NOTE: the pointer must stay a plain local derived from the extern array (no integer
// round-trip, never stored in memory) so that the compiler keeps it in the LDS address space
// and emits ds_read/ds_write instead of flat_* accesses.
#define CHEM_DYN_LDS(R) \
  extern __shared__ __attribute__((aligned(16))) unsigned char chem_dyn_lds[]; \
  Vec4<R>* const sx = reinterpret_cast<Vec4<R>*>(chem_dyn_lds)

// Descriptor tables of one tile, computed ONCE per rebuild by k_tile_desc and kept in HBM
// (~1.3 KB per tile); every later launch copies them into LDS with one coalesced read instead
// of re-deriving them through dependent global loads and three barrier phases.
template <typename R>
__device__ __forceinline__ void tile_tables(TileLDS<R>& T, const int CAP, int tile, const int* __restrict__ cell_start,
                                            const Box<R>& box, DevCtl* ctl, const int* __restrict__ cell_sub = nullptr) {
  const int nx = box.nc[0], ny = box.nc[1], nz = box.nc[2];
  const int ntx = tile_ntx(nx, box.xs_nb, box.xs_w), nty = (ny + HY - 1) / HY;
  const int tx = tile % ntx, ty = (tile / ntx) % nty, tz = tile / (ntx * nty);
  // z-ghost mode: layers 1..nz-2 are own, the stencil reaches into the ghost layers 0 and nz-1
  const int zg = box.zghost;
  int cx0, hx;
  tile_xrange(tx, nx, box.xs_nb, box.xs_w, cx0, hx);
  const int cy0 = ty * HY, cz0 = tz * HZ + zg;
  const int hy = min(HY, ny - cy0), hz = min(HZ, nz - zg - cz0);
  const int t = threadIdx.x;
  if (t < NROW * SX) {
    const int r = t / SX, k = t % SX, ry = r % SY, rz = r / SY;
    int cnt = 0, g = 0, sub = 0; R shx = 0;
    if (ry < hy + 2 && rz < hz + 2 && k < hx + 2) {
      int ox = cx0 - 1 + k, oy = cy0 - 1 + ry, oz = cz0 - 1 + rz;
      R shy = 0, shz = 0;
      if (ox < 0) { ox += nx; shx = pos_shift(-1, 0, box); } else if (ox >= nx) { ox -= nx; shx = pos_shift(1, 0, box); }
      if (oy < 0) { oy += ny; shy = pos_shift(-1, 1, box); } else if (oy >= ny) { oy -= ny; shy = pos_shift(1, 1, box); }
      if (zg) { shz = oz == 0 ? box.shz_lo : (oz == nz - 1 ? box.shz_hi : (R)0); }
      else if (oz < 0) { oz += nz; shz = pos_shift(-1, 2, box); } else if (oz >= nz) { oz -= nz; shz = pos_shift(1, 2, box); }
      const int oc = (oz * ny + oy) * nx + ox;
      g = cell_start[oc]; cnt = cell_start[oc + 1] - g;
      sub = cell_sub ? cell_sub[oc] : -1;
      if (k == 0) { T.rowshy[r] = shy; T.rowshz[r] = shz; }
    } else if (k == 0) { T.rowshy[r] = 0; T.rowshz[r] = 0; }
    T.cellg[r][k] = g; T.cellshx[r][k] = shx; T.cellsub[r][k] = sub;
    T.celloff[r][k + 1] = cnt;   // turned into a prefix below
  }
  __syncthreads();
  if (t < NROW) {
    int o = 0; T.celloff[t][0] = 0;
    for (int k = 0; k < SX; ++k) { o += T.celloff[t][k + 1]; T.celloff[t][k + 1] = o; }
  }
  __syncthreads();
  // Third step, three waves side by side, everything from the LDS tables of the first two steps (the home cells are
  // among the 125 stencil cells): a single thread walking 25 rows and 9 home runs with dependent loads cost ~10 us
  // per tile in the rebuild, where these tables are computed (the force kernel reads them back from HBM).
  const int wv = t >> 6, ln = t & 63;
  if (wv == 0) {          // first slot of every stencil row
    const int len = ln < NROW ? T.celloff[ln][SX] : 0;
    int incl = len;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { const int u = __shfl_up(incl, o); if (ln >= o) incl += u; }
    if (ln < NROW) T.rowoff[ln] = incl - len;
    if (ln == NROW - 1) {
      T.rowoff[NROW] = incl;
      if (incl > CAP) atomicMax(&ctl->stage_overflow, incl);
      T.geom[3] = incl < CAP ? incl : CAP;
    }
  } else if (wv == 1) {   // (blocks of >= 128 threads) home x-runs (contiguous in memory): global start, prefix of counts
    const int hyi = ln % HY, hzi = ln / HY;
    int st = 0, cn = 0;
    if (ln < NHSEG && hyi < hy && hzi < hz) {
      const int hr = (hzi + 1) * SY + (hyi + 1);
      st = T.cellg[hr][1]; cn = T.celloff[hr][hx + 1] - T.celloff[hr][1];
    }
    int incl = cn;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { const int u = __shfl_up(incl, o); if (ln >= o) incl += u; }
    if (ln < NHSEG) { T.hstart[ln] = st; T.hoff[ln] = incl - cn; }
    if (ln == NHSEG - 1) { T.hoff[NHSEG] = incl; T.geom[4] = incl; }
  }
  if (t == (int)blockDim.x - 1) {
    T.geom[0] = hx; T.geom[1] = hy; T.geom[2] = hz;
    T.geom[6] = cx0 | (cy0 << 10) | (cz0 << 20);   // first home cell (list build: slot of an excluded partner)
    T.geom[7] = 0;                                 // set by list_stage_f32: a stencil cell without x-slice information
    // geometry of the staged image (list build, x-window of a row).  In z-ghost mode layer l of the slab is the
    // global layer z0g + l - 1; the staged z of the ghost layers carries shz_lo / shz_hi, which continues the
    // same affine map across the periodic boundary.
    const double clx = 2.0 * box.qh[0] / nx, cly = 2.0 * box.qh[1] / ny, clz = 2.0 * box.qh[2] / (zg ? box.nzg : nz);   // (qh = L / 2 in fp64)
    T.clen[0] = (R)clx; T.clen[1] = (R)cly; T.clen[2] = (R)clz;
    T.org[0] = pos_enc(((double)(cx0 - 1) + kRefCells) * clx, 0, box); T.org[1] = pos_enc(((double)(cy0 - 1) + kRefCells) * cly, 1, box);
    T.org[2] = pos_enc(((double)(zg ? box.z0g + cz0 - 2 : cz0 - 1) + kRefCells) * clz, 2, box);
    T.subinv = (R)((double)NSUB / clx); T.pad_ = 0;
    T.qs[0] = box.qsf[0]; T.qs[1] = box.qsf[1]; T.qs[2] = box.qsf[2]; T.pad2_ = 0;
  }
  __syncthreads();
}

template <typename R>
__global__ __launch_bounds__(128) void k_tile_desc(int ntiles, int CAP, const int* __restrict__ cell_start, Box<R> box,
                                                   TileLDS<R>* __restrict__ desc, DevCtl* ctl, const int* __restrict__ cell_sub) {
  if (!ctl->need_rebuild) return;
  __shared__ TileLDS<R> T;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    tile_tables<R>(T, CAP, tile, cell_start, box, ctl, cell_sub);
    const int* src = reinterpret_cast<const int*>(&T);
    int* dst = reinterpret_cast<int*>(&desc[tile]);
    for (int k = threadIdx.x; k < (int)(sizeof(TileLDS<R>) / 4); k += blockDim.x) dst[k] = src[k];
  }
}

template <typename R>
__device__ __forceinline__ void tile_load_desc(TileLDS<R>& T, const TileLDS<R>* __restrict__ desc, int tile) {
  const int* src = reinterpret_cast<const int*>(&desc[tile]);
  int* dst = reinterpret_cast<int*>(&T);
  for (int k = threadIdx.x; k < (int)(sizeof(TileLDS<R>) / 4); k += blockDim.x) dst[k] = src[k];
}

// stages the stencil of the tile described by T into sx.  wmode 0: .w = particle type (force
// kernel), 1: .w = (global index << 5 | type) (list build; bit 4 stays clear so that v_bfe can take .w as its offset).  Caller synchronises afterwards.
// All global loads of a wave are issued before the first LDS write so that the staging costs one
// memory latency, not one per row chunk.
// D3 (fp64 force kernel): 24-byte slots (x, y, z) with the types in a byte array behind the image instead of 32-byte
// (x, y, z, type) slots -- 70 KB instead of 90 KB at C5, i.e. two workgroups per CU instead of one
template <typename R> __device__ __forceinline__ unsigned char* d3_types(Vec4<R>* sx, int CAP) { return reinterpret_cast<unsigned char*>(sx) + (size_t)(CAP + 1) * 24; }
template <typename R, bool D3>
__device__ __forceinline__ void slot_store(Vec4<R>* const sx, const int CAP, int dst, const Vec4<R>& p) {
  if constexpr (D3) {
    R* s3 = reinterpret_cast<R*>(sx) + 3 * dst;
    s3[0] = p.x; s3[1] = p.y; s3[2] = p.z;
    d3_types<R>(sx, CAP)[dst] = (unsigned char)(int)p.w;
  } else sx[dst] = p;
}
template <typename R, int BS, bool LEAN = false, bool D3 = false>
__device__ __forceinline__ void tile_fill(const TileLDS<R>& T, Vec4<R>* const sx, const int CAP,
                                          const Vec4<R>* __restrict__ x4, int wmode) {
  constexpr int NW = BS / 64, RPW = (NROW + NW - 1) / NW;
  const int t = threadIdx.x;
  const int w = t >> 6, l = t & 63;
  if (t == 0) slot_store<R, D3>(sx, CAP, T.geom[3], mk4<R>((R)1e18, (R)1e18, (R)1e18, (R)0));
  if constexpr (LEAN) {
    // register-lean variant (list build: staging is <2 % of the tile's time, occupancy matters more)
    for (int r = w; r < NROW; r += NW) {
      const int len = T.celloff[r][SX], o0 = T.rowoff[r];
      for (int e = l; e < len; e += 64) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < SX; ++q) k += (e >= T.celloff[r][q]) ? 1 : 0;
        const int g = T.cellg[r][k] + (e - T.celloff[r][k]);
        const int dst = o0 + e;
        if (dst < CAP) {
          Vec4<R> p = x4[g];
          p.x = pos_local(p.x, T.cellshx[r][k], T.org[0], T.qs[0]); p.y = pos_local(p.y, T.rowshy[r], T.org[1], T.qs[1]); p.z = pos_local(p.z, T.rowshz[r], T.org[2], T.qs[2]);
          if (wmode) p.w = idx_as_real((g << 5) | (int)p.w, (R)0);
          slot_store<R, D3>(sx, CAP, dst, p);
        }
      }
    }
    return;
  }
  Vec4<R> pv[RPW][2];
  int pk_[RPW][2], pg[RPW][2];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int r = w + rr * NW;
    const int len = r < NROW ? T.celloff[r][SX] : 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int e = l + 64 * c;
      pk_[rr][c] = -1;
      if (e < len) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < SX; ++q) k += (e >= T.celloff[r][q]) ? 1 : 0;
        const int g = T.cellg[r][k] + (e - T.celloff[r][k]);
        pk_[rr][c] = k; pg[rr][c] = g;
        pv[rr][c] = x4[g];
      }
    }
  }
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int r = w + rr * NW;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int k = pk_[rr][c];
      if (k >= 0) {
        const int dst = T.rowoff[r] + l + 64 * c;
        if (dst < CAP) {
          Vec4<R> p = pv[rr][c];
          p.x = pos_local(p.x, T.cellshx[r][k], T.org[0], T.qs[0]); p.y = pos_local(p.y, T.rowshy[r], T.org[1], T.qs[1]); p.z = pos_local(p.z, T.rowshz[r], T.org[2], T.qs[2]);
          if (wmode) p.w = idx_as_real((pg[rr][c] << 5) | (int)p.w, (R)0);
          slot_store<R, D3>(sx, CAP, dst, p);
        }
      }
    }
  }
  // rows longer than 128 particles (crowded cells): remainder, serial
  for (int r = w; r < NROW; r += NW) {
    const int len = T.celloff[r][SX], o0 = T.rowoff[r];
    for (int e = l + 128; e < len; e += 64) {
      int k = 0;
#pragma unroll
      for (int q = 1; q < SX; ++q) k += (e >= T.celloff[r][q]) ? 1 : 0;
      const int g = T.cellg[r][k] + (e - T.celloff[r][k]);
      const int dst = o0 + e;
      if (dst < CAP) {
        Vec4<R> p = x4[g];
        p.x = pos_local(p.x, T.cellshx[r][k], T.org[0], T.qs[0]); p.y = pos_local(p.y, T.rowshy[r], T.org[1], T.qs[1]); p.z = pos_local(p.z, T.rowshz[r], T.org[2], T.qs[2]);
        if (wmode) p.w = idx_as_real((g << 5) | (int)p.w, (R)0);
        slot_store<R, D3>(sx, CAP, dst, p);
      }
    }
  }
}

// List build, fp32: LDS layout of one staged tile (inside the same dynamic LDS block the force kernel uses for its
// float4 image -- 12 + ntypes/8 bytes per slot instead of 16):
//   img   groups of FOUR slots in SoA order, 64 bytes (+16 of padding, kGrpF) per group: u0..u3 | v0..v3 | w0..w3 | q0..q3 with (u,v,w) the
//         position relative to the lower corner of the tile's stencil (|u| <= 5 cell edges) and q = u^2+v^2+w^2.  The
//         test of a candidate is  rl^2 + delta - |ri|^2 - q + 2 ri.(u,v,w) >= 0 : one packed add and three packed FMAs per
//         TWO candidates instead of 3 sub + 1 mul + 2 fma + 1 sub each -- the loop is bound by VALU issue.  The
//         expanded form loses ~6e-5 absolute on r^2 (coordinates <= 14), which `delta` covers: the 16-bit force list
//         is a SUPERSET by a shell of 5e-5 (the force kernel applies the exact cutoff); the int32 rows re-test their
//         hits with the difference form and stay exact;
//   tmask one bit per slot and home type ti: "the pair (ti, type of the slot) carries a potential" (MSB first: slot s
//         is bit 31 - (s & 31) of word s >> 5).  The type filter of a 32-candidate segment is one funnel shift of two
//         words instead of two instructions per candidate;
//   bnd   slot offset (inside its row) of every x-slice boundary of every stencil row: 5 cells x 4 slices + 1 values,
//         so that the x-window of a row is two 16-bit reads (was: two table reads and bit-field arithmetic per end).
// geom[7] of the tile tables is set when a stencil cell has no slice information (crowded cell): whole-cell windows then.
struct ActMask { unsigned int row[kMaxTypes]; };   // bit tj of row[ti]: pair (ti,tj) has a potential
#define CHEM_LDS __attribute__((address_space(3)))
constexpr int NBND = SX * NSUB + 1;
#ifndef CHEM_LIST_DELTA
#define CHEM_LIST_DELTA 2.5e-4f
#endif
constexpr float kListDelta = CHEM_LIST_DELTA;
// floats per group of four slots in the list-build image: 16 used + 4 of padding.  With a stride of 16 words the reads
// of lanes on groups g and g + 4 hit the same banks (SQ_LDS_BANK_CONFLICT was 47 % of the LDS cycles of the rebuild);
// 20 words put 16 consecutive groups on 16 different bank quads.
constexpr int kGrpF = 20;   // see the layout note below: what the expanded distance form may lose on r^2
struct ListLDS { int img_bytes, nwords, tmask_off, bnd_off; };
__device__ __forceinline__ ListLDS list_lds_layout(int CAP, int ntypes) {
  ListLDS L;
  L.img_bytes = (((CAP + 3) >> 2) + 3) * kGrpF * 4;          // (+ three groups: the pipelined reads run up to two groups past the last slot)
  L.nwords = ((CAP + 31) >> 5) + 2;
  L.tmask_off = L.img_bytes;
  L.bnd_off = L.tmask_off + ntypes * L.nwords * 4;
  return L;
}
__host__ __device__ constexpr size_t list_lds_bytes(int CAP, int ntypes) {
  return (size_t)(((CAP + 3) >> 2) + 3) * kGrpF * 4 + (size_t)ntypes * (((CAP + 31) >> 5) + 2) * 4 + (size_t)NROW * NBND * 2 + 16;
}

// zeroes the type masks (call before the workgroup barriers of tile_tables) ...
template <int BS>
__device__ __forceinline__ void list_stage_clear(unsigned char* lds, const ListLDS& L, int ntypes) {
  CHEM_LDS unsigned int* tm = (CHEM_LDS unsigned int*)((CHEM_LDS unsigned char*)lds + L.tmask_off);
  for (int k = threadIdx.x; k < ntypes * L.nwords; k += BS) tm[k] = 0u;
}
// ... and stages coordinates, type masks and slice boundaries of the tile described by T (caller synchronises afterwards)
// RS = precision of the particle arrays and tile tables: the image itself is always fp32 (tile-local coordinates, the
// subtraction of the stencil corner done in RS) -- the force list it yields may be a superset, so the fp64 build uses
// this list build too and keeps its exact fp64 one for the int32 Verlet rows only
template <typename RS, int BS>
__device__ __forceinline__ void list_stage_f32(TileLDS<RS>& T, unsigned char* lds, const ListLDS& L, const int CAP,
                                               const Vec4<RS>* __restrict__ x4, const ActMask& act, int ntypes) {
  constexpr int NW = BS / 64;
  // explicit LDS address space: through a generic pointer these become flat_* accesses (and flat atomics)
  CHEM_LDS unsigned char* const l3 = (CHEM_LDS unsigned char*)lds;
  CHEM_LDS float* img = (CHEM_LDS float*)l3;
  CHEM_LDS unsigned int* tm = (CHEM_LDS unsigned int*)(l3 + L.tmask_off);
  CHEM_LDS unsigned short* bnd = (CHEM_LDS unsigned short*)(l3 + L.bnd_off);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int t = threadIdx.x; t < NROW * NBND; t += BS) {
    const int r = t / NBND, f = t - r * NBND, k = f >> 2, j = f & 3;
    int v = T.celloff[r][k < SX ? k : SX];
    if (k < SX && j) {
      const int pk = T.cellsub[r][k];
      if (pk < 0) { if (T.celloff[r][k + 1] > T.celloff[r][k]) T.geom[7] = 1; }
      else v += (pk >> (8 * (j - 1))) & 0xff;
    }
    bnd[t] = (unsigned short)v;
  }
  for (int r = w; r < NROW; r += NW) {
    const int len = T.celloff[r][SX], o0 = T.rowoff[r];
    for (int e0 = 0; e0 < len; e0 += 64) {          // (wave-uniform trip count: the type masks are built with ballots)
      const int e = e0 + l;
      int tj = -1;
      if (e < len) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < SX; ++q) k += (e >= T.celloff[r][q]) ? 1 : 0;
        const int g = T.cellg[r][k] + (e - T.celloff[r][k]);
        const int dst = o0 + e;
        if (dst < CAP) {
          const Vec4<RS> p = x4[g];
          CHEM_LDS float* grp = img + (dst >> 2) * kGrpF + (dst & 3);
          const float u = (float)pos_local(p.x, T.cellshx[r][k], T.org[0], T.qs[0]), v = (float)pos_local(p.y, T.rowshy[r], T.org[1], T.qs[1]),
                      w_ = (float)pos_local(p.z, T.rowshz[r], T.org[2], T.qs[2]);
          grp[0] = u; grp[4] = v; grp[8] = w_; grp[12] = fmaf(u, u, fmaf(v, v, w_ * w_));   // (explicit: the same rounding in every kernel this is inlined into)
          tj = (int)p.w & 15;
        }
      }
      // 64 consecutive slots = 2 or 3 mask words: one ballot per home type, the three words assembled by shifts and
      // bit reversal, one atomic each (a per-slot atomic OR is a 32-way same-address conflict in the LDS)
      const int slot0 = o0 + e0, sh = slot0 & 31, wbase = slot0 >> 5;
      for (int ti = 0; ti < ntypes; ++ti) {
        const unsigned long long m = __ballot(tj >= 0 && ((act.row[ti] >> tj) & 1u));
        if (m == 0ull) continue;
        const unsigned long long lo = m << sh;
        const unsigned int w2 = sh ? (unsigned int)(m >> (64 - sh)) : 0u;
        const unsigned int part = l == 0 ? (unsigned int)lo : (l == 1 ? (unsigned int)(lo >> 32) : w2);
        if (l < 3 && part) __hip_atomic_fetch_or(&tm[ti * L.nwords + wbase + l], __brev(part), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load_u4(const uint4* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// XCD-aware work-item order (speed only, never correctness): workgroups are dealt round-robin
// over the 8 XCDs, so block b and b+8 share an L2.  Remapping gives every XCD one contiguous
// range of tiles (a z-slab of the box): its 4 MiB L2 then holds the slab's positions and the
// stencil staging hits L2 instead of pulling the whole position array through the fabric.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


// ---- list build on tiles.  Two products from one sweep:
//   nl16/nnh  : 16-bit LDS slots of the pairs that carry a non-bonded potential (type-pair mask
//               `act`; e.g. A-B and A-D of chain_growth_catalytic/topol.top:16-17 are off) -- read by
//               the force kernel every step.  Layout is tile-major and transposed: the 8-slot chunk c
//               of home particle q of a tile lives at  base + (c*nhome + q)*8,  so that consecutive
//               lanes (= consecutive home particles) read consecutive 16-byte words.
//   nlist/nn  : int32 global indices of ALL non-excluded pairs within rc+skin (the Verlet list the
//               reaction scan and diagnostics traverse) -- only when `nlist` is non-null.
// One lane per home particle walks the nine x-runs of its 27-cell stencil in the staged LDS image:
// no cross-lane traffic, every lane does useful work (the wave-per-particle ballot version spent
// ~7x more instructions per accepted pair).

// list build of ONE staged tile (T and sx filled, block synchronised): see k_nlist_tiles
template <typename R, int BS>
__device__ __forceinline__ void dev_nlist_tile(const TileLDS<R>& T, Vec4<R>* const sx, const int* tag, const R rl2,
                                               const int* excl_start, const int* excl_list, const int has_excl, const ActMask& act,
                                               unsigned short* nl16, const int S16, int* nnh, int* nlist, const int S, int* nn, DevCtl* ctl,
                                               const Box<R>* bx = nullptr, const int* rtag = nullptr, const Vec4<R>* x4 = nullptr, const R rl2_rows_ = (R)-1) {
    const R rl2_rows = rl2_rows_ > (R)0 ? rl2_rows_ : rl2;   // int32 Verlet rows: the workload's rc+skin (see dev_nlist_tile_f32)
    const int hx = T.geom[0], total = T.geom[3], nhome = T.geom[4], hbase = T.geom[5];
    unsigned short* reg16 = nl16 + (size_t)hbase * S16;
    for (int q = threadIdx.x; q < nhome; q += BS) {
      int sgi = 0;
#pragma unroll
      for (int k = 1; k < NHSEG; ++k) sgi += (q >= T.hoff[k]) ? 1 : 0;
      const int inrun = q - T.hoff[sgi];
      const int p = T.hstart[sgi] + inrun;
      const int ly = sgi % HY, lz = sgi / HY;
      const int hr = (lz + 1) * SY + (ly + 1);
      const int eh = inrun + T.celloff[hr][1];
      int lx = 0;
      for (int k = 2; k <= hx; ++k) lx += (eh >= T.celloff[hr][k]) ? 1 : 0;
      const int sself = T.rowoff[hr] + eh;
      const Vec4<R> xi = sx[sself];
      const unsigned int arow = act.row[real_as_idx(xi.w) & 15];
      int e0 = 0, e1 = 0;
      if (has_excl) { const int tg = tag[p]; e0 = excl_start[tg]; e1 = excl_start[tg + 1]; }
      int cnt = 0, cnt16 = 0;
      int* row32 = nlist ? nlist + (size_t)p * S : nullptr;
      uint4* regq = reinterpret_cast<uint4*>(reg16) + q;   // chunk c of this particle: regq[c * nhome]
      // accepted slots are shifted into a 128-bit register; every 8th append stores one whole
      // 16-byte chunk (instead of eight scattered 2-byte stores)
      uint4 acc = make_uint4(0, 0, 0, 0);
      auto push = [&](unsigned int sl) {
        acc.x = __builtin_amdgcn_alignbit(acc.y, acc.x, 16);
        acc.y = __builtin_amdgcn_alignbit(acc.z, acc.y, 16);
        acc.z = __builtin_amdgcn_alignbit(acc.w, acc.z, 16);
        acc.w = (acc.w >> 16) | (sl << 16);
        if ((cnt16 & 7) == 7 && cnt16 < S16) regq[(size_t)(cnt16 >> 3) * nhome] = acc;
        ++cnt16;
      };
      auto hit = [&](int s, int jw) {
        const int j = jw >> 5;
        bool ok = j != p;
        if (ok && e1 > e0) {
          const int tgj = tag[j];
          for (int e = e0; e < e1; ++e) if (excl_list[e] == tgj) { ok = false; break; }
        }
        if (ok) {
          if ((arow >> (jw & 15)) & 1u) push((unsigned int)s);
          if (row32) {
            bool in_rows = true;
            if (rl2_rows < rl2) { const Vec4<R> xj = sx[s]; const R dx = xi.x - xj.x, dy_ = xi.y - xj.y, dz_ = xi.z - xj.z; in_rows = dx * dx + dy_ * dy_ + dz_ * dz_ <= rl2_rows; }
            if (in_rows) { if (cnt < S) row32[cnt] = j; ++cnt; }
          }
        }
      };
      // Two passes per x-run segment of <= 64 staged candidates:
      //  1. distance tests only, results OR-ed into a 64-bit per-lane hit mask (no branches);
      //  2. the few hits (~15 %) are peeled off the mask one by one for the type / exclusion /
      //     activity filters and the chunked store.
      // The branchy hit handling no longer sits in the loop that runs 475 times per particle.
      // x-window of every stencil row: a candidate in row (dy, dz) is at least (dyv, dzv) away in y and z
      // (distance of the home particle from that row's slab), so only x within sqrt(rl^2 - dyv^2 - dzv^2)
      // can be a neighbour -- on average 46 % of the 3-cell run.  The cells are sorted by x sub-bin
      // (dev_sort_gather), whose prefix counts turn the window into a slot range; `weps` absorbs every
      // rounding difference between the binning arithmetic and this one (bins only prune, the distance test
      // below decides membership).
      auto suboff = [&](int r, int f, bool lower) {
        const int k = f >> 2, j = f & 3;
        const int base = T.celloff[r][k];
        if (j == 0) return base;
        const int pk = T.cellsub[r][k];
        if (pk < 0) return lower ? base : T.celloff[r][k + 1];
        return base + ((pk >> (8 * (j - 1))) & 0xff);
      };
#pragma unroll 1
      for (int dzy = 0; dzy < 9; ++dzy) {
        const int dz = dzy / 3, dy = dzy - 3 * dz;
        const int r = (lz + dz) * SY + (ly + dy);
        // (everything is recomputed per row from the LDS tables: keeping it live across the row loop spills)
        const R weps = T.clen[0] * (R)2e-4;
        const R ylo = ((R)(ly + 1) - (R)kRefCells) * T.clen[1], zlo = ((R)(lz + 1) - (R)kRefCells) * T.clen[2];   // (staged coordinates are relative to T.org)
        R ddy = dy == 0 ? xi.y - ylo - weps : (dy == 2 ? ylo + T.clen[1] - xi.y - weps : (R)0);
        R ddz = dz == 0 ? xi.z - zlo - weps : (dz == 2 ? zlo + T.clen[2] - xi.z - weps : (R)0);
        ddy = ddy > 0 ? ddy : (R)0; ddz = ddz > 0 ? ddz : (R)0;
        const R w2 = rl2 - ddy * ddy - ddz * ddz;
        if (w2 < (R)0) continue;
        const R ws = (sqrt_r(w2) + weps) * T.subinv, sxi = (xi.x + (R)kRefCells * T.clen[0]) * T.subinv;
        int f_lo = (int)(sxi - ws), f_hi = (int)(sxi + ws);     // truncation == floor where it matters (clamped below at >= 0)
        f_lo = f_lo > lx * NSUB ? f_lo : lx * NSUB;
        f_hi = f_hi < (lx + 3) * NSUB - 1 ? f_hi : (lx + 3) * NSUB - 1;
        const int a = T.rowoff[r] + suboff(r, f_lo, true);
        int b = T.rowoff[r] + suboff(r, f_hi + 1, false);
        b = b < total ? b : total;
        for (int s0 = a; s0 < b; s0 += 64) {
          const int len = (b - s0) < 64 ? (b - s0) : 64;
          unsigned int mlo = 0, mhi = 0;
          for (int k = 0; k < len; k += 4) {
            Vec4<R> xj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xj[u] = sx[(k + u < len) ? s0 + k + u : total];   // past the run: far-away dummy
            unsigned int bits = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const R dx = xi.x - xj[u].x, dy_ = xi.y - xj[u].y, dz_ = xi.z - xj[u].z;
              const R r2 = dx * dx + dy_ * dy_ + dz_ * dz_;
              bits |= (r2 <= rl2) ? (1u << u) : 0u;
            }
            if (k < 32) mlo |= bits << k; else mhi |= bits << (k - 32);
          }
          unsigned long long m = ((unsigned long long)mhi << 32) | mlo;
          while (m) {
            const int k = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int s = s0 + k;
            hit(s, real_as_idx(sx[s].w));
          }
        }
      }
      // pad the last chunk with the far-away dummy slot (chunks are read whole)
      const int real16 = cnt16;
      if (real16 <= S16) while (cnt16 & 7) push((unsigned int)total);
      nnh[hbase + q] = real16 < S16 ? real16 : S16;
      if (real16 > S16) atomicMax(&ctl->nl_overflow, real16);
      if (row32) {
        const int c32 = cnt < S ? cnt : S;
        for (int k = c32; k < ((c32 + 3) & ~3); ++k) row32[k] = p;
        nn[p] = c32;
        if (cnt > S) atomicMax(&ctl->nl_overflow, cnt);
      }
    }
}

// fp32 list build of ONE tile staged by list_stage_f32 (workgroup synchronised).  Same products as dev_nlist_tile.
// Per home particle (one lane each) and stencil row: x-window [a, b) from the slice-boundary table, then segments of
// up to 32 candidates = 8 groups of four slots, two groups per trip on ping-pong registers: the four ds_read_b128 of the
// group after next are in flight while one is tested, and the window of the NEXT row is looked up before this row's
// candidates are walked, so that the LDS latencies hide behind arithmetic instead of adding up (16 waves per CU: the
// rebuild kernel runs two workgroups per CU at 128 registers).  A test (expanded form, see the layout note above)
// leaves its result in the sign bit of rl^2 + delta - r^2, which one v_alignbit per candidate
// shifts into the segment's miss mask.  Hits surviving the type mask (and, located as slots, the self pair and up to
// four excluded partners) are peeled off and appended to the lane's 16-byte chunk register.
template <typename RS, int BS>
__device__ __forceinline__ void dev_nlist_tile_f32(const TileLDS<RS>& T, unsigned char* lds, const ListLDS& L, const int* tag, const float rl2,
                                                   const int* excl_start, const int* excl_list, const int has_excl,
                                                   unsigned short* nl16, const int S16, int* nnh, int* nlist, const int S, int* nn, DevCtl* ctl,
                                                   const Box<RS>* bx, const int* rtag, const Vec4<RS>* x4, const int ablate = 0, const float rl2_rows_ = -1.f,
                                                   uint4* bslots = nullptr, const int* gtag = nullptr, const int real0 = 0, const int real1 = 0x7fffffff) {
  // (gtag, real0, real1: slab decomposition -- index of a tag's GHOST copy on this rank, range of the real particles)
  const float rl2_rows = rl2_rows_ > 0.f ? rl2_rows_ : rl2;   // int32 Verlet rows: the workload's rc+skin (the 16-bit force list may use a wider list skin)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) const volatile f32x4 lds_f32x4;   // volatile: keeps the 16-byte reads and their order
  CHEM_LDS unsigned char* const l3 = (CHEM_LDS unsigned char*)lds;   // (explicit LDS address space: ds_read, not flat_load)
  const CHEM_LDS float* img = (const CHEM_LDS float*)l3;
  const CHEM_LDS unsigned int* tmask = (const CHEM_LDS unsigned int*)(l3 + L.tmask_off);
  const CHEM_LDS unsigned short* bnd = (const CHEM_LDS unsigned short*)(l3 + L.bnd_off);
  const int hx = T.geom[0], total = T.geom[3], nhome = T.geom[4], hbase = T.geom[5];
  const bool nowin = T.geom[7] != 0;
  unsigned short* reg16 = nl16 + (size_t)hbase * S16;
  for (int q = threadIdx.x; q < nhome; q += BS) {
    int sgi = 0;
#pragma unroll
    for (int k = 1; k < NHSEG; ++k) sgi += (q >= T.hoff[k]) ? 1 : 0;
    const int inrun = q - T.hoff[sgi];
    const int p = T.hstart[sgi] + inrun;
    const int ly = sgi % HY, lz = sgi / HY;
    const int hr = (lz + 1) * SY + (ly + 1);
    const int eh = inrun + T.celloff[hr][1];
    int lx = 0;
    for (int k = 2; k <= hx; ++k) lx += (eh >= T.celloff[hr][k]) ? 1 : 0;
    const int sself = T.rowoff[hr] + eh;
    const CHEM_LDS float* self = img + (sself >> 2) * kGrpF + (sself & 3);
    const float xix = self[0], xiy = self[4], xiz = self[8];             // relative to the stencil's lower corner
    const float cq = rl2 + kListDelta - self[12];
    const f32x2 ax = {2.f * xix, 2.f * xix}, ay = {2.f * xiy, 2.f * xiy}, az = {2.f * xiz, 2.f * xiz}, cc = {cq, cq};
    const int ti = (int)x4[p].w & 15;                 // home cells carry no periodic shift: x4[p] is the staged particle
    const CHEM_LDS unsigned int* tmrow = tmask + ti * L.nwords;
    int e0 = 0, e1 = 0;
    if (has_excl && ablate != 5) { const int tg = tag[p]; e0 = excl_start[tg]; e1 = excl_start[tg + 1]; }      // (ablate 5, diagnostic: exclusions ignored)
    int cnt = 0, cnt16 = 0;
    int* row32 = nlist ? nlist + (size_t)p * S : nullptr;
    // Exclusions without leaving the plain path: the few excluded partners of a particle are located
    // in the staged tile ONCE (tag -> index -> position -> cell -> slot, the binning arithmetic repeated on the same
    // bits) and their bits are cleared from the hit masks, like the self pair.  More than 4 exclusions: generic path.
    // On a slab (z-ghost mode) a partner across the slab boundary is a GHOST copy with its own index: rtag names the real
    // particle where it is owned here, else the ghost; gtag names the ghost copy in any case (a one-rank slab holds both).
    // The copy that counts is the one in a cell layer next to the home particle's (a partner is closer than one cell edge).
    int xs0 = -1, xs1 = -1, xs2 = -1, xs3 = -1;
    bool fastx = false;
    if (bx && e1 > e0 && e1 - e0 <= 4 && !row32) {
      fastx = true;
      const int org = T.geom[6];
      const int nx = bx->nc[0], ny = bx->nc[1], nz = bx->nc[2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int slot = -1;
        if (e0 + k < e1) {
          const int tj = excl_list[e0 + k];
          int g = rtag[tj];
          for (int copy = 0; copy < 2 && slot < 0; ++copy) {      // (second trip: slabs only, the tag's ghost copy)
            if (copy) { if (!gtag) break; const int g2 = gtag[tj]; if (g2 == g) break; g = g2; }
            if (g < 0) continue;
            const Vec4<RS> xg = x4[g];
            const int cx = pos_cell(xg.x, 0, nx, *bx), cy = pos_cell(xg.y, 1, ny, *bx);   // (the binning arithmetic on the same bits)
            int kx = cx - (org & 1023) + 1, ky = cy - ((org >> 10) & 1023) + 1, kz;
            kx += kx < 0 ? nx : 0; kx -= kx >= nx ? nx : 0;
            ky += ky < 0 ? ny : 0; ky -= ky >= ny ? ny : 0;
            if (bx->zghost) {      // local layer: ghosts by their index range, reals by the binning arithmetic (dev_bin)
              const int lzc = g < real0 ? 0 : (g >= real1 ? nz - 1 : pos_cell(xg.z, 2, bx->nzg, *bx) - bx->z0g + 1);
              kz = lzc - (org >> 20) + 1;
              const int dl = kz - (lz + 1);
              if (kz < 0 || dl < -1 || dl > 1) continue;      // the far copy (or none here)
            } else {
              kz = pos_cell(xg.z, 2, nz, *bx) - (org >> 20) + 1;
              kz += kz < 0 ? nz : 0; kz -= kz >= nz ? nz : 0;
            }
            if (kx < hx + 2 && ky < T.geom[1] + 2 && kz < T.geom[2] + 2) {
              const int rr = kz * SY + ky, off = g - T.cellg[rr][kx];
              if (off >= 0 && off < T.celloff[rr][kx + 1] - T.celloff[rr][kx]) slot = T.rowoff[rr] + T.celloff[rr][kx] + off;
              else ctl->excl_slot_error = 1;
            }
          }
        }
        if (k == 0) xs0 = slot; else if (k == 1) xs1 = slot; else if (k == 2) xs2 = slot; else xs3 = slot;
      }
    }
    // Inline bonds (force kernel epilogue).  The host enables this only where the exclusion set IS the bond set and all bonds
    // share one harmonic parameter set (chain-growth systems: every reaction bond is excluded, nothing else is): the LDS slots
    // of the excluded partners are the bonded partners -- recorded as they are (eight 32-bit words per home particle, ~0 = none),
    // no bonded table is consulted.  Located partners (<= 4 exclusions) are written here; on the generic path (5..8 exclusions,
    // or int32 rows wanted) every excluded hit is recorded as it is met in the sweep below.  More than kBondSlots exclusions:
    // the particle's bonds stay with the work list.
    unsigned int* const bsw = bslots ? reinterpret_cast<unsigned int*>(bslots) + (size_t)p * kBondSlots : nullptr;      // (indexed by the particle's place in the sorted arrays)
    int nbw = 0;
    const bool bond_rec = bsw && !fastx && e1 > e0 && e1 - e0 <= kBondSlots;      // generic path records while sweeping
    if (bsw && (fastx || e1 == e0)) {
      uint4 bw = make_uint4(~0u, ~0u, ~0u, ~0u);
      if (fastx) {
        if (xs0 >= 0) bw.x = (unsigned int)xs0; else if (e0 + 0 < e1) ctl->bond_slot_miss = 1;
        if (xs1 >= 0) bw.y = (unsigned int)xs1; else if (e0 + 1 < e1) ctl->bond_slot_miss = 1;
        if (xs2 >= 0) bw.z = (unsigned int)xs2; else if (e0 + 2 < e1) ctl->bond_slot_miss = 1;
        if (xs3 >= 0) bw.w = (unsigned int)xs3; else if (e0 + 3 < e1) ctl->bond_slot_miss = 1;
      }
      *reinterpret_cast<uint4*>(bsw) = bw;
      if (bw.w != ~0u) *reinterpret_cast<uint4*>(bsw + 4) = make_uint4(~0u, ~0u, ~0u, ~0u);   // (the second quad is only read behind a full first one)
    }
    const bool plain = (e1 == e0 || fastx) && !row32;
    uint4* regq = reinterpret_cast<uint4*>(reg16) + q;   // chunk c of this particle: regq[c * nhome]
    // accepted slots are shifted into a 128-bit register; every 8th append stores one whole 16-byte chunk (one
    // predicated 2-byte store per hit instead was measured slower: tile phase 269 -> 284 us)
    uint4 acc = make_uint4(0, 0, 0, 0);
    auto push = [&](unsigned int sl) {
      acc.x = __builtin_amdgcn_alignbit(acc.y, acc.x, 16);
      acc.y = __builtin_amdgcn_alignbit(acc.z, acc.y, 16);
      acc.z = __builtin_amdgcn_alignbit(acc.w, acc.z, 16);
      acc.w = (acc.w >> 16) | (sl << 16);
      if ((cnt16 & 7) == 7 && cnt16 < S16 && ablate != 2) regq[(size_t)(cnt16 >> 3) * nhome] = acc;
      ++cnt16;
    };
    // x-window of stencil row dzy: a candidate in row (dy, dz) is at least (ddy, ddz) away in y and z (distance of the
    // home particle from that row's slab), so only x within sqrt(rl^2 - ddy^2 - ddz^2) can be a neighbour -- 46 % of the
    // 3-cell run on average.  `weps` absorbs every rounding difference between the binning arithmetic and this one (the
    // slices only prune, the distance test decides membership).
    const float clen0 = (float)T.clen[0], clen1 = (float)T.clen[1], clen2 = (float)T.clen[2], subinv = (float)T.subinv;
    const float weps = clen0 * 2e-4f;
    const float ylo = ((float)(ly + 1) - kRefCells) * clen1, zlo = ((float)(lz + 1) - kRefCells) * clen2;   // (image coordinates are relative to T.org)
    const float dylo = fmaxf(xiy - ylo - weps, 0.f), dyhi = fmaxf(ylo + clen1 - xiy - weps, 0.f);
    const float dzlo = fmaxf(xiz - zlo - weps, 0.f), dzhi = fmaxf(zlo + clen2 - xiz - weps, 0.f);
    const float sxi = (xix + kRefCells * clen0) * subinv;
    const int fmin = lx * NSUB, fmax = (lx + 3) * NSUB - 1;
    auto window = [&](int dzy, int& a, int& b) {
      const int dz = dzy / 3, dy = dzy - 3 * dz;
      const int r = (lz + dz) * SY + (ly + dy);
      const float ddy = dy == 0 ? dylo : (dy == 2 ? dyhi : 0.f), ddz = dz == 0 ? dzlo : (dz == 2 ? dzhi : 0.f);
      const float w2 = rl2 - ddy * ddy - ddz * ddz;
      int f_lo = fmin, f_hi = fmax;
      if (!nowin) {
        const float ws = (__builtin_amdgcn_sqrtf(fmaxf(w2, 0.f)) + weps) * subinv;
        const int g_lo = (int)(sxi - ws), g_hi = (int)(sxi + ws);     // truncation == floor where it matters (clamped at >= fmin)
        f_lo = g_lo > fmin ? g_lo : fmin; f_lo = f_lo < fmax ? f_lo : fmax; f_hi = g_hi < fmax ? g_hi : fmax; f_hi = f_hi > fmin ? f_hi : fmin;
      }
      const int ro = T.rowoff[r];
      a = ro + (int)((const volatile CHEM_LDS unsigned short*)bnd)[r * NBND + f_lo];
      b = ro + (int)((const volatile CHEM_LDS unsigned short*)bnd)[r * NBND + f_hi + 1];
      b = b < total ? b : total;
      if (w2 < 0.f) b = a;                                            // the whole row is out of reach
    };
    int a_n, b_n;
    window(0, a_n, b_n);
#pragma unroll 1
    for (int dzy = 0; dzy < 9; ++dzy) {
      const int a = a_n, b = b_n;
      if (dzy < 8) window(dzy + 1, a_n, b_n);
      const int r = (lz + dzy / 3) * SY + (ly + dzy % 3);
      for (int s0 = a & ~3; s0 < b; s0 += 32) {
        const int len = (b - s0) < 32 ? (b - s0) : 32;
        const int ng = (len + 3) >> 2;
        lds_f32x4* gp = (lds_f32x4*)(l3) + (kGrpF / 4) * (s0 >> 2);
        const unsigned int w0 = tmrow[s0 >> 5], w1 = tmrow[(s0 >> 5) + 1];
        // two groups per trip, ping-pong registers: the reads of the group after next are in flight while one is tested
        f32x4 X0 = gp[0], Y0 = gp[1], Z0 = gp[2], Q0 = gp[3];
        unsigned int miss = 0;
        auto test4 = [&](const f32x4& X, const f32x4& Y, const f32x4& Z, const f32x4& Q) {
          f32x2 lo = cc - Q.xy, hi = cc - Q.zw;                       // sign bit set at the end: outside
          lo = __builtin_elementwise_fma(az, Z.xy, lo); hi = __builtin_elementwise_fma(az, Z.zw, hi);
          lo = __builtin_elementwise_fma(ay, Y.xy, lo); hi = __builtin_elementwise_fma(ay, Y.zw, hi);
          lo = __builtin_elementwise_fma(ax, X.xy, lo); hi = __builtin_elementwise_fma(ax, X.zw, hi);
          miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(lo.x), 31);
          miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(lo.y), 31);
          miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(hi.x), 31);
          miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(hi.y), 31);
        };
        const int ng2 = (ng + 1) & ~1;                                // (the odd group's bits fall behind `len` and are masked)
        for (int g = 0; g < ng2; g += 2) {
          const f32x4 X1 = gp[kGrpF / 4], Y1 = gp[kGrpF / 4 + 1], Z1 = gp[kGrpF / 4 + 2], Q1 = gp[kGrpF / 4 + 3];
          test4(X0, Y0, Z0, Q0);
          gp += 2 * (kGrpF / 4);
          X0 = gp[0]; Y0 = gp[1]; Z0 = gp[2]; Q0 = gp[3];             // (up to two groups past the run: allocated, never used)
          test4(X1, Y1, Z1, Q1);
        }
        unsigned int m = ng2 < 8 ? ~miss << (32 - 4 * ng2) : ~miss;  // candidate u -> bit 31-u
        m &= ~(len < 32 ? (0xffffffffu >> len) : 0u);                 // slots behind the run
        m &= 0xffffffffu >> (a > s0 ? a - s0 : 0);                    // slots in front of a run that starts inside a group
        const int sh = s0 & 31;
        const unsigned int tm = sh ? __builtin_amdgcn_alignbit(w0, w1, 32 - sh) : w0;   // type-pair filter of these 32 slots
        const unsigned int ks = (unsigned int)(sself - s0);
        if (ks < 32u) m &= ~(0x80000000u >> ks);
        if (ablate == 3) { cnt16 += __popc(m & tm); continue; }
        if (plain) {
          // no exclusions (or located as slots) and no int32 row: a hit needs nothing from memory
          m &= tm;
          if (fastx) {
            const unsigned int k0 = (unsigned int)(xs0 - s0), k1 = (unsigned int)(xs1 - s0), k2 = (unsigned int)(xs2 - s0), k3 = (unsigned int)(xs3 - s0);
            if (k0 < 32u) m &= ~(0x80000000u >> k0);
            if (k1 < 32u) m &= ~(0x80000000u >> k1);
            if (k2 < 32u) m &= ~(0x80000000u >> k2);
            if (k3 < 32u) m &= ~(0x80000000u >> k3);
          }
          while (m) {
            const int k = __clz((int)m);
            m &= ~(0x80000000u >> k);
            push((unsigned int)(s0 + k));
          }
        } else {
          while (m) {
            const int k = __clz((int)m);
            const unsigned int bit = 0x80000000u >> k;
            m &= ~bit;
            const int sl = s0 + k;
            // global index of the slot from the row's cell tables (this path is rare: int32 rows wanted, or > 4 exclusions)
            const int e = sl - T.rowoff[r];
            int kc = 0;
#pragma unroll
            for (int qq = 1; qq < SX; ++qq) kc += (e >= T.celloff[r][qq]) ? 1 : 0;
            const int j = T.cellg[r][kc] + (e - T.celloff[r][kc]);
            const CHEM_LDS float* cj = img + (sl >> 2) * kGrpF + (sl & 3);
            const float ddx = xix - cj[0], ddy = xiy - cj[4], ddz = xiz - cj[8];
            const bool exact = ddx * ddx + ddy * ddy + ddz * ddz <= rl2_rows;  // difference form: the mask is a superset by the delta shell
            bool ok = true;
            if (e1 > e0) {
              const int tgj = tag[j];
              for (int ee = e0; ee < e1; ++ee) if (excl_list[ee] == tgj) { ok = false; break; }
              if (!ok && bond_rec && nbw < kBondSlots) bsw[nbw++] = (unsigned int)sl;      // inline bonds: an excluded hit is a bonded partner
            }
            if (ok) {
              if (tm & bit) push((unsigned int)sl);      // (shell pairs included, as on the plain path: the force list does not depend on the path)
              if (row32 && exact) { if (cnt < S) row32[cnt] = j; ++cnt; }
            }
          }
        }
      }
    }
    if (bsw && !fastx && e1 > e0) {       // generic path: close the record (or leave it empty for a particle the work list keeps)
      if (bond_rec && nbw != e1 - e0) ctl->bond_slot_miss = 1;      // every excluded (= bonded) partner sits within the list radius
      for (int k = bond_rec ? nbw : 0; k < kBondSlots; ++k) bsw[k] = ~0u;
    }
    // pad the last chunk with the far-away dummy slot (chunks are read whole)
    const int real16 = cnt16;
    if (real16 <= S16) while (cnt16 & 7) push((unsigned int)total);
    nnh[hbase + q] = ablate >= 2 ? (real16 < 0 ? 1 : 0) : (real16 < S16 ? real16 : S16);   // (diagnostic builds of the list leave no usable chunks)
    if (real16 > S16) atomicMax(&ctl->nl_overflow, real16);
    if (row32) {
      const int c32 = cnt < S ? cnt : S;
      for (int k = c32; k < ((c32 + 3) & ~3); ++k) row32[k] = p;
      nn[p] = c32;
      if (cnt > S) atomicMax(&ctl->nl_overflow, cnt);
    }
  }
}

template <typename R, int BS>
__global__ __launch_bounds__(BS, 4) void k_nlist_tiles(int ntiles, int CAP, const Vec4<R>* __restrict__ x4, const int* __restrict__ tag,
                                                    const TileLDS<R>* __restrict__ desc, R rl2,
                                                    const int* __restrict__ excl_start, const int* __restrict__ excl_list, int has_excl,
                                                    ActMask act, int ntypes, unsigned short* __restrict__ nl16, int S16, int* __restrict__ nnh,
                                                    int* __restrict__ nlist, int S, int* __restrict__ nn, DevCtl* ctl, R rl2_rows, uint4* __restrict__ bslots,
                                                    const Box<R>* __restrict__ boxp, const int* __restrict__ rtag, const int* __restrict__ gtag, int real0, int real1) {
  // (rl2: the radius the 16-bit force list is built for, rc + list skin; rl2_rows: the int32 Verlet rows', rc + skin;
  //  boxp/rtag/gtag/real0/real1: located-partner path of the exclusions on a slab, boxp == nullptr: generic path)
  if (!ctl->need_rebuild) return;
  __shared__ TileLDS<R> T;
  __shared__ Box<R> sbox;
  if (boxp) { const int* bs_ = reinterpret_cast<const int*>(boxp); int* bd_ = reinterpret_cast<int*>(&sbox); for (int k = threadIdx.x; k < (int)(sizeof(Box<R>) / 4); k += BS) bd_[k] = bs_[k]; }
  CHEM_DYN_LDS(R);
  for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
    const int tile = xcd_remap(vb, ntiles);     // gridDim.x is a multiple of 8: vb % 8 == blockIdx.x % 8
    __syncthreads();
    tile_load_desc<R>(T, desc, tile);
    const bool f32list = sizeof(R) == 4 || !nlist;      // fp64: the exact list build only where the int32 rows are wanted
    if (f32list) list_stage_clear<BS>(chem_dyn_lds, list_lds_layout(CAP, ntypes), ntypes);
    __syncthreads();
    if (f32list) {
      const ListLDS L = list_lds_layout(CAP, ntypes);
      list_stage_f32<R, BS>(T, chem_dyn_lds, L, CAP, x4, act, ntypes);
      __syncthreads();
      dev_nlist_tile_f32<R, BS>(T, chem_dyn_lds, L, tag, (float)rl2, excl_start, excl_list, has_excl, nl16, S16, nnh, nlist, S, nn, ctl,
                                boxp ? (const Box<R>*)&sbox : (const Box<R>*)nullptr, rtag, x4, 0, (float)rl2_rows, bslots, gtag, real0, real1);      // (bslots: inline bonds, see dev_nlist_tile_f32)
    } else {
      tile_fill<R, BS, true>(T, sx, CAP, x4, 1);
      __syncthreads();
      dev_nlist_tile<R, BS>(T, sx, tag, rl2, excl_start, excl_list, has_excl, act, nl16, S16, nnh, nlist, S, nn, ctl, nullptr, nullptr, nullptr, rl2_rows);
    }
  }
}

// exclusive scan of the home-particle counts of all tiles -> geom[5] (base of the tile's region
// in the transposed 16-bit list / count array); single block
template <typename R>
__global__ __launch_bounds__(1024) void k_tile_scan(int ntiles, TileLDS<R>* __restrict__ desc, const DevCtl* ctl) {
  if (!ctl->need_rebuild) return;
  __shared__ int wsum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int w = threadIdx.x >> 6;
  for (int base = 0; base < ntiles; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < ntiles ? desc[i].geom[4] : 0;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o); if (lane_id() >= o) incl += t; }
    if (lane_id() == 63) wsum[w] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int k = 0; k < 16; ++k) { if (k < w) woff += wsum[k]; tot += wsum[k]; }
    if (i < ntiles) desc[i].geom[5] = carry_s + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
}

// ---- pair forces on tiles --------------------------------------------------------------
template <typename R, bool ENERGY, bool LJONLY>
__device__ __forceinline__ void pair_accum(const PairCore<R> pc, const PairExt<R>* __restrict__ pext, int pidx,
                                           const Vec4<R>* __restrict__ tab, R r2, R dx, R dy, R dz,
                                           R& fx, R& fy, R& fz, double& e_lj, double& e_tab, double& vir) {
  if (LJONLY && !ENERGY) {
    const R r2i = rcp_r(r2), r6i = r2i * r2i * r2i;
    R ff = r6i * (pc.lj1 * r6i - pc.lj2) * r2i;
    ff = (r2 <= pc.rc2) ? ff * pc.kind : (R)0;   // kind (1 for LJ) keeps the read a single ds_read_b128 (b96 costs 2x the LDS cycles)
    fx += ff * dx; fy += ff * dy; fz += ff * dz;
  } else {
    pair_term<R, ENERGY>(pc, pext, pidx, tab, r2, dx, dy, dz, fx, fy, fz, e_lj, e_tab, vir);
  }
}

// One neighbour = one ds_read_b128.  Where .w is not used (uniform LJ) the compiler narrows a plain float4 read to
// ds_read_b96, which takes 8 LDS-array cycles per wave instead of 4 (MI355X_MICROARCH.md, LDS table): the volatile
// 16-byte access keeps the width.
__device__ __forceinline__ float4 lds_gather4(const float4* sx, unsigned int slot) {
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  const f32x4_ v = *((const volatile __attribute__((address_space(3))) f32x4_*)sx + slot);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double4 lds_gather4(const double4* sx, unsigned int slot) { return sx[slot]; }
// fp64 force kernel, 24-byte slots: coordinates only (the type is read from the byte array where a mode needs it)
__device__ __forceinline__ double4 lds_gather3d(const double4* sx, unsigned int slot) {
  const double* s3 = reinterpret_cast<const double*>(sx) + 3 * slot;
  return make_double4(s3[0], s3[1], s3[2], 0.0);
}
template <typename R> __device__ __forceinline__ Vec4<R> lds_gather4(const Vec4<R>* sx, unsigned int slot) { return lds_gather4(sx, slot); }

struct UniLJ { float rc2, lj1, lj2, pad; double drc2, dlj1, dlj2; };   // all listed pairs share one LJ parameter set

// One workgroup per tile; TPP lanes per home particle (lane `sub` takes chunks sub, sub+TPP, ...).
// MODE 0: general (type-pair table in LDS, tables allowed)  1: LJ/off pairs only, branch-free
//      2: uniform LJ -- every listed pair has the same parameters (kernel arguments / SGPRs)
// Tiles of one launch: [base1, base1+n1) followed by [base2, ...).  The decomposed path launches the
// tiles whose stencil stays inside the own layers ("interior": they need no ghost) while the halo
// exchange is still in flight on the communication stream, and the two boundary tile layers after it.
struct TileSub { int base1, n1, base2; };
// guard == 2 (decomposed path): every workgroup of the force kernel takes the rebuild decision itself from the P
// gathered step maxima (same inputs, same arithmetic: same result), workgroup 0 publishes it -- control block, pinned
// host words the host polls -- and all leave at once when a rebuild is due: no one-block decision launch between the
// halo exchange and the forces.  The accumulated distance is double-buffered by step parity (read [par], written [par^1]).
// LDS slot of particle g (sorted index, position xg) in the staged image of the tile described by T; -1: outside its stencil.
// (cell of g by the binning arithmetic on the same bits, relative to the tile's first home cell with the periodic wrap)
// Slabs (bx.zghost): the local layer of a ghost follows from its index range (below real0: lower ghost layer, from real1 on:
// upper), that of a real particle from the binning arithmetic; hz = stencil row (z) of the home particle -- only a copy in a
// layer next to it counts (a one-rank slab holds a particle and its ghost, possibly both inside one stencil).
template <typename R>
__device__ __forceinline__ int tile_partner_slot(const TileLDS<R>& T, int g, const Vec4<R>& xg, const Box<R>& bx, DevCtl* ctl,
                                                 const int real0 = 0, const int real1 = 0x7fffffff, const int hz = 0) {
  const int nx = bx.nc[0], ny = bx.nc[1], nz = bx.nc[2], org = T.geom[6];
  int kx = pos_cell(xg.x, 0, nx, bx) - (org & 1023) + 1, ky = pos_cell(xg.y, 1, ny, bx) - ((org >> 10) & 1023) + 1, kz;
  kx += kx < 0 ? nx : 0; kx -= kx >= nx ? nx : 0;
  ky += ky < 0 ? ny : 0; ky -= ky >= ny ? ny : 0;
  if (bx.zghost) {
    const int lzc = g < real0 ? 0 : (g >= real1 ? nz - 1 : pos_cell(xg.z, 2, bx.nzg, bx) - bx.z0g + 1);
    kz = lzc - (org >> 20) + 1;
    if (kz < 0 || kz - hz < -1 || kz - hz > 1) return -1;
  } else {
    kz = pos_cell(xg.z, 2, nz, bx) - (org >> 20) + 1;
    kz += kz < 0 ? nz : 0; kz -= kz >= nz ? nz : 0;
  }
  if (kx < T.geom[0] + 2 && ky < T.geom[1] + 2 && kz < T.geom[2] + 2) {
    const int rr = kz * SY + ky, off = g - T.cellg[rr][kx];
    if (off >= 0 && off < T.celloff[rr][kx + 1] - T.celloff[rr][kx]) return T.rowoff[rr] + T.celloff[rr][kx] + off;
    ctl->excl_slot_error = 1;
  }
  return -1;
}
// what the force kernel needs to record the bonded partners' slots itself (bond_mode 2, launch of a rebuild step)
template <typename R> struct BondRec { const int *tag, *excl_start, *excl_list, *rtag, *gtag; int real0, real1; Box<R> box; };   // (gtag, real0, real1: slabs)
// The force launch behind a rebuild records, for home particle p of the tile described by *T, the LDS slots of its bonded
// (= excluded) partners: tag -> exclusion row -> partner index -> partner position -> cell -> slot through the tile tables;
// written out for the launches up to the next rebuild, first quad returned.  A real function call on purpose: inlined, its
// registers cost the hot loop of k_pair_tiles 84-116 bytes of spills per lane; it runs once per list lifetime.  The partners are
// walked one after the other (the launch behind a rebuild takes 128 us instead of 79): issuing each level of the chain for
// four partners at once needs more registers in the callee, and THAT slowed every launch (same box, melt / late stage:
// 62.7 / 85.8 us -> 66.7 / 93.8 us per force launch; tools/ab_late_lib.sh).
template <typename R>
__device__ __noinline__ uint4 bond_record(const TileLDS<R>* T, int p, const BondRec<R>* __restrict__ brec_p, const Vec4<R>* __restrict__ x4,
                                          uint4* __restrict__ bslots, DevCtl* ctl) {
  const BondRec<R>& brec = *brec_p;
  unsigned int w[kBondSlots];
#pragma unroll
  for (int k = 0; k < kBondSlots; ++k) w[k] = ~0u;
  const int tg = brec.tag[p];
  const int e0 = brec.excl_start[tg], e1 = brec.excl_start[tg + 1];
  int hz = 0;      // stencil row (z) of the home particle: from the home run that holds p
  if (brec.box.zghost) {
#pragma unroll 1
    for (int k = 0; k < NHSEG; ++k) if (p >= T->hstart[k] && p < T->hstart[k] + (T->hoff[k + 1] - T->hoff[k])) hz = k / HY + 1;
  }
  if (e1 > e0 && e1 - e0 <= kBondSlots) {
#pragma unroll
    for (int k = 0; k < kBondSlots; ++k) {
      if (e0 + k >= e1) break;
      const int tj = brec.excl_list[e0 + k];
      const int g = brec.rtag[tj];
      int sl = g >= 0 ? tile_partner_slot<R>(*T, g, x4[g], brec.box, ctl, brec.real0, brec.real1, hz) : -1;
      if (sl < 0 && brec.gtag) {      // slabs: the tag's ghost copy
        const int g2 = brec.gtag[tj];
        if (g2 >= 0 && g2 != g) sl = tile_partner_slot<R>(*T, g2, x4[g2], brec.box, ctl, brec.real0, brec.real1, hz);
      }
      if (sl < 0) ctl->bond_slot_miss = 1;      // every bonded partner sits inside the stencil (bond length << cell edge)
      else w[k] = (unsigned int)sl;
    }
  }
  const uint4 q0 = make_uint4(w[0], w[1], w[2], w[3]);
  bslots[2 * (size_t)p] = q0;
  if (w[3] != ~0u) bslots[2 * (size_t)p + 1] = make_uint4(w[4], w[5], w[6], w[7]);      // (only read behind a full first quad)
  return q0;
}

// (fold != nullptr: `gathered` holds n * kFoldSlots bit patterns of squared displacements -- every rank's fold words, see
//  k_integrate -- instead of n doubles; `fold` = this rank's words, cleared here for the next step)
struct DecideArgs { const double* gathered; int n; volatile int* host_flag; int ticket, par, criterion; double half_skin_ref; unsigned long long* fold; };

// (fp64: the 32-byte-per-slot image allows one or two workgroups per CU anyway -- 128 registers instead of 80 and spills)
// DIAG = true: diagnostic instantiation with per-block phase stamps (`dbg`) and early exits (`ablate`: 1 stop after
// staging, 2 skip staging, 3 descriptor only, 4 dispatch only); the production instantiation carries neither.
// guard != 0: speculative launch of the decomposed path -- leave at once while a rebuild is pending.
template <typename R, int TPP, bool ENERGY, int BS, int MODE, bool DIAG = false>
__global__ __launch_bounds__(BS, sizeof(R) == 8 ? 4 : (BS == 1024 ? 2048 : 1536) / 256) void k_pair_tiles(int ntiles, int CAP, const Vec4<R>* __restrict__ x4, Vec4<R>* __restrict__ f4,
                                                   const TileLDS<R>* __restrict__ desc, const unsigned short* __restrict__ nl16,
                                                   const int* __restrict__ nnh, int S16,
                                                   const PairCore<R>* __restrict__ pcore, const PairExt<R>* __restrict__ pext,
                                                   int ntypes, const Vec4<R>* __restrict__ tab, UniLJ uni, double* __restrict__ eout,
                                                   double half_skin, DevCtl* ctl, int guard, int ablate, long long* __restrict__ dbg, TileSub sub_,
                                                   DecideArgs da = DecideArgs{}, uint4* __restrict__ bslots = nullptr,
                                                   double bond_K = 0.0, double bond_r0 = 0.0, int bond_mode = 0, ActMask bact = ActMask{},
                                                   const BondRec<R>* __restrict__ brec = nullptr) {      // (device copy: by value it was spilled to every lane's stack at kernel start)
  constexpr bool LJONLY = MODE >= 1;
#ifndef CHEM_NCH
#define CHEM_NCH 3
#endif
  constexpr int NCH = TPP == 1 ? CHEM_NCH : (TPP == 2 ? 3 : 2);   // chunks (8 slots) each lane preloads before the staging barrier
  long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
  if (DIAG && dbg) st0 = wall_clock64();
  if (DIAG && ablate == 4) return;   // diagnostic: dispatch cost only
  if (guard == 2) {
    double m2;
    if (da.fold) {      // every wave for itself: lane l takes words l, l + 64, ... of all ranks, then a wave maximum
      const unsigned long long* gw = reinterpret_cast<const unsigned long long*>(da.gathered);
      unsigned long long mb = 0;
      for (int q = lane_id(); q < da.n * kFoldSlots; q += 64) { const unsigned long long v = gw[q]; mb = v > mb ? v : mb; }
      for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(mb, o); mb = v > mb ? v : mb; }
      m2 = sizeof(R) == 4 ? bits_real_f(mb) : bits_real_d(mb);
      if (blockIdx.x == 0 && threadIdx.x < kFoldSlots) da.fold[threadIdx.x] = 0ull;      // (sent already: the exchange is ahead of this launch in the stream)
    } else {
      m2 = da.gathered[0];
      for (int q = 1; q < da.n; ++q) m2 = da.gathered[q] > m2 ? da.gathered[q] : m2;
    }
    const double acc = da.criterion ? sqrt(m2) : ctl->acc_pp[da.par] + sqrt(m2);
    const int forced = ctl->force_rebuild;
    const int need = (acc > half_skin) || forced;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ctl->step_m2 = m2; ctl->acc_pp[da.par ^ 1] = need ? 0.0 : acc; ctl->acc_maxdist = need ? 0.0 : acc;
      {      // the reference rule on the workload's skin (reported rebuild count; these two words are this thread's alone)
        const double accr = da.criterion ? sqrt(m2) : ctl->acc_ref + sqrt(m2);
        const bool rn = (accr > da.half_skin_ref) || forced;
        ctl->acc_ref = rn ? 0.0 : accr;
        if (rn) ctl->ref_rebuilds++;
      }
      // (force_rebuild is NOT cleared here: workgroups that start later must read the same value -- the host clears it,
      //  stream-ordered, at the top of the slab rebuild this decision triggers)
      if (need) ctl->rebuild_count++;
      ctl->need_rebuild = need;
      da.host_flag[0] = need;
      __threadfence_system();
      da.host_flag[1] = da.ticket;
    }
    if (need) return;
  } else if (guard && ctl->need_rebuild) return;   // speculative launch (decomposed path): the host rebuilds first and launches again
  if (ctl->halt) return;
  __shared__ TileLDS<R> T;
  __shared__ PairCore<R> spc[kMaxTypes * kMaxTypes];
  CHEM_DYN_LDS(R);
  if (MODE != 2) for (int k = threadIdx.x; k < ntypes * ntypes; k += BS) spc[k] = pcore[k];
  if (guard != 2 && blockIdx.x == 0 && threadIdx.x == 0 && ctl->acc_maxdist > half_skin) ctl->skin_violation = 1;
  const R u_rc2 = sizeof(R) == 4 ? (R)uni.rc2 : (R)uni.drc2, u_lj1 = sizeof(R) == 4 ? (R)uni.lj1 : (R)uni.dlj1,
          u_lj2 = sizeof(R) == 4 ? (R)uni.lj2 : (R)uni.dlj2;
  const int vtile = xcd_remap(blockIdx.x, ntiles);   // ntiles = tiles of THIS launch (all of them, or one of the two subsets below)
  const int tile = vtile < sub_.n1 ? sub_.base1 + vtile : sub_.base2 + (vtile - sub_.n1);
  tile_load_desc<R>(T, desc, tile);
  __syncthreads();
  if (DIAG && ablate == 3) return;   // diagnostic: descriptor load only
  if (DIAG && dbg) st1 = wall_clock64();
  const int nhome = T.geom[4], hbase = T.geom[5];
  const int slice = threadIdx.x / TPP, sub = threadIdx.x % TPP;
  constexpr int NSL = BS / TPP;
  const uint4* reg = reinterpret_cast<const uint4*>(nl16 + (size_t)hbase * S16);
  // first pass's list/count loads are issued before the staging barrier so that their latency
  // overlaps the stencil loads
  int p = -1, cnt = 0, hslot = 0, qq = 0;
  uint4 pkv[NCH];
  uint4 bwv = make_uint4(~0u, ~0u, ~0u, ~0u);
  // (uniform.  Single domain: set by the rebuild of THIS step, cleared by the next idle decision; slabs: the host rebuilds, it
  //  asks for the record with bond_mode 3 = mode 2 + record now)
  const bool bond_rec = bslots && brec && ((bond_mode == 2 && ctl->need_rebuild != 0) || bond_mode == 3);
#ifndef CHEM_NT_PRE
#define CHEM_NT_PRE 1
#endif
  auto locate = [&](int q) {
    int sgi = 0;
#pragma unroll
    for (int k = 1; k < NHSEG; ++k) sgi += (q >= T.hoff[k]) ? 1 : 0;
    const int inrun = q - T.hoff[sgi];
    p = T.hstart[sgi] + inrun; qq = q;
    const int hr = (sgi / HY + 1) * SY + (sgi % HY + 1);
    hslot = T.rowoff[hr] + T.celloff[hr][1] + inrun;      // the home particle's own slot in the staged tile
    cnt = nnh[hbase + q];
    if (bslots && !bond_rec) bwv = bslots[2 * (size_t)p];   // (inline bonds: first quad, issued with the list chunks, consumed after the pair loop)
#pragma unroll
    for (int c = 0; c < NCH; ++c)                           // the tile's region is always allocated: safe before cnt is known
      pkv[c] = (sub + c * TPP) * 8 < S16 ? (CHEM_NT_PRE ? nt_load_u4(&reg[(size_t)(sub + c * TPP) * nhome + q]) : reg[(size_t)(sub + c * TPP) * nhome + q]) : make_uint4(0, 0, 0, 0);
  };
  if (slice < nhome) locate(slice);
  constexpr bool D3 = sizeof(R) == 8;     // fp64: 24-byte slots + type bytes (see tile_fill)
  if (!DIAG || ablate != 2) tile_fill<R, BS, false, D3>(T, sx, CAP, x4, 0);
  __syncthreads();
  if (DIAG && ablate == 1) return;   // diagnostic: staging only
  if (DIAG && dbg) st2 = wall_clock64();
  double e_lj = 0, e_tab = 0, vir = 0;
  for (int q0 = 0; q0 < nhome; q0 += NSL) {
    const int q = q0 + slice;
    R fx = 0, fy = 0, fz = 0;
    if (q0 > 0) { p = -1; if (q < nhome) locate(q); }
    if (p >= 0) {
      Vec4<R> xi;
      if constexpr (D3) { xi = lds_gather3d(sx, hslot); xi.w = (R)d3_types<R>(sx, CAP)[hslot]; } else xi = sx[hslot];
      const int pbase = (int)xi.w * ntypes;
      // Uniform-LJ fp32 path: (x, y) of one neighbour live in an aligned register pair (the 16-byte LDS read returns
      // x y z w in four consecutive VGPRs), so the differences, their squares and the force accumulation of the x and y
      // components are PACKED fp32 instructions on that pair -- no register shuffling as when the compiler pairs up two
      // neighbours (its SLP form of this loop spent 27.6 VALU instructions per neighbour, profiles/round3_pmc_sq_counters.txt)
      typedef float f32x2_ __attribute__((ext_vector_type(2)));
      typedef float f32x4_ __attribute__((ext_vector_type(4)));
      f32x2_ fxy = {0.f, 0.f};
      [[maybe_unused]] const f32x2_ xixy = {(float)xi.x, (float)xi.y};
      auto do_chunk = [&](const uint4 pk) {
        const unsigned int wds[4] = {pk.x, pk.y, pk.z, pk.w};
#ifndef CHEM_PK
#define CHEM_PK 1
#endif
        if constexpr (CHEM_PK && MODE == 2 && !ENERGY && sizeof(R) == 4) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x4_ pj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // 4 LDS gathers in flight
              const unsigned int sl = (wds[2 * h + (u >> 1)] >> ((u & 1) * 16)) & 0xffff;
              pj[u] = *((const volatile __attribute__((address_space(3))) f32x4_*)sx + sl);
            }
#pragma unroll
            for (int u = 0; u < 4; u += 2) {      // two neighbours: distances per neighbour on its (x, y) pair, the LJ chain packed over the two
              const f32x2_ dxy0 = xixy - pj[u].xy, dxy1 = xixy - pj[u + 1].xy;
              const float dz0 = (float)xi.z - pj[u].z, dz1 = (float)xi.z - pj[u + 1].z;
              const f32x2_ sq0 = dxy0 * dxy0, sq1 = dxy1 * dxy1;
              const f32x2_ r2 = {__builtin_fmaf(dz0, dz0, sq0.x + sq0.y), __builtin_fmaf(dz1, dz1, sq1.x + sq1.y)};
              const f32x2_ r2i = {__builtin_amdgcn_rcpf(r2.x), __builtin_amdgcn_rcpf(r2.y)};
              const f32x2_ r6i = r2i * r2i * r2i;
              const f32x2_ lj1v = {(float)u_lj1, (float)u_lj1}, lj2v = {-(float)u_lj2, -(float)u_lj2};
              f32x2_ ff = r6i * __builtin_elementwise_fma(lj1v, r6i, lj2v) * r2i;
              ff.x = (r2.x <= (float)u_rc2) ? ff.x : 0.f;
              ff.y = (r2.y <= (float)u_rc2) ? ff.y : 0.f;
              const f32x2_ f0 = {ff.x, ff.x}, f1 = {ff.y, ff.y};
              fxy = __builtin_elementwise_fma(f0, dxy0, fxy);
              fxy = __builtin_elementwise_fma(f1, dxy1, fxy);
              fz = (R)__builtin_fmaf(ff.y, dz1, __builtin_fmaf(ff.x, dz0, (float)fz));
            }
          }
          return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          Vec4<R> xs[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {   // 4 LDS gathers in flight
            const unsigned int sl = (wds[2 * h + (u >> 1)] >> ((u & 1) * 16)) & 0xffff;
            if constexpr (D3) { xs[u] = lds_gather3d(sx, sl); if (MODE != 2 || ENERGY) xs[u].w = (R)d3_types<R>(sx, CAP)[sl]; }
            else xs[u] = lds_gather4<R>(sx, sl);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const Vec4<R> xj = xs[u];
            const R dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
            const R r2 = dx * dx + dy * dy + dz * dz;
            if (MODE == 2 && !ENERGY) {
              const R r2i = rcp_r(r2), r6i = r2i * r2i * r2i;
              R ff = r6i * (u_lj1 * r6i - u_lj2) * r2i;
              ff = (r2 <= u_rc2) ? ff : (R)0;
              fx += ff * dx; fy += ff * dy; fz += ff * dz;
            } else {
              const int pidx = pbase + (int)xj.w;
              pair_accum<R, ENERGY, LJONLY>(spc[pidx], pext, pidx, tab, r2, dx, dy, dz, fx, fy, fz, e_lj, e_tab, vir);
            }
          }
        }
      };
#pragma unroll
      for (int c = 0; c < NCH; ++c) if ((sub + c * TPP) * 8 < cnt) do_chunk(pkv[c]);
      // rows longer than the preloaded chunks (the rule with a list skin: ~52 entries = 6.5 chunks): the load of the chunk
      // after next is issued before a chunk is worked on, so that a lane waits for one L2 round trip per row, not per chunk
      {
        int c = sub + NCH * TPP;
        bool have = c * 8 < cnt;
        uint4 cur = make_uint4(0, 0, 0, 0);
#ifndef CHEM_NT_TAIL
#define CHEM_NT_TAIL 0
#endif
        if (have) cur = CHEM_NT_TAIL ? nt_load_u4(&reg[(size_t)c * nhome + qq]) : reg[(size_t)c * nhome + qq];
        while (have) {
          const int cn = c + TPP;
          const bool hn = cn * 8 < cnt;
          uint4 nx = make_uint4(0, 0, 0, 0);
          if (hn) nx = CHEM_NT_TAIL ? nt_load_u4(&reg[(size_t)cn * nhome + qq]) : reg[(size_t)cn * nhome + qq];
          do_chunk(cur);
          cur = nx; have = hn; c = cn;
        }
      }
      fx += (R)fxy.x; fy += (R)fxy.y;      // (packed x/y accumulators of the uniform-LJ fp32 path; zero otherwise)
      if (bslots && sub == 0) {
        // Inline harmonic bonds (FixedPairListHarmonic, gromacs_topology.py:949-961; reaction bonds reaction_setup.py:449-467):
        // the LDS slots of the bonded (= excluded) partners are recorded at every rebuild, the geometry comes from the staged
        // image (tile-local coordinates: 2.4e-7 in the fp32 build, exact in fp64) -- no bonded launch, no second pass over f4.
        // bond_mode 1: the list build removed the excluded pairs from the force list (decomposed path) -- add the bonds.
        // bond_mode 2: the list build IGNORED the exclusions (single domain: locating the partners inside the list phase cost
        //   196 us of a 709-us late-stage rebuild, profiles/round3_rebuild_ablation.txt; a streaming pass behind it records the
        //   slots instead) -- the partners' pair term is in the sums above: take it out again, then add the bonds.  An excluded
        //   pair within the cutoff now was within the list radius at the build, i.e. it IS in the list if its type pair is active.
        if (bond_rec) bwv = bond_record<R>(&T, p, brec, x4, bslots, ctl);      // (the launch behind a rebuild, bond_mode 2)
        const R m2K = (R)(-2.0 * bond_K), r0b = (R)bond_r0;              // (one parameter set: kernel arguments, no table)
        auto bond_quad = [&](const uint4 bq) {
          const unsigned int bws[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (bws[k] == ~0u) continue;
            const unsigned int sl = bws[k];
            Vec4<R> xj;
            if constexpr (D3) { xj = lds_gather3d(sx, sl); xj.w = (R)d3_types<R>(sx, CAP)[sl]; } else xj = sx[sl];
            const R dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
            const R r2 = dx * dx + dy * dy + dz * dz;
            if (bond_mode >= 2) {
              if (MODE == 2 && !ENERGY) {
                if (r2 <= u_rc2 && ((bact.row[(int)xi.w & 15] >> ((int)xj.w & 15)) & 1u)) {
                  const R r2i = rcp_r(r2), r6i = r2i * r2i * r2i;
                  const R ffp = r6i * (u_lj1 * r6i - u_lj2) * r2i;
                  fx -= ffp * dx; fy -= ffp * dy; fz -= ffp * dz;
                }
              } else {
                R tx = 0, ty = 0, tz = 0; double te1 = 0, te2 = 0, tv = 0;
                const int pidx = pbase + (int)xj.w;
                pair_accum<R, ENERGY, LJONLY>(spc[pidx], pext, pidx, tab, r2, dx, dy, dz, tx, ty, tz, te1, te2, tv);
                fx -= tx; fy -= ty; fz -= tz; e_lj -= te1; e_tab -= te2; vir -= tv;
              }
            }
            if (!ENERGY) {
              const R r = sqrt_r(r2);
              const R ffb = m2K * (r - r0b) / r;                           // U = K (r - r0)^2
              fx += ffb * dx; fy += ffb * dy; fz += ffb * dz;
            }
          }
        };
        bond_quad(bwv);
        if (bwv.w != ~0u) bond_quad(bslots[2 * (size_t)p + 1]);   // five to eight bonds: the second quad (rare)
      }
    }
    if (TPP > 1) {
#pragma unroll
      for (int o = TPP / 2; o > 0; o >>= 1) { fx += __shfl_xor(fx, o); fy += __shfl_xor(fy, o); fz += __shfl_xor(fz, o); }
    }
    if (p >= 0 && sub == 0) f4[p] = mk4<R>(fx, fy, fz, (R)0);
  }
  if (DIAG && dbg && threadIdx.x == 0) {   // diagnostic instantiation only: per-block phase stamps (100 MHz wall clock)
    st3 = wall_clock64();
    long long* o = dbg + 6 * (size_t)blockIdx.x;
    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = 0; o[5] = tile;
  }
  if (ENERGY) {
    __shared__ double red[3][BS / 64];
    for (int o = 32; o > 0; o >>= 1) { e_lj += __shfl_xor(e_lj, o); e_tab += __shfl_xor(e_tab, o); vir += __shfl_xor(vir, o); }
    if (lane_id() == 0) { red[0][threadIdx.x >> 6] = e_lj; red[1][threadIdx.x >> 6] = e_tab; red[2][threadIdx.x >> 6] = vir; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double a = 0, b = 0, c = 0;
      for (int k = 0; k < BS / 64; ++k) { a += red[0][k]; b += red[1][k]; c += red[2][k]; }
      eout[3 * blockIdx.x + 0] = 0.5 * a; eout[3 * blockIdx.x + 1] = 0.5 * b; eout[3 * blockIdx.x + 2] = 0.5 * c;
    }
  }
}

// =======================================================================================
// K4-K6  bonded forces (FixedPairList/TripleList/QuadrupleList interactions,
//     gromacs_topology.py:949-961,1086-1096,1206-1224).  Per-particle CSR (by tag): every
//     member of a tuple evaluates the term and keeps its own force -> no atomics,
//     deterministic.  Geometry in fp64 in both precision modes.
// =======================================================================================

__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot3(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross3(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct BoxD { double L[3], invL[3]; double qs[3], qh[3]; };   // qs, qh: fixed-point decoding of the fp32 build's positions (Box<R>)
__device__ __forceinline__ D3 minimgD(const BoxD& b, D3 d) {
  return {d.x - b.L[0] * rint(d.x * b.invL[0]), d.y - b.L[1] * rint(d.y * b.invL[1]), d.z - b.L[2] * rint(d.z * b.invL[2])};
}
// |d|^2 exactly as the CPU oracle rounds it: three products, two sums, NO fused multiply-add.  (__dmul_rn / __dadd_rn are plain
// operators in the HIP headers and -ffp-contract=fast fused them; invisible as long as the test positions were fp32 values,
// whose products are exact in fp64.)
__device__ __forceinline__ double dist2_unfused(const D3& d) {
#pragma clang fp contract(off)
  const double xx = d.x * d.x, yy = d.y * d.y, zz = d.z * d.z;
  return (xx + yy) + zz;
}
template <typename R> __device__ __forceinline__ D3 posD(const Vec4<R>& v, const BoxD& b) { return pos_real(v, b.qs, b.qh); }   // exact in both builds

// bond tables (chem_table_create): rows = (e, f) pairs of all tables back to back, info[h] = (first row, rows, r0, 1/dr)
struct BTab { const double2* rows; const double4* info; };

// one bonded term seen from member `me` of the tuple (j0..j3 = particle indices of the tuple in order)
// BONDS_ONLY: the caller guarantees arity 2 and an analytic kind (harmonic, FENE, FENE+LJ, LJ pair) -- the angle, dihedral
// and table code is not compiled in, which is what brings the per-step kernel from 209 to a few dozen registers
template <typename R, bool ENERGY, bool BONDS_ONLY = false>
__device__ __forceinline__ void bonded_term(const BondedParam& bp, const int me, const int j0, const int j1, const int j2, const int j3,
                                            const Vec4<R>* __restrict__ x4, const BoxD& box, D3& f, double* __restrict__ elist, DevCtl* ctl, const BTab& bt) {
    const double* p = bp.p;
    double u = 0;
    if (BONDS_ONLY || bp.arity == 2) {
      // tuple (t0,t1); r_ij = x_t0 - x_t1
      if ((j0 | j1) < 0) { ctl->bonded_missing = 1; return; }
      const D3 x0 = posD<R>(x4[j0], box), x1 = posD<R>(x4[j1], box);
      const D3 d = minimgD(box, x0 - x1);
      const double r = sqrt(dot3(d, d));
      double ff = 0;
      if (bp.kind == CHEM_POT_HARMONIC) { const double dr = r - p[1]; u = p[0] * dr * dr; ff = -2.0 * p[0] * dr / r; }
      else if (bp.kind == CHEM_POT_FENE) { const double dr = r - p[1], q = dr / p[2], den = 1.0 - q * q; u = -0.5 * p[0] * p[2] * p[2] * log(den); ff = -p[0] * dr / den / r; }
      else if (bp.kind == CHEM_POT_FENE_LJ) {   // FENELennardJones(K, r0, rMax, sigma, epsilon)
        const double dr = r - p[1], q = dr / p[2], den = 1.0 - q * q;
        const double s2 = p[3] * p[3] / (r * r), s6 = s2 * s2 * s2;
        u = -0.5 * p[0] * p[2] * p[2] * log(den) + 4.0 * p[4] * (s6 * s6 - s6);
        ff = -p[0] * dr / den / r + 24.0 * p[4] * (2.0 * s6 * s6 - s6) / (r * r);
      }
      else if (bp.kind == CHEM_POT_LJ_BOND) {   // FixedPairListLennardJones(epsilon, sigma, cutoff): 1-4 pairs
        if (r <= p[2]) {
          const double s2 = p[1] * p[1] / (r * r), s6 = s2 * s2 * s2;
          // energy shifted to zero at the cutoff: espressopp's LennardJones defaults to shift = 'auto' (SURVEY App. C), also on the 1-4 lists
          const double c2 = p[1] * p[1] / (p[2] * p[2]), c6 = p[2] < 1e29 ? c2 * c2 * c2 : 0.0;
          u = 4.0 * p[0] * ((s6 * s6 - s6) - (c6 * c6 - c6));
          ff = 24.0 * p[0] * (2.0 * s6 * s6 - s6) / (r * r);
        }
      }
      else if (!BONDS_ONLY && bp.kind == CHEM_POT_TABULATED) {   // Tabulated(itype=1): linear interpolation of e(r), f(r); end rows beyond the grid
        const double4 ti = bt.info[(int)p[0]];
        const double2* row = bt.rows + (size_t)ti.x;
        const int nrow = (int)ti.y;
        const double t = (r - ti.z) * ti.w;
        double fv;
        if (t <= 0) { u = row[0].x; fv = row[0].y; }
        else if (t >= (double)(nrow - 1)) { u = row[nrow - 1].x; fv = row[nrow - 1].y; }
        else { const int k = (int)t; const double w = t - (double)k; const double2 a = row[k], b = row[k + 1]; u = a.x + w * (b.x - a.x); fv = a.y + w * (b.y - a.y); }
        ff = fv / r;
      }
      const double sgn = me == 0 ? 1.0 : -1.0;
      f = f + (sgn * ff) * d;
    } else if (BONDS_ONLY) {
    } else if (bp.arity == 3) {
      if ((j0 | j1 | j2) < 0) { ctl->bonded_missing = 1; return; }
      const D3 x0 = posD<R>(x4[j0], box), x1 = posD<R>(x4[j1], box), x2 = posD<R>(x4[j2], box);
      const D3 r1 = minimgD(box, x0 - x1), r2 = minimgD(box, x2 - x1);
      const double n1 = sqrt(dot3(r1, r1)), n2 = sqrt(dot3(r2, r2));
      double c = dot3(r1, r2) / (n1 * n2);
      c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
      const double th = acos(c);
      double s = sqrt(1.0 - c * c);
      if (s < 1e-9) s = 1e-9;
      double dU = 0;
      if (bp.kind == CHEM_POT_ANG_HARMONIC) { const double d = th - p[1]; u = p[0] * d * d; dU = 2.0 * p[0] * d; }
      else if (bp.kind == CHEM_POT_ANG_COSINE) { u = p[0] * (1.0 + cos(th - p[1])); dU = -p[0] * sin(th - p[1]); }
      else if (bp.kind == CHEM_POT_ANG_TABULATED) {   // TabulatedAngular(itype=1): U(theta), -dU/dtheta on a uniform grid in radians
        const double4 ti = bt.info[(int)p[0]];
        const double2* row = bt.rows + (size_t)ti.x;
        const int nrow = (int)ti.y;
        const double t = (th - ti.z) * ti.w;
        double fv;
        if (t <= 0) { u = row[0].x; fv = row[0].y; }
        else if (t >= (double)(nrow - 1)) { u = row[nrow - 1].x; fv = row[nrow - 1].y; }
        else { const int k = (int)t; const double w = t - (double)k; const double2 a = row[k], b = row[k + 1]; u = a.x + w * (b.x - a.x); fv = a.y + w * (b.y - a.y); }
        dU = -fv;
      }
      const double a = dU / s;
      const D3 fi = a * ((1.0 / (n1 * n2)) * r2 - (c / (n1 * n1)) * r1);
      const D3 fk = a * ((1.0 / (n1 * n2)) * r1 - (c / (n2 * n2)) * r2);
      if (me == 0) f = f + fi; else if (me == 2) f = f + fk; else f = f - (fi + fk);
    } else {
      if ((j0 | j1 | j2 | j3) < 0) { ctl->bonded_missing = 1; return; }
      const D3 x0 = posD<R>(x4[j0], box), x1 = posD<R>(x4[j1], box), x2 = posD<R>(x4[j2], box), x3 = posD<R>(x4[j3], box);
      const D3 b1 = minimgD(box, x1 - x0), b2 = minimgD(box, x2 - x1), b3 = minimgD(box, x3 - x2);
      const D3 m = cross3(b1, b2), nn = cross3(b2, b3);
      const double m2 = dot3(m, m), n2 = dot3(nn, nn), lb2 = dot3(b2, b2), lb = sqrt(lb2);
      if (m2 < 1e-30 || n2 < 1e-30) return;
      const double phi = atan2(lb * dot3(b1, nn), dot3(m, nn));
      double dU = 0;
      if (bp.kind == CHEM_POT_DIH_NCOS) { u = p[0] * (1.0 + cos(p[2] * phi - p[1])); dU = -p[0] * p[2] * sin(p[2] * phi - p[1]); }
      else if (bp.kind == CHEM_POT_DIH_RB) {
        const double psi = phi - M_PI, cp = cos(psi), sp = sin(psi);
        double pw = 1.0, dsum = 0;
        for (int k = 0; k < 6; ++k) { u += p[k] * pw; if (k < 5) dsum += (k + 1) * p[k + 1] * pw; pw *= cp; }
        dU = -sp * dsum;
      }
      else if (bp.kind == CHEM_POT_DIH_HARMONIC) {   // DihedralHarmonic(K, phi0): U = K/2 (phi - phi0)^2, difference wrapped
        double d = phi - p[1];
        d -= 2.0 * M_PI * rint(d / (2.0 * M_PI));
        u = 0.5 * p[0] * d * d; dU = p[0] * d;
      }
      else if (bp.kind == CHEM_POT_DIH_TABULATED) {   // TabulatedDihedral(itype=1): U(phi), -dU/dphi on a uniform grid over [-pi, pi]
        const double4 ti = bt.info[(int)p[0]];
        const double2* row = bt.rows + (size_t)ti.x;
        const int nrow = (int)ti.y;
        const double t = (phi - ti.z) * ti.w;
        double fv;
        if (t <= 0) { u = row[0].x; fv = row[0].y; }
        else if (t >= (double)(nrow - 1)) { u = row[nrow - 1].x; fv = row[nrow - 1].y; }
        else { const int k = (int)t; const double w = t - (double)k; const double2 a = row[k], b = row[k + 1]; u = a.x + w * (b.x - a.x); fv = a.y + w * (b.y - a.y); }
        dU = -fv;
      }
      const D3 g1 = (-lb / m2) * m, g4 = (lb / n2) * nn;
      const double s12 = dot3(b1, b2) / lb2, s32 = dot3(b3, b2) / lb2;
      D3 g;
      if (me == 0) g = g1;
      else if (me == 3) g = g4;
      else if (me == 1) g = (-1.0 - s12) * g1 + s32 * g4;
      else g = (-1.0 - s32) * g4 + s12 * g1;
      f = f - dU * g;
    }
    if (ENERGY && me == 0) atomicAdd(&elist[bp.list], u);
}

template <typename R, bool ENERGY>
__global__ __launch_bounds__(256) void k_bonded(int i0, int n, const Vec4<R>* __restrict__ x4, Vec4<R>* __restrict__ f4,
                                                const int* __restrict__ tag, const int* __restrict__ rtag,
                                                const int* __restrict__ bstart, const BondedEntry* __restrict__ bent,
                                                const BondedParam* __restrict__ bpar, BoxD box, double* __restrict__ elist, DevCtl* ctl, BTab bt) {
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= i0 + n) return;
  const int tg = tag[i];
  const int e0 = bstart[tg], e1 = bstart[tg + 1];
  if (e0 == e1) return;
  D3 f = {0, 0, 0};
  for (int e = e0; e < e1; ++e) {
    const BondedEntry be = bent[e];
    const int slot = be.meta & 0x0fffffff, me = (be.meta >> 28) & 3;
    const BondedParam& bp = bpar[slot];
    const int ar = bp.arity;
    const int j0 = rtag[be.t0], j1 = rtag[be.t1], j2 = ar > 2 ? rtag[be.t2] : 0;
    int j3 = 0;
    if (ar == 4) { j3 = rtag[bent[e + 1].t0]; ++e; }   // quadruples occupy two consecutive entries: (t0,t1,t2,meta),(t3,-,-,-)
    bonded_term<R, ENERGY>(bp, me, j0, j1, j2, j3, x4, box, f, elist, ctl, bt);
  }
  Vec4<R> fo = f4[i];
  fo.x += (R)f.x; fo.y += (R)f.y; fo.z += (R)f.z;
  f4[i] = fo;
}

// exclusive scan over the block (BS threads, BS/64 <= 16 waves); returns the exclusive prefix of v,
// *total = block sum.  Two barriers; safe to call back to back.
template <int BS>
__device__ __forceinline__ int block_scan_excl(int v, int* total) {
  __shared__ int ws[16];
  const int lane = lane_id(), w = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  __syncthreads();
  if (lane == 63) ws[w] = incl;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < BS / 64; ++k) { if (k < w) off += ws[k]; tot += ws[k]; }
  *total = tot;
  return off + incl - v;
}

// ---- bonded work list: rebuilt together with the Verlet list (the particle order only changes
// there).  bwork = compact list of the particles that own bonded entries (i, e0, e1), bj = the
// entries' partner tags already resolved to particle indices.  The per-step kernel runs over the
// owners only, with full waves (the fp64 term code is long: a wave with one bonded lane costs as
// much as a full one), and has two dependent loads in front of the arithmetic instead of five.
// one chunk of BS particles starting at ib (block-uniform; ends with a workgroup barrier)
template <int BS>
// over_excl != nullptr (inline bonds): only the owners with MORE than kBondSlots exclusions enter the work list -- the force kernel
// evaluates the bonds of everybody else from its staged image (excl_start rows: the same criterion the list build applies)
__device__ __forceinline__ void dev_bonded_prep_chunk(int ib, int iend, const int* tag, const int* rtag, const int* bstart, const BondedEntry* bent,
                                                      int4* bwork, int4* bj, DevCtl* ctl, const int* over_excl = nullptr) {
  __shared__ int s_base, s_ebase;
  {
    const int i = ib + threadIdx.x;
    int e0 = 0, e1 = 0;
    if (i < iend) {
      const int tg = tag[i]; e0 = bstart[tg]; e1 = bstart[tg + 1];
      if (over_excl && over_excl[tg + 1] - over_excl[tg] <= kBondSlots) e1 = e0;
    }
    const int has = e1 > e0 ? 1 : 0;
    int tot, etot;
    const int rank = block_scan_excl<BS>(has, &tot);       // one global atomic per block and pass, not per wave:
    const int erank = block_scan_excl<BS>(e1 - e0, &etot); // (same-address atomics cost ~12 ns each)
    if (threadIdx.x == 0) {
      const unsigned long long old = tot ? atomicAdd(&ctl->bw64, ((unsigned long long)etot << 32) | (unsigned long long)tot) : 0ull;
      s_base = (int)(old & 0xffffffffull); s_ebase = (int)(old >> 32);
    }
    __syncthreads();
    // the entries of an owner follow each other IN WORK-LIST ORDER (the per-tag CSR is in tag order, i.e. scattered for
    // neighbouring particles): the per-step kernel reads one coalesced 16-byte record per entry -- partner indices and
    // the parameter slot / own position -- and nothing else
    const int eo = s_ebase + erank;
    if (has) bwork[s_base + rank] = make_int4(i, eo, e1 - e0, 0);
    for (int e = e0; e < e1; ++e) { const BondedEntry be = bent[e]; bj[eo + (e - e0)] = make_int4(rtag[be.t0], rtag[be.t1], rtag[be.t2], be.meta); }
    __syncthreads();
  }
}
template <int BS>
__device__ __forceinline__ void dev_bonded_prep(int i0, int n, const int* tag, const int* rtag, const int* bstart, const BondedEntry* bent,
                                                int4* bwork, int4* bj, DevCtl* ctl, const int* over_excl = nullptr) {
  for (int ib = i0 + blockIdx.x * BS; ib < i0 + n; ib += gridDim.x * BS)    // block-uniform bound (block scan inside)
    dev_bonded_prep_chunk<BS>(ib, i0 + n, tag, rtag, bstart, bent, bwork, bj, ctl, over_excl);
}

__global__ __launch_bounds__(256) void k_bonded_prep(int i0, int n, const int* __restrict__ tag, const int* __restrict__ rtag, const int* __restrict__ bstart,
                                                     const BondedEntry* __restrict__ bent, int4* __restrict__ bwork, int4* __restrict__ bj, DevCtl* ctl,
                                                     const int* __restrict__ over_excl) {
  dev_bonded_prep<256>(i0, n, tag, rtag, bstart, bent, bwork, bj, ctl, over_excl);
}

template <typename R, bool BONDS_ONLY = false>
__global__ __launch_bounds__(256) void k_bonded_work(const Vec4<R>* __restrict__ x4, Vec4<R>* __restrict__ f4, const int4* __restrict__ bwork, const int4* __restrict__ bj,
                                                     const BondedEntry* __restrict__ bent, const BondedParam* __restrict__ bpar, BoxD box, DevCtl* ctl, int guard, BTab bt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (int)(ctl->bw64 & 0xffffffffull) || (guard && ctl->need_rebuild) || ctl->halt) return;
  const int4 wk = bwork[k];
  D3 f = {0, 0, 0};
  for (int e = wk.y; e < wk.y + wk.z; ++e) {
    const int4 jj = bj[e];
    const int meta = jj.w;
    const int slot = meta & 0x0fffffff, me = (meta >> 28) & 3;
    const BondedParam& bp = bpar[slot];
    int j3 = 0;
    if (!BONDS_ONLY && bp.arity == 4) { j3 = bj[e + 1].x; ++e; }
    bonded_term<R, false, BONDS_ONLY>(bp, me, jj.x, jj.y, (!BONDS_ONLY && bp.arity > 2) ? jj.z : 0, j3, x4, box, f, nullptr, ctl, bt);
  }
  Vec4<R> fo = f4[wk.x];
  fo.x += (R)f.x; fo.y += (R)f.y; fo.z += (R)f.z;
  f4[wk.x] = fo;
}

// =======================================================================================
// Fused rebuild: decision + binning + cell scan + placement + canonical sort + copy-back + tile
// descriptors + list build in ONE persistent launch with grid barriers.
// Why: on this part a dependent kernel-to-kernel hand-over costs ~4.6 us, and the separate chain is
// 9 launches that all early-exit on ~7 of 8 steps -- ~40 us of a 160 us step spent launching
// nothing.  Here the idle path is one launch whose workgroups each fold the step's displacement
// maxima themselves (identical inputs, identical arithmetic => identical decision, no barrier
// needed to agree) and exit.
// The grid is sized by the host to be fully co-resident (occupancy query x CU count, exclusive
// device); a barrier that is not released within ~1 s sets ctl->barrier_timeout and every
// workgroup leaves (the host reports a fatal error) instead of spinning forever.
// =======================================================================================
struct GridBar {
  unsigned int grp[8][32];     // arrivals per XCD group (128-byte spacing)
  unsigned int top[32];        // groups arrived
  unsigned int gen[32];        // generation (release flag)
  unsigned int tq[8][32];      // tile queue heads, one per XCD
  unsigned int prepq[32];      // head of the bonded-work-list chunk queue
  long long stamp[16];         // wall_clock64 of workgroup 0 at the phase boundaries of the last rebuild (diagnostics)
};

__device__ __forceinline__ bool grid_barrier(GridBar* gb, DevCtl* ctl) {
  __shared__ int ok_s;
  // Every wave waits until its own stores are acknowledged by the L2; ONE thread of the workgroup
  // then executes the device-scope release (L2 write-back, needed across XCDs) and, after the wait,
  // the acquire (L1/L2 invalidate).  Both act on the caches, not on the issuing wave, so one per
  // workgroup is enough -- fences from all 6144 waves made a barrier cost ~160 us.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    int ok = 1;
    const unsigned int g = blockIdx.x & 7u;
    const unsigned int ngrp = gridDim.x < 8u ? gridDim.x : 8u;
    const unsigned int gsize = (gridDim.x - g + 7u) >> 3;
    const unsigned int gen = __hip_atomic_load(&gb->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (atomicAdd(&gb->grp[g][0], 1u) == gsize - 1u) {
      __hip_atomic_store(&gb->grp[g][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (atomicAdd(&gb->top[0], 1u) == ngrp - 1u) {
        __hip_atomic_store(&gb->top[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&gb->gen[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    unsigned int spins = 0;
    while (__hip_atomic_load(&gb->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 20)) { ctl->barrier_timeout = 1; ok = 0; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop stale lines before anyone reads what other XCDs wrote
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

template <typename R> struct FusedArgs {
  int n, ncell, ntiles, CAP, S, has_excl, criterion, par, seg_shift, tseg_shift, nblk, want32, ntypes, ablate;
  double half_skin; R rl2;
  long long istep;                     // step this launch belongs to (recorded in ctl->halt_step when the launch has to stop the run)
  double half_skin_ref; R rl2_rows;    // the workload's skin: reference rebuild count, radius of the int32 Verlet rows (rl2 / half_skin: list skin)
  Vec4<R> *x4, *v4, *x4o, *v4o, *x0;
  int *tag, *tago, *rtag; int4 *img4, *img4o;
  int *cell_cnt, *cell_of, *slot_of, *cell_start, *cell_loc, *btot, *perm, *tn, *tloc, *tbtot, *cell_sub, *cell_n, *bucket; int bcap;
  TileLDS<R>* desc; const int *excl_start, *excl_list;
  int* tile_ord;         // the same order as a list (inverse of tile_pos): the tile queues of the list phase hand out the large tiles first.
                         // Both arrays are double-buffered ([parity of the rebuild count][ntiles]): workgroups 0..7 rewrite the other
                         // half from THIS rebuild's home counts (gelation makes the tiles unequal), the next rebuild uses it
  int* tile_pos;         // where the descriptor of tile t goes: inside every XCD's range the full-size tiles first, the short edge
                         // tiles last, so that the one-shot force launch (workgroup v reads desc[xcd_remap(v)]) ends on short blocks
  unsigned short* nl16; int *nnh, *nlist, *nn;
  unsigned long long* blockmax; DevCtl* ctl; GridBar* gb;
  const int* bstart; const BondedEntry* bent; int4 *bwork, *bj; int nbent;
  uint4* bslots;         // inline bonds (non-null: the bonded partners' LDS slots are recorded, eight words per particle)
  int bond_pass;         // 1: the list build ignores the exclusions (= bonds); the force launch of the rebuild step records the partner
                         //    slots itself and every force launch takes the partners' pair term out again (k_pair_tiles bond_mode 2);
                         //    0: the list build removes the excluded pairs and records their slots
  Box<R> box; ActMask act;
  long long* wgst;   // diagnostics (option debug_stamps=2): 8 wall-clock stamps per workgroup of the last rebuilding launch
};

// segment offsets: s_off[k] = base + sum of tot[0..k), s_off[nseg] = grand total (nseg <= 1024)
template <int BS>
__device__ __forceinline__ void seg_offsets(const int* tot, int nseg, int* s_off) {
  int carry = 0;
  for (int base = 0; base < nseg; base += BS) {
    const int k = base + (int)threadIdx.x;
    const int v = k < nseg ? tot[k] : 0;
    int t;
    const int ex = block_scan_excl<BS>(v, &t);
    if (k < nseg) s_off[k] = carry + ex;
    carry += t;
  }
  if (threadIdx.x == 0) s_off[nseg] = carry;
  __syncthreads();
}

// block-parallel local scans: segment `seg` = 2^shift consecutive items; loc[] = exclusive prefix inside
// the segment, tot[seg] = segment sum.  zero != 0: the counts are cleared for the next rebuild.
template <int BS>
__device__ __forceinline__ void seg_scan(int* cnt, int nitem, int shift, int* loc, int* tot, bool zero, int* cnt_copy = nullptr) {
  const int per = 1 << shift, nseg = (nitem + per - 1) >> shift;
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int lo = seg << shift, hi = min(nitem, lo + per);
    int carry = 0;
    for (int base = lo; base < hi; base += BS) {
      const int k = base + (int)threadIdx.x;
      int v = 0;
      if (k < hi) { v = cnt[k]; if (zero) cnt[k] = 0; if (cnt_copy) cnt_copy[k] = v; }
      int t;
      const int ex = block_scan_excl<BS>(v, &t);
      if (k < hi) loc[k] = carry + ex;
      carry += t;
    }
    if (threadIdx.x == 0) tot[seg] = carry;
  }
}

#ifndef CHEM_FUSED_WAVES
#define CHEM_FUSED_WAVES 4
#endif
// DIAG = true: diagnostic instantiation (option debug_stamps) with the per-workgroup phase stamps and the ablation switch of
// the list build, also used when the int32 Verlet rows are wanted (chem_get_verlet_pairs); the production instantiation carries none of it (their uniform branches in the peel loop cost 7 % of the
// launch: 386 -> 360 us)
template <typename R, int BS, bool DIAG = false>
__global__ __launch_bounds__(BS, CHEM_FUSED_WAVES) void k_rebuild_fused(const FusedArgs<R> a) {
  __shared__ TileLDS<R> T;
  __shared__ int s_off[1025];
  __shared__ unsigned long long s_m[BS / 64];
  __shared__ int s_tile;
  CHEM_DYN_LDS(R);
  const int t = threadIdx.x, b = blockIdx.x, NB = gridDim.x, lane = lane_id(), w = t >> 6;
  DevCtl* const ctl = a.ctl;

  if (ctl->halt) return;   // a previous launch of this run stopped it (set one launch ago at the earliest: every workgroup sees it)
  // ---- P0: decision, computed redundantly by every workgroup ----
  unsigned long long m = 0;
  for (int k = t; k < a.nblk; k += BS) { const unsigned long long v = a.blockmax[k]; m = v > m ? v : m; }
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(m, o); m = v > m ? v : m; }
  if (lane == 0) s_m[w] = m;
  __syncthreads();
  m = s_m[0];
#pragma unroll
  for (int k = 1; k < BS / 64; ++k) m = s_m[k] > m ? s_m[k] : m;
  const double m2 = sizeof(R) == 4 ? bits_real_f(m) : bits_real_d(m);
  double acc = a.criterion ? sqrt(m2) : ctl->acc_pp[a.par] + sqrt(m2);
  const int forced = ctl->force_rebuild;
  const int need = (acc > a.half_skin) || forced;
  // reference rule on the workload's skin (bookkeeping of thread (0,0) only: nobody else reads these two words)
  auto ref_update = [&]() {
    const double accr = a.criterion ? sqrt(m2) : ctl->acc_ref + sqrt(m2);
    const bool rn = (accr > a.half_skin_ref) || forced;
    ctl->acc_ref = rn ? 0.0 : accr;
    if (rn) ctl->ref_rebuilds++;
  };
  if (!need) {
    if (b == 0 && t == 0) { ctl->step_m2 = m2; ctl->acc_pp[a.par ^ 1] = acc; ctl->acc_maxdist = acc; ctl->need_rebuild = 0; ref_update(); }
    return;
  }

  // ---- P1: bin.  Members go straight into the bucket row of their cell; per-segment particle totals are
  //          accumulated beside the cell counts, so that no scan phase is needed afterwards ----
  if (b == 0 && t == 0) a.gb->stamp[0] = wall_clock64();
  long long* const wst = (DIAG && a.wgst) ? a.wgst + 8 * (size_t)b : nullptr;
  const int ablate = DIAG ? a.ablate : 0;
#define WGST(K) do { if (wst && t == 0) wst[K] = wall_clock64(); } while (0)
  WGST(0);
  if (b == 0 && t < 8) a.gb->tq[t][0] = 0u;
  if (b == 0 && t == 8) { ctl->bw64 = 0ull; a.gb->prepq[0] = 0u; }
  { MigBuf<R> none{}; dev_bin<R>(0, a.n, a.x4, a.v4, a.tag, a.img4, a.box, a.cell_cnt, a.cell_of, a.slot_of, none, none, ctl, a.bucket, a.bcap, a.btot, a.seg_shift); }
  WGST(1);
  if (!grid_barrier(a.gb, ctl)) return; WGST(2); if (b == 0 && t == 0) { const long long st = wall_clock64(); a.gb->stamp[1] = st; a.gb->stamp[2] = st; a.gb->stamp[3] = st; }
  if (ctl->bucket_overflow) {
    // a cell is fuller than a bucket row: nothing but the counters has been touched yet -- clear them and leave; the
    // host redoes this rebuild with the unfused chain (force_rebuild stays set)
    for (int k = b * BS + t; k < a.ncell; k += NB * BS) a.cell_cnt[k] = 0;
    if (b == 0) for (int k = t; k < 1024; k += BS) a.btot[k] = 0;
    if (b == 0 && t == 0) { ctl->need_rebuild = 1; ctl->force_rebuild = 1; ctl->halt_step = a.istep; ctl->halt = 1; }
    return;
  }
  // every workgroup has taken its decision: the control block may change now
  if (b == 0 && t == 0) {
    ctl->step_m2 = m2; ctl->acc_pp[a.par ^ 1] = 0.0; ctl->acc_maxdist = 0.0; ctl->force_rebuild = 0;
    ctl->rebuild_count++; ctl->need_rebuild = 1;
    ref_update();
  }

  // ---- P4: cell_start, canonical order inside every cell + gather + tag -> index map; home count of every tile ----
  {
    const int nseg = (a.ncell + (1 << a.seg_shift) - 1) >> a.seg_shift;
    seg_offsets<BS>(a.btot, nseg, s_off);
    if (b == 0 && t == 0) a.cell_start[a.ncell] = s_off[nseg];
    dev_sort_gather_bucket<R>(a.ncell, a.cell_cnt, s_off, a.seg_shift, a.bucket, a.bcap, a.cell_start, a.x4, a.v4, a.tag, a.img4,
                              a.x4o, a.v4o, a.tago, a.img4o, a.rtag, a.box, a.cell_sub);
    const int nx = a.box.nc[0], ny = a.box.nc[1], nz = a.box.nc[2];
    const int ntx = tile_ntx(nx, a.box.xs_nb, a.box.xs_w), nty = (ny + HY - 1) / HY;
    for (int tile = b * BS + t; tile < a.ntiles; tile += NB * BS) {
      const int tx = tile % ntx, ty = (tile / ntx) % nty, tz = tile / (ntx * nty);
      int cx0, hx;
      tile_xrange(tx, nx, a.box.xs_nb, a.box.xs_w, cx0, hx);
      const int cy0 = ty * HY, cz0 = tz * HZ;
      const int hy = min(HY, ny - cy0), hz = min(HZ, nz - cz0);
      int nh = 0;
      for (int hzi = 0; hzi < hz; ++hzi) for (int hyi = 0; hyi < hy; ++hyi) {
        const int c0 = ((cz0 + hzi) * ny + (cy0 + hyi)) * nx + cx0;
        for (int k = 0; k < hx; ++k) nh += a.cell_cnt[c0 + k];
      }
      a.tn[tile] = nh;
    }
  }
  WGST(3);
  if (!grid_barrier(a.gb, ctl)) return; WGST(4); if (b == 0 && t == 0) { const long long st = wall_clock64(); a.gb->stamp[4] = st; a.gb->stamp[5] = st; }

  // ---- P6: tile descriptors + list build.  Tiles are handed out dynamically, each XCD first drains its own contiguous
  //          range of tiles (L2 locality, see xcd_remap) and then helps the others.  The counts of this rebuild are
  //          cleared for the next one; the sorted arrays (x4o/v4o/tago/img4o) are read in place and copied back into
  //          the live arrays by workgroups that have run out of tiles ----
  {
    for (int k = b * BS + t; k < a.ncell; k += NB * BS) a.cell_cnt[k] = 0;
    if (b == 0) for (int k = t; k < 1024; k += BS) a.btot[k] = 0;
    // base of every tile's region in the transposed list = exclusive prefix of the home counts: segment totals of
    // 2^tseg_shift tiles are summed by every workgroup for itself (a few KB of reads), the rest at claim time
    const int tper = 1 << a.tseg_shift, ntseg = (a.ntiles + tper - 1) >> a.tseg_shift;
    {
      int carry = 0;
      for (int base = 0; base < ntseg; base += BS) {
        const int k = base + t;
        int v = 0;
        if (k < ntseg) for (int j = k << a.tseg_shift; j < min(a.ntiles, (k + 1) << a.tseg_shift); ++j) v += a.tn[j];
        int tot;
        const int ex = block_scan_excl<BS>(v, &tot);
        if (k < ntseg) s_off[k] = carry + ex;
        carry += tot;
      }
      __syncthreads();
    }
    const int q = a.ntiles >> 3, r = a.ntiles & 7, myx = b & 7;
    const int lpar = a.tile_ord ? (ctl->rebuild_count & 1) : 0;      // (incremented behind the first barrier: the same for everybody here)
    const int* const ord_r = a.tile_ord ? a.tile_ord + (size_t)lpar * a.ntiles : nullptr;
    const int* const pos_r = a.tile_pos ? a.tile_pos + (size_t)lpar * a.ntiles : nullptr;
    int ndone = 0;
    WGST(5);
    // thread 0 claims a tile: own XCD's queue first, then the others'
    auto claim = [&]() {
      int tile = -1;
      for (int d = 0; d < 8 && tile < 0; ++d) {
        const int x = (myx + d) & 7;
        const int cntx = q + (x < r ? 1 : 0);
        if (__hip_atomic_load(&a.gb->tq[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned int)cntx) continue;
        const int k = (int)atomicAdd(&a.gb->tq[x][0], 1u);
        if (k < cntx) { tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k; if (ord_r) tile = ord_r[tile]; }
      }
      return tile;
    };
    for (;;) {
      __syncthreads();
      if (t == 0) s_tile = claim();    // (claiming ahead of time, beside the previous tile's work, hides this round trip but
      __syncthreads();                 //  unbalances the tail of the phase: measured 307 -> 367 us)
      const int tile = s_tile;
      if (tile < 0) break;
      ++ndone;
      // base of the tile's list region = prefix of its segment + the home counts of the tiles in front of it inside the
      // segment: loaded here by the last wave, reduced after the tables (their latency overlaps)
      int hb_part = 0;
      if (w == BS / 64 - 1) {
        for (int j = ((tile >> a.tseg_shift) << a.tseg_shift) + lane; j < tile; j += 64) hb_part += a.tn[j];
      }
      const bool f32list = sizeof(R) == 4 || !(DIAG && a.want32);   // fp64: the exact list build only where the int32 rows are wanted
      if (f32list) list_stage_clear<BS>(chem_dyn_lds, list_lds_layout(a.CAP, a.ntypes), a.ntypes);   // (tile_tables synchronises)
      tile_tables<R>(T, a.CAP, tile, a.cell_start, a.box, ctl, a.cell_sub);      // ends with a workgroup barrier
      if (w == BS / 64 - 1) {
        for (int o = 32; o > 0; o >>= 1) hb_part += __shfl_xor(hb_part, o);
        if (lane == 0) T.geom[5] = s_off[tile >> a.tseg_shift] + hb_part;
      }
      __syncthreads();
      {   // the descriptor every force launch until the next rebuild reads
        const int* src = reinterpret_cast<const int*>(&T);
        int* dst = reinterpret_cast<int*>(&a.desc[pos_r ? pos_r[tile] : tile]);
        for (int k = t; k < (int)(sizeof(TileLDS<R>) / 4); k += BS) dst[k] = src[k];
      }
      if (f32list) {
        const ListLDS L = list_lds_layout(a.CAP, a.ntypes);
        list_stage_f32<R, BS>(T, chem_dyn_lds, L, a.CAP, a.x4o, a.act, a.ntypes);
        __syncthreads();
        if (ablate == 1) { for (int q = t; q < T.geom[4]; q += BS) a.nnh[T.geom[5] + q] = 0; }
        else
        dev_nlist_tile_f32<R, BS>(T, chem_dyn_lds, L, a.tago, (float)a.rl2, a.excl_start, a.excl_list, a.bond_pass ? 0 : a.has_excl, a.nl16, a.S, a.nnh,
                               (DIAG && a.want32) ? a.nlist : (int*)nullptr, a.S, a.nn, ctl, &a.box, a.rtag, a.x4o, ablate, (float)a.rl2_rows,
                               a.bond_pass ? (uint4*)nullptr : a.bslots);
      } else {
        tile_fill<R, BS, true>(T, sx, a.CAP, a.x4o, 1);
        __syncthreads();
        dev_nlist_tile<R, BS>(T, sx, a.tago, a.rl2, a.excl_start, a.excl_list, a.has_excl, a.act, a.nl16, a.S, a.nnh,
                              a.want32 ? a.nlist : (int*)nullptr, a.S, a.nn, ctl, &a.box, a.rtag, a.x4o, a.rl2_rows);
      }
    }
    WGST(6);
    if (wst && t == 0) wst[7] = ndone;
    // order of the NEXT rebuild (and of the force launches behind it): wave 0 of workgroup x < 8 sorts the tiles of XCD
    // range x by home count, largest first -- eight classes relative to the mean, stable inside a class (ballot prefix)
    if (a.tile_ord && b < 8 && w == 0) {
      const int cntx = q + (b < r ? 1 : 0), off = b < r ? b * (q + 1) : r * (q + 1) + (b - r) * q;
      int* const ord_w = a.tile_ord + (size_t)(lpar ^ 1) * a.ntiles;
      int* const pos_w = a.tile_pos + (size_t)(lpar ^ 1) * a.ntiles;
      const int mean4 = (int)(((long long)a.n * 4 + a.ntiles - 1) / a.ntiles);      // class = 4 tn / mean, clamped: 7 classes above 7/4 of the mean ... 0
      auto cls_of = [&](int tile) { const int c = a.tn[tile] * 16 / (mean4 > 0 ? mean4 : 1); return 7 - (c > 7 ? 7 : c); };   // 0 = largest
      int tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int k0 = 0; k0 < cntx; k0 += 64) {
        const int k = k0 + lane, c = k < cntx ? cls_of(off + k) : -1;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) tot[cc] += __popcll(__ballot(c == cc));
      }
      int base[8]; { int acc_ = 0;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) { base[cc] = acc_; acc_ += tot[cc]; } }
      for (int k0 = 0; k0 < cntx; k0 += 64) {
        const int k = k0 + lane, c = k < cntx ? cls_of(off + k) : -1;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const unsigned long long m = __ballot(c == cc);
          if (c == cc) { const int p_ = off + base[cc] + __popcll(m & lanemask_lt()); ord_w[p_] = off + k; pos_w[off + k] = p_; }
          base[cc] += __popcll(m);
        }
      }
    }
    // bonded work list (nothing in this launch reads it): chunks of BS particles from a queue, taken by workgroups that
    // have run out of tiles -- as a fixed share in front of the tiles it was 25 us on every workgroup's critical path,
    // here it fills the idle tail of the phase (rtag is complete since the last barrier)
    if (a.nbent > 0) {
      for (;;) {
        __syncthreads();
        if (t == 0) s_tile = (int)atomicAdd(&a.gb->prepq[0], 1u);
        __syncthreads();
        const int ib = s_tile * BS;
        if (ib >= a.n) break;
        dev_bonded_prep_chunk<BS>(ib, a.n, a.tago, a.rtag, a.bstart, a.bent, a.bwork, a.bj, ctl, a.bslots ? a.excl_start : (const int*)nullptr);
      }
    }
    // copy-back (+ reference positions): every workgroup moves its share once it has no tile left
    for (int k = b * BS + t; k < a.n; k += NB * BS) {
      const Vec4<R> xk = a.x4o[k];
      a.x4[k] = xk; if (a.x0) a.x0[k] = xk;
      a.v4[k] = a.v4o[k]; a.img4[k] = a.img4o[k]; a.tag[k] = a.tago[k];
    }
  }
  // Lists this launch could not build -- a stencil beyond the LDS tile capacity, a row beyond its stride -- stop the run like a
  // full bucket row does (the particles are sorted and consistent, the forces of this step never evaluated): the host grows
  // the capacity at its next synchronisation and re-enters at this step (chem_api.hip: halted(), rebuild_now()).
  if (t == 0) {
    const volatile DevCtl* vc = ctl;
    if (vc->stage_overflow | vc->nl_overflow) { ctl->halt_step = a.istep; ctl->halt = 1; }
  }
  if (b == 0 && t == 0) a.gb->stamp[6] = wall_clock64();
  if (wst) { __syncthreads(); if (t == 0) wst[7] |= (long long)(wall_clock64() - wst[0]) << 16; }
#undef WGST
}

// =======================================================================================
// K11 observables (analysis.Temperature/KineticEnergy, start_simulation.py:453-493)
// =======================================================================================
template <typename R>
__global__ __launch_bounds__(256) void k_kinetic(int i0, int n, const Vec4<R>* __restrict__ v4, double* __restrict__ out) {
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  double ek = 0, px = 0, py = 0, pz = 0;
  if (i < i0 + n) {
    const Vec4<R> v = v4[i];
    const double m = (double)v.w, vx = (double)v.x, vy = (double)v.y, vz = (double)v.z;
    ek = 0.5 * m * (vx * vx + vy * vy + vz * vz); px = m * vx; py = m * vy; pz = m * vz;
  }
  __shared__ double red[4][4];
  for (int o = 32; o > 0; o >>= 1) { ek += __shfl_xor(ek, o); px += __shfl_xor(px, o); py += __shfl_xor(py, o); pz += __shfl_xor(pz, o); }
  if (lane_id() == 0) { int w = threadIdx.x >> 6; red[0][w] = ek; red[1][w] = px; red[2][w] = py; red[3][w] = pz; }
  __syncthreads();
  if (threadIdx.x < 4) {
    double s = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[threadIdx.x][k];
    out[4 * blockIdx.x + threadIdx.x] = s;
  }
}

// ---- velocity-rescaling thermostats (Berendsen, Isokinetic; start_simulation.py:341-348) --------
// k_kinetic leaves per-block (Ekin, p) partials; one block folds them into the scale factor, a
// streaming kernel applies it.  Decomposed path: the fold writes the local Ekin, the ranks sum it
// (transport all-reduce), k_rescale_lambda is launched with nblk = 0 and reads the sum.
__global__ __launch_bounds__(256) void k_rescale_lambda(const double* __restrict__ part, int nblk, double* __restrict__ ekin_io, double* __restrict__ lam_out,
                                                        int kind, double kT, double pref /* dt/tau */, double ntot, uint64_t svr_seed, uint64_t step) {
  __shared__ double red[4];
  double ek = 0;
  for (int k = threadIdx.x; k < nblk; k += blockDim.x) ek += part[4 * k];
  for (int o = 32; o > 0; o >>= 1) ek += __shfl_xor(ek, o);
  if (lane_id() == 0) red[threadIdx.x >> 6] = ek;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (nblk > 0) { ek = red[0] + red[1] + red[2] + red[3]; *ekin_io = ek; }
    else ek = *ekin_io;
    if (lam_out) {
      const double kTnow = 2.0 * ek / (3.0 * ntot);
      // kind 3: StochasticVelocityRescaling (one scalar draw per step, include/chem_philox.h; taut = tau/dt = 1/pref)
      *lam_out = kind == 1 ? sqrt(1.0 + pref * (kT / kTnow - 1.0)) : kind == 2 ? sqrt(kT / kTnow)
               : chem_philox::svr_lambda(svr_seed, step, ek, 1.5 * ntot * kT, (int64_t)(3.0 * ntot), 1.0 / pref);
    }
  }
}
template <typename R>
__global__ __launch_bounds__(256) void k_scale_v(int i0, int n, Vec4<R>* __restrict__ v4, const double* __restrict__ lam, const DevCtl* __restrict__ ctl) {
  if (ctl->halt) return;   // (a stopped run must not keep rescaling its frozen velocities)
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= i0 + n) return;
  const R s = (R)*lam;
  Vec4<R> v = v4[i];
  v.x *= s; v.y *= s; v.z *= s;
  v4[i] = v;
}

// =======================================================================================
// K9/K10  reaction scan + resolve (integrator.ChemicalReaction / Reaction,
//     reaction_setup.py:71-165,416-427)
// =======================================================================================
struct ReactionDev {
  int type_1, type_2, delta_1, delta_2;
  int min1, max1, min2, max2;
  int intramolecular, intraresidual, active, restricted;   // restricted: RestrictReaction -- only pairs of the connection table react
  int cons_role, pad_;   // ReactionConstraintNeighbourState on role 1 | 2: the host evaluates it per particle into a bit table (ConnTable::cons_ok)
  double cut2, mincut2, prob;
};
struct ReactSet { int n; uint64_t seed; uint64_t step; int nearest; ReactionDev r[CHEM_MAX_REACTIONS]; };
// RestrictReaction.define_connection (reaction_setup.py:115-128): per-tag CSR of the allowed partners, one bit per reaction
struct ConnTable { const int* start; const int* partner; const unsigned int* mask; const unsigned int* cons_ok; };   // cons_ok[tag]: bit q = constraint of reaction q holds
__device__ __forceinline__ bool conn_allows(const ConnTable& ct, int ta, int tb, int q) {
  if (!ct.start) return false;
  for (int e = ct.start[ta]; e < ct.start[ta + 1]; ++e) if (ct.partner[e] == tb) return (ct.mask[e] >> q) & 1u;
  return false;
}

struct Candidate { int a, b, r; unsigned int h; double d2; };  // a: role type_1, b: role type_2 (tags)

// Traverses the Verlet list once (pairs with tag_i < tag_j), applies the type/state/res_id
// filters, evaluates the distance criterion in fp64 without FMA contraction (bit-comparable
// with the CPU oracle) and appends the hits with one atomic per wave (ballot compaction).
template <typename R>
__global__ __launch_bounds__(256) void k_react_scan(int i0, int n, const Vec4<R>* __restrict__ x4, const int* __restrict__ tag,
                                                    const int* __restrict__ nlist, const int* __restrict__ nn, int S,
                                                    const int* __restrict__ state, const int* __restrict__ res_id,
                                                    const int* __restrict__ mol_id, BoxD box, const ReactSet* __restrict__ rs_g,
                                                    Candidate* __restrict__ cand, int cand_cap, DevCtl* ctl, ConnTable conn) {
  __shared__ ReactSet rs;
  {
    const int* src = reinterpret_cast<const int*>(rs_g);
    int* dst = reinterpret_cast<int*>(&rs);
    for (int k = threadIdx.x; k < (int)(sizeof(ReactSet) / 4); k += blockDim.x) dst[k] = src[k];
  }
  __syncthreads();
  // 8 lanes per particle row; hits are staged per wave in LDS (wave-private region)
  constexpr int TPP = 8;
  constexpr int kWaveBuf = 192;
  __shared__ Candidate sbuf[4 * kWaveBuf];
  Candidate* wbuf = sbuf + (threadIdx.x >> 6) * kWaveBuf;
  int wcount = 0;   // wave-uniform
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = i0 + gid / TPP, sub = gid % TPP;
  int cnt = 0, ti = 0, si = 0, tgi = 0, ri = 0, mi = 0;
  Vec4<R> xi = mk4<R>(0, 0, 0, 0);
  const int* row = nullptr;
  if (i < i0 + n) {
    xi = x4[i]; ti = (int)xi.w; tgi = tag[i]; si = state[tgi]; ri = res_id[tgi]; mi = mol_id[tgi];
    cnt = nn[i]; row = nlist + (size_t)i * S;
    // quick reject: particle cannot take part in any active reaction
    bool any = false;
    for (int q = 0; q < rs.n; ++q) {
      const ReactionDev& R_ = rs.r[q];
      if (!R_.active) continue;
      any |= (ti == R_.type_1 && si >= R_.min1 && si < R_.max1) || (ti == R_.type_2 && si >= R_.min2 && si < R_.max2);
    }
    if (!any) cnt = 0;
  }
  float maxcut2f = 0;
  for (int q = 0; q < rs.n; ++q) if (rs.r[q].active) maxcut2f = fmaxf(maxcut2f, (float)rs.r[q].cut2);
  maxcut2f *= 1.001f;
  const float bLf[3] = {(float)box.L[0], (float)box.L[1], (float)box.L[2]}, biLf[3] = {(float)box.invL[0], (float)box.invL[1], (float)box.invL[2]};
  // uniform trip count inside a wave so that __ballot is convergent
  int maxcnt = cnt;
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(maxcnt, o); maxcnt = t > maxcnt ? t : maxcnt; }
  for (int k0 = 0; k0 < maxcnt; k0 += TPP) {
    const int k = k0 + sub;
    int j = -1, tj = 0, sj = 0, tgj = 0, rj = 0, mj = 0;
    bool live = k < cnt;
    Vec4<R> xj = xi;
    if (live) {
      j = row[k]; tgj = tag[j];
      live = tgi < tgj;
      if (live) {
        xj = x4[j];
        // cheap reject before the three by-tag gathers and the fp64 arithmetic: the reaction radii are
        // far inside the list radius (1.2 vs 2.8: 8 % of the neighbours).  Same inputs as the fp64
        // distance below, 1e-3 relative margin >> any rounding difference.
        const D3 pi_ = posD<R>(xi, box), pj_ = posD<R>(xj, box);
        const float fx = (float)(pi_.x - pj_.x), fy = (float)(pi_.y - pj_.y), fz = (float)(pi_.z - pj_.z);
        const float gx = fx - bLf[0] * rintf(fx * biLf[0]), gy = fy - bLf[1] * rintf(fy * biLf[1]), gz = fz - bLf[2] * rintf(fz * biLf[2]);
        live = gx * gx + gy * gy + gz * gz <= maxcut2f;
      }
      if (live) { tj = (int)xj.w; sj = state[tgj]; rj = res_id[tgj]; mj = mol_id[tgj]; }
    }
    double d2 = 0;
    if (__any(live)) {
      if (live) {
        D3 d = minimgD(box, posD<R>(xi, box) - posD<R>(xj, box));
        d2 = dist2_unfused(d);
      }
      for (int q = 0; q < rs.n; ++q) {
        const ReactionDev& R_ = rs.r[q];
        bool hit = false; int a = 0, b = 0; unsigned int h = 0;
        if (live && R_.active) {
          const bool fwd = ti == R_.type_1 && si >= R_.min1 && si < R_.max1 && tj == R_.type_2 && sj >= R_.min2 && sj < R_.max2;
          const bool rev = tj == R_.type_1 && sj >= R_.min1 && sj < R_.max1 && ti == R_.type_2 && si >= R_.min2 && si < R_.max2;
          if (fwd || rev) {
            hit = true;
            if (fwd) { a = tgi; b = tgj; } else { a = tgj; b = tgi; }   // tgi < tgj: forward role assignment wins
            if (!R_.intraresidual && ri == rj) hit = false;
            if (!R_.intramolecular && mi == mj) hit = false;
            if (!(d2 >= R_.mincut2 && d2 < R_.cut2)) hit = false;
            if (hit && R_.restricted && !conn_allows(conn, tgi, tgj, q)) hit = false;
            if (hit && R_.cons_role && !((conn.cons_ok[R_.cons_role == 1 ? a : b] >> q) & 1u)) hit = false;
            if (hit) {
              uint32_t rr[4];
              chem_philox::reaction_draw(rs.seed, rs.step, (uint32_t)tgi, (uint32_t)tgj, (uint32_t)q, rr);
              if (R_.prob < 1.0 && !(chem_philox::u01(rr[0]) < R_.prob)) hit = false;
              h = rr[1];
            }
          }
        }
        const unsigned long long m = __ballot(hit);
        if (m) {
          const int nh = __popcll(m);
          if (wcount + nh > kWaveBuf) {   // flush this wave's staged hits: one global atomic per flush
            int base = 0;
            if (lane_id() == 0) base = atomicAdd(&ctl->cand_count, wcount);
            base = __shfl(base, 0);
            for (int t = lane_id(); t < wcount; t += 64) { if (base + t < cand_cap) cand[base + t] = wbuf[t]; else ctl->cand_overflow = 1; }
            wcount = 0;
          }
          if (hit) wbuf[wcount + __popcll(m & lanemask_lt())] = Candidate{a, b, q, h, d2};
          wcount += nh;
        }
      }
    }
  }
  if (wcount > 0) {
    int base = 0;
    if (lane_id() == 0) base = atomicAdd(&ctl->cand_count, wcount);
    base = __shfl(base, 0);
    for (int t = lane_id(); t < wcount; t += 64) { if (base + t < cand_cap) cand[base + t] = wbuf[t]; else ctl->cand_overflow = 1; }
  }
}

// ---- reaction scan on the staged tiles, WITHOUT rebuilding anything.
// Role words of the reaction scan on tiles: for every particle slot (position order) one bit per reaction and role -- bit q:
// "type and state fit role 1 of (active) reaction q", bit 16 + q: role 2 (CHEM_MAX_REACTIONS = 16).  One streaming pass in
// front of the scan (tag -> state is the only gather); the scan stages the words beside the positions (4 bytes per slot), so
// that the role compatibility of a candidate pair is decided in LDS and only the pairs that can react reach the dependent
// global loads (tags, exclusion row, labels, exact fp64 distance).
template <typename R>
__global__ __launch_bounds__(256) void k_react_roles(int nslot, int ntag, const Vec4<R>* __restrict__ x4, const int* __restrict__ tag,
                                                     const int* __restrict__ state, const ReactSet* __restrict__ rs_g, unsigned int* __restrict__ role) {
  __shared__ ReactSet rs;
  {
    const int* src = reinterpret_cast<const int*>(rs_g);
    int* dst = reinterpret_cast<int*>(&rs);
    for (int k = threadIdx.x; k < (int)(sizeof(ReactSet) / 4); k += 256) dst[k] = src[k];
  }
  __syncthreads();
  for (int g = blockIdx.x * 256 + threadIdx.x; g < nslot; g += gridDim.x * 256) {
    const int tg = tag[g];
    unsigned int bits = 0;
    if (tg >= 0 && tg < ntag) {
      const int t = (int)x4[g].w, st = state[tg];
      for (int q = 0; q < rs.n; ++q) {
        const ReactionDev& R_ = rs.r[q];
        if (!R_.active) continue;
        if (t == R_.type_1 && st >= R_.min1 && st < R_.max1) bits |= 1u << q;
        if (t == R_.type_2 && st >= R_.min2 && st < R_.max2) bits |= 1u << (16 + q);
      }
    }
    role[g] = bits;
  }
}
// stages role[g] of every stencil slot (same addressing as tile_fill's register-lean form)
template <typename R, int BS>
__device__ __forceinline__ void tile_fill_roles(const TileLDS<R>& T, unsigned int* const sr, const int CAP, const unsigned int* __restrict__ role) {
  constexpr int NW = BS / 64;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int r = w; r < NROW; r += NW) {
    const int len = T.celloff[r][SX], o0 = T.rowoff[r];
    for (int e = l; e < len; e += 64) {
      int k = 0;
#pragma unroll
      for (int q = 1; q < SX; ++q) k += (e >= T.celloff[r][q]) ? 1 : 0;
      const int g = T.cellg[r][k] + (e - T.celloff[r][k]);
      if (o0 + e < CAP) sr[o0 + e] = role[g];
    }
  }
}
__host__ __device__ constexpr size_t scan_roles_offset(int CAP, size_t slot_bytes) { return ((size_t)(CAP + 5) * slot_bytes + 15) / 16 * 16; }

// The int32-list scan above needs a fresh list, i.e. a forced rebuild at the reaction step -- and a
// rebuild re-sorts the particles but not their forces, which the next half-kick still needs (the
// reaction sits between the force evaluation of step s and the first kick of step s+1).  This
// kernel needs no list: two particles within a reaction radius (<= rc) now were within
// radius + skin <= rc + skin = one cell edge at the last rebuild, so the 27-cell stencil of the
// tile tables of THAT rebuild still contains every candidate, with the current positions staged
// into LDS.  Per home particle: fp32 pre-test of the stencil candidates against the largest
// reaction radius, then for the few survivors (tag order, states, labels) the fp64 distance from
// the global arrays exactly as k_react_scan computes it, the reaction filters and the Philox draw.
// Round 3: tag and role bits are staged beside the positions (k_react_roles), so a home particle without a role costs
// nothing and a candidate reaches the global loads only if tag order and roles allow a reaction.
// Output: every tile appends to its own fixed region of `region` through an LDS counter (no
// contended global atomic); k_cand_offsets / k_cand_gather compact the regions afterwards.
template <typename R, int BS>
__global__ __launch_bounds__(BS, 4) void k_react_scan_tiles(int ntiles, int CAP, const Vec4<R>* __restrict__ x4, const int* __restrict__ tag,
                                                            const TileLDS<R>* __restrict__ desc, const int* __restrict__ state,
                                                            const int* __restrict__ res_id, const int* __restrict__ mol_id, BoxD box,
                                                            const ReactSet* __restrict__ rs_g, Candidate* __restrict__ region, int region_cap,
                                                            int* __restrict__ tile_count, DevCtl* ctl, R slack,
                                                            const int* __restrict__ excl_start, const int* __restrict__ excl_list, ConnTable conn,
                                                            const unsigned int* __restrict__ role, int roles_staged) {      // (roles_staged 0: the image leaves no room behind it -- role words from global memory)
  __shared__ TileLDS<R> T;
  __shared__ ReactSet rs;
  __shared__ int s_cnt;
  CHEM_DYN_LDS(R);
  unsigned int* const sr = reinterpret_cast<unsigned int*>(chem_dyn_lds + scan_roles_offset(CAP, sizeof(Vec4<R>)));   // role words behind the image
  {
    const int* src = reinterpret_cast<const int*>(rs_g);
    int* dst = reinterpret_cast<int*>(&rs);
    for (int k = threadIdx.x; k < (int)(sizeof(ReactSet) / 4); k += BS) dst[k] = src[k];
  }
  __syncthreads();
  R maxcut2 = 0;
  for (int q = 0; q < rs.n; ++q) if (rs.r[q].active) maxcut2 = rs.r[q].cut2 > (double)maxcut2 ? (R)rs.r[q].cut2 : maxcut2;
  maxcut2 *= (R)1.001;   // the staged positions carry the periodic shift: one more rounding than the global ones
  for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
    const int tile = xcd_remap(vb, ntiles);
    __syncthreads();
    tile_load_desc<R>(T, desc, tile);
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    tile_fill<R, BS, true>(T, sx, CAP, x4, 1);
    if (roles_staged) tile_fill_roles<R, BS>(T, sr, CAP, role);
    __syncthreads();
    const int hx = T.geom[0], total = T.geom[3], nhome = T.geom[4];
    if (roles_staged && threadIdx.x == 0 && total < CAP) sr[total] = 0u;      // (the far-away dummy slot: no role)
    __syncthreads();
    Candidate* out = region + (size_t)tile * region_cap;
    for (int q = threadIdx.x; q < nhome; q += BS) {
      int sgi = 0;
#pragma unroll
      for (int k = 1; k < NHSEG; ++k) sgi += (q >= T.hoff[k]) ? 1 : 0;
      const int inrun = q - T.hoff[sgi];
      const int p = T.hstart[sgi] + inrun;
      const int ly = sgi % HY, lz = sgi / HY;
      const int hr = (lz + 1) * SY + (ly + 1);
      const int eh = inrun + T.celloff[hr][1];
      int lx = 0;
      for (int k = 2; k <= hx; ++k) lx += (eh >= T.celloff[hr][k]) ? 1 : 0;
      const int sself = T.rowoff[hr] + eh;
      const unsigned int rwi = roles_staged ? sr[sself] : role[p];
      if (!rwi) continue;                                     // no role in any active reaction
      const Vec4<R> xi = sx[sself];
      const int tgi = tag[p];
      const unsigned int ri1 = rwi & 0xffffu, ri2 = rwi >> 16;
      const int ri = res_id[tgi], mi = mol_id[tgi];
      const Vec4<R> xgi = x4[p];
      // x-window of every stencil row, as in the list build (dev_nlist_tile) but for the largest reaction radius and
      // with the tables of the LAST rebuild: a candidate sits within `slack` (= skin/2) of where it was binned, so the
      // row distances shrink and the window grows by that much
      const R rmax = sqrt_r(maxcut2), weps = T.clen[0] * (R)2e-4;
      auto suboff = [&](int r, int f, bool lower) {
        const int k = f >> 2, j = f & 3;
        const int base = T.celloff[r][k];
        if (j == 0) return base;
        const int pk = T.cellsub[r][k];
        if (pk < 0) return lower ? base : T.celloff[r][k + 1];
        return base + ((pk >> (8 * (j - 1))) & 0xff);
      };
#pragma unroll 1
      for (int dzy = 0; dzy < 9; ++dzy) {
        const int dz = dzy / 3, dy = dzy - 3 * dz;
        const int r = (lz + dz) * SY + (ly + dy);
        const R ylo = ((R)(ly + 1) - (R)kRefCells) * T.clen[1], zlo = ((R)(lz + 1) - (R)kRefCells) * T.clen[2];   // (staged coordinates are relative to T.org)
        R ddy = dy == 0 ? xi.y - ylo : (dy == 2 ? ylo + T.clen[1] - xi.y : (R)0);
        R ddz = dz == 0 ? xi.z - zlo : (dz == 2 ? zlo + T.clen[2] - xi.z : (R)0);
        ddy -= slack + weps; ddz -= slack + weps;
        ddy = ddy > 0 ? ddy : (R)0; ddz = ddz > 0 ? ddz : (R)0;
        const R w2 = rmax * rmax - ddy * ddy - ddz * ddz;
        if (w2 < (R)0) continue;
        const R ws = (sqrt_r(w2) + slack + weps) * T.subinv, sxi = (xi.x + (R)kRefCells * T.clen[0]) * T.subinv;
        int f_lo = (int)(sxi - ws), f_hi = (int)(sxi + ws);
        f_lo = f_lo > lx * NSUB ? f_lo : lx * NSUB;
        f_hi = f_hi < (lx + 3) * NSUB - 1 ? f_hi : (lx + 3) * NSUB - 1;
        const int a = T.rowoff[r] + suboff(r, f_lo, true);
        int b = T.rowoff[r] + suboff(r, f_hi + 1, false);
        b = b < total ? b : total;
        // four candidates per trip: their LDS reads are issued together (one latency per four tests; slots behind the
        // run read the far-away dummy at `total`)
        for (int sl0 = a; sl0 < b; sl0 += 4) {
          Vec4<R> xq[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) xq[u] = lds_gather4<R>(sx, sl0 + u < b ? sl0 + u : total);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
          const int sl = sl0 + u;
          const Vec4<R> xj = xq[u];
          const R dx = xi.x - xj.x, dy_ = xi.y - xj.y, dz_ = xi.z - xj.z;
          if (dx * dx + dy_ * dy_ + dz_ * dz_ > maxcut2 || sl == sself || sl >= b) continue;
          const unsigned int rwj = roles_staged ? sr[sl] : role[real_as_idx(xj.w) >> 5];
          const unsigned int fwdm = ri1 & (rwj >> 16), revm = ri2 & (rwj & 0xffffu);   // bit q: reaction q with (i, j) as roles (1, 2) / (2, 1)
          if (!(fwdm | revm)) continue;
          const int j = real_as_idx(xj.w) >> 5;
          const int tgj = tag[j];
          if (!(tgi < tgj)) continue;          // every pair once, from its lower tag (on the rank that owns it)
          if (excl_start) {                    // candidates are Verlet-list pairs: an excluded (e.g. bonded) pair never reacts
            bool ex = false;
            for (int e = excl_start[tgi]; e < excl_start[tgi + 1]; ++e) ex |= excl_list[e] == tgj;
            if (ex) continue;
          }
          const int rj = res_id[tgj], mj = mol_id[tgj];
          const D3 d = minimgD(box, posD<R>(xgi, box) - posD<R>(x4[j], box));
          const double d2 = dist2_unfused(d);
          for (int k = 0; k < rs.n; ++k) {
            const ReactionDev& R_ = rs.r[k];
            const bool fwd = (fwdm >> k) & 1u, rev = (revm >> k) & 1u;      // (bits exist for active reactions only)
            if (!(fwd || rev)) continue;
            if (!R_.intraresidual && ri == rj) continue;
            if (!R_.intramolecular && mi == mj) continue;
            if (!(d2 >= R_.mincut2 && d2 < R_.cut2)) continue;
            if (R_.restricted && !conn_allows(conn, tgi, tgj, k)) continue;
            if (R_.cons_role && !((conn.cons_ok[(R_.cons_role == 1) == fwd ? tgi : tgj] >> k) & 1u)) continue;   // (role 1 is tgi on a forward match)
            uint32_t rr[4];
            chem_philox::reaction_draw(rs.seed, rs.step, (uint32_t)tgi, (uint32_t)tgj, (uint32_t)k, rr);
            if (R_.prob < 1.0 && !(chem_philox::u01(rr[0]) < R_.prob)) continue;
            const int idx = atomicAdd(&s_cnt, 1);
            if (idx < region_cap) out[idx] = fwd ? Candidate{tgi, tgj, k, rr[1], d2} : Candidate{tgj, tgi, k, rr[1], d2};   // tgi < tgj: forward role assignment wins
            else ctl->cand_overflow = 1;
          }
          }
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) tile_count[tile] = s_cnt < region_cap ? s_cnt : region_cap;
  }
}

// exclusive scan of the per-tile candidate counts (single block) -> tile_off, total -> ctl->cand_count
__global__ __launch_bounds__(1024) void k_cand_offsets(int ntiles, const int* __restrict__ tile_count, int* __restrict__ tile_off, DevCtl* ctl) {
  __shared__ int ws[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int w = threadIdx.x >> 6, lane = lane_id();
  for (int base = 0; base < ntiles; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < ntiles ? tile_count[i] : 0;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) ws[w] = incl;
    __syncthreads();
    int off = 0, tot = 0;
    for (int k = 0; k < 16; ++k) { if (k < w) off += ws[k]; tot += ws[k]; }
    if (i < ntiles) tile_off[i] = carry_s + off + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) ctl->cand_count = carry_s;
}

__global__ __launch_bounds__(256) void k_cand_gather(int ntiles, const Candidate* __restrict__ region, int region_cap, const int* __restrict__ tile_count,
                                                     const int* __restrict__ tile_off, Candidate* __restrict__ cand, int cand_cap, DevCtl* ctl) {
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int c = tile_count[tile], o = tile_off[tile];
    const Candidate* src = region + (size_t)tile * region_cap;
    for (int k = threadIdx.x; k < c; k += blockDim.x) { if (o + k < cand_cap) cand[o + k] = src[k]; else ctl->cand_overflow = 1; }
  }
}

// ---- resolve: UniqueA, UniqueB, one-event-per-particle (parallel greedy) -------------
__device__ __forceinline__ unsigned long long cand_key1(const Candidate& c, int nearest) {
  return nearest ? (unsigned long long)__double_as_longlong(c.d2) : (unsigned long long)c.h;
}

// pass 1 of a segmented arg-min: best1[p] = min key1; side 0: p = a, side 1: p = b
__global__ void k_res_min1(int nc, const Candidate* __restrict__ c, const int* __restrict__ alive, int side, int nearest,
                           unsigned long long* __restrict__ best1) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nc || !alive[k]) return;
  atomicMin(&best1[side ? c[k].b : c[k].a], cand_key1(c[k], nearest));
}
// pass 2: tie-break on (partner tag, reaction)
__global__ void k_res_min2(int nc, const Candidate* __restrict__ c, const int* __restrict__ alive, int side, int nearest,
                           const unsigned long long* __restrict__ best1, unsigned long long* __restrict__ best2) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nc || !alive[k]) return;
  const int p = side ? c[k].b : c[k].a, o = side ? c[k].a : c[k].b;
  if (cand_key1(c[k], nearest) == best1[p]) atomicMin(&best2[p], ((unsigned long long)(unsigned)o << 8) | (unsigned)c[k].r);
}
__global__ void k_res_keep(int nc, const Candidate* __restrict__ c, int* __restrict__ alive, int side, int nearest,
                           const unsigned long long* __restrict__ best1, const unsigned long long* __restrict__ best2) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nc || !alive[k]) return;
  const int p = side ? c[k].b : c[k].a, o = side ? c[k].a : c[k].b;
  const bool keep = cand_key1(c[k], nearest) == best1[p] && ((((unsigned long long)(unsigned)o << 8) | (unsigned)c[k].r) == best2[p]);
  if (!keep) alive[k] = 0;
}
// after UniqueA+UniqueB every particle is 'a' of at most one and 'b' of at most one survivor
__global__ void k_res_index(int nc, const Candidate* __restrict__ c, const int* __restrict__ alive, int* __restrict__ asA,
                            int* __restrict__ asB) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nc || !alive[k]) return;
  asA[c[k].a] = k; asB[c[k].b] = k;
}
__device__ __forceinline__ bool cand_less(const Candidate& p, const Candidate& q, int nearest) {
  const unsigned long long kp = cand_key1(p, nearest), kq = cand_key1(q, nearest);
  return kp != kq ? kp < kq : p.a < q.a;
}
// one round of priority greedy matching: status 1 alive, 2 accepted, 0 dead
__global__ void k_res_round(int nc, const Candidate* __restrict__ c, const int* __restrict__ st_in, int* __restrict__ st_out,
                            const int* __restrict__ asA, const int* __restrict__ asB, int nearest, DevCtl* ctl) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nc) return;
  int s = st_in[k];
  if (s == 1) {
    // neighbours: pairs sharing a particle with k
    int nb[4] = {asB[c[k].a], asA[c[k].b], asA[c[k].a], asB[c[k].b]};
    bool ismin = true, dead = false;
    for (int q = 0; q < 4; ++q) {
      const int o = nb[q];
      if (o < 0 || o == k) continue;
      const int so = st_in[o];
      if (so == 2) dead = true;
      else if (so == 1 && cand_less(c[o], c[k], nearest)) ismin = false;
    }
    if (dead) s = 0; else if (ismin) s = 2; else ctl->alive = 1;
  }
  st_out[k] = s;
}

// apply accepted events: state/type/mass on the device arrays, emit compact event records
struct ReactApply { int delta_1, delta_2, new_type_1, new_type_2; double new_mass_1, new_mass_2; };
struct ReactApplySet { ReactApply r[CHEM_MAX_REACTIONS]; };
template <typename R>
__global__ void k_react_apply(int nc, const Candidate* __restrict__ c, const int* __restrict__ st, ReactApplySet ras,
                              int* __restrict__ state, const int* __restrict__ rtag, Vec4<R>* __restrict__ x4,
                              Vec4<R>* __restrict__ v4, Candidate* __restrict__ out, int* __restrict__ out_count) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const bool acc = k < nc && st[k] == 2;
  // output slot: one atomic per wave (a same-address atomic per event serialises in the L2: 250 us for 2.8e5 events)
  const unsigned long long m = __ballot(acc);
  if (!m) return;
  int base = 0;
  if (lane_id() == __ffsll((long long)m) - 1) base = atomicAdd(out_count, __popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1);
  if (!acc) return;
  const Candidate cd = c[k];
  const ReactApply ra = ras.r[cd.r];
  state[cd.a] += ra.delta_1; state[cd.b] += ra.delta_2;
  if (ra.new_type_1 >= 0) { int i = rtag[cd.a]; if (i >= 0) { x4[i].w = (R)ra.new_type_1; v4[i].w = (R)ra.new_mass_1; } }
  if (ra.new_type_2 >= 0) { int i = rtag[cd.b]; if (i >= 0) { x4[i].w = (R)ra.new_type_2; v4[i].w = (R)ra.new_mass_2; } }
  out[base + __popcll(m & lanemask_lt())] = cd;
}

// property changes computed by the host topology manager (PostProcessChangeNeighboursProperty): by tag
struct PropChangeDev { int tag, type, set_state, state; double mass, q; };
template <typename R>
__global__ void k_apply_props(int nchg, const PropChangeDev* __restrict__ chg, int* __restrict__ state, const int* __restrict__ rtag,
                              Vec4<R>* __restrict__ x4, Vec4<R>* __restrict__ v4) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nchg) return;
  const PropChangeDev c = chg[k];
  if (c.set_state) state[c.tag] = c.state;
  const int i = rtag[c.tag];
  if (i >= 0) { x4[i].w = (R)c.type; v4[i].w = (R)c.mass; }
}

// =======================================================================================
// Per-tag CSR tables (bonded entries, exclusions) built ON THE DEVICE from flat, append-only arrays.
// A bond-forming reaction step used to rebuild both tables on the host (a pass over all 10^6 rows, ~14 MB of
// uploads, 3-4 ms); now the host uploads only the NEW tuples / pairs and four small kernels rebuild the CSR:
// count (atomics by tag) -> exclusive scan -> fill (atomic cursor) -> canonical order inside every row
// (one thread per tag, insertion sort: rows hold a handful of entries).  The row order is a pure function of
// the flat arrays, so the tables -- and with them the force sums -- are run-to-run deterministic.
// =======================================================================================
// exclusive scan of n counters -> start[0..n] in three launches (4096 counters per block; block totals scanned by one
// block; offsets added); the counters are cleared on the way (they become the fill cursors)
constexpr int kScanItems = 4096;
__global__ __launch_bounds__(1024) void k_scan_local(int n, int* __restrict__ cnt, int* __restrict__ start, int* __restrict__ btot) {
  __shared__ int ws[16];
  const int i0 = blockIdx.x * kScanItems + threadIdx.x * 4;
  int v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[k] = 0; if (i0 + k < n) { v[k] = cnt[i0 + k]; cnt[i0 + k] = 0; } }
  const int sum = v[0] + v[1] + v[2] + v[3];
  const int lane = lane_id(), w = threadIdx.x >> 6;
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  if (lane == 63) ws[w] = incl;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { if (k < w) off += ws[k]; tot += ws[k]; }
  int run = off + incl - sum;
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (i0 + k < n) start[i0 + k] = run; run += v[k]; }
  if (threadIdx.x == 0) btot[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void k_scan_blocks(int nb, int* __restrict__ btot, int n, int* __restrict__ start) {
  __shared__ int ws[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = lane_id(), w = threadIdx.x >> 6;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nb ? btot[i] : 0;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) ws[w] = incl;
    __syncthreads();
    int off = 0, tot = 0;
    for (int k = 0; k < 16; ++k) { if (k < w) off += ws[k]; tot += ws[k]; }
    if (i < nb) btot[i] = carry_s + off + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) start[n] = carry_s;
}
__global__ __launch_bounds__(1024) void k_scan_add(int n, const int* __restrict__ btot, int* __restrict__ start) {
  const int off = btot[blockIdx.x];
  if (off == 0) return;
  const int i0 = blockIdx.x * kScanItems + threadIdx.x * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) if (i0 + k < n) start[i0 + k] += off;
}

// ---- exclusions: flat pairs (a, b) -> symmetric CSR, rows ascending (= HostTopology::build_excl) ----
__global__ void k_ex_count(int m, const int2* __restrict__ pairs, int* __restrict__ cnt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const int2 p = pairs[k];
  atomicAdd(&cnt[p.x], 1); atomicAdd(&cnt[p.y], 1);
}
__global__ void k_ex_fill(int m, const int2* __restrict__ pairs, const int* __restrict__ start, int* __restrict__ cursor, int* __restrict__ list) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const int2 p = pairs[k];
  list[start[p.x] + atomicAdd(&cursor[p.x], 1)] = p.y;
  list[start[p.y] + atomicAdd(&cursor[p.y], 1)] = p.x;
}
__global__ void k_ex_sort(int n, const int* __restrict__ start, int* __restrict__ cursor, int* __restrict__ list) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  cursor[t] = 0;
  const int s = start[t], e = start[t + 1];
  for (int i = s + 1; i < e; ++i) {
    const int v = list[i];
    int j = i - 1;
    while (j >= s && list[j] > v) { list[j + 1] = list[j]; --j; }
    list[j + 1] = v;
  }
}

// ---- bonded entries: flat tuples (tags t0..t3, list index) -> per-tag CSR of BondedEntry ----
// parameter slot of a tuple: plain lists have one slot, typed lists one per registered type tuple (matched forwards
// or backwards against the CURRENT particle types); -1 = no parameters for these types (tuple skipped, as
// HostTopology::build_bonded does)
struct SlotKey { int list, t0, t1, t2, t3, by_types, arity, pad; };
template <typename R>
__global__ void k_bt_count(int ne, const int4* __restrict__ fent, const int* __restrict__ flist, int nslot, const SlotKey* __restrict__ keys,
                           const Vec4<R>* __restrict__ x4, const int* __restrict__ rtag, const int* __restrict__ type_by_tag,
                           int* __restrict__ eslot, int* __restrict__ cnt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const int4 tt = fent[e];
  const int li = flist[e];
  const int tg[4] = {tt.x, tt.y, tt.z, tt.w};
  int slot = -1, arity = 2;
  for (int s = 0; s < nslot && slot < 0; ++s) {
    const SlotKey k = keys[s];
    if (k.list != li) continue;
    arity = k.arity;
    if (!k.by_types) { slot = s; break; }
    int ty[4] = {-1, -1, -1, -1};
    for (int q = 0; q < k.arity; ++q) {
      if (type_by_tag) ty[q] = type_by_tag[tg[q]];
      else { const int i = rtag[tg[q]]; ty[q] = i >= 0 ? (int)x4[i].w : -1; }
    }
    const int kt[4] = {k.t0, k.t1, k.t2, k.t3};
    bool fwd = true, rev = true;
    for (int q = 0; q < k.arity; ++q) { fwd &= ty[q] == kt[q]; rev &= ty[k.arity - 1 - q] == kt[q]; }
    if (fwd || rev) slot = s;
  }
  eslot[e] = slot;
  if (slot < 0) return;
  const int w = arity == 4 ? 2 : 1;
  for (int q = 0; q < arity; ++q) atomicAdd(&cnt[tg[q]], w);
}
__global__ void k_bt_fill(int ne, const int4* __restrict__ fent, const int* __restrict__ eslot, const SlotKey* __restrict__ keys,
                          const int* __restrict__ start, int* __restrict__ cursor, BondedEntry* __restrict__ bent, int* __restrict__ bkey) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const int slot = eslot[e];
  if (slot < 0) return;
  const int4 tt = fent[e];
  const int arity = keys[slot].arity, w = arity == 4 ? 2 : 1;
  const int tg[4] = {tt.x, tt.y, tt.z, tt.w};
  for (int q = 0; q < arity; ++q) {
    const int pos = start[tg[q]] + atomicAdd(&cursor[tg[q]], w);
    bent[pos] = BondedEntry{tt.x, tt.y, arity > 2 ? tt.z : 0, slot | (q << 28)};
    bkey[pos] = 2 * e;
    if (arity == 4) { bent[pos + 1] = BondedEntry{tt.w, 0, 0, 0}; bkey[pos + 1] = 2 * e + 1; }
  }
}
// rows in flat-array order (key = 2 e + half); counts[0] = tags that own entries
__global__ void k_bt_sort(int n, const int* __restrict__ start, int* __restrict__ cursor, BondedEntry* __restrict__ bent, int* __restrict__ bkey, int* __restrict__ counts) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  cursor[t] = 0;
  const int s = start[t], e = start[t + 1];
  if (e > s) atomicAdd(&counts[0], 1);
  for (int i = s + 1; i < e; ++i) {
    const int kv = bkey[i]; const BondedEntry bv = bent[i];
    int j = i - 1;
    while (j >= s && bkey[j] > kv) { bkey[j + 1] = bkey[j]; bent[j + 1] = bent[j]; --j; }
    bkey[j + 1] = kv; bent[j + 1] = bv;
  }
}

template <typename T> __global__ void k_fill(T* p, T v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace chem
