// md_oracle.cpp -- CPU restatement (fp64, scalar) of the ESPResSo++ algorithms that ChemLab's
// hot path relies on.
//
// *** TEST INFRASTRUCTURE ONLY ***  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library.  The product (chemlab_amd/, libchem_mi355.so)
// never links, imports or falls back to it.
//
// *** PARITY UNPINNED ***  The arithmetic of this path lives in the external, un-pinned
// ESPResSo++ fork cgchemlab/espressopp (reference README.md:7-8), which is not vendored in
// /root/reference and cannot be built or imported here.  The reference's own
// tests pin nothing numerically on this path (src/tests/* only check parsing/replication
// counts).  This file therefore restates the published algorithm from the ChemLab call-site
// contract and the formulas in doc/topology.rst; it is validated by analytic known answers,
// finite differences and conservation laws (tests/test_oracle_*.py), not by reference output.
//
// Citations (paths relative to /root/reference):
//   velocity-Verlet loop / skin trigger ...... src/start_simulation.py:165-167,780 ; SURVEY 3.3
//   Verlet list, exclusions ................... src/start_simulation.py:189-197
//   Lennard-Jones ............................. doc/topology.rst:12-14 ; gromacs_topology.py:715-721
//   Tabulated (itype=1, linear) ............... gromacs_topology.py:696-707 ; tools/convert_gromacs2espp.py:84-107
//   Harmonic / FENE bonds ..................... doc/topology.rst:60 ; gromacs_topology.py:918-931
//   AngularHarmonic / Cosine .................. doc/topology.rst:89-99 ; gromacs_topology.py:1073,1082
//   Dihedrals ................................. doc/topology.rst:101-129 ; gromacs_topology.py:1185-1202
//   Langevin thermostat ....................... src/start_simulation.py:330-336
//   ChemicalReaction / Reaction ............... src/chemlab/reaction_setup.py:71-165,416-427
//   TopologyManager / DynamicExcludeList ...... src/start_simulation.py:189,378-441
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
// -ffp-contract=off keeps r^2 = dx*dx+dy*dy+dz*dz free of FMA so the reaction distance
// test is bit-comparable with the device path.
//
// All-core variant (bench.py's cpu_baseline leg, SURVEY 8d "(ii) all physical cores, OpenMP over
// cells"): the same source built with -O3 -march=native -fopenmp (oracle/Makefile target
// liboracle_omp.so) and option "threads" > 1.  Then the list build runs over cells in parallel
// (pairs kept in cell order, deterministic for any thread count, no global sort), the pair-force
// loop accumulates into per-thread force arrays that are summed afterwards, and the integrator /
// thermostat loops are plain parallel-for.  With threads == 1 (the default; what every parity test
// uses) the code path below is the scalar one, unchanged; option "threads"=1 on the OpenMP build
// runs the threaded code path with one thread (the baseline's 1-core figure: no global pair sort, which
// is an artefact of this checker, not of the algorithm).

#ifdef _OPENMP
#include <omp.h>
#endif
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <ctime>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../include/chem_mi355.h"
#include "../include/chem_philox.h"

namespace {

struct Vec3 { double x, y, z; };
static inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline Vec3 cross(Vec3 a, Vec3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct PairPot {
  int kind = 0;  // 0 none, 1 LJ, 2 table
  double eps = 0, sig = 0, rc = 0, shift = 0;
  // table
  double r0 = 0, dr = 0;
  std::vector<double> e, f;
};

struct BondedList {
  int arity = 2, kind = 0, by_types = 0;
  std::vector<int32_t> ent;             // arity tags per entry
  std::set<std::vector<int32_t>> seen;  // canonical (orientation-free) keys for de-duplication
  bool has_plain = false;
  double plain[CHEM_MAX_POT_PARAMS] = {0};
  std::map<std::vector<int>, std::vector<double>> typed;  // type tuple -> params
  std::vector<std::vector<int>> registered;               // topology-manager type tuples
};

struct Orc {
  std::string err;
  double L[3] = {0, 0, 0};
  double rc = 0, skin = 0, dt = 0;
  int64_t n = 0;
  std::vector<int64_t> id;      // tag -> external id (ascending)
  std::unordered_map<int64_t, int32_t> id2tag;
  std::vector<int32_t> type, state, res_id, mol_id;
  std::vector<int32_t> img;     // 3 per particle
  std::vector<Vec3> x, v, f;
  std::vector<double> mass, q;
  std::vector<std::set<int32_t>> excl;   // per tag
  std::vector<std::set<int32_t>> graph;  // bond graph per tag
  int ntypes = 0;
  PairPot pp[CHEM_MAX_TYPES][CHEM_MAX_TYPES];
  std::vector<BondedList> lists;
  // thermostat
  bool lang = false; double kT = 0, gamma = 0; uint64_t lang_seed = 0; uint32_t lang_tmask = 0;   // thermal groups: start_simulation.py:312-336
  // reactions
  bool react_init = false, react_on = false;
  int interval = 0, nearest = 1, max_per_interval = 0; uint64_t react_seed = 0;
  std::vector<chem_reaction_desc> reactions;
  std::vector<chem_nb_change> nb_rules;   // PostProcessChangeNeighboursProperty
  std::map<std::pair<int32_t, int32_t>, uint32_t> restrict_map;   // RestrictReaction.define_connection: (tag lo, tag hi) -> reaction bits
  uint32_t restricted_mask = 0;
  struct NbCons { int role = 0, nb_type = 0, min_state = 0, max_state = 0; };   // ReactionConstraintNeighbourState, per reaction
  std::vector<NbCons> constraints;
  // integrator.ATRPActivator (reaction_post_process.py:380-426)
  struct AtrpCenter { int type, state, is_activator, new_type, delta_state; double new_mass, new_q; };
  bool atrp_on = false; chem_atrp_desc atrp{}; std::vector<AtrpCenter> atrp_centers; std::vector<chem_atrp_stats> atrp_stats;
  std::vector<chem_event> events;
  // integrator state
  int64_t step = 0;
  bool resort = true;
  double maxdist = 0;
  int criterion = 0;            // 0: accumulated per-step maxima (ESPResSo++), 1: max true displacement since the last build
  std::vector<Vec3> x0;         // positions at the last list build (criterion 1)
  std::vector<std::pair<int32_t, int32_t>> pairs;  // half Verlet list
  int64_t rebuilds = 0, reaction_steps = 0;
  double cap_force = 0;
  int resc_kind = 0; double resc_kT = 0, resc_param = 0;   // Berendsen / Isokinetic (chem_thermostat_rescale), 3 = SVR (chem_thermostat_svr)
  uint64_t svr_seed = 0;
  bool count_intra_inter = false;
  struct BTable { double r0, dr; std::vector<double> e, f; };
  std::vector<BTable> btables;   // chem_table_create registry (tabulated bonds)
  double e_lj = 0, e_tab = 0, virial = 0;
  double e_list[CHEM_MAX_LISTS] = {0};
  int threads = 1; bool par = false;   // par: OpenMP code paths (option "threads", also with 1 thread: the baseline's 1-core leg), see the header
  std::vector<Vec3> fpriv;      // per-thread force arrays of the threaded pair loop
  // threaded mode: the half list lives in per-thread segments (cell order: segment t, then t+1, ...); their
  // capacity is reused by every rebuild.  `pairs` is only materialised when something walks the whole list
  // (reaction step, read-back)
  std::vector<std::vector<std::pair<int32_t, int32_t>>> parts;
  bool pairs_stale = false;
};

static std::string g_err;

static inline double wrap1(double d, double L) { return d - L * std::nearbyint(d / L); }
static inline Vec3 minimg(const Orc& o, Vec3 d) {
  return {wrap1(d.x, o.L[0]), wrap1(d.y, o.L[1]), wrap1(d.z, o.L[2])};
}

// fold positions into [0,L) and keep image counters (storage.decompose())
static void fold(Orc& o) {
  for (int64_t i = 0; i < o.n; ++i) {
    double* p = &o.x[i].x;
    for (int d = 0; d < 3; ++d) {
      double s = std::floor(p[d] / o.L[d]);
      if (s != 0.0) { p[d] -= s * o.L[d]; o.img[3 * i + d] += (int)s; }
      if (p[d] >= o.L[d]) { p[d] -= o.L[d]; o.img[3 * i + d] += 1; }
      if (p[d] < 0) { p[d] += o.L[d]; o.img[3 * i + d] -= 1; }
    }
  }
}

// Half Verlet list: every unordered pair with r^2 <= (rc+skin)^2 that is not excluded.
struct Orc;
static void materialise_pairs(Orc& o);
static double wall_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static const bool g_orc_trace = getenv("ORC_TRACE") != nullptr;

static void build_pairs(Orc& o) {
  const double tb0 = wall_s();
  fold(o);
  o.pairs.clear();
  const double rl = o.rc + o.skin, rl2 = rl * rl;
  int nc[3];
  for (int d = 0; d < 3; ++d) nc[d] = std::max(1, (int)std::floor(o.L[d] / rl));
  auto test = [&](int32_t i, int32_t j) {
    Vec3 d = minimg(o, o.x[i] - o.x[j]);
    double r2 = d.x * d.x + d.y * d.y + d.z * d.z;
    if (r2 > rl2) return;
    int32_t a = std::min(i, j), b = std::max(i, j);
    if (o.excl[a].count(b)) return;
    o.pairs.emplace_back(a, b);
  };
  if (nc[0] < 3 || nc[1] < 3 || nc[2] < 3) {
    for (int32_t i = 0; i < o.n; ++i)
      for (int32_t j = i + 1; j < o.n; ++j) test(i, j);
  } else {
    std::vector<std::vector<int32_t>> cells((size_t)nc[0] * nc[1] * nc[2]);
    auto cid = [&](int cx, int cy, int cz) { return ((size_t)cz * nc[1] + cy) * nc[0] + cx; };
    for (int32_t i = 0; i < o.n; ++i) {
      int c[3];
      const double* p = &o.x[i].x;
      for (int d = 0; d < 3; ++d) {
        c[d] = (int)(p[d] / o.L[d] * nc[d]);
        if (c[d] >= nc[d]) c[d] = nc[d] - 1;
        if (c[d] < 0) c[d] = 0;
      }
      cells[cid(c[0], c[1], c[2])].push_back(i);
    }
    const int ncell = nc[0] * nc[1] * nc[2];
    if (g_orc_trace) fprintf(stderr, "[orc] fold+cells %.3f s\n", wall_s() - tb0);
    // pairs of one home cell (own cell + 13 forward neighbours) appended to `out`
    auto cell_pairs = [&](int c, std::vector<std::pair<int32_t, int32_t>>& out) {
      const int cx = c % nc[0], cy = (c / nc[0]) % nc[1], cz = c / (nc[0] * nc[1]);
      auto test_to = [&](int32_t i, int32_t j) {
        Vec3 d = minimg(o, o.x[i] - o.x[j]);
        double r2 = d.x * d.x + d.y * d.y + d.z * d.z;
        if (r2 > rl2) return;
        int32_t a = std::min(i, j), b = std::max(i, j);
        if (!o.excl[a].empty() && o.excl[a].count(b)) return;
        out.emplace_back(a, b);
      };
      const auto& A = cells[cid(cx, cy, cz)];
      for (size_t a = 0; a < A.size(); ++a)
        for (size_t b = a + 1; b < A.size(); ++b) test_to(A[a], A[b]);
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            if (dz < 0 || (dz == 0 && dy < 0) || (dz == 0 && dy == 0 && dx <= 0)) continue;   // 13 forward neighbours
            int ox = (cx + dx + nc[0]) % nc[0], oy = (cy + dy + nc[1]) % nc[1], oz = (cz + dz + nc[2]) % nc[2];
            const auto& B = cells[cid(ox, oy, oz)];
            for (int32_t i : A)
              for (int32_t j : B) test_to(i, j);
          }
    };
    if (!o.par) {
      for (int c = 0; c < ncell; ++c) cell_pairs(c, o.pairs);
    } else {
#ifdef _OPENMP
      // OpenMP over cells.  Same pair set as above, organised the way a production CPU code would:
      // cell-contiguous coordinate copies, the periodic image chosen per neighbour CELL (a shift vector)
      // instead of per pair, every thread walking a contiguous range of cells and appending to its own
      // vector; the vectors are concatenated in thread (= cell) order, so the list does not depend on
      // the number of threads.
      const int nt = o.threads;
      std::vector<int32_t> cstart(ncell + 1, 0), order(o.n);
      for (int c = 0; c < ncell; ++c) cstart[c + 1] = cstart[c] + (int32_t)cells[c].size();
      std::vector<Vec3> xs(o.n);
#pragma omp parallel for schedule(static) num_threads(nt)
      for (int c = 0; c < ncell; ++c)
        for (size_t k = 0; k < cells[c].size(); ++k) { order[cstart[c] + k] = cells[c][k]; xs[cstart[c] + k] = o.x[cells[c][k]]; }
      if (g_orc_trace) fprintf(stderr, "[orc] xs/order %.3f s\n", wall_s() - tb0);
      auto& part = o.parts;
      part.resize(nt);
      bool any_excl = false;
      for (int64_t i = 0; i < o.n && !any_excl; ++i) any_excl = !o.excl[i].empty();
#pragma omp parallel num_threads(nt)
      {
        const int t = omp_get_thread_num();
        auto& out = part[t];
        out.clear();
        if (out.capacity() == 0) out.reserve((size_t)(40.0 * o.n / nt));
        const int c0 = (int)((int64_t)ncell * t / nt), c1 = (int)((int64_t)ncell * (t + 1) / nt);
        for (int c = c0; c < c1; ++c) {
          const int cx = c % nc[0], cy = (c / nc[0]) % nc[1], cz = c / (nc[0] * nc[1]);
          const int a0 = cstart[c], a1 = cstart[c + 1];
          auto emit = [&](int32_t i, int32_t j) {
            const int32_t a = std::min(i, j), b = std::max(i, j);
            if (any_excl && !o.excl[a].empty() && o.excl[a].count(b)) return;
            out.emplace_back(a, b);
          };
          for (int ia = a0; ia < a1; ++ia)
            for (int ib = ia + 1; ib < a1; ++ib) {
              const Vec3 d = xs[ia] - xs[ib];
              if (d.x * d.x + d.y * d.y + d.z * d.z <= rl2) emit(order[ia], order[ib]);
            }
          for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
              for (int dx = -1; dx <= 1; ++dx) {
                if (dz < 0 || (dz == 0 && dy < 0) || (dz == 0 && dy == 0 && dx <= 0)) continue;   // 13 forward neighbours
                int ox = cx + dx, oy = cy + dy, oz = cz + dz;
                Vec3 sh = {0, 0, 0};   // image of the neighbour cell next to this one
                if (ox < 0) { ox += nc[0]; sh.x = -o.L[0]; } else if (ox >= nc[0]) { ox -= nc[0]; sh.x = o.L[0]; }
                if (oy < 0) { oy += nc[1]; sh.y = -o.L[1]; } else if (oy >= nc[1]) { oy -= nc[1]; sh.y = o.L[1]; }
                if (oz < 0) { oz += nc[2]; sh.z = -o.L[2]; } else if (oz >= nc[2]) { oz -= nc[2]; sh.z = o.L[2]; }
                const int oc = (int)cid(ox, oy, oz);
                for (int ia = a0; ia < a1; ++ia) {
                  const Vec3 xi = xs[ia] - sh;
                  for (int ib = cstart[oc]; ib < cstart[oc + 1]; ++ib) {
                    const Vec3 d = xi - xs[ib];
                    if (d.x * d.x + d.y * d.y + d.z * d.z <= rl2) emit(order[ia], order[ib]);
                  }
                }
              }
        }
      }
      o.pairs_stale = true;
      if (g_orc_trace) fprintf(stderr, "[orc] pairs found %.3f s\n", wall_s() - tb0);
#endif
    }
  }
  if (g_orc_trace) fprintf(stderr, "[orc] list complete %.3f s\n", wall_s() - tb0);
  if (!o.par) std::sort(o.pairs.begin(), o.pairs.end());
  o.x0 = o.x;
  o.maxdist = 0;
  o.resort = false;
  o.rebuilds++;
}

static void materialise_pairs(Orc& o) {
  if (!o.pairs_stale) return;
  size_t tot = 0;
  for (auto& p : o.parts) tot += p.size();
  o.pairs.clear(); o.pairs.reserve(tot);
  for (auto& p : o.parts) o.pairs.insert(o.pairs.end(), p.begin(), p.end());
  o.pairs_stale = false;
}

// ---- potentials ---------------------------------------------------------------------

// returns force factor ff such that F_i = ff * r_ij (r_ij = x_i - x_j); adds energy to *e
static inline bool pair_eval(const PairPot& p, double r2, double* ff, double* e) {
  if (p.kind == 1) {
    if (r2 > p.rc * p.rc) return false;
    double frac2 = 1.0 / r2, s2 = p.sig * p.sig * frac2, s6 = s2 * s2 * s2;
    *ff = 24.0 * p.eps * (2.0 * s6 * s6 - s6) * frac2;
    *e = 4.0 * p.eps * (s6 * s6 - s6) + p.shift;
    return true;
  }
  if (p.kind == 2) {
    if (r2 > p.rc * p.rc) return false;
    double r = std::sqrt(r2);
    double t = (r - p.r0) / p.dr;
    int64_t nrow = (int64_t)p.e.size();
    double fe, ffv;
    if (t <= 0) { fe = p.e[0]; ffv = p.f[0]; }
    else if (t >= (double)(nrow - 1)) { fe = p.e[nrow - 1]; ffv = p.f[nrow - 1]; }
    else {
      int64_t k = (int64_t)t; double w = t - (double)k;
      fe = p.e[k] + w * (p.e[k + 1] - p.e[k]);
      ffv = p.f[k] + w * (p.f[k + 1] - p.f[k]);
    }
    *ff = ffv / r; *e = fe;
    return true;
  }
  return false;
}

static const double* list_params(const Orc& o, const BondedList& l, const int32_t* tags) {
  if (!l.by_types) return l.has_plain ? l.plain : nullptr;
  std::vector<int> key(l.arity), rkey(l.arity);
  for (int k = 0; k < l.arity; ++k) { key[k] = o.type[tags[k]]; rkey[l.arity - 1 - k] = key[k]; }
  auto it = l.typed.find(key);
  if (it == l.typed.end()) it = l.typed.find(rkey);
  if (it == l.typed.end()) return nullptr;
  return it->second.data();
}

static void bonded_forces(Orc& o) {
  for (size_t li = 0; li < o.lists.size(); ++li) {
    BondedList& l = o.lists[li];
    double etot = 0;
    size_t ne = l.ent.size() / l.arity;
    for (size_t e = 0; e < ne; ++e) {
      const int32_t* t = &l.ent[e * l.arity];
      const double* p = list_params(o, l, t);
      if (!p) continue;
      if (l.arity == 2) {
        Vec3 d = minimg(o, o.x[t[0]] - o.x[t[1]]);
        double r = std::sqrt(dot(d, d)), ff = 0, u = 0;
        if (l.kind == CHEM_POT_HARMONIC) {
          double dr = r - p[1]; u = p[0] * dr * dr; ff = -2.0 * p[0] * dr / r;
        } else if (l.kind == CHEM_POT_FENE) {
          double dr = r - p[1], q = dr / p[2], den = 1.0 - q * q;
          u = -0.5 * p[0] * p[2] * p[2] * std::log(den);
          ff = -p[0] * dr / den / r;
        } else if (l.kind == CHEM_POT_FENE_LJ) {   // FENELennardJones(K, r0, rMax, sigma, epsilon), doc/topology.rst:68-75
          double dr = r - p[1], q = dr / p[2], den = 1.0 - q * q;
          double s2 = p[3] * p[3] / (r * r), s6 = s2 * s2 * s2;
          u = -0.5 * p[0] * p[2] * p[2] * std::log(den) + 4.0 * p[4] * (s6 * s6 - s6);
          ff = -p[0] * dr / den / r + 24.0 * p[4] * (2.0 * s6 * s6 - s6) / (r * r);
        } else if (l.kind == CHEM_POT_LJ_BOND) {   // FixedPairListLennardJones(epsilon, sigma, cutoff): 1-4 pairs
          if (r <= p[2]) {
            double s2 = p[1] * p[1] / (r * r), s6 = s2 * s2 * s2;
            // shift = 'auto' (espressopp's LennardJones default): U(rc) = 0
            double c2 = p[1] * p[1] / (p[2] * p[2]), c6 = p[2] < 1e29 ? c2 * c2 * c2 : 0.0;
            u = 4.0 * p[0] * ((s6 * s6 - s6) - (c6 * c6 - c6));
            ff = 24.0 * p[0] * (2.0 * s6 * s6 - s6) / (r * r);
          }
        } else if (l.kind == CHEM_POT_TABULATED) {   // Tabulated(itype=1): linear interpolation, gromacs_topology.py:919-925
          const Orc::BTable& tb = o.btables[(size_t)p[0]];
          const int64_t nrow = (int64_t)tb.e.size();
          const double t = (r - tb.r0) / tb.dr;
          double fv;
          if (t <= 0) { u = tb.e[0]; fv = tb.f[0]; }
          else if (t >= (double)(nrow - 1)) { u = tb.e[nrow - 1]; fv = tb.f[nrow - 1]; }
          else { const int64_t k = (int64_t)t; const double w = t - (double)k; u = tb.e[k] + w * (tb.e[k + 1] - tb.e[k]); fv = tb.f[k] + w * (tb.f[k + 1] - tb.f[k]); }
          ff = fv / r;
        }
        o.f[t[0]] = o.f[t[0]] + ff * d; o.f[t[1]] = o.f[t[1]] - ff * d; etot += u;
      } else if (l.arity == 3) {
        Vec3 r1 = minimg(o, o.x[t[0]] - o.x[t[1]]), r2 = minimg(o, o.x[t[2]] - o.x[t[1]]);
        double n1 = std::sqrt(dot(r1, r1)), n2 = std::sqrt(dot(r2, r2));
        double c = dot(r1, r2) / (n1 * n2);
        c = std::max(-1.0, std::min(1.0, c));
        double th = std::acos(c), s = std::sqrt(1.0 - c * c);
        if (s < 1e-9) s = 1e-9;
        double u = 0, dU = 0;
        if (l.kind == CHEM_POT_ANG_HARMONIC) { double d = th - p[1]; u = p[0] * d * d; dU = 2.0 * p[0] * d; }
        else if (l.kind == CHEM_POT_ANG_COSINE) { u = p[0] * (1.0 + std::cos(th - p[1])); dU = -p[0] * std::sin(th - p[1]); }
        else if (l.kind == CHEM_POT_ANG_TABULATED) {   // TabulatedAngular(itype=1): table of U(theta), -dU/dtheta, gromacs_topology.py:1074-1080
          const Orc::BTable& tb = o.btables[(size_t)p[0]];
          const int64_t nrow = (int64_t)tb.e.size();
          const double t = (th - tb.r0) / tb.dr;
          double fv;
          if (t <= 0) { u = tb.e[0]; fv = tb.f[0]; }
          else if (t >= (double)(nrow - 1)) { u = tb.e[nrow - 1]; fv = tb.f[nrow - 1]; }
          else { const int64_t k = (int64_t)t; const double w = t - (double)k; u = tb.e[k] + w * (tb.e[k + 1] - tb.e[k]); fv = tb.f[k] + w * (tb.f[k + 1] - tb.f[k]); }
          dU = -fv;
        }
        double a = dU / s;
        Vec3 fi = a * ((1.0 / (n1 * n2)) * r2 - (c / (n1 * n1)) * r1);
        Vec3 fk = a * ((1.0 / (n1 * n2)) * r1 - (c / (n2 * n2)) * r2);
        o.f[t[0]] = o.f[t[0]] + fi; o.f[t[2]] = o.f[t[2]] + fk;
        o.f[t[1]] = o.f[t[1]] - (fi + fk); etot += u;
      } else {
        Vec3 b1 = minimg(o, o.x[t[1]] - o.x[t[0]]), b2 = minimg(o, o.x[t[2]] - o.x[t[1]]),
             b3 = minimg(o, o.x[t[3]] - o.x[t[2]]);
        Vec3 m = cross(b1, b2), nn = cross(b2, b3);
        double m2 = dot(m, m), n2 = dot(nn, nn), lb2 = dot(b2, b2), lb = std::sqrt(lb2);
        if (m2 < 1e-30 || n2 < 1e-30) continue;
        double phi = std::atan2(lb * dot(b1, nn), dot(m, nn));
        double u = 0, dU = 0;
        if (l.kind == CHEM_POT_DIH_NCOS) {
          u = p[0] * (1.0 + std::cos(p[2] * phi - p[1])); dU = -p[0] * p[2] * std::sin(p[2] * phi - p[1]);
        } else if (l.kind == CHEM_POT_DIH_RB) {
          double psi = phi - M_PI, cp = std::cos(psi), sp = std::sin(psi), pw = 1.0, dsum = 0;
          for (int k = 0; k < 6; ++k) { u += p[k] * pw; if (k < 5) { dsum += (k + 1) * p[k + 1] * pw; } pw *= cp; }
          dU = -sp * dsum;
        } else if (l.kind == CHEM_POT_DIH_HARMONIC) {   // DihedralHarmonic(K, phi0): U = K/2 (phi - phi0)^2, doc/topology.rst:120-128
          double d = phi - p[1];
          d -= 2.0 * M_PI * std::nearbyint(d / (2.0 * M_PI));
          u = 0.5 * p[0] * d * d; dU = p[0] * d;
        } else if (l.kind == CHEM_POT_DIH_TABULATED) {   // TabulatedDihedral(itype=1): U(phi), -dU/dphi, gromacs_topology.py:1192-1198
          const Orc::BTable& tb = o.btables[(size_t)p[0]];
          const int64_t nrow = (int64_t)tb.e.size();
          const double tt = (phi - tb.r0) / tb.dr;
          double fv;
          if (tt <= 0) { u = tb.e[0]; fv = tb.f[0]; }
          else if (tt >= (double)(nrow - 1)) { u = tb.e[nrow - 1]; fv = tb.f[nrow - 1]; }
          else { const int64_t k = (int64_t)tt; const double w = tt - (double)k; u = tb.e[k] + w * (tb.e[k + 1] - tb.e[k]); fv = tb.f[k] + w * (tb.f[k + 1] - tb.f[k]); }
          dU = -fv;
        }
        // dphi/dx (Blondel & Karplus)
        Vec3 g1 = (-lb / m2) * m, g4 = (lb / n2) * nn;
        double s12 = dot(b1, b2) / lb2, s32 = dot(b3, b2) / lb2;
        Vec3 g2 = {(-1.0 - s12) * g1.x + s32 * g4.x, (-1.0 - s12) * g1.y + s32 * g4.y, (-1.0 - s12) * g1.z + s32 * g4.z};
        Vec3 g3 = {(-1.0 - s32) * g4.x + s12 * g1.x, (-1.0 - s32) * g4.y + s12 * g1.y, (-1.0 - s32) * g4.z + s12 * g1.z};
        o.f[t[0]] = o.f[t[0]] - dU * g1; o.f[t[1]] = o.f[t[1]] - dU * g2;
        o.f[t[2]] = o.f[t[2]] - dU * g3; o.f[t[3]] = o.f[t[3]] - dU * g4; etot += u;
      }
    }
    o.e_list[li] = etot;
  }
}

// phase: 0 = evaluation at run() start, 1 = in-loop evaluation of step index `istep`
static void update_forces(Orc& o, int64_t istep, int phase) {
  for (auto& f : o.f) f = {0, 0, 0};
  o.e_lj = o.e_tab = o.virial = 0;
  if (!o.par) {
    for (auto& pr : o.pairs) {
      int32_t i = pr.first, j = pr.second;
      const PairPot& p = o.pp[o.type[i]][o.type[j]];
      if (!p.kind) continue;
      Vec3 d = minimg(o, o.x[i] - o.x[j]);
      double r2 = d.x * d.x + d.y * d.y + d.z * d.z, ff, e;
      if (!pair_eval(p, r2, &ff, &e)) continue;
      o.f[i] = o.f[i] + ff * d; o.f[j] = o.f[j] - ff * d;
      (p.kind == 1 ? o.e_lj : o.e_tab) += e;
      o.virial += ff * r2;
    }
  } else {
#ifdef _OPENMP
    // threaded pair loop: contiguous chunks of the (cell-ordered) half list per thread, Newton's third
    // law into a per-thread force array, arrays summed in thread order afterwards
    const int nt = o.threads;
    const int64_t n = o.n;
    o.fpriv.resize((size_t)nt * n);
    double elj = 0, etab = 0, vir = 0;
#pragma omp parallel num_threads(nt) reduction(+ : elj, etab, vir)
    {
      const int t = omp_get_thread_num();
      Vec3* f = o.fpriv.data() + (size_t)t * n;
      for (int64_t i = 0; i < n; ++i) f[i] = {0, 0, 0};
      const auto& seg = o.parts[t];   // the segment this thread built (same thread count)
      const int64_t hi = (int64_t)seg.size();
      for (int64_t k = 0; k < hi; ++k) {
        const int32_t i = seg[k].first, j = seg[k].second;
        const PairPot& p = o.pp[o.type[i]][o.type[j]];
        if (!p.kind) continue;
        Vec3 d = minimg(o, o.x[i] - o.x[j]);
        double r2 = d.x * d.x + d.y * d.y + d.z * d.z, ff, e;
        if (!pair_eval(p, r2, &ff, &e)) continue;
        f[i] = f[i] + ff * d; f[j] = f[j] - ff * d;
        if (p.kind == 1) elj += e; else etab += e;
        vir += ff * r2;
      }
#pragma omp barrier
#pragma omp for schedule(static)
      for (int64_t i = 0; i < n; ++i) {
        Vec3 a = {0, 0, 0};
        for (int q = 0; q < nt; ++q) a = a + o.fpriv[(size_t)q * n + i];
        o.f[i] = a;
      }
    }
    o.e_lj = elj; o.e_tab = etab; o.virial = vir;
#endif
  }
  bonded_forces(o);
  if (o.cap_force > 0) {   // integrator.CapForce, connected before the thermostat (start_simulation.py:320-324)
    for (int64_t i = 0; i < o.n; ++i) {
      const double f2 = o.f[i].x * o.f[i].x + o.f[i].y * o.f[i].y + o.f[i].z * o.f[i].z;
      if (f2 > o.cap_force * o.cap_force) { const double s = o.cap_force / std::sqrt(f2); o.f[i] = s * o.f[i]; }
    }
  }
  if (o.lang) {
#pragma omp parallel for schedule(static) num_threads(o.threads) if (o.threads > 1)
    for (int64_t i = 0; i < o.n; ++i) {
      if (o.lang_tmask && !((o.lang_tmask >> (o.type[i] & 31)) & 1u)) continue;   // add_valid_types: only the listed (current) types
      uint32_t r[4];
      chem_philox::langevin_draw(o.lang_seed, (uint64_t)istep, (uint32_t)phase, (uint32_t)i, r);
      double m = o.mass[i], pref = std::sqrt(24.0 * o.kT * o.gamma * m / o.dt);
      double* f = &o.f[i].x; const double* v = &o.v[i].x;
      for (int d = 0; d < 3; ++d) f[d] += -o.gamma * m * v[d] + pref * (chem_philox::u01(r[d]) - 0.5);
    }
  }
}

// ---- topology manager ---------------------------------------------------------------

static std::vector<int32_t> canon_key(const int32_t* t, int arity) {
  std::vector<int32_t> a(t, t + arity), b(arity);
  for (int k = 0; k < arity; ++k) b[arity - 1 - k] = a[k];
  return std::min(a, b);
}

static bool list_insert(BondedList& l, const int32_t* t) {
  auto key = canon_key(t, l.arity);
  if (!l.seen.insert(key).second) return false;
  l.ent.insert(l.ent.end(), t, t + l.arity);
  return true;
}

static void exclude(Orc& o, int32_t a, int32_t b) {
  if (a == b) return;
  o.excl[std::min(a, b)].insert(std::max(a, b));
}

// relabel the bonded cluster containing `start` with the minimum res_id / tag it holds
static void merge_cluster(Orc& o, int32_t a, int32_t b) {
  int32_t new_res = std::min(o.res_id[a], o.res_id[b]);
  int32_t new_mol = std::min(o.mol_id[a], o.mol_id[b]);
  // flood fill from a over the bond graph (a-b is already in the graph)
  std::vector<int32_t> stack{a};
  std::set<int32_t> vis{a};
  while (!stack.empty()) {
    int32_t p = stack.back(); stack.pop_back();
    o.res_id[p] = new_res; o.mol_id[p] = new_mol;
    for (int32_t nb : o.graph[p]) if (vis.insert(nb).second) stack.push_back(nb);
  }
}

// try to place (t[0..arity)) into the first list of that arity whose registered type tuples match
static void spawn_tuple(Orc& o, int arity, const int32_t* t) {
  for (auto& l : o.lists) {
    if (l.arity != arity) continue;
    for (auto& reg : l.registered) {
      bool fwd = true, rev = true;
      for (int k = 0; k < arity; ++k) {
        if (o.type[t[k]] != reg[k]) fwd = false;
        if (o.type[t[arity - 1 - k]] != reg[k]) rev = false;
      }
      if (!fwd && !rev) continue;
      int32_t tt[4];
      for (int k = 0; k < arity; ++k) tt[k] = fwd ? t[k] : t[arity - 1 - k];
      if (list_insert(l, tt)) exclude(o, tt[0], tt[arity - 1]);
      return;
    }
  }
}

static void on_new_bonds(Orc& o, const std::vector<std::pair<int32_t, int32_t>>& nb) {
  for (auto& e : nb) { o.graph[e.first].insert(e.second); o.graph[e.second].insert(e.first); }
  for (auto& e : nb) merge_cluster(o, e.first, e.second);
  for (auto& e : nb) {
    int32_t a = e.first, b = e.second;
    exclude(o, a, b);
    // angles (n,a,b), (a,b,m)
    for (int32_t n : o.graph[a]) if (n != b) { int32_t t[3] = {n, a, b}; spawn_tuple(o, 3, t); }
    for (int32_t m : o.graph[b]) if (m != a) { int32_t t[3] = {a, b, m}; spawn_tuple(o, 3, t); }
    // dihedrals (n',n,a,b), (n,a,b,m), (a,b,m,m')
    for (int32_t n : o.graph[a]) if (n != b) {
      for (int32_t n2 : o.graph[n]) if (n2 != a && n2 != b) { int32_t t[4] = {n2, n, a, b}; spawn_tuple(o, 4, t); }
      for (int32_t m : o.graph[b]) if (m != a && m != n) { int32_t t[4] = {n, a, b, m}; spawn_tuple(o, 4, t); }
    }
    for (int32_t m : o.graph[b]) if (m != a)
      for (int32_t m2 : o.graph[m]) if (m2 != b && m2 != a) { int32_t t[4] = {a, b, m, m2}; spawn_tuple(o, 4, t); }
  }
}

// ---- reactions ----------------------------------------------------------------------

struct Cand { int32_t a, b, r; double d2; uint32_t h; };

static void react(Orc& o) {
  materialise_pairs(o);
  o.reaction_steps++;
  std::vector<Cand> c;
  for (auto& pr : o.pairs) {
    int32_t lo = pr.first, hi = pr.second;
    for (size_t ri = 0; ri < o.reactions.size(); ++ri) {
      const chem_reaction_desc& R = o.reactions[ri];
      if (!R.active) continue;
      auto ok = [&](int32_t a, int32_t b) {
        return o.type[a] == R.type_1 && o.state[a] >= R.min_state_1 && o.state[a] < R.max_state_1 &&
               o.type[b] == R.type_2 && o.state[b] >= R.min_state_2 && o.state[b] < R.max_state_2;
      };
      int32_t a, b;
      if (ok(lo, hi)) { a = lo; b = hi; } else if (ok(hi, lo)) { a = hi; b = lo; } else continue;
      if (!R.intraresidual && o.res_id[a] == o.res_id[b]) continue;
      if (!R.intramolecular && o.mol_id[a] == o.mol_id[b]) continue;
      Vec3 d = minimg(o, o.x[a] - o.x[b]);
      double d2 = d.x * d.x + d.y * d.y + d.z * d.z;
      if (!(d2 >= R.min_cutoff * R.min_cutoff && d2 < R.cutoff * R.cutoff)) continue;
      if (ri < o.constraints.size() && o.constraints[ri].role) {   // add_constraint(ReactionConstraintNeighbourState) (reaction_setup.py:203-204)
        const Orc::NbCons& cs = o.constraints[ri];
        const int32_t who = cs.role == 1 ? a : b;
        bool ok_ = false;
        for (int32_t nb : o.graph[who]) ok_ |= o.type[nb] == cs.nb_type && o.state[nb] >= cs.min_state && o.state[nb] < cs.max_state;
        if (!ok_) continue;
      }
      if ((o.restricted_mask >> ri) & 1u) {   // RestrictReaction: only the connections of the map (reaction_setup.py:115-128)
        auto it = o.restrict_map.find({lo, hi});
        if (it == o.restrict_map.end() || !((it->second >> ri) & 1u)) continue;
      }
      double prob = R.rate * o.dt * (double)o.interval;
      uint32_t rr[4];
      chem_philox::reaction_draw(o.react_seed, (uint64_t)o.step, (uint32_t)lo, (uint32_t)hi, (uint32_t)ri, rr);
      if (prob < 1.0 && !(chem_philox::u01(rr[0]) < prob)) continue;
      c.push_back({a, b, (int32_t)ri, d2, rr[1]});
    }
  }
  const bool nearest = o.nearest != 0;
  // UniqueA: every A keeps one partner (nearest, or pseudo-random by hash)
  auto keyA = [&](const Cand& q) { return nearest ? std::make_tuple(q.d2, 0u, q.b, q.r) : std::make_tuple(0.0, q.h, q.b, q.r); };
  auto keyB = [&](const Cand& q) { return nearest ? std::make_tuple(q.d2, 0u, q.a, q.r) : std::make_tuple(0.0, q.h, q.a, q.r); };
  {
    std::map<int32_t, Cand> best;
    for (auto& q : c) { auto it = best.find(q.a); if (it == best.end() || keyA(q) < keyA(it->second)) best[q.a] = q; }
    c.clear(); for (auto& kv : best) c.push_back(kv.second);
  }
  {  // UniqueB
    std::map<int32_t, Cand> best;
    for (auto& q : c) { auto it = best.find(q.b); if (it == best.end() || keyB(q) < keyB(it->second)) best[q.b] = q; }
    c.clear(); for (auto& kv : best) c.push_back(kv.second);
  }
  // one event per particle: greedy over (d2|hash, a) order
  std::sort(c.begin(), c.end(), [&](const Cand& p, const Cand& q) {
    return nearest ? std::make_tuple(p.d2, p.a) < std::make_tuple(q.d2, q.a)
                   : std::make_tuple(p.h, p.a) < std::make_tuple(q.h, q.a);
  });
  std::vector<char> used(o.n, 0);
  std::vector<Cand> acc;
  for (auto& q : c) { if (used[q.a] || used[q.b]) continue; used[q.a] = used[q.b] = 1; acc.push_back(q); }
  // ChemicalReaction.max_per_interval (reaction_setup.py:426-427): at most that many events per reaction step.
  // `acc` is in priority order here (nearest mode: r^2, then A's tag; random mode: pair hash, then A's tag): the
  // first max_per_interval survive [EXT-RECALL: ESPResSo++ keeps a random subset; this is its deterministic counterpart]
  if (o.max_per_interval > 0 && (int64_t)acc.size() > o.max_per_interval) acc.resize(o.max_per_interval);
  std::sort(acc.begin(), acc.end(), [](const Cand& p, const Cand& q) {
    return std::make_pair(std::min(p.a, p.b), std::max(p.a, p.b)) < std::make_pair(std::min(q.a, q.b), std::max(q.a, q.b));
  });
  // apply
  std::vector<std::pair<int32_t, int32_t>> newbonds;
  std::vector<int> bond_list_of;
  for (auto& q : acc) {
    const chem_reaction_desc& R = o.reactions[q.r];
    o.state[q.a] += R.delta_1; o.state[q.b] += R.delta_2;
    if (R.new_type_1 >= 0 && R.new_type_1 != o.type[q.a]) { o.type[q.a] = R.new_type_1; o.mass[q.a] = R.new_mass_1; o.q[q.a] = R.new_q_1; }
    if (R.new_type_2 >= 0 && R.new_type_2 != o.type[q.b]) { o.type[q.b] = R.new_type_2; o.mass[q.b] = R.new_mass_2; o.q[q.b] = R.new_q_2; }
    o.events.push_back({o.step, o.id[q.a], o.id[q.b], q.r, o.count_intra_inter && o.mol_id[q.a] == o.mol_id[q.b] ? 1 : 0, q.d2});   // pad: intra-cluster flag (option count_intra_inter)
    if (!R.is_virtual) {
      int32_t t[2] = {q.a, q.b};
      if (list_insert(o.lists[R.bond_list], t)) newbonds.emplace_back(q.a, q.b);
    }
  }
  if (!newbonds.empty()) { on_new_bonds(o, newbonds); o.resort = true; }
  // PostProcessChangeNeighboursProperty (reaction_post_process.py:76-115): `acc` is in canonical order
  if (!o.nb_rules.empty()) {
    std::vector<int32_t> frontier, next;
    std::set<int32_t> seen;
    for (auto& q : acc)
      for (int role = 1; role <= 2; ++role)
        for (auto& rl : o.nb_rules) {
          if (rl.reaction != q.r || !(rl.invoke_on & role)) continue;
          const int32_t root = role == 1 ? q.a : q.b;
          frontier.assign(1, root); seen.clear(); seen.insert(root);
          for (int lvl = 0; lvl < rl.nb_level; ++lvl) {   // breadth-first shells of the bond graph
            next.clear();
            for (int32_t p : frontier) for (int32_t nb : o.graph[p]) if (seen.insert(nb).second) next.push_back(nb);
            frontier.swap(next);
          }
          std::sort(frontier.begin(), frontier.end());
          for (int32_t p : frontier) {
            if (o.type[p] != rl.old_type) continue;
            if (rl.min_state < rl.max_state && !(o.state[p] >= rl.min_state && o.state[p] < rl.max_state)) continue;   // set_min_max_state
            o.type[p] = rl.new_type; o.mass[p] = rl.new_mass; o.q[p] = rl.new_q;
            if (rl.set_state == 1) o.state[p] = rl.new_state; else if (rl.set_state == 2) o.state[p] += rl.new_state;   // incr_state
          }
        }
  }
}

// ---- ATRPActivator (reaction_post_process.py:380-426; rule set in include/chem_mi355.h) ----
static void atrp_step(Orc& o) {
  struct Sel { uint32_t key; int32_t tag; uint32_t u; int center; };
  auto center_of = [&](int32_t t) {
    for (size_t c = 0; c < o.atrp_centers.size(); ++c) if (o.atrp_centers[c].type == o.type[t] && o.atrp_centers[c].state == o.state[t]) return (int)c;
    return -1;
  };
  std::vector<Sel> pool;
  int64_t ncand = 0;
  for (int32_t t = 0; t < (int32_t)o.n; ++t) {
    const int c = center_of(t);
    if (c >= 0) ++ncand;
    if (c < 0 && !o.atrp.select_from_all) continue;
    uint32_t r[4];
    chem_philox::atrp_draw(o.atrp.seed, (uint64_t)o.step, (uint32_t)t, r);
    pool.push_back({r[0], t, r[1], c});
  }
  std::sort(pool.begin(), pool.end(), [](const Sel& a, const Sel& b) { return std::make_pair(a.key, a.tag) < std::make_pair(b.key, b.tag); });
  if ((int64_t)pool.size() > o.atrp.num_particles) pool.resize((size_t)o.atrp.num_particles);
  const double dc = o.atrp.delta_catalyst / (double)o.atrp.num_particles;
  chem_atrp_stats st{}; st.step = o.step; st.candidates = ncand; st.selected = (int64_t)pool.size();
  bool changed = false;
  for (auto& s : pool) {
    if (s.center < 0) continue;
    const Orc::AtrpCenter& c = o.atrp_centers[s.center];
    const double p = c.is_activator ? o.atrp.k_deactivate * o.atrp.ratio_deactivator : o.atrp.k_activate * o.atrp.ratio_activator;
    if (!(chem_philox::u01(s.u) < p)) continue;
    if (c.new_type >= 0 && c.new_type != o.type[s.tag]) { o.type[s.tag] = c.new_type; o.mass[s.tag] = c.new_mass; o.q[s.tag] = c.new_q; }
    o.state[s.tag] += c.delta_state;
    if (c.is_activator) { const double m = std::min(dc, o.atrp.ratio_deactivator); o.atrp.ratio_deactivator -= m; o.atrp.ratio_activator += m; st.deactivated++; }
    else { const double m = std::min(dc, o.atrp.ratio_activator); o.atrp.ratio_activator -= m; o.atrp.ratio_deactivator += m; st.activated++; }
    changed = true;
  }
  st.ratio_activator = o.atrp.ratio_activator; st.ratio_deactivator = o.atrp.ratio_deactivator;
  o.atrp_stats.push_back(st);
  if (changed) o.resort = true;
}

// ---- integrator ---------------------------------------------------------------------

static void run(Orc& o, int64_t nsteps) {
  if (o.resort) build_pairs(o);
  update_forces(o, o.step, 0);
  for (int64_t s = 0; s < nsteps; ++s) {
    double max2 = 0;
#pragma omp parallel for schedule(static) num_threads(o.threads) reduction(max : max2) if (o.threads > 1)
    for (int64_t i = 0; i < o.n; ++i) {
      double hm = 0.5 * o.dt / o.mass[i];
      o.v[i] = o.v[i] + hm * o.f[i];
      Vec3 dx = o.dt * o.v[i];
      o.x[i] = o.x[i] + dx;
      if (o.criterion == 1) dx = o.x[i] - o.x0[i];
      double d2 = dx.x * dx.x + dx.y * dx.y + dx.z * dx.z;
      if (d2 > max2) max2 = d2;
    }
    if (o.criterion == 1) o.maxdist = std::sqrt(max2); else o.maxdist += std::sqrt(max2);
    if (o.maxdist > 0.5 * o.skin || o.resort) build_pairs(o);
    update_forces(o, o.step, 1);
#pragma omp parallel for schedule(static) num_threads(o.threads) if (o.threads > 1)
    for (int64_t i = 0; i < o.n; ++i) o.v[i] = o.v[i] + (0.5 * o.dt / o.mass[i]) * o.f[i];
    o.step++;
    if (o.resc_kind == 1 || o.resc_kind == 3 || (o.resc_kind == 2 && o.step % (int64_t)o.resc_param == 0)) {   // aftIntV, start_simulation.py:337-348
      double ek = 0;
      for (int64_t i = 0; i < o.n; ++i) ek += 0.5 * o.mass[i] * dot(o.v[i], o.v[i]);
      const double kTnow = 2.0 * ek / (3.0 * (double)o.n);
      const double lam = o.resc_kind == 1 ? std::sqrt(1.0 + o.dt / o.resc_param * (o.resc_kT / kTnow - 1.0))
                       : o.resc_kind == 2 ? std::sqrt(o.resc_kT / kTnow)
                       // StochasticVelocityRescaling: 3N degrees of freedom, K_ref = 3N kT / 2, taut = coupling / dt
                       : chem_philox::svr_lambda(o.svr_seed, (uint64_t)o.step, ek, 1.5 * (double)o.n * o.resc_kT, 3 * o.n, o.resc_param / o.dt);
      for (int64_t i = 0; i < o.n; ++i) o.v[i] = lam * o.v[i];
    }
    if (o.react_on && o.interval > 0 && o.step % o.interval == 0) react(o);
    if (o.atrp_on && o.step % o.atrp.interval == 0) atrp_step(o);   // added to the integrator behind `ar` (start_simulation.py:737-740)
  }
}

}  // namespace

// =====================================================================================
// C API (mirrors include/chem_mi355.h with the orc_ prefix)
// =====================================================================================
#define O(ctx) (*reinterpret_cast<Orc*>(ctx))
#define FAIL(code, msg) do { o.err = (msg); return (code); } while (0)

extern "C" {

void* orc_create() { return new Orc(); }
void orc_destroy(void* c) { delete reinterpret_cast<Orc*>(c); }
const char* orc_last_error(void* c) { return c ? O(c).err.c_str() : g_err.c_str(); }

int orc_set_box(void* c, const double L[3]) { Orc& o = O(c); for (int d = 0; d < 3; ++d) { if (!(L[d] > 0)) FAIL(CHEM_EINVAL, "box"); o.L[d] = L[d]; } return 0; }
int orc_set_cutoff(void* c, double rc, double skin) { Orc& o = O(c); if (!(rc > 0) || skin < 0) FAIL(CHEM_EINVAL, "cutoff"); o.rc = rc; o.skin = skin; o.resort = true; return 0; }
int orc_set_dt(void* c, double dt) { O(c).dt = dt; return 0; }

int orc_set_particles(void* c, int64_t n, const int64_t* id, const int32_t* type, const double* pos,
                      const double* vel, const double* mass, const double* q, const int32_t* state,
                      const int32_t* res_id) {
  Orc& o = O(c);
  if (n <= 0 || !id || !type || !pos || !mass) FAIL(CHEM_EINVAL, "set_particles: null/empty");
  std::vector<int64_t> order(n);
  for (int64_t i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return id[a] < id[b]; });
  o.n = n; o.id.resize(n); o.type.resize(n); o.state.resize(n); o.res_id.resize(n); o.mol_id.resize(n);
  o.x.resize(n); o.v.resize(n); o.f.assign(n, {0, 0, 0}); o.mass.resize(n); o.q.resize(n);
  o.img.assign(3 * n, 0); o.excl.assign(n, {}); o.graph.assign(n, {}); o.id2tag.clear();
  for (int64_t t = 0; t < n; ++t) {
    int64_t s = order[t];
    if (t && id[s] == o.id[t - 1]) FAIL(CHEM_EINVAL, "duplicate particle id");
    if (type[s] < 0 || type[s] >= CHEM_MAX_TYPES) FAIL(CHEM_EINVAL, "type out of range");
    o.id[t] = id[s]; o.id2tag[id[s]] = (int32_t)t; o.type[t] = type[s];
    o.x[t] = {pos[3 * s], pos[3 * s + 1], pos[3 * s + 2]};
    o.v[t] = vel ? Vec3{vel[3 * s], vel[3 * s + 1], vel[3 * s + 2]} : Vec3{0, 0, 0};
    o.mass[t] = mass[s]; o.q[t] = q ? q[s] : 0.0;
    o.state[t] = state ? state[s] : 0; o.res_id[t] = res_id ? res_id[s] : (int32_t)id[s];
    o.mol_id[t] = (int32_t)t;
  }
  o.resort = true;
  return 0;
}

static int tag_of(Orc& o, int64_t id) { auto it = o.id2tag.find(id); return it == o.id2tag.end() ? -1 : it->second; }

int orc_modify_particle(void* c, int64_t id, int what, double value) {
  Orc& o = O(c); int t = tag_of(o, id); if (t < 0) FAIL(CHEM_EINVAL, "unknown id");
  if (what == CHEM_STATE_TYPE) o.type[t] = (int)value; else if (what == CHEM_STATE_STATE) o.state[t] = (int)value;
  else if (what == CHEM_STATE_MASS) o.mass[t] = value; else if (what == CHEM_STATE_RESID) o.res_id[t] = (int)value;
  else FAIL(CHEM_EINVAL, "modify: what"); return 0;
}

int orc_set_exclusions(void* c, int64_t n, const int64_t* p) {
  Orc& o = O(c);
  for (auto& s : o.excl) s.clear();
  for (int64_t k = 0; k < n; ++k) {
    int a = tag_of(o, p[2 * k]), b = tag_of(o, p[2 * k + 1]);
    if (a < 0 || b < 0) FAIL(CHEM_EINVAL, "exclusion: unknown id");
    exclude(o, a, b);
  }
  o.resort = true; return 0;
}

int orc_nb_lj(void* c, int t1, int t2, double eps, double sig, double rc, int shift_auto) {
  Orc& o = O(c);
  if (t1 < 0 || t2 < 0 || t1 >= CHEM_MAX_TYPES || t2 >= CHEM_MAX_TYPES) FAIL(CHEM_EINVAL, "type");
  PairPot p; p.kind = (sig > 0 && rc > 0) ? 1 : 0; p.eps = eps; p.sig = sig; p.rc = rc;
  if (p.kind && shift_auto) { double s2 = sig * sig / (rc * rc), s6 = s2 * s2 * s2; p.shift = -4.0 * eps * (s6 * s6 - s6); }
  o.pp[t1][t2] = p; o.pp[t2][t1] = p; return 0;
}

int orc_nb_table(void* c, int t1, int t2, int64_t nrow, double r0, double dr, const double* e, const double* f, double rc) {
  Orc& o = O(c);
  if (t1 < 0 || t2 < 0 || t1 >= CHEM_MAX_TYPES || t2 >= CHEM_MAX_TYPES || nrow < 2 || !(dr > 0)) FAIL(CHEM_EINVAL, "table");
  PairPot p; p.kind = 2; p.rc = rc; p.r0 = r0; p.dr = dr; p.e.assign(e, e + nrow); p.f.assign(f, f + nrow);
  o.pp[t1][t2] = p; o.pp[t2][t1] = p; return 0;
}

int orc_list_create(void* c, int arity, int kind, int by_types) {
  Orc& o = O(c);
  if (arity < 2 || arity > 4 || (int)o.lists.size() >= CHEM_MAX_LISTS) FAIL(CHEM_EINVAL, "list_create");
  BondedList l; l.arity = arity; l.kind = kind; l.by_types = by_types; o.lists.push_back(l);
  return (int)o.lists.size() - 1;
}

int orc_list_add(void* c, int list, int64_t n, const int64_t* ids) {
  Orc& o = O(c);
  if (list < 0 || list >= (int)o.lists.size()) FAIL(CHEM_EINVAL, "list handle");
  BondedList& l = o.lists[list];
  std::vector<std::pair<int32_t, int32_t>> nb;
  for (int64_t e = 0; e < n; ++e) {
    int32_t t[4];
    for (int k = 0; k < l.arity; ++k) { int tg = tag_of(o, ids[e * l.arity + k]); if (tg < 0) FAIL(CHEM_EINVAL, "list_add: unknown id"); t[k] = tg; }
    if (list_insert(l, t) && l.arity == 2) nb.emplace_back(t[0], t[1]);
  }
  // bonds feed the topology graph (TopologyManager.observe_tuple + initialize_topology)
  for (auto& e : nb) { o.graph[e.first].insert(e.second); o.graph[e.second].insert(e.first); }
  for (auto& e : nb) {
    int32_t new_mol = std::min(o.mol_id[e.first], o.mol_id[e.second]);
    std::vector<int32_t> st{e.first}; std::set<int32_t> vis{e.first};
    while (!st.empty()) { int32_t p = st.back(); st.pop_back(); o.mol_id[p] = new_mol; for (int32_t q : o.graph[p]) if (vis.insert(q).second) st.push_back(q); }
  }
  return 0;
}

int orc_list_set_params(void* c, int list, int t1, int t2, int t3, int t4, const double* p, int np) {
  Orc& o = O(c);
  if (list < 0 || list >= (int)o.lists.size() || np < 1 || np > CHEM_MAX_POT_PARAMS) FAIL(CHEM_EINVAL, "list_set_params");
  BondedList& l = o.lists[list];
  if (!l.by_types) { std::fill(l.plain, l.plain + CHEM_MAX_POT_PARAMS, 0.0); std::copy(p, p + np, l.plain); l.has_plain = true; return 0; }
  int tt[4] = {t1, t2, t3, t4};
  std::vector<int> key(tt, tt + l.arity);
  std::vector<double> v(CHEM_MAX_POT_PARAMS, 0.0); std::copy(p, p + np, v.begin());
  l.typed[key] = v; return 0;
}

int64_t orc_get_list(void* c, int list, int64_t* out, int64_t cap) {
  Orc& o = O(c);
  if (list < 0 || list >= (int)o.lists.size()) FAIL(CHEM_EINVAL, "list handle");
  BondedList& l = o.lists[list]; int64_t ne = (int64_t)l.ent.size() / l.arity;
  if (!out) return ne; if (cap < ne) FAIL(CHEM_ENOSPC, "get_list cap");
  for (size_t k = 0; k < l.ent.size(); ++k) out[k] = o.id[l.ent[k]];
  return ne;
}

int orc_table_create(void* c, int64_t nrow, double r0, double dr, const double* e, const double* f) {
  Orc& o = O(c);
  if (nrow < 2 || !(dr > 0) || !e || !f) FAIL(CHEM_EINVAL, "table_create");
  Orc::BTable t; t.r0 = r0; t.dr = dr; t.e.assign(e, e + nrow); t.f.assign(f, f + nrow);
  o.btables.push_back(std::move(t));
  return (int)o.btables.size() - 1;
}
int orc_thermostat_rescale(void* c, int kind, double kT, double param) {
  Orc& o = O(c);
  if (kind < 0 || kind > 2 || (kind && !(kT > 0)) || (kind == 1 && !(param > 0)) || (kind == 2 && param < 1)) FAIL(CHEM_EINVAL, "thermostat_rescale");
  o.resc_kind = kind; o.resc_kT = kT; o.resc_param = kind == 2 ? std::floor(param) : param; return 0;
}
int orc_thermostat_svr(void* c, double kT, double coupling, uint64_t seed) {
  Orc& o = O(c);
  if (coupling > 0 && !(kT > 0)) FAIL(CHEM_EINVAL, "thermostat_svr");
  o.resc_kind = coupling > 0 ? 3 : 0; o.resc_kT = kT; o.resc_param = coupling; o.svr_seed = seed; return 0;
}
int orc_cap_force(void* c, double max_force) { O(c).cap_force = max_force > 0 ? max_force : 0; return 0; }
int orc_thermostat_langevin(void* c, double kT, double gamma, uint64_t seed) {
  Orc& o = O(c); o.lang = (gamma > 0 && kT >= 0); o.kT = kT; o.gamma = gamma; o.lang_seed = seed; return 0;
}

int orc_thermostat_langevin_types(void* c, int n, const int32_t* types) {
  Orc& o = O(c); uint32_t m = 0;
  for (int k = 0; k < n; ++k) { if (types[k] < 0 || types[k] >= CHEM_MAX_TYPES) FAIL(CHEM_EINVAL, "thermal group type id"); m |= 1u << types[k]; }
  o.lang_tmask = m; return 0;
}

int orc_reaction_init(void* c, int interval, int nearest, int max_per_interval, uint64_t seed) {
  Orc& o = O(c); if (interval <= 0) FAIL(CHEM_EINVAL, "interval");
  o.react_init = true; o.interval = interval; o.nearest = nearest; o.max_per_interval = max_per_interval; o.react_seed = seed; return 0;
}

int orc_reaction_add(void* c, const chem_reaction_desc* d) {
  Orc& o = O(c);
  if (!o.react_init) FAIL(CHEM_ESTATE, "reaction_init first");
  if (!d->is_virtual && (d->bond_list < 0 || d->bond_list >= (int)o.lists.size() || o.lists[d->bond_list].arity != 2)) FAIL(CHEM_EINVAL, "bond_list");
  o.reactions.push_back(*d); return (int)o.reactions.size() - 1;
}

int orc_reaction_neighbour_change(void* c, const chem_nb_change* r) {
  Orc& o = O(c);
  if (!r || r->reaction < 0 || r->reaction >= (int)o.reactions.size() || r->invoke_on < 1 || r->invoke_on > 3 || r->nb_level < 1 ||
      r->old_type < 0 || r->old_type >= CHEM_MAX_TYPES || r->new_type < 0 || r->new_type >= CHEM_MAX_TYPES || !(r->new_mass > 0))
    FAIL(CHEM_EINVAL, "reaction_neighbour_change");
  o.nb_rules.push_back(*r); return 0;
}

int orc_reaction_constraint(void* c, int reaction, int role, int nb_type, int min_state, int max_state) {
  Orc& o = O(c);
  if (reaction < 0 || reaction >= (int)o.reactions.size() || (role != 1 && role != 2) || nb_type < 0 || nb_type >= CHEM_MAX_TYPES) FAIL(CHEM_EINVAL, "reaction_constraint");
  if (o.constraints.size() < o.reactions.size()) o.constraints.resize(o.reactions.size());
  o.constraints[reaction] = Orc::NbCons{role, nb_type, min_state, max_state}; return 0;
}
int orc_reaction_restrict(void* c, int reaction, int64_t n, const int64_t* p) {
  Orc& o = O(c);
  if (reaction < 0 || reaction >= (int)o.reactions.size() || reaction >= 32) FAIL(CHEM_EINVAL, "reaction_restrict: reaction index");
  for (int64_t k = 0; k < n; ++k) {
    auto a = o.id2tag.find(p[2 * k]), b = o.id2tag.find(p[2 * k + 1]);
    if (a == o.id2tag.end() || b == o.id2tag.end()) FAIL(CHEM_EINVAL, "reaction_restrict: unknown id");
    o.restrict_map[{std::min(a->second, b->second), std::max(a->second, b->second)}] |= 1u << reaction;
  }
  o.restricted_mask |= 1u << reaction; return 0;
}
int orc_atrp_init(void* c, const chem_atrp_desc* d) {
  Orc& o = O(c);
  if (!d) { o.atrp_on = false; return 0; }
  if (d->interval <= 0 || d->num_particles <= 0) FAIL(CHEM_EINVAL, "atrp_init");
  o.atrp = *d; o.atrp_on = true; return 0;
}
int orc_atrp_add_center(void* c, int type, int state, int is_activator, int new_type, double new_mass, double new_q, int delta_state) {
  Orc& o = O(c);
  if (type < 0 || type >= CHEM_MAX_TYPES || new_type >= CHEM_MAX_TYPES) FAIL(CHEM_EINVAL, "atrp_add_center");
  o.atrp_centers.push_back({type, state, is_activator, new_type, delta_state, new_mass, new_q}); return 0;
}
int64_t orc_atrp_get_stats(void* c, chem_atrp_stats* out, int64_t cap) {
  Orc& o = O(c); const int64_t n = (int64_t)o.atrp_stats.size();
  if (!out) return n; if (cap < n) FAIL(CHEM_ENOSPC, "atrp stats cap");
  std::copy(o.atrp_stats.begin(), o.atrp_stats.end(), out); return n;
}

int orc_topology_register(void* c, int arity, int list, const int32_t* types) {
  Orc& o = O(c);
  if (list < 0 || list >= (int)o.lists.size() || o.lists[list].arity != arity) FAIL(CHEM_EINVAL, "topology_register");
  o.lists[list].registered.emplace_back(types, types + arity); return 0;
}

int orc_set_option(void* c, const char* name, double value) {
  Orc& o = O(c); std::string k = name ? name : "";
  if (k == "rebuild_criterion") { o.criterion = value != 0 ? 1 : 0; o.resort = true; return 0; }
  if (k == "count_intra_inter") { o.count_intra_inter = value != 0; return 0; }
  if (k == "threads") {   // all-core CPU baseline (needs the -fopenmp build, otherwise stays scalar)
#ifdef _OPENMP
    o.threads = value >= 1 ? (int)value : 1; o.par = true; o.resort = true; return 0;
#else
    FAIL(CHEM_ENOTIMPL, "threads: this oracle build has no OpenMP (use liboracle_omp.so)");
#endif
  }
  return 0;   // device tuning knobs have no meaning for the scalar restatement
}

int orc_reactions_enable(void* c, int on) { O(c).react_on = on != 0; return 0; }
int orc_reaction_set_rate(void* c, int r, double rate) { Orc& o = O(c); if (r < 0 || r >= (int)o.reactions.size()) FAIL(CHEM_EINVAL, "reaction"); o.reactions[r].rate = rate; return 0; }

int orc_run(void* c, int64_t nsteps) {
  Orc& o = O(c);
  if (o.n == 0 || !(o.L[0] > 0) || !(o.rc > 0) || !(o.dt > 0)) FAIL(CHEM_ESTATE, "run: system incomplete");
  run(o, nsteps); return 0;
}

int64_t orc_num_particles(void* c) { return O(c).n; }
int64_t orc_get_step(void* c) { return O(c).step; }

int64_t orc_get_state(void* c, int what, void* out, int64_t cap) {
  Orc& o = O(c); int64_t n = o.n;
  int per = (what == CHEM_STATE_POS || what == CHEM_STATE_VEL || what == CHEM_STATE_FORCE || what == CHEM_STATE_IMAGE || what == CHEM_STATE_POS_UNFOLDED) ? 3 : 1;
  if (cap < n * per) FAIL(CHEM_ENOSPC, "get_state cap");
  double* d = (double*)out; int32_t* i32 = (int32_t*)out; int64_t* i64 = (int64_t*)out;
  for (int64_t i = 0; i < n; ++i) {
    switch (what) {
      case CHEM_STATE_POS: { Vec3 p = o.x[i]; double* q = &p.x; for (int k = 0; k < 3; ++k) { double s = std::floor(q[k] / o.L[k]); q[k] -= s * o.L[k]; if (q[k] >= o.L[k]) q[k] -= o.L[k]; d[3 * i + k] = q[k]; } break; }
      case CHEM_STATE_POS_UNFOLDED: { const double* q = &o.x[i].x; for (int k = 0; k < 3; ++k) d[3 * i + k] = q[k] + o.img[3 * i + k] * o.L[k]; break; }
      case CHEM_STATE_VEL: d[3 * i] = o.v[i].x; d[3 * i + 1] = o.v[i].y; d[3 * i + 2] = o.v[i].z; break;
      case CHEM_STATE_FORCE: d[3 * i] = o.f[i].x; d[3 * i + 1] = o.f[i].y; d[3 * i + 2] = o.f[i].z; break;
      case CHEM_STATE_TYPE: i32[i] = o.type[i]; break;
      case CHEM_STATE_STATE: i32[i] = o.state[i]; break;
      case CHEM_STATE_RESID: i32[i] = o.res_id[i]; break;
      case CHEM_STATE_MOLID: i32[i] = (int32_t)o.id[o.mol_id[i]]; break;
      case CHEM_STATE_MASS: d[i] = o.mass[i]; break;
      case CHEM_STATE_ID: i64[i] = o.id[i]; break;
      case CHEM_STATE_IMAGE: for (int k = 0; k < 3; ++k) i32[3 * i + k] = o.img[3 * i + k]; break;
      default: FAIL(CHEM_EINVAL, "get_state: what");
    }
  }
  return n;
}

int64_t orc_get_events(void* c, chem_event* out, int64_t cap) {
  Orc& o = O(c); int64_t n = (int64_t)o.events.size();
  if (!out) return n; if (cap < n) FAIL(CHEM_ENOSPC, "events cap");
  std::copy(o.events.begin(), o.events.end(), out); return n;
}

int64_t orc_get_exclusions(void* c, int64_t* out, int64_t cap) {
  Orc& o = O(c); int64_t n = 0; for (auto& s : o.excl) n += (int64_t)s.size();
  if (!out) return n; if (cap < n) FAIL(CHEM_ENOSPC, "excl cap");
  int64_t k = 0; for (int64_t a = 0; a < o.n; ++a) for (int32_t b : o.excl[a]) { out[2 * k] = o.id[a]; out[2 * k + 1] = o.id[b]; ++k; }
  return n;
}

int64_t orc_get_verlet_pairs(void* c, int64_t* out, int64_t cap) {
  Orc& o = O(c);
  if (o.resort) build_pairs(o);
  materialise_pairs(o);
  int64_t n = (int64_t)o.pairs.size();
  if (!out) return n; if (cap < n) FAIL(CHEM_ENOSPC, "pairs cap");
  for (int64_t k = 0; k < n; ++k) { out[2 * k] = o.id[o.pairs[k].first]; out[2 * k + 1] = o.id[o.pairs[k].second]; }
  return n;
}

// forces/energies of the current configuration without advancing time (no thermostat noise)
int orc_observe(void* c, chem_obs* out) {
  Orc& o = O(c);
  if (o.resort) build_pairs(o);
  bool lang = o.lang; o.lang = false;
  std::vector<Vec3> fsave = o.f;
  update_forces(o, o.step, 0);
  o.f = fsave; o.lang = lang;
  std::memset(out, 0, sizeof(*out));
  out->step = o.step; out->npart = o.n;
  double ek = 0, p[3] = {0, 0, 0};
  for (int64_t i = 0; i < o.n; ++i) { ek += 0.5 * o.mass[i] * dot(o.v[i], o.v[i]); p[0] += o.mass[i] * o.v[i].x; p[1] += o.mass[i] * o.v[i].y; p[2] += o.mass[i] * o.v[i].z; }
  out->ekin = ek; out->temperature = 2.0 * ek / (3.0 * (double)o.n);
  out->epot_lj = o.e_lj; out->epot_tab = o.e_tab; out->virial_nb = o.virial;
  for (size_t l = 0; l < o.lists.size(); ++l) { out->epot_list[l] = o.e_list[l]; out->list_size[l] = (int64_t)o.lists[l].ent.size() / o.lists[l].arity; }
  for (int k = 0; k < 3; ++k) out->momentum[k] = p[k];
  return 0;
}

// evaluate forces of the current configuration into the force array (for force parity tests)
int orc_compute_forces(void* c) {
  Orc& o = O(c);
  if (o.resort) build_pairs(o);
  update_forces(o, o.step, 0);
  return 0;
}

int orc_get_timers(void* c, chem_timers* t) { Orc& o = O(c); materialise_pairs(o); std::memset(t, 0, sizeof(*t)); t->steps = o.step; t->rebuilds = o.rebuilds; t->list_rebuilds = o.rebuilds; t->reaction_steps = o.reaction_steps; t->nlist_entries = 2 * (int64_t)o.pairs.size(); return 0; }

}  // extern "C"

// one StochasticVelocityRescaling factor (tests/test_oracle_analytic.py)
extern "C" double orc_svr_lambda(uint64_t seed, uint64_t step, double K, double K_ref, int64_t ndeg, double taut) {
  return chem_philox::svr_lambda(seed, step, K, K_ref, ndeg, taut);
}

// raw Philox block for the known-answer test (tests/test_philox.py)
extern "C" void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  chem_philox::philox4x32_10(ctr, key, out);
}
