// chem_api.hip -- context, run loop and the C ABI of libchem_mi355.so (include/chem_mi355.h).
//
// One context == one GPU.  chem_run() enqueues the whole velocity-Verlet loop on one HIP
// stream with no host round trip per step: the skin-triggered neighbour rebuild is decided on
// the device (k_rebuild_decide) and the rebuild chain early-exits when it is not needed.
// The host only synchronises at reaction steps (every `interval` steps) and at the end.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <thread>
#include <mutex>
#include <exception>
#include <tuple>

#include "chem_comm.hpp"
#include "chem_host.hpp"
#include "md_kernels.hpp"

namespace chem {

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      throw ChemError(CHEM_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));           \
  } while (0)

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// std::vector storage in pinned host memory: the device tables rebuilt at reaction steps (bonded and
// exclusion CSR, ~14 MB at 10^6 particles) are uploaded at PCIe speed instead of through the
// pageable-copy bounce buffers (2.5 ms -> 0.5 ms)
template <typename T> struct PinnedAlloc {
  using value_type = T;
  PinnedAlloc() = default;
  template <class U> PinnedAlloc(const PinnedAlloc<U>&) {}
  T* allocate(size_t n) { void* p = nullptr; if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) throw std::bad_alloc(); return (T*)p; }
  void deallocate(T* p, size_t) { (void)hipHostFree(p); }
  template <class U> bool operator==(const PinnedAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const PinnedAlloc<U>&) const { return false; }
};
template <typename T> using PinnedVec = std::vector<T, PinnedAlloc<T>>;

template <typename T> struct DBuf {
  T* p = nullptr; size_t n = 0;
  void alloc(size_t count) {
    if (count <= n && p) return;
    free();
    HIPCHK(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)));
    n = count;
  }
  void free() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
  ~DBuf() { free(); }
  template <class V> void upload(const V& h, hipStream_t s) {
    alloc(h.size());
    if (!h.empty()) HIPCHK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void download(std::vector<T>& h, size_t count, hipStream_t s) {
    h.resize(count);
    if (count) { HIPCHK(hipMemcpyAsync(h.data(), p, count * sizeof(T), hipMemcpyDeviceToHost, s)); }
    HIPCHK(hipStreamSynchronize(s));
  }
};

static const bool g_trace = getenv("CHEM_TRACE") != nullptr;
struct Trace {
  double t; const char* what;
  Trace(const char* w) : t(now_s()), what(w) {}
  void lap(const char* tag) { if (g_trace) { double n = now_s(); fprintf(stderr, "[chem trace] %s/%s %.3f ms\n", what, tag, 1e3 * (n - t)); t = n; } }
};

static inline int cdiv(long long a, int b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------
struct Ctx {
  std::string err;
  int device = 0, prec = 32;
  double L[3] = {0, 0, 0}, rc = 0, skin = 0, dt = 0, cap_force = 0;
  int resc_kind = 0; double resc_kT = 0, resc_param = 0;   // Berendsen (1) / Isokinetic (2) velocity rescaling, StochasticVelocityRescaling (3)
  uint64_t svr_seed = 0;
  HostTopology top;
  std::vector<double> pos0, vel0;  // staged particle data (tag order) until first upload
  int ntypes = 1;
  HostPairPot pp[CHEM_MAX_TYPES][CHEM_MAX_TYPES];
  bool lang = false; double kT = 0, gamma = 0; uint64_t lang_seed = 0; uint32_t lang_tmask = 0;
  bool react_init = false, react_on = false;
  int interval = 0, nearest = 1, max_per_interval = 0; uint64_t react_seed = 0;
  std::vector<chem_reaction_desc> reactions;
  std::vector<chem_nb_change> nb_rules;   // PostProcessChangeNeighboursProperty (chem_reaction_neighbour_change)
  // RestrictReaction.define_connection (chem_reaction_restrict): (tag lo, tag hi) -> reaction bits; CSR rebuilt when it changed
  std::map<std::pair<int32_t, int32_t>, uint32_t> restrict_map; uint32_t restricted_mask = 0; bool restrict_dirty = false;
  struct NbCons { int role = 0, nb_type = 0, min_state = 0, max_state = 0; };   // ReactionConstraintNeighbourState (chem_reaction_constraint)
  std::vector<NbCons> constraints;
  // integrator.ATRPActivator (chem_atrp_init; reaction_post_process.py:380-426)
  struct AtrpCenter { int type, state, is_activator, new_type, delta_state; double new_mass, new_q; };
  bool atrp_on = false; chem_atrp_desc atrp{}; std::vector<AtrpCenter> atrp_centers; std::vector<chem_atrp_stats> atrp_stats;
  std::vector<chem_event> events;   // expanded, canonical order (filled lazily from the arena by chem_get_events)
  // event arena: SoA copy of the device records of every reaction step, appended in device order (amortised growth:
  // fresh vectors per step cost their page faults every time -- 4 ms for 2.8e5 events); blocks = (step, first index).
  // The host's type/mass/charge mirrors are brought up to date from it lazily (sync_type_mirrors): chemical states and
  // types live on the device, the mirrors are only read by typed lists, tuple spawning and read-back paths.
  struct EventArena {
    std::vector<int32_t> a, b, r; std::vector<double> d2; std::vector<int8_t> intra;
    std::vector<std::pair<int64_t, size_t>> blocks; size_t mirror_pos = 0, expanded_blocks = 0;
    size_t size() const { return a.size(); }
    void clear() { a.clear(); b.clear(); r.clear(); d2.clear(); intra.clear(); blocks.clear(); mirror_pos = 0; expanded_blocks = 0; }
  } arena;
  int opt_intra_inter = 0;   // classify every event as intra-/inter-cluster at event time (ar.save_intra_inter_counter)
  // records of one reaction step (device order) -> arena; runs on the bookkeeping thread beside the MD steps unless the
  // host needs the type mirrors within the reaction step itself
  template <class Rec> void append_events(const Rec* ev, size_t m, int64_t at_step, const std::vector<int8_t>& intra_flags) {
    EventArena& A = arena;
    const size_t n0 = A.size();
    A.blocks.emplace_back(at_step, n0);
    A.a.resize(n0 + m); A.b.resize(n0 + m); A.r.resize(n0 + m); A.d2.resize(n0 + m);
    if (!intra_flags.empty()) { A.intra.resize(n0 + m); std::copy(intra_flags.begin(), intra_flags.end(), A.intra.begin() + n0); }
    for (size_t k = 0; k < m; ++k) { A.a[n0 + k] = ev[k].a; A.b[n0 + k] = ev[k].b; A.r[n0 + k] = ev[k].r; A.d2[n0 + k] = ev[k].d2; }
  }
  void sync_type_mirrors() {
    for (size_t k = arena.mirror_pos; k < arena.size(); ++k) {
      const chem_reaction_desc& d = reactions[arena.r[k]];
      const int32_t ea = arena.a[k], eb = arena.b[k];
      if (d.new_type_1 >= 0 && d.new_type_1 != top.type[ea]) { top.type[ea] = d.new_type_1; top.mass[ea] = d.new_mass_1; top.q[ea] = d.new_q_1; }
      if (d.new_type_2 >= 0 && d.new_type_2 != top.type[eb]) { top.type[eb] = d.new_type_2; top.mass[eb] = d.new_mass_2; top.q[eb] = d.new_q_2; }
    }
    arena.mirror_pos = arena.size();
  }
  int64_t step = 0;
  bool resort = true;
  bool geom_dirty = true, particles_dirty = true, pair_dirty = true, bonded_dirty = true, excl_dirty = true, labels_dirty = true;
  int nl_capacity_user = 0;
  int opt_tpp = 0;          // 0 = automatic
  int opt_time_pair = 0;    // HIP-event timing of every N-th pair-force launch (0 = off)
  int64_t pair_launch_no = 0;
  int opt_fuse = 1;         // fused integrate2+integrate1
  int opt_lpt = 1;          // descriptors of the fused rebuild in largest-tile-first order per XCD range (force launch tail)
  bool opt_dd_fold = true;   // decomposed path: block maxima folded by atomics of the integrate kernel (no fold launch per step)
  int opt_dd_merge = 1;     // decomposed path: displacement fold in the integrate kernel, decision in the force kernel (no one-block launches)
  bool opt_dd_fastx = true;  // decomposed path: excluded partners located as slots by the standalone list kernel (ghost copies through gtag)
  int opt_ablate_list = 0;  // diagnostics (with debug_stamps): parts of the list build left out, see tools/rebuild_stamps.py
  int opt_tiles = 1;        // LDS-tiled list/force kernels when the cell grid allows
  int opt_fused = 1;        // rebuild chain as one persistent launch with grid barriers (single domain, tiles)
  int pair_guard = 0;       // 256 while the force kernels are launched speculatively (decomposed path)
  int pair_subset = 0;      // 0 all tiles, 1 interior tiles of the slab, 2 its two boundary tile layers
  int opt_overlap = -1;     // decomposed path: interior forces while the halo exchange is in flight (-1: automatic, by slab size)
  hipStream_t cstream = nullptr; hipEvent_t ev_fold = nullptr, ev_halo = nullptr;
  int opt_criterion = 0;     // 1: rebuild when max |x - x(last build)| > skin/2 ; 0: reference's accumulated per-step maxima
  // Internal list skin (>= the workload's skin): cells, tiles and the 16-bit force list are built for rc + skin_list and live
  // until the accumulated displacement exceeds skin_list / 2.  Forces do not depend on it (exact cutoff in the force kernel);
  // the reported rebuild count stays the reference rule's (DevCtl::ref_rebuilds).  opt_list_skin: < 0 automatic, 0 off
  // (= the workload's skin), > 0 explicit.  Only the fused single-domain tile path uses it.
  double opt_list_skin = -1.0, skin_list = 0.0;
  double skin_eff() const { return skin_list > skin ? skin_list : skin; }
  int opt_ablate = 0;        // diagnostic only: 1 = pair kernel stops after staging, 2 = skips staging
  int opt_skip_inactive = 1; // force list omits type pairs without a potential
  int opt_bonds_inline = 1;  // harmonic 2-body bonds evaluated by the force kernel from its staged image (when the host can prove it applicable)
  int opt_tile_split = 0;    // narrow tiles at the end of every tile row (pick_tile_split): 0 off, nb * 10 + w
  int opt_bucket_cap = 0;    // testing: bucket rows of the fused rebuild narrower than 64 (provokes the mid-run overflow recovery)
  int64_t halts_recovered = 0;
  // slab domain decomposition (chem_comm_init)
  bool dd_on = false; int P = 1, rk = 0;
  std::unique_ptr<Transport> tr;
  chem_timers tm{};
  virtual ~Ctx() {}
  virtual void run(int64_t nsteps) = 0;
  virtual int64_t get_state(int what, void* out, int64_t cap) = 0;
  virtual void observe(chem_obs* out) = 0;
  virtual int64_t verlet_pairs(int64_t* out, int64_t cap) = 0;
  virtual void modify_particle(int tag, int what, double value) = 0;
  virtual void sync() = 0;
  virtual void refresh_timers() {}
  virtual void set_pair_bs(int) {}
  virtual void set_bond_pass(bool) {}
  virtual int64_t debug_dump(long long*, int64_t) { return 0; }
  virtual void debug_enable(int) {}
  virtual int64_t debug_dump_rebuild(long long*, int64_t) { return 0; }
  virtual int64_t debug_force_list(int, int32_t*, int64_t) { return -1; }
  virtual void debug_tiles(int32_t* out) = 0;
  virtual void join_async() {}
};

template <typename R> struct CtxT : Ctx {
  using V4 = Vec4<R>;
  int pair_bs = 512;
  bool lj_only = true;
  bool state_mirror_stale = false;   // top.state lags the device after reaction steps
  int tile_cap = 0;       // LDS slots of one staged tile (dynamic LDS)
  size_t tile_lds_bytes() const { return (size_t)(tile_cap + 5) * sizeof(V4) + 16; }
  // force kernel: fp64 stages 24-byte slots + type bytes (two workgroups per CU instead of one)
  size_t pair_lds_bytes() const { return sizeof(R) == 8 ? (size_t)(tile_cap + 5) * 25 + 64 : tile_lds_bytes(); }
  // the list build has its own layout of the same block (fp32: SoA groups + type masks + slice boundaries)
  // (fp64 builds use the fp32 list image too -- the force list may be a superset -- and need their own 32-byte-per-slot
  //  image only where the exact int32 rows are built)
  static constexpr size_t kTileLdsBudget = 150 * 1024;   // of 160 KB per CU: the rest is the kernels' static __shared__ (tile tables, scan scratch)
  // reaction scan: the staged image + 4 bytes per slot of role words (k_react_roles)
  size_t scan_lds_bytes() const { return scan_roles_offset(tile_cap, sizeof(V4)) + (size_t)(tile_cap + 1) * sizeof(unsigned int); }
  bool scan_roles_staged() const { return scan_lds_bytes() <= kTileLdsBudget; }      // (otherwise the scan reads the role words from global memory)
  size_t tile_lds_need() const { return std::max(std::max(tile_lds_bytes(), pair_lds_bytes()), list_lds_need()); }
  size_t list_lds_need(bool exact_rows = true) const {
    const size_t lb = list_lds_bytes(tile_cap, kMaxTypes);
    return (sizeof(R) == 4 || exact_rows) ? std::max(tile_lds_bytes(), lb) : lb;
  }
  // ---- position codec on the host (md_kernels.hpp "position codec"): fp32 build = int32 fixed point, fp64 build = reals
  static constexpr bool kFixed = sizeof(R) == 4;
  double q_scale(int d) const { return L[d] / 2147483648.0; }
  R enc_pos(double x, int d) const {      // x already folded into [0, L)
    if constexpr (kFixed) { const int32_t q = (int32_t)(long long)std::llrint((x - 0.5 * L[d]) / q_scale(d)); R r; std::memcpy(&r, &q, 4); return r; }
    else return (R)x;
  }
  double dec_pos(R v, int d) const {
    if constexpr (kFixed) { int32_t q; std::memcpy(&q, &v, 4); return (double)q * q_scale(d) + 0.5 * L[d]; }
    else return (double)v;
  }
  R enc_shift(double boxes, int d) const {   // periodic shift of +-1 box length (slab ghost layers)
    if constexpr (kFixed) { const uint32_t q = boxes != 0.0 ? 0x80000000u : 0u; R r; std::memcpy(&r, &q, 4); return r; }
    else return (R)(boxes * L[d]);
  }
  static double fold_coord(double x, double Ld) { double s = std::floor(x / Ld); x -= s * Ld; if (x >= Ld) x -= Ld; if (x < 0) x = 0; return x; }
  PosScale<R> pos_scale() const {
    PosScale<R> ps{};
    for (int d = 0; d < 3; ++d) { ps.s[d] = (R)q_scale(d); ps.inv[d] = (R)(1.0 / q_scale(d)); }
    return ps;
  }
  hipStream_t stream = nullptr;
  int n = 0;
  DBuf<V4> x4, v4, f4, x4o, v4o, tab, x0;
  DBuf<int> tag, tago, rtag, state, res_id, mol_id;
  DBuf<int4> img4, img4o;
  DBuf<int> cell_cnt, cell_start, cell_of, slot_of, perm, cell_sub;
  // fused rebuild (single domain, tiles): segment scans + grid barrier state
  DBuf<int> cell_loc, seg_tot, tile_n, tile_loc, tseg_tot, cell_n, bucket; int bcap = 0; DBuf<GridBar> gbar;
  DBuf<int> tile_cnt, tile_off;   // reaction scan on tiles: candidates per tile, their offsets
  DBuf<unsigned int> scan_role;   // ... and the role bits of every particle slot (k_react_roles)
  DBuf<int4> bwork, bj; int nb_owner = 0; bool bwork_dirty = true;   // bonded work list (see dev_bonded_prep)
  // Inline bonds: all bonded terms are 2-body bonds of ONE harmonic parameter set and the exclusion set is exactly the bond
  // set (chain-growth systems) -> the LDS slots of the excluded partners, which the list build locates anyway, ARE the bonded
  // partners: it writes them out (bslots), the force kernel evaluates the bonds from its staged image, the per-step bonded
  // launch disappears.  Particles with more than kBondSlots (8) exclusions stay with the work list (then launched for them alone).
  // Decided on the host from what it knows exactly (bonds_inline()).
  DBuf<uint4> bslots; bool harmonic_only = false, bonds_excluded = false;
  std::vector<uint8_t> excl_deg; size_t excl_deg_upto = 0; int64_t excl_over = 0;   // particles with more than kBondSlots (8) exclusions: their bonds stay with the work-list kernel
  void update_excl_deg() {
    if (excl_deg.size() != (size_t)top.n) { excl_deg.assign((size_t)top.n, 0); excl_deg_upto = 0; excl_over = 0; }
    if (excl_deg_upto > top.excl_log.size()) { std::fill(excl_deg.begin(), excl_deg.end(), 0); excl_deg_upto = 0; excl_over = 0; }
    for (; excl_deg_upto < top.excl_log.size(); ++excl_deg_upto) {
      const auto& e = top.excl_log[excl_deg_upto];
      for (int32_t t : {e.first, e.second}) { if (excl_deg[t] < 255) { ++excl_deg[t]; if (excl_deg[t] == kBondSlots + 1) ++excl_over; } }
    }
  }
  double inline_K = 0, inline_r0 = 0;
  // ... and where no particle has more than kBondSlots of them (single domain, fused rebuild), the list build does not look at
  // the exclusions at all: a streaming pass behind the tiles records the partner slots, the force kernel takes the partners'
  // pair term out of its sums again (k_pair_tiles bond_mode 2).  Option bond_pass = 0: the list build removes the pairs itself.
  bool opt_bond_pass = true;
  void set_bond_pass(bool v) override { opt_bond_pass = v; }
  DBuf<BondRec<R>> brec_dev; BondRec<R> brec_host{}; bool brec_valid = false;
  bool bond_by_pass() { return opt_bond_pass && ((use_fused && !dd_on) || (dd_on && use_tiles && opt_dd_fastx)) && bonds_inline() && excl_over == 0; }
  bool dd_record_bonds = false;      // slabs, bond pass: the next force launch records the partner slots (set by the slab rebuild)
  bool bonds_inline() {
    if (!opt_bonds_inline || !(use_fused || (dd_on && use_tiles)) || nbent <= 0 || !harmonic_only || !bonds_excluded) return false;
    // the exclusion set must BE the bond set (bonds are a subset: bonds_excluded; both are duplicate-free): equal counts
    size_t nb2 = 0;
    for (const auto& l : top.lists) if (l.arity == 2) nb2 += (size_t)l.size();
    if (nb2 != top.excl_log.size()) return false;
    update_excl_deg();
    return true;
  }
  int fused_grid = 0, fused_grid_diag = 0, fused_par = 0, seg_shift = 0, tseg_shift = 0; bool use_fused = false;
  DBuf<int> nlist, nn, nnh;
  DBuf<unsigned short> nl16;
  DBuf<TileLDS<R>> tdesc;
  DBuf<long long> dbgbuf, wgst; bool dbg_on = false;
  Candidate* pin_ev = nullptr; size_t pin_ev_cap = 0;   // pinned host staging for reaction events
  // cluster labels of a reaction step are merged on a host thread beside the following MD steps;
  // joined (and uploaded) before the next reaction scan and before any other API call
  std::thread label_thr; bool labels_pending = false; std::exception_ptr label_err;
  std::vector<std::pair<int32_t, int32_t>> label_bonds; std::vector<int32_t> label_touched;
  std::vector<int> label_seen_lists;      // bonded list of every bond in label_bonds whose de-duplication key the thread still has to insert
  void join_async() override {
    join_thread();
    sync_type_mirrors();      // any API call other than chem_run sees current type/mass/charge mirrors
  }
  // the bookkeeping thread of the last reaction step (event log, bond graph, cluster labels, exclusion rows); the
  // reaction step itself joins only this -- replaying the type mirrors there would cost the 2-3 ms it just saved
  void join_thread() {
    if (label_thr.joinable()) label_thr.join();
    if (label_err) { std::exception_ptr e = label_err; label_err = nullptr; labels_pending = false; std::rethrow_exception(e); }
    if (labels_pending) {
      res_id.upload(top.res_id, stream); mol_id.upload(top.mol_id, stream);
      HIPCHK(hipStreamSynchronize(stream));
      labels_pending = false;
    }
  }
  // ---- slab decomposition state: reals live at [G, G+n), lower ghosts right-aligned in front of
  // them, upper ghosts behind; cap = allocated particles
  int nglob = 0, lower = 0, upper = 0, nzg = 0, z0 = 0, ncz = 0;
  int G = 0, nglo = 0, ngup = 0, cap = 0, mcap = 0;
  DBuf<unsigned char> mig[4];          // send down, send up, recv from upper, recv from lower
  DBuf<int> lcnt_dn, lcnt_up, gcnt_lo, gcnt_up;
  int halo_dn_off = 0, halo_dn_cnt = 0, halo_up_off = 0, halo_up_cnt = 0;   // slices of the own boundary layers
  DBuf<double> redbuf;                 // small device buffer for cross-rank reductions
  DBuf<Candidate> cand_loc; DBuf<int> cnt_all;   // reaction candidates before the all-gather
  int acap() const { return dd_on ? cap : n; }
  int64_t dd_rebuilds = 0, dd_direct_rebuilds = 0;   // slab rebuilds: all of them / those the host called for itself (rebuild_now)
  int* hflag = nullptr; int* hflag_dev = nullptr; int hticket = 0;   // pinned decision word + ticket
  DBuf<double> dd_vals;
  DBuf<unsigned long long> foldmax;    // decomposed path: atomic fold of the step's displacement maxima (kFoldSlots words, md_kernels.hpp)
  bool fold_on() const { return dd_merged() && opt_dd_fold; }
  // (allocated here if need be: the first drift of a context runs before its first dd_step_sync, and a launch without the words
  //  would leave that step's displacement out of the accumulated distance)
  unsigned long long* fold_arg() { if (!(dd_on && fold_on())) return nullptr; ensure_hflag(); return foldmax.p; }
  DBuf<int> gtag;                      // slab: index of a tag's ghost copy on this rank (-1: none)
  DBuf<Box<R>> box_dev; Box<R> box_dev_host; bool box_dev_valid = false;   // device copy of the box for the standalone list kernel
  int S = 0;
  bool use_tiles = false, want32 = false;   // int32 list only built on demand (reaction steps, diagnostics)
  ActMask act{}; UniLJ uni{}; bool uniform_lj = false; bool all_active = false;
  int ntiles = 0;
  DBuf<int> excl_start, excl_list; int has_excl = 0;
  DBuf<int> bstart; DBuf<BondedEntry> bent; DBuf<BondedParam> bpar; int64_t nbent = 0; bool bonds_only = false;
  DBuf<PairCore<R>> pcore; DBuf<PairExt<R>> pext;
  DBuf<DevCtl> ctl;
  DBuf<unsigned long long> blockmax;
  int dd_par = 0; DecideArgs pair_da{};   // decomposed path: parity of the accumulated distance, decision arguments of the force launch
  DBuf<double> eout, ekout, elist;
  // reactions
  DBuf<Candidate> cand, evout; int cand_cap = 0;
  DBuf<int> st0, st1, asA, asB, evcount;
  DBuf<unsigned long long> best1, best2;
  DBuf<ReactSet> rs_dev;
  DBuf<int> conn_start, conn_partner; DBuf<unsigned int> conn_mask, cons_ok;
  Box<R> box{}; BoxD boxd{};
  bool device_ready = false;
  // per-kernel HIP-event samples of the timed region (option time_pair_kernel = N: every N-th step)
  std::vector<hipEvent_t> ev;  // pool, used in pairs
  size_t ev_used = 0;
  std::vector<int> ev_kind;    // kind of sample k (events 2k, 2k+1): 0 pair, 1 neighbour kernel, 2 integrate, 3 bonded
  bool timed_step = false;
  // A sampled step brackets the force kernel and ONE of the other per-step kernels (rotating) with events: the records between
  // the launches cost ~3 us each, and eight of them on every sampled step showed in short timed regions
  int timed_extra = 1;
  void tbeg(int kind) { if (timed_step && (kind == 0 || kind == timed_extra) && ev_used + 2 <= ev.size()) { HIPCHK(hipEventRecord(ev[ev_used], stream)); ev_kind.push_back(kind); } }
  void tend() { if (timed_step && ev_used + 2 <= ev.size() && ev_kind.size() == ev_used / 2 + 1) { HIPCHK(hipEventRecord(ev[ev_used + 1], stream)); ev_used += 2; } }

  CtxT() { HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); }
  ~CtxT() override {
    if (label_thr.joinable()) label_thr.join();
    if (stream) (void)hipStreamSynchronize(stream);   // nothing may still write the pinned words or the buffers
    tr.reset();
    for (auto e : ev) (void)hipEventDestroy(e);
    if (pin_ev) (void)hipHostFree(pin_ev);
    if (hflag) (void)hipHostFree(hflag);
    if (cstream) { (void)hipStreamSynchronize(cstream); (void)hipStreamDestroy(cstream); (void)hipEventDestroy(ev_fold); (void)hipEventDestroy(ev_halo); }
    if (stream) (void)hipStreamDestroy(stream);
  }
  void sync() override { HIPCHK(hipStreamSynchronize(stream)); }

  // ---- uploads ------------------------------------------------------------------------
  void setup_box() {
    const double rl = rc + skin_eff();
    int nc[3]; bool cells = true;
    for (int d = 0; d < 3; ++d) { nc[d] = (int)std::floor(L[d] / rl); if (nc[d] < 3) cells = false; }
    for (int d = 0; d < 3; ++d) {
      box.L[d] = (R)L[d]; box.invL[d] = (R)(1.0 / L[d]);
      box.nc[d] = cells ? nc[d] : 0;
      box.cell_inv[d] = (R)((cells ? nc[d] : 1) / L[d]);
      boxd.L[d] = L[d]; boxd.invL[d] = 1.0 / L[d];
      box.qs[d] = boxd.qs[d] = q_scale(d); box.qh[d] = boxd.qh[d] = 0.5 * L[d];
      box.qsf[d] = (R)q_scale(d); box.qinvf[d] = (R)(1.0 / q_scale(d));
    }
    box.ncell = cells ? nc[0] * nc[1] * nc[2] : 1;
    box.xs_nb = 1 << 20; box.xs_w = 1;      // (all tiles HX wide until pick_tile_split says otherwise)
    box.zghost = 0; box.z0g = 0; box.nzg = cells ? nc[2] : 1; box.shz_lo = 0; box.shz_hi = 0;
    if (dd_on) {
      if (!cells) throw ChemError(CHEM_EINVAL, "domain decomposition needs at least 3 cells of edge rc+skin per axis");
      nzg = nc[2];
      const int base = nzg / P, rem = nzg % P;
      if (base < 2) throw ChemError(CHEM_EINVAL, "domain decomposition: fewer than 2 cell layers per rank along z");
      ncz = base + (rk < rem ? 1 : 0);
      z0 = rk * base + std::min(rk, rem);
      lower = (rk + P - 1) % P; upper = (rk + 1) % P;
      box.zghost = 1; box.z0g = z0; box.nzg = nzg; box.nc[2] = ncz + 2;
      box.ncell = nc[0] * nc[1] * (ncz + 2);
      box.shz_lo = enc_shift(z0 == 0 ? -1.0 : 0.0, 2);
      box.shz_hi = enc_shift(z0 + ncz == nzg ? 1.0 : 0.0, 2);
    }
  }

  // cells, neighbour-list capacity and everything else that depends on box/cutoff/skin
  // Automatic list skin (measured on the 1M-particle melt, profiles/round3_list_skin.txt): two cells fewer per axis than the
  // workload's skin would give, i.e. ~0.2 sigma more skin at rc + skin = 2.8 -- the lists live ~60 % longer for ~20 % more
  // entries -- and always the whole cell edge (the slack between L / floor(L / rl) and rl is free skin).
  double pick_list_skin() const {
    if (opt_list_skin == 0.0 || opt_criterion != 0 || !opt_tiles || (!dd_on && !opt_fused)) return 0.0;
    const double rl = rc + skin;
    double edge = 1e300;
    for (int d = 0; d < 3; ++d) {
      int nc = (int)std::floor(L[d] / (opt_list_skin > 0 ? rc + opt_list_skin : rl));
      if (opt_list_skin < 0) { if ((dd_on ? nglob : n) < 100000) return 0.0; nc -= 2; }
      if (nc < 5) return 0.0;
      if (dd_on && d == 2 && nc / P < 2) return 0.0;      // (a slab needs two cell layers)
      edge = std::min(edge, L[d] / nc);
    }
    const double s = edge * (1.0 - 1e-9) - rc;
    return s > skin ? s : 0.0;
  }
  // Narrow tiles (Box::xs_nb / xs_w, md_kernels.hpp tile_xrange): option tile_split = nb * 10 + w, 0 = off (default).
  // Built to fill the last round of the force launch (1728 equal one-shot workgroups on 768 resident slots: 2.25 rounds of
  // work) with shorter jobs, and measured: no gain at any mix -- C5: 7811 steps/s unsplit, 7549 with 10 wide + 6 one-cell
  // tiles per row, 7721 with 11 + 3; 125k particles: 22980 unsplit, 20134 all one cell wide (profiles/round3_tile_split.txt).
  // The workgroups of the last round run faster on their emptier CUs than the model assumed; the extra staging is not paid back.
  void debug_tiles(int32_t* out) override {
    if (geom_dirty) setup_geometry();
    const int ntx = use_tiles ? tile_ntx(box.nc[0], box.xs_nb, box.xs_w) : 0;
    out[0] = ntiles; out[1] = box.nc[0]; out[2] = use_tiles ? tile_nbx(box.nc[0], box.xs_nb) : 0; out[3] = box.xs_w;
    out[4] = ntx ? ntiles / ntx : 0; out[5] = tile_cap;
  }
  void pick_tile_split() {
    box.xs_nb = 1 << 20; box.xs_w = 1;
    if (!use_tiles || opt_tile_split <= 0 || HX < 2) return;
    const int nb = opt_tile_split / 10, w = std::max(1, std::min(opt_tile_split % 10, HX));
    if (nb * HX < box.nc[0]) { box.xs_nb = nb; box.xs_w = w; }
  }
  void setup_geometry() {
    skin_list = pick_list_skin();
    setup_geometry_once();
    if (skin_list > 0 && !use_fused && !dd_on) { skin_list = 0.0; setup_geometry_once(); }   // the wider skin only pays on the fused tile path
  }
  void setup_geometry_once() {
    setup_box();
    cell_cnt.alloc(box.ncell + 1); cell_start.alloc(box.ncell + 1); cell_sub.alloc(box.ncell + 2);
    HIPCHK(hipMemsetAsync(cell_cnt.p, 0, sizeof(int) * (box.ncell + 1), stream));
    const double vol = L[0] * L[1] * L[2], rl = rc + skin_eff();
    const double expect = 4.0 / 3.0 * M_PI * rl * rl * rl * (dd_on ? nglob : n) / vol;
    int ncap = nl_capacity_user > 0 ? nl_capacity_user : (int)(expect * 1.6 + 48);
    ncap = std::min(ncap, std::max((dd_on ? nglob : n) - 1, 1));
    S = (ncap + 15) / 16 * 16;
    use_tiles = opt_tiles && box.nc[0] >= HX + 2 && box.nc[1] >= HY + 2 && (dd_on ? true : box.nc[2] >= HZ + 2);
    if (dd_on && !(box.nc[0] >= HX + 2 && box.nc[1] >= HY + 2)) throw ChemError(CHEM_EINVAL, "domain decomposition needs >= 5 cells along x and y");
    // (the per-cell kernels know nothing of ghost layers: with tiles=0 a slab used to run on and return wrong forces -- found by
    //  tests/test_gpu_sweep.py case 100)
    if (dd_on && !use_tiles) throw ChemError(CHEM_EINVAL, "domain decomposition needs the LDS-staged tiles: option tiles=0 is a single-domain switch");
    if (use_tiles) {
      // LDS capacity from the mean stencil occupancy (+12 % for density fluctuations), in 256-slot steps
      const double per_cell = dd_on ? (double)nglob / ((double)box.nc[0] * box.nc[1] * nzg) : (double)n / box.ncell;
      const int need = (int)(SX * SY * SZ * per_cell * 1.12) + 64;
      tile_cap = std::max(1024, (need + 255) / 256 * 256);
      // every kernel that stages a tile must fit: the force kernel's image AND the list build's (SoA groups + type masks +
      // slice boundaries: ~22 B per slot against 16), next to the static __shared__ of k_rebuild_fused / k_nlist_tiles
      if (tile_lds_need() > kTileLdsBudget) {   // cells too crowded: per-cell kernels
        if (dd_on) throw ChemError(CHEM_ENOSPC, "domain decomposition needs the LDS-staged tiles, and a stencil of this density does not fit the LDS");
        use_tiles = false;
      }
      else set_tile_lds_attr();
    }
    pick_tile_split();
    ntiles = use_tiles ? tile_ntx(box.nc[0], box.xs_nb, box.xs_w) * ((box.nc[1] + HY - 1) / HY) * (((dd_on ? ncz : box.nc[2]) + HZ - 1) / HZ) : 0;
    alloc_lists();
    setup_fused();
    setup_tile_order();
    HIPCHK(hipStreamSynchronize(stream));
    geom_dirty = false; resort = true;
  }

  // Fused rebuild kernel: the grid must be co-resident (grid barriers), so it is sized from the
  // occupancy of the kernel itself on this device.
  void setup_fused() {
    use_fused = false;
    if (!opt_fused || !use_tiles || dd_on) return;
    const void* fn = reinterpret_cast<const void*>(&k_rebuild_fused<R, 512>);
    // (a block the device refuses leaves the fused launch off -- the unfused chain takes over -- instead of failing the set-up)
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)list_lds_need()) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rebuild_fused<R, 512, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)list_lds_need()) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, list_lds_need(false)) != hipSuccess || per_cu < 1) return;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    int g = std::min(per_cu * prop.multiProcessorCount, 1024) / 8 * 8;
    if (g < 8) return;
    fused_grid = g;
    // the diagnostic instantiation (stamps, int32 rows) has its own resource usage and, in fp64, the larger LDS block
    int per_cu_d = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_d, reinterpret_cast<const void*>(&k_rebuild_fused<R, 512, true>), 512, list_lds_need(true)) != hipSuccess || per_cu_d < 1) return;
    fused_grid_diag = std::min(std::min(per_cu_d * prop.multiProcessorCount, 1024) / 8 * 8, fused_grid);
    if (fused_grid_diag < 8) return;
    auto shift_for = [&](int nitem, int lo, int maxseg) { int sh = lo; while (((nitem + (1 << sh) - 1) >> sh) > maxseg) ++sh; return sh; };
    // cell segments: 64 cells (the prefix inside a segment is one wave reduction in the sort phase), more when that
    // would give more than 1024 segments; tile segments: <= 1024 of them
    seg_shift = shift_for(box.ncell, 6, 1024); tseg_shift = shift_for(ntiles, 0, 1024);
    cell_loc.alloc(box.ncell + 1); seg_tot.alloc(1024); tile_n.alloc(ntiles + 1); tile_loc.alloc(ntiles + 1); tseg_tot.alloc(1024);
    // bucket rows of the binning pass: 64 members per cell = what one wave sorts (the LDS tiles already require a mean
    // cell occupancy far below that); a fuller cell raises ctl->bucket_overflow and the unfused chain takes over
    cell_n.alloc(box.ncell + 1);
    bcap = opt_bucket_cap > 0 ? std::min(opt_bucket_cap, 64) : 64;
    bucket.alloc((size_t)box.ncell * bcap);
    HIPCHK(hipMemsetAsync(bucket.p, 0, sizeof(int) * (size_t)box.ncell * bcap, stream));   // (the sort phase reads whole rows: entries must always be valid indices)
    if (!gbar.p) { gbar.alloc(1); HIPCHK(hipMemsetAsync(gbar.p, 0, sizeof(GridBar), stream)); }
    HIPCHK(hipMemsetAsync(seg_tot.p, 0, sizeof(int) * 1024, stream));
    use_fused = true;
  }
  DBuf<int> tile_pos, tile_ord;
  void setup_tile_order() {
    if (!use_tiles || dd_on || ntiles < 8) { tile_pos.free(); tile_ord.free(); return; }
    const int ntx = tile_ntx(box.nc[0], box.xs_nb, box.xs_w), nty = (box.nc[1] + HY - 1) / HY;
    auto home_cells = [&](int tile) {
      const int tx = tile % ntx, ty = (tile / ntx) % nty, tz = tile / (ntx * nty);
      int cx0, hx;
      tile_xrange(tx, box.nc[0], box.xs_nb, box.xs_w, cx0, hx);
      return hx * std::min(HY, box.nc[1] - ty * HY) * std::min(HZ, box.nc[2] - tz * HZ);
    };
    std::vector<int> ord(ntiles), pos(ntiles);
    const int q = ntiles >> 3, r = ntiles & 7;
    for (int x = 0, off = 0; x < 8; ++x) {      // the contiguous ranges xcd_remap hands to the XCDs
      const int cnt = q + (x < r ? 1 : 0);
      for (int k = 0; k < cnt; ++k) ord[off + k] = off + k;
      std::stable_sort(ord.begin() + off, ord.begin() + off + cnt, [&](int a_, int b_) { return home_cells(a_) > home_cells(b_); });
      for (int k = 0; k < cnt; ++k) pos[ord[off + k]] = off + k;
      off += cnt;
    }
    // (two copies each: the rebuild kernel double-buffers them by the parity of its rebuild count and rewrites them from the home counts)
    pos.insert(pos.end(), pos.begin(), pos.begin() + ntiles); ord.insert(ord.end(), ord.begin(), ord.begin() + ntiles);
    tile_pos.upload(pos, stream); tile_ord.upload(ord, stream);
  }
  void launch_rebuild_fused() {
    FusedArgs<R> a{};
    a.n = n; a.ncell = box.ncell; a.ntiles = ntiles; a.CAP = tile_cap; a.S = S; a.has_excl = has_excl; a.criterion = opt_criterion;
    a.par = fused_par; a.seg_shift = seg_shift; a.tseg_shift = tseg_shift; a.nblk = cdiv(n, kIntPerBlock); a.want32 = want32 ? 1 : 0; a.ntypes = ntypes; a.ablate = dbg_on ? opt_ablate_list : 0;
    a.half_skin = 0.5 * skin_eff(); a.rl2 = (R)((rc + skin_eff()) * (rc + skin_eff()));
    a.half_skin_ref = 0.5 * skin; a.rl2_rows = (R)((rc + skin) * (rc + skin));
    a.istep = step;
    a.x4 = x4.p; a.v4 = v4.p; a.x4o = x4o.p; a.v4o = v4o.p; a.x0 = opt_criterion ? x0.p : nullptr;      // (reference positions: only the displacement criterion reads them -- 16 MB of writes per rebuild otherwise)
    a.tag = tag.p; a.tago = tago.p; a.rtag = rtag.p; a.img4 = img4.p; a.img4o = img4o.p;
    a.cell_cnt = cell_cnt.p; a.cell_of = cell_of.p; a.slot_of = slot_of.p; a.cell_start = cell_start.p; a.cell_loc = cell_loc.p;
    a.cell_sub = cell_sub.p; a.cell_n = cell_n.p; a.bucket = bucket.p; a.bcap = bcap; a.btot = seg_tot.p; a.perm = perm.p; a.tn = tile_n.p; a.tloc = tile_loc.p; a.tbtot = tseg_tot.p;
    a.tile_pos = opt_lpt ? tile_pos.p : nullptr; a.tile_ord = opt_lpt ? tile_ord.p : nullptr;
    a.desc = tdesc.p; a.excl_start = excl_start.p; a.excl_list = excl_list.p; a.nl16 = nl16.p; a.nnh = nnh.p; a.nlist = nlist.p; a.nn = nn.p;
    a.blockmax = blockmax.p; a.ctl = ctl.p; a.gb = gbar.p; a.box = box; a.act = act;
    a.wgst = dbg_on && wgst.p ? wgst.p : nullptr;
    a.bstart = bstart.p; a.bent = bent.p; a.bwork = bwork.p; a.bj = bj.p; a.nbent = (int)std::min<int64_t>(nbent, 1 << 30);
    a.bslots = nullptr; a.bond_pass = 0;
    if (bonds_inline() && !(sizeof(R) == 8 && want32)) {   // (the exact fp64 builder of the int32 rows locates no slots: ensure_list32 rebuilds again)
      if (bslots.n < 2 * (size_t)n) bslots.alloc(2 * (size_t)n + 1024);      // two quads (kBondSlots = 8 words) per particle
      a.bslots = bslots.p;
      a.bond_pass = (bond_by_pass() && !want32) ? 1 : 0;    // (the int32 rows must leave the excluded pairs out: ensure_list32 rebuilds again behind them)
      if (excl_over == 0) a.nbent = 0;                      // no work list needed; otherwise it holds the owners with > kBondSlots exclusions only
    }
    if (dbg_on || want32) hipLaunchKernelGGL((k_rebuild_fused<R, 512, true>), dim3(fused_grid_diag), dim3(512), list_lds_need(true), stream, a);
    else hipLaunchKernelGGL((k_rebuild_fused<R, 512>), dim3(fused_grid), dim3(512), list_lds_need(false), stream, a);
    fused_par ^= 1;
  }

  void set_tile_lds_attr() {
    const int bytes = (int)tile_lds_bytes();
#define SETA(K) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, bytes))
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nlist_tiles<R, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)list_lds_need()));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_react_scan_tiles<R, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(scan_roles_staged() ? scan_lds_bytes() : tile_lds_bytes())));
#define SETB(T, E, M) SETA((k_pair_tiles<R, T, E, 256, M>)); SETA((k_pair_tiles<R, T, E, 512, M>)); SETA((k_pair_tiles<R, T, E, 1024, M>)); \
                      if (T == 1 && !E) SETA((k_pair_tiles<R, 1, false, 512, M, true>))
#define SETT(E, M) SETB(1, E, M); SETB(2, E, M); SETB(4, E, M); SETB(8, E, M)
    SETT(false, 2); SETT(false, 1); SETT(false, 0); SETT(true, 0);
#undef SETT
#undef SETB
#undef SETA
  }

  void alloc_lists() {
    nlist.free(); nl16.free();
    const size_t ac = (size_t)acap();
    nlist.alloc(ac * S);
    nnh.alloc(ac);
    if (use_tiles) { nl16.alloc(ac * S); HIPCHK(hipMemsetAsync(nl16.p, 0, ac * S * 2, stream)); tdesc.alloc(ntiles); }
    tm.nlist_capacity = S;
  }

  void upload_particles() {
    nglob = (int)top.n;
    n = nglob; G = 0; nglo = ngup = 0; cap = nglob;
    std::vector<V4> hx, hv; std::vector<int> ht; std::vector<int4> hi;
    if (dd_on) {
      skin_list = pick_list_skin();   // (the slab bounds below must be those of the geometry set up afterwards)
      setup_box();   // slab bounds (z0, ncz, nzg)
      // capacities: ghost layers are one cell layer each; reals fluctuate with the slab occupancy
      const double per_layer = (double)nglob / nzg;
      G = (int)(per_layer * 1.5) + 1024;
      mcap = std::max(4096, (int)(per_layer / 4));
      cap = 2 * G + (int)(per_layer * ncz * 1.2) + 2 * mcap + 4096;
      hx.assign(cap, V4{}); hv.assign(cap, V4{}); ht.assign(cap, 0); hi.assign(cap, make_int4(0, 0, 0, 0));
      int k = G;
      for (int t = 0; t < nglob; ++t) {
        double z = pos0[3 * t + 2];
        const double s = std::floor(z / L[2]);
        z -= s * L[2]; if (z >= L[2]) z -= L[2];
        int gz = (int)std::floor(z * nzg / L[2]); gz = std::min(std::max(gz, 0), nzg - 1);
        if (gz < z0 || gz >= z0 + ncz) continue;
        if (k >= cap - G) throw ChemError(CHEM_ENOSPC, "domain decomposition: slab holds more particles than the allocated capacity");
        // (x and y are folded by the first binning pass in the fp64 build; the fixed-point encoding wants them in the box now)
        const double fx_ = kFixed ? fold_coord(pos0[3 * t], L[0]) : pos0[3 * t], fy_ = kFixed ? fold_coord(pos0[3 * t + 1], L[1]) : pos0[3 * t + 1];
        hx[k].x = enc_pos(fx_, 0); hx[k].y = enc_pos(fy_, 1); hx[k].z = enc_pos(z, 2); hx[k].w = (R)top.type[t];
        if (kFixed) { hi[k].x = (int)std::llrint((pos0[3 * t] - fx_) / L[0]); hi[k].y = (int)std::llrint((pos0[3 * t + 1] - fy_) / L[1]); }
        hv[k].x = (R)vel0[3 * t]; hv[k].y = (R)vel0[3 * t + 1]; hv[k].z = (R)vel0[3 * t + 2]; hv[k].w = (R)top.mass[t];
        ht[k] = t; hi[k].z = (int)s;
        ++k;
      }
      n = k - G;
    } else {
      hx.resize(n); hv.resize(n); ht.resize(n); hi.assign(n, make_int4(0, 0, 0, 0));
      for (int t = 0; t < n; ++t) {
        if constexpr (kFixed) {   // folded on the host, images counted (the fp64 build leaves that to the first binning pass)
          for (int d = 0; d < 3; ++d) {
            const double x = pos0[3 * t + d], xf = fold_coord(x, L[d]);
            (&hx[t].x)[d] = enc_pos(xf, d); (&hi[t].x)[d] = (int)std::llrint((x - xf) / L[d]);
          }
        } else { hx[t].x = (R)pos0[3 * t]; hx[t].y = (R)pos0[3 * t + 1]; hx[t].z = (R)pos0[3 * t + 2]; }
        hx[t].w = (R)top.type[t];
        hv[t].x = (R)vel0[3 * t]; hv[t].y = (R)vel0[3 * t + 1]; hv[t].z = (R)vel0[3 * t + 2]; hv[t].w = (R)top.mass[t];
        ht[t] = t;
      }
    }
    const int ac = acap();
    x4.upload(hx, stream); v4.upload(hv, stream); tag.upload(ht, stream); img4.upload(hi, stream);
    {
      std::vector<int> hr(nglob, -1);
      for (int k = 0; k < n; ++k) hr[ht[G + k]] = G + k;
      rtag.upload(hr, stream);
    }
    f4.alloc(ac); x4o.alloc(ac); v4o.alloc(ac); tago.alloc(ac); img4o.alloc(ac); x0.alloc(ac);
    HIPCHK(hipMemcpyAsync(x0.p, x4.p, sizeof(V4) * ac, hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemsetAsync(f4.p, 0, sizeof(V4) * ac, stream));
    cell_of.alloc(ac); slot_of.alloc(ac); perm.alloc(ac); nn.alloc(ac);
    HIPCHK(hipMemsetAsync(nn.p, 0, sizeof(int) * ac, stream));
    ctl.alloc(1);
    HIPCHK(hipMemsetAsync(ctl.p, 0, sizeof(DevCtl), stream));
    blockmax.alloc(cdiv(ac, 256));
    HIPCHK(hipMemsetAsync(blockmax.p, 0, sizeof(unsigned long long) * cdiv(ac, 256), stream));
    eout.alloc(3 * (size_t)cdiv((long long)ac * 64, 256) + 8);
    ekout.alloc(4 * (size_t)cdiv(ac, 256) + 8);
    elist.alloc(CHEM_MAX_LISTS);
    redbuf.alloc(64);
    state.alloc(nglob); res_id.alloc(nglob); mol_id.alloc(nglob);
    if (dd_on) {
      const size_t mb = mig_bytes();
      for (int k = 0; k < 4; ++k) { mig[k].alloc(mb); HIPCHK(hipMemsetAsync(mig[k].p, 0, mb, stream)); }
    }
    HIPCHK(hipStreamSynchronize(stream));
    pos0.clear(); pos0.shrink_to_fit(); vel0.clear(); vel0.shrink_to_fit();
    particles_dirty = false; labels_dirty = true; geom_dirty = true; resort = true; device_ready = true;
  }

  // migration buffer layout: [count, pad x3][x4 x mcap][v4 x mcap][img4 x mcap][tag x mcap]
  size_t mig_bytes() const { return 16 + (size_t)mcap * (2 * sizeof(V4) + sizeof(int4) + sizeof(int)); }
  MigBuf<R> mig_view(int k) {
    unsigned char* b = mig[k].p;
    MigBuf<R> m;
    m.count = reinterpret_cast<int*>(b);
    m.x = reinterpret_cast<V4*>(b + 16);
    m.v = m.x + mcap;
    m.img = reinterpret_cast<int4*>(m.v + mcap);
    m.tag = reinterpret_cast<int*>(m.img + mcap);
    m.cap = mcap;
    return m;
  }

  void upload_labels() {
    if (state_mirror_stale) { std::vector<int> hs; state.download(hs, nglob, stream); top.state.assign(hs.begin(), hs.end()); state_mirror_stale = false; }
    state.upload(top.state, stream); res_id.upload(top.res_id, stream); mol_id.upload(top.mol_id, stream);
    HIPCHK(hipStreamSynchronize(stream));
    labels_dirty = false;
  }

  void upload_pair() {
    int nt = 1;
    for (int t : top.type) nt = std::max(nt, t + 1);
    for (int a = 0; a < CHEM_MAX_TYPES; ++a) for (int b = 0; b < CHEM_MAX_TYPES; ++b) if (pp[a][b].kind) nt = std::max(nt, std::max(a, b) + 1);
    for (auto& r : reactions) { nt = std::max(nt, std::max(r.new_type_1, r.new_type_2) + 1); }
    for (auto& r : nb_rules) nt = std::max(nt, r.new_type + 1);
    for (auto& c : atrp_centers) nt = std::max(nt, std::max(c.type, c.new_type) + 1);
    ntypes = nt;
    std::vector<PairCore<R>> hc((size_t)nt * nt);
    std::vector<PairExt<R>> he((size_t)nt * nt);
    std::vector<V4> htab;
    for (int a = 0; a < nt; ++a) for (int b = 0; b < nt; ++b) {
      const HostPairPot& p = pp[a][b];
      PairCore<R> c{(R)-1, 0, 0, 0}; PairExt<R> e{0, 0, 0, 0, 0, 0, 0, 0};
      if (p.kind == 1) {
        const double s6 = std::pow(p.sig, 6), s12 = s6 * s6;
        c.rc2 = (R)(p.rc * p.rc); c.lj1 = (R)(48.0 * p.eps * s12); c.lj2 = (R)(24.0 * p.eps * s6); c.kind = (R)1;
        e.e1 = (R)(4.0 * p.eps * s12); e.e2 = (R)(4.0 * p.eps * s6); e.shift = (R)p.shift;
      } else if (p.kind == 2) {
        c.rc2 = (R)(p.rc * p.rc); c.kind = (R)2;
        e.r0 = (R)p.r0; e.inv_dr = (R)(1.0 / p.dr); e.toff = (int)htab.size(); e.nrow = (int)p.e.size();
        for (size_t k = 0; k < p.e.size(); ++k) {
          const double fk = p.f[k], ek = p.e[k];
          const double fn = k + 1 < p.f.size() ? p.f[k + 1] : fk, en = k + 1 < p.e.size() ? p.e[k + 1] : ek;
          V4 row; row.x = (R)fk; row.y = (R)(fn - fk); row.z = (R)ek; row.w = (R)(en - ek);
          htab.push_back(row);
        }
      }
      hc[(size_t)a * nt + b] = c; he[(size_t)a * nt + b] = e;
    }
    lj_only = htab.empty();
    // type pairs with a potential; uniform LJ = every active pair shares one parameter set
    std::memset(&act, 0, sizeof(act));
    uniform_lj = lj_only;
    const HostPairPot* first = nullptr;
    for (int a = 0; a < CHEM_MAX_TYPES; ++a) for (int b = 0; b < CHEM_MAX_TYPES; ++b) {
      const HostPairPot& p = pp[a][b];
      if (!opt_skip_inactive || p.kind) act.row[a] |= 1u << b;
      if (p.kind == 1) {
        if (!first) first = &p;
        else if (p.eps != first->eps || p.sig != first->sig || p.rc != first->rc) uniform_lj = false;
      }
    }
    if (!opt_skip_inactive || !first) uniform_lj = false;
    all_active = true;   // every pair of types in use carries a potential: the list build skips the per-hit type filter
    for (int a = 0; a < nt; ++a) for (int b = 0; b < nt; ++b) if (!((act.row[a] >> b) & 1u)) all_active = false;
    if (uniform_lj) {
      const double s6 = std::pow(first->sig, 6), s12 = s6 * s6;
      uni.drc2 = first->rc * first->rc; uni.dlj1 = 48.0 * first->eps * s12; uni.dlj2 = 24.0 * first->eps * s6;
      uni.rc2 = (float)uni.drc2; uni.lj1 = (float)uni.dlj1; uni.lj2 = (float)uni.dlj2;
    }
    resort = true;   // the force list depends on the type-pair mask
    pcore.upload(hc, stream); pext.upload(he, stream); tab.upload(htab, stream);
    HIPCHK(hipStreamSynchronize(stream));
    pair_dirty = false;
  }

  // ---- per-tag CSR tables built on the device from flat, append-only arrays (md_kernels.hpp "Per-tag CSR tables") ----
  DBuf<int4> fent; DBuf<int> flist, eslot, bkey, tcnt, tcounts, type_tag; DBuf<SlotKey> skeys; DBuf<int2> epairs;
  size_t fent_n = 0, epairs_n = 0; std::vector<size_t> list_uploaded; int nslot = 0;
  DBuf<double2> btab_rows; DBuf<double4> btab_info; size_t btab_uploaded = 0;
  BTab btab_view() const { return BTab{btab_rows.p, btab_info.p}; }
  void upload_bond_tables() {
    if (btab_uploaded == top.btables.size()) return;
    std::vector<double2> rows; std::vector<double4> info;
    for (const HostBondTable& t : top.btables) {
      info.push_back(make_double4((double)rows.size(), (double)t.e.size(), t.r0, 1.0 / t.dr));
      for (size_t k = 0; k < t.e.size(); ++k) rows.push_back(make_double2(t.e[k], t.f[k]));
    }
    btab_rows.upload(rows, stream); btab_info.upload(info, stream);
    HIPCHK(hipStreamSynchronize(stream));
    btab_uploaded = top.btables.size();
  }
  template <typename T> static void grow(DBuf<T>& b, size_t used, size_t need, hipStream_t st) {   // keeps the first `used` elements
    if (need <= b.n) return;
    DBuf<T> nb; nb.alloc(std::max(need, b.n * 2 + 1024));
    if (used) HIPCHK(hipMemcpyAsync(nb.p, b.p, used * sizeof(T), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    std::swap(b.p, nb.p); std::swap(b.n, nb.n);
  }
  DBuf<int> scan_tot;
  void scan_counts(int n, int* cnt, int* start) {
    const int nb = cdiv(n, kScanItems);
    scan_tot.alloc(std::max(nb, 1));
    hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(1024), 0, stream, n, cnt, start, scan_tot.p);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, nb, scan_tot.p, n, start);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(1024), 0, stream, n, (const int*)scan_tot.p, start);
  }
  // full = true: every tuple of every list again (set-up, parameter changes); false: only what the lists gained
  // since the last call (reaction steps).  Parameter slots are re-resolved against the current particle types on
  // the device either way.
  // wait = false: everything is only enqueued (reaction step: the host goes on with its own bookkeeping meanwhile);
  // the table sizes the host needs (entry and owner counts) are upper bounds from the flat arrays, the kernels
  // themselves read the exact figures from device memory
  size_t fent_members = 0;     // sum of arity * (2 for quadruples) over the flat tuples
  std::vector<HBondedParam> stage_hp; std::vector<HostTopology::HSlotKey> stage_hk;
  PinnedVec<int4> stage_ne; PinnedVec<int> stage_nl; PinnedVec<int2> stage_ep;   // pinned: a pageable async copy blocks for ~0.1 ms
  void upload_bonded(bool full = true, bool wait = true) {
    upload_bond_tables();
    // (host staging lives in members: with wait = false the copies may still be in flight when this returns)
    std::vector<HBondedParam>& hp = stage_hp; std::vector<HostTopology::HSlotKey>& hk = stage_hk;
    top.build_params(hp, hk);
    static_assert(sizeof(HBondedParam) == sizeof(BondedParam) && sizeof(HostTopology::HSlotKey) == sizeof(SlotKey), "layout");
    nslot = (int)hp.size();
    // analytic pair terms only (the chain-growth systems): the per-step kernel without the angle / dihedral / table code
    bonds_only = nslot > 0;
    for (const auto& q : hp) bonds_only &= q.arity == 2 && (q.kind == CHEM_POT_HARMONIC || q.kind == CHEM_POT_FENE || q.kind == CHEM_POT_FENE_LJ || q.kind == CHEM_POT_LJ_BOND);
    harmonic_only = nslot == 1 && hp[0].arity == 2 && hp[0].kind == CHEM_POT_HARMONIC;      // ONE harmonic parameter set (kernel arguments of the force launch)
    if (harmonic_only) { inline_K = hp[0].p[0]; inline_r0 = hp[0].p[1]; }
    if (full && harmonic_only) {
      // every listed pair must be an excluded pair (ChemLab's DynamicExcludeList observes the bond lists; a caller of the C ABI
      // need not): checked once against the host's exclusion rows; bonds made by reactions are excluded by construction
      bonds_excluded = true;
      for (const auto& l : top.lists) {
        if (l.arity != 2) continue;
        for (size_t e = 0; e < (size_t)l.size() && bonds_excluded; ++e) {
          const int32_t a_ = l.ent[2 * e], b_ = l.ent[2 * e + 1];
          bonds_excluded = std::binary_search(top.excl[a_].begin(), top.excl[a_].end(), b_);
        }
      }
    }
    bpar.alloc(std::max<size_t>(hp.size(), 1)); skeys.alloc(std::max<size_t>(hk.size(), 1));
    if (nslot) {
      HIPCHK(hipMemcpyAsync(bpar.p, hp.data(), hp.size() * sizeof(BondedParam), hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemcpyAsync(skeys.p, hk.data(), hk.size() * sizeof(SlotKey), hipMemcpyHostToDevice, stream));
    }
    if (full || list_uploaded.size() != top.lists.size()) { fent_n = 0; fent_members = 0; list_uploaded.assign(top.lists.size(), 0); }
    PinnedVec<int4>& ne = stage_ne; PinnedVec<int>& nl = stage_nl;
    ne.clear(); nl.clear();
    bool typed = false;
    for (size_t li = 0; li < top.lists.size(); ++li) {
      const HostList& l = top.lists[li];
      typed |= l.by_types != 0;
      const size_t have = (size_t)l.size();
      for (size_t e = list_uploaded[li]; e < have; ++e) {
        const int32_t* t = &l.ent[e * l.arity];
        ne.push_back(make_int4(t[0], t[1], l.arity > 2 ? t[2] : -1, l.arity > 3 ? t[3] : -1)); nl.push_back((int)li);
      }
      fent_members += (have - list_uploaded[li]) * (size_t)(l.arity == 4 ? 8 : l.arity);
      list_uploaded[li] = have;
    }
    grow(fent, fent_n, fent_n + ne.size(), stream); grow(flist, fent_n, fent_n + ne.size(), stream);
    if (!ne.empty()) {
      HIPCHK(hipMemcpyAsync(fent.p + fent_n, ne.data(), ne.size() * sizeof(int4), hipMemcpyHostToDevice, stream));
      HIPCHK(hipMemcpyAsync(flist.p + fent_n, nl.data(), nl.size() * sizeof(int), hipMemcpyHostToDevice, stream));
      fent_n += ne.size();
    }
    const int nt = nglob > 0 ? nglob : (int)top.n;
    if (tcnt.n < (size_t)nt + 1) { tcnt.alloc((size_t)nt + 1); HIPCHK(hipMemsetAsync(tcnt.p, 0, sizeof(int) * ((size_t)nt + 1), stream)); }
    bstart.alloc((size_t)nt + 1); tcounts.alloc(2);
    if (typed) type_tag.upload(top.type, stream);      // typed lists: the host keeps the type mirrors current (react_step)
    HIPCHK(hipMemsetAsync(tcounts.p, 0, 2 * sizeof(int), stream));
    const size_t ub = (fent_n && nslot) ? fent_members : 0;      // upper bound of the entry count (tuples without parameters drop out)
    if (ub) {
      const int ne_i = (int)fent_n;
      if (eslot.n < fent_n) eslot.alloc(fent_n + fent_n / 2 + 1024);          // (amortised: these grow at every reaction step)
      if (bent.n < ub) { bent.alloc(ub + ub / 2 + 1024); bkey.alloc(ub + ub / 2 + 1024); }
      hipLaunchKernelGGL((k_bt_count<R>), dim3(cdiv(ne_i, 256)), dim3(256), 0, stream, ne_i, fent.p, flist.p, nslot, skeys.p, x4.p, rtag.p,
                         typed ? (const int*)type_tag.p : (const int*)nullptr, eslot.p, tcnt.p);
      scan_counts(nt, tcnt.p, bstart.p);
      hipLaunchKernelGGL(k_bt_fill, dim3(cdiv(ne_i, 256)), dim3(256), 0, stream, ne_i, fent.p, eslot.p, skeys.p, bstart.p, tcnt.p, bent.p, bkey.p);
      hipLaunchKernelGGL(k_bt_sort, dim3(cdiv(nt, 256)), dim3(256), 0, stream, nt, bstart.p, tcnt.p, bent.p, bkey.p, tcounts.p);
    } else HIPCHK(hipMemsetAsync(bstart.p, 0, sizeof(int) * ((size_t)nt + 1), stream));
    nbent = (int64_t)ub; nb_owner = (int)std::min<size_t>((size_t)nt, ub);
    if (bwork.n < (size_t)std::max(nb_owner, 1)) bwork.alloc((size_t)nb_owner + nb_owner / 2 + 1024);
    if (bj.n < (size_t)std::max<int64_t>(nbent, 1)) bj.alloc((size_t)nbent + nbent / 2 + 1024);
    bwork_dirty = true;
    bonded_dirty = false;
    if (wait) HIPCHK(hipStreamSynchronize(stream));
  }

  // Reaction steps append to these tables; their first growth (device reallocation + copy, pinned reallocation of the
  // staging vectors: milliseconds each) is paid here, once, when a run with reactions starts -- sized for one bond
  // per particle, beyond that they grow geometrically as before.
  bool react_tables_reserved = false;
  std::vector<Candidate> bond_ev;       // bond-forming events of a reaction step in canonical order
  std::vector<Candidate> radix_tmp;     // scratch of the event sort (kept: a fresh 6 MB vector per step is 1.5 ms of page faults)
  void reserve_reaction_tables() {
    if (react_tables_reserved) return;
    react_tables_reserved = true;
    const size_t nt = (size_t)(nglob > 0 ? nglob : (int)top.n);
    const size_t per = top.spawns_tuples() ? 4 : 1;      // tuples a new bond may add (itself + spawned angles / dihedrals)
    for (auto& d : reactions) if (!d.is_virtual && d.bond_list >= 0 && d.bond_list < (int)top.lists.size()) {
      HostList& l = top.lists[d.bond_list];
      l.seen.reserve(l.seen.used + nt); l.ent.reserve(l.ent.size() + 2 * nt);
    }
    radix_tmp.reserve(nt / 2 + 1024); bond_ev.reserve(nt / 2 + 1024);
    stage_ne.reserve(nt / 2 * per + 1024); stage_nl.reserve(nt / 2 * per + 1024); stage_ep.reserve(nt / 2 * per + 1024);
    grow(fent, fent_n, fent_n + nt * per, stream); grow(flist, fent_n, fent_n + nt * per, stream);
    grow(epairs, epairs_n, epairs_n + nt * per, stream);
    const size_t ub = fent_members + 2 * nt * per;
    if (eslot.n < fent_n + nt * per) eslot.alloc(fent_n + nt * per);
    if (bent.n < ub) { DBuf<BondedEntry> nb_; nb_.alloc(ub); if (nbent > 0) HIPCHK(hipMemcpyAsync(nb_.p, bent.p, (size_t)nbent * sizeof(BondedEntry), hipMemcpyDeviceToDevice, stream));
                       HIPCHK(hipStreamSynchronize(stream)); std::swap(bent.p, nb_.p); std::swap(bent.n, nb_.n); bkey.alloc(ub); }
    if (bwork.n < nt) { bwork.alloc(nt); bwork_dirty = true; }
    if (bj.n < ub) { bj.alloc(ub); bwork_dirty = true; }
    if (excl_list.n < 3 * (epairs_n + nt * per) + 1024) {
      DBuf<int> nl_; nl_.alloc(3 * (epairs_n + nt * per) + 1024);
      if (epairs_n) HIPCHK(hipMemcpyAsync(nl_.p, excl_list.p, std::min(excl_list.n, 2 * epairs_n) * sizeof(int), hipMemcpyDeviceToDevice, stream));
      HIPCHK(hipStreamSynchronize(stream)); std::swap(excl_list.p, nl_.p); std::swap(excl_list.n, nl_.n);
    }
  }

  void upload_excl(bool full = true, bool wait = true) {
    if (full) epairs_n = 0;
    const size_t have = top.excl_log.size();
    const size_t add = have - epairs_n;
    grow(epairs, epairs_n, have, stream);
    static_assert(sizeof(std::pair<int32_t, int32_t>) == sizeof(int2), "layout");
    if (add) {
      stage_ep.resize(add);
      std::memcpy(stage_ep.data(), top.excl_log.data() + epairs_n, add * sizeof(int2));
      HIPCHK(hipMemcpyAsync(epairs.p + epairs_n, stage_ep.data(), add * sizeof(int2), hipMemcpyHostToDevice, stream));
    }
    epairs_n = have;
    const int nt = nglob > 0 ? nglob : (int)top.n;
    if (tcnt.n < (size_t)nt + 1) { tcnt.alloc((size_t)nt + 1); HIPCHK(hipMemsetAsync(tcnt.p, 0, sizeof(int) * ((size_t)nt + 1), stream)); }
    excl_start.alloc((size_t)nt + 1);
    if (excl_list.n < std::max<size_t>(2 * have, 1)) excl_list.alloc(3 * have + 1024);
    if (have) {
      const int m = (int)have;
      hipLaunchKernelGGL(k_ex_count, dim3(cdiv(m, 256)), dim3(256), 0, stream, m, epairs.p, tcnt.p);
      scan_counts(nt, tcnt.p, excl_start.p);
      hipLaunchKernelGGL(k_ex_fill, dim3(cdiv(m, 256)), dim3(256), 0, stream, m, epairs.p, excl_start.p, tcnt.p, excl_list.p);
      hipLaunchKernelGGL(k_ex_sort, dim3(cdiv(nt, 256)), dim3(256), 0, stream, nt, excl_start.p, tcnt.p, excl_list.p);
    } else HIPCHK(hipMemsetAsync(excl_start.p, 0, sizeof(int) * ((size_t)nt + 1), stream));
    if (wait) HIPCHK(hipStreamSynchronize(stream));
    has_excl = have ? 1 : 0;
    excl_dirty = false; resort = true;
  }

  void flush_host_state() {
    if (top.n <= 0 || !(L[0] > 0) || !(rc > 0)) throw ChemError(CHEM_ESTATE, "system incomplete: need box, cutoff and particles");
    if (particles_dirty) upload_particles();
    if (geom_dirty) setup_geometry();
    if (labels_dirty) upload_labels();
    if (pair_dirty) upload_pair();
    if (bonded_dirty) upload_bonded();
    if (excl_dirty) upload_excl();
  }

  // ---- rebuild chain (every kernel early-exits unless ctl->need_rebuild) ---------------
  // Grids are capped ("persistent" grid-stride kernels) so that the early-exit launches of the
  // steps that do not rebuild cost ~2 us each instead of a full-size dispatch.
  void launch_sort_chain(int i0, int npart) {
    const int nb = std::max(1, std::min(cdiv(npart, 256), 2048));
    DevCtl* c = ctl.p;
    MigBuf<R> mdn{}, mup{};
    if (dd_on) { mdn = mig_view(0); mup = mig_view(1); }
    hipLaunchKernelGGL(k_bin<R>, dim3(nb), dim3(256), 0, stream, i0, npart, x4.p, v4.p, tag.p, img4.p, box, cell_cnt.p, cell_of.p, slot_of.p, mdn, mup, c);
  }
  void launch_list_chain() {
    DevCtl* c = ctl.p;
    const R rl2 = (R)((rc + skin_eff()) * (rc + skin_eff())), rl2_rows = (R)((rc + skin) * (rc + skin));   // (they differ on the tile path only)
    if (use_tiles) {
      uint4* bs = nullptr;      // inline bonds (decomposed path: the standalone list kernel records the partner slots ...)
      int list_excl = has_excl;
      if (dd_on && bonds_inline()) {
        if (bslots.n < 2 * (size_t)acap()) bslots.alloc(2 * (size_t)acap() + 1024);
        bs = bslots.p;
        // (... unless the bond pass is on: the list build ignores the exclusions = bonds, the force launch behind this rebuild
        //  records the partner slots and every force launch takes the partners' pair term out again, as on the fused path)
        if (bond_by_pass() && !want32) { bs = nullptr; list_excl = 0; dd_record_bonds = true; }
      }
      hipLaunchKernelGGL((k_tile_desc<R>), dim3(std::min(ntiles, 2048)), dim3(128), 0, stream, ntiles, tile_cap, cell_start.p, box, tdesc.p, c, (const int*)cell_sub.p);
      hipLaunchKernelGGL((k_tile_scan<R>), dim3(1), dim3(1024), 0, stream, ntiles, tdesc.p, c);
      // slabs: the excluded partners of a particle are located as slots like on the fused path (ghost copies through gtag) --
      // without it every hit of a particle with exclusions paid a tag gather and a row scan (late stage: this launch 0.36 -> 1.59 ms)
      const Box<R>* bxp = nullptr;
      if (dd_on && has_excl && opt_dd_fastx && gtag.p && !(bond_by_pass() && !want32)) {
        if (!box_dev.p) box_dev.alloc(1);
        if (!box_dev_valid || std::memcmp(&box, &box_dev_host, sizeof(box)) != 0) {
          std::memcpy(&box_dev_host, &box, sizeof(box)); box_dev_valid = true;
          HIPCHK(hipMemcpyAsync(box_dev.p, &box_dev_host, sizeof(box), hipMemcpyHostToDevice, stream));
        }
        bxp = box_dev.p;
      }
      hipLaunchKernelGGL((k_nlist_tiles<R, 512>), dim3(ntiles), dim3(512), list_lds_need(want32), stream, ntiles, tile_cap, x4.p, tag.p, tdesc.p, rl2,
                         excl_start.p, excl_list.p, list_excl, act, ntypes, nl16.p, S, nnh.p, want32 ? nlist.p : (int*)nullptr, S, nn.p, c, rl2_rows, bs,
                         bxp, (const int*)rtag.p, (const int*)gtag.p, G, G + n);
    } else if (box.nc[0] > 0)
      hipLaunchKernelGGL((k_nlist_cells<R, 1536>), dim3(std::min(box.ncell, 2560)), dim3(256), 0, stream, n, x4.p, tag.p, cell_start.p, box, rl2,
                         excl_start.p, excl_list.p, has_excl, nlist.p, nn.p, S, c);
    else
      hipLaunchKernelGGL(k_nlist_brute<R>, dim3(cdiv(n, 4)), dim3(256), 0, stream, n, x4.p, tag.p, box, rl2, excl_start.p,
                         excl_list.p, has_excl, nlist.p, nn.p, S, c);
  }
  void launch_rebuild_chain() {   // single domain
    const int nb = std::min(cdiv(n, 256), 2048);
    DevCtl* c = ctl.p;
    launch_sort_chain(0, n);
    hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(1024), 0, stream, box.ncell, 0, cell_cnt.p, cell_start.p, c);
    hipLaunchKernelGGL(k_place, dim3(nb), dim3(256), 0, stream, 0, n, cell_of.p, slot_of.p, cell_start.p, perm.p, c);
    hipLaunchKernelGGL(k_sort_gather<R>, dim3(std::min(cdiv(box.ncell, 4), 2048)), dim3(256), 0, stream, box.ncell, cell_start.p, perm.p,
                       x4.p, v4.p, tag.p, img4.p, x4o.p, v4o.p, tago.p, img4o.p, c, box, cell_sub.p);
    hipLaunchKernelGGL(k_copyback<R>, dim3(nb), dim3(256), 0, stream, cell_start.p, 0, box.ncell, x4o.p, v4o.p, tago.p, img4o.p, x4.p, v4.p, tag.p, img4.p, rtag.p, x0.p, c);
    launch_list_chain();
  }

  void decide_and_rebuild() {
    if (use_fused) { tbeg(1); launch_rebuild_fused(); tend(); return; }
    hipLaunchKernelGGL(k_rebuild_decide<R>, dim3(1), dim3(1024), 0, stream, ctl.p, blockmax.p, cdiv(n, kIntPerBlock), 0.5 * skin_eff(), opt_criterion, 3, (const double*)nullptr, 0, (volatile int*)nullptr, 0, 0.5 * skin);
    launch_rebuild_chain();
  }

  // ---- slab decomposition ---------------------------------------------------------------
  void ensure_hflag() {
    if (hflag) return;
    HIPCHK(hipHostMalloc((void**)&hflag, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(hflag, 0, 4096);
    HIPCHK(hipHostGetDevicePointer((void**)&hflag_dev, hflag, 0));
    dd_vals.alloc(64 * kFoldSlots);
    foldmax.alloc(kFoldSlots);
    HIPCHK(hipMemsetAsync(foldmax.p, 0, kFoldSlots * sizeof(unsigned long long), stream));
  }
  void poll_ticket(volatile int* word, int ticket, const char* what) {
    long long spins = 0;
    while (*word != ticket) {
      if ((++spins & 0xfffff) == 0 && hipStreamQuery(stream) != hipErrorNotReady) {   // stream drained or failed: re-check once, then give up
        if (*word == ticket) break;
        HIPCHK(hipStreamSynchronize(stream));
        if (*word != ticket) throw ChemError(CHEM_EDEVICE, std::string(what) + " never arrived from the device");
      }
    }
  }
  // up to 8 device integers with one poll instead of one blocking copy each
  int collect_ticket = 0;
  void collect_ints(std::initializer_list<const int*> src, int* out) {
    ensure_hflag();
    CollectArgs a{}; a.n = 0;
    for (const int* p : src) a.src[a.n++] = p;
    const int ticket = ++collect_ticket;
    hipLaunchKernelGGL(k_collect_ints, dim3(1), dim3(64), 0, stream, a, (volatile int*)(hflag_dev + 16), ticket);
    poll_ticket(hflag + 16, ticket, "slab rebuild counters");
    for (int k = 0; k < a.n; ++k) out[k] = hflag[17 + k];
  }
  void set_need_rebuild_async(int v) { HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&ctl.p->need_rebuild, v, 1, stream)); }

  int read_int(const int* dev) {
    int v = 0;
    HIPCHK(hipMemcpyAsync(&v, dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return v;
  }

  // Host-synchronous rebuild of one slab: migrate leavers to the z-neighbours, sort the reals,
  // exchange the boundary layers as ghosts (contiguous slices of the cell-sorted arrays), rebuild
  // cells/tiles/lists.  Every rank enters together (the trigger is a cross-rank maximum).
  void rebuild_dd() {
    Trace trd("dd");
    const int nxy = box.nc[0] * box.nc[1];
    // (need_rebuild on, force_rebuild off: the request is being served; migration headers cleared)
    hipLaunchKernelGGL(k_dd_begin, dim3(1), dim3(64), 0, stream, ctl.p, reinterpret_cast<int*>(mig[0].p), reinterpret_cast<int*>(mig[1].p));
    // 1. bin the reals; leavers go to the migration buffers
    launch_sort_chain(G, n);
    // 2. migration exchange (fixed-capacity buffers, count in the header)
    tr->exchange(mig[0].p, mig_bytes(), mig[1].p, mig_bytes(), mig[2].p, mig_bytes(), mig[3].p, mig_bytes(), lower, upper, stream);
    int cnt2[2];
    collect_ints({reinterpret_cast<const int*>(mig[2].p), reinterpret_cast<const int*>(mig[3].p)}, cnt2);
    const int from_up = cnt2[0], from_lo = cnt2[1];
    trd.lap("bin+migrate+poll");
    if (from_up > mcap || from_lo > mcap) throw ChemError(CHEM_ENOSPC, "domain decomposition: migration buffer overflow");
    if (G + n + from_lo + from_up > cap - G) throw ChemError(CHEM_ENOSPC, "domain decomposition: slab capacity exceeded by arrivals");
    // 3. arrivals behind the current reals, then bin them too (they are in their own slab now)
    if (from_lo) hipLaunchKernelGGL(k_append_arrivals<R>, dim3(cdiv(from_lo, 256)), dim3(256), 0, stream, mig_view(3), from_lo, G + n, x4.p, v4.p, tag.p, img4.p);
    if (from_up) hipLaunchKernelGGL(k_append_arrivals<R>, dim3(cdiv(from_up, 256)), dim3(256), 0, stream, mig_view(2), from_up, G + n + from_lo, x4.p, v4.p, tag.p, img4.p);
    const int npend = n + from_lo + from_up;
    HIPCHK(hipMemsetAsync(mig[0].p, 0, 16, stream)); HIPCHK(hipMemsetAsync(mig[1].p, 0, 16, stream));
    if (from_lo + from_up) launch_sort_chain(G + n, from_lo + from_up);
    // 4. sort the reals into [G, G + n_new)
    DevCtl* c = ctl.p;
    const int nb = std::max(1, std::min(cdiv(npend, 256), 2048));
    hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(1024), 0, stream, box.ncell, G, cell_cnt.p, cell_start.p, c);
    hipLaunchKernelGGL(k_place, dim3(nb), dim3(256), 0, stream, G, npend, cell_of.p, slot_of.p, cell_start.p, perm.p, c);
    hipLaunchKernelGGL(k_sort_gather<R>, dim3(std::min(cdiv(box.ncell, 4), 2048)), dim3(256), 0, stream, box.ncell, cell_start.p, perm.p,
                       x4.p, v4.p, tag.p, img4.p, x4o.p, v4o.p, tago.p, img4o.p, c, box, cell_sub.p);
    // the ghost layers are filled by the neighbours' particles afterwards: no sub-bin information for their cells
    if (gtag.n < (size_t)nglob) gtag.alloc(nglob);
    hipLaunchKernelGGL(k_dd_clear, dim3(cdiv(std::max(nglob, nxy), 256)), dim3(256), 0, stream, rtag.p, gtag.p, nglob, cell_sub.p, cell_sub.p + (size_t)(ncz + 1) * nxy, nxy);
    hipLaunchKernelGGL(k_copyback<R>, dim3(nb), dim3(256), 0, stream, cell_start.p, nxy, (ncz + 1) * nxy, x4o.p, v4o.p, tago.p, img4o.p, x4.p, v4.p, tag.p, img4.p, rtag.p, x0.p, c);
    // 5. boundary-layer counts to the neighbours
    lcnt_dn.alloc(nxy + 1); lcnt_up.alloc(nxy + 1); gcnt_lo.alloc(nxy + 1); gcnt_up.alloc(nxy + 1);
    hipLaunchKernelGGL(k_layer_counts, dim3(cdiv(nxy + 1, 256)), dim3(256), 0, stream, nxy, ncz, cell_start.p, lcnt_dn.p, lcnt_up.p);
    tr->exchange(lcnt_dn.p, (nxy + 1) * sizeof(int), lcnt_up.p, (nxy + 1) * sizeof(int), gcnt_up.p, (nxy + 1) * sizeof(int), gcnt_lo.p,
                 (nxy + 1) * sizeof(int), lower, upper, stream);
    int c7[7];
    collect_ints({cell_start.p + (size_t)(ncz + 1) * nxy, reinterpret_cast<const int*>(mig[0].p), reinterpret_cast<const int*>(mig[1].p),
                  lcnt_dn.p + nxy, lcnt_up.p + nxy, gcnt_lo.p + nxy, gcnt_up.p + nxy}, c7);
    trd.lap("sort+counts+poll");
    n = c7[0] - G;
    if (c7[1] || c7[2]) throw ChemError(CHEM_ESTATE, "domain decomposition: a migrated particle left its new slab immediately");
    halo_dn_off = G; halo_dn_cnt = c7[3];
    halo_up_cnt = c7[4]; halo_up_off = G + n - halo_up_cnt;
    nglo = c7[5]; ngup = c7[6];
    if (nglo > G || G + n + ngup > cap) throw ChemError(CHEM_ENOSPC, "domain decomposition: ghost layer exceeds the reserved capacity");
    // 6. ghost particles: contiguous slices, received in place (lower ghosts right-aligned in front of the reals)
    tr->exchange2(Transport::Msg{x4.p + halo_dn_off, halo_dn_cnt * sizeof(V4), x4.p + halo_up_off, halo_up_cnt * sizeof(V4), x4.p + G + n, ngup * sizeof(V4),
                                 x4.p + G - nglo, nglo * sizeof(V4)},
                  Transport::Msg{tag.p + halo_dn_off, halo_dn_cnt * sizeof(int), tag.p + halo_up_off, halo_up_cnt * sizeof(int), tag.p + G + n, ngup * sizeof(int),
                                 tag.p + G - nglo, nglo * sizeof(int)}, lower, upper, stream);
    hipLaunchKernelGGL(k_ghost_cells, dim3(1), dim3(1024), 0, stream, nxy, ncz, gcnt_lo.p, gcnt_up.p, cell_start.p);
    if (nglo + ngup) hipLaunchKernelGGL(k_ghost_rtag, dim3(cdiv(nglo + ngup, 256)), dim3(256), 0, stream, G - nglo, nglo, G + n, ngup, tag.p, rtag.p, gtag.p);
    // 7. tiles + lists over the own layers.  A stencil beyond the LDS tile capacity or a row beyond its stride is a LOCAL matter
    //    (both capacities are per rank, no collective depends on them): grow and build the lists again, here, instead of
    //    carrying the flag to the end of the call and failing there (tests/test_gpu_sweep.py: a trimer melt at rho = 0.3
    //    outgrew the capacity estimated from its first configuration in the middle of a run)
    for (int attempt = 0; attempt < 4; ++attempt) {
      launch_list_chain();
      if (!use_tiles) break;
      int ov[2];
      collect_ints({&ctl.p->stage_overflow, &ctl.p->nl_overflow}, ov);
      if (!ov[0] && !ov[1]) break;
      if (ov[0]) {
        const int want = (ov[0] + ov[0] / 8 + 255) / 256 * 256, old_cap = tile_cap;
        if (want <= tile_cap) break;
        tile_cap = want;
        if (tile_lds_need() > kTileLdsBudget) { tile_cap = old_cap; break; }      // (stays flagged: reported by check_flags / rebuild_now)
        set_tile_lds_attr();
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&ctl.p->stage_overflow, 0, 1, stream));
        if (g_trace) fprintf(stderr, "[chem trace] slab rebuild: staged-tile capacity %d -> %d slots\n", old_cap, tile_cap);
      }
      if (ov[1]) {
        if (nl_capacity_user > 0) break;
        S = std::min(((int)(ov[1] * 1.25) + 31) / 16 * 16, std::max((nglob + 15) / 16 * 16, 16));
        alloc_lists();
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&ctl.p->nl_overflow, 0, 1, stream));
        if (g_trace) fprintf(stderr, "[chem trace] slab rebuild: list rows grown to %d entries\n", S);
      }
    }
    set_need_rebuild_async(0);
    if (g_trace) { HIPCHK(hipStreamSynchronize(stream)); trd.lap("ghosts+tiles+list"); }
    bwork_dirty = true;   // particle order and ghosts changed: the bonded work list is rebuilt before the next force evaluation
    ++dd_rebuilds;
  }

  // per-step ghost position update: the same contiguous slices, straight into the ghost ranges
  void halo_update() {
    tr->exchange(x4.p + halo_dn_off, halo_dn_cnt * sizeof(V4), x4.p + halo_up_off, halo_up_cnt * sizeof(V4), x4.p + G + n, ngup * sizeof(V4),
                 x4.p + G - nglo, nglo * sizeof(V4), lower, upper, stream);
  }

  DevCtl read_ctl() {
    DevCtl h;
    HIPCHK(hipMemcpyAsync(&h, ctl.p, sizeof(DevCtl), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return h;
  }
  template <typename F> void set_ctl_field(F DevCtl::*field, F value) {
    DevCtl* base = nullptr;
    size_t off = (size_t)((char*)&(base->*field) - (char*)base);
    HIPCHK(hipMemcpyAsync((char*)ctl.p + off, &value, sizeof(F), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
  }

  // forced synchronous rebuild (run() prologue, observe); grows the row stride on overflow
  void rebuild_now() {
    const double t0 = now_s();
    for (int attempt = 0; attempt < 6; ++attempt) {
      if (dd_on) {      // (a slab rebuild the host calls for, not one of the device's decisions: counted on the host)
        set_ctl_field(&DevCtl::force_rebuild, 0); set_ctl_field(&DevCtl::acc_maxdist, 0.0); set_ctl_field(&DevCtl::acc_ref, 0.0);
        HIPCHK(hipMemsetAsync(ctl.p->acc_pp, 0, 2 * sizeof(double), stream));
        if (attempt == 0) ++dd_direct_rebuilds;
        rebuild_dd();
      }
      else { set_ctl_field(&DevCtl::force_rebuild, 1); decide_and_rebuild(); }
      DevCtl h = read_ctl();
      if (dd_on) agree_flags(h);
      if (h.halt) set_ctl_field(&DevCtl::halt, 0);       // (a launch that could not finish its lists also stopped the run: this IS the recovery)
      if (h.mig_error) throw ChemError(CHEM_ESTATE, "domain decomposition: particle migration error " + std::to_string(h.mig_error));
      if (h.barrier_timeout) throw ChemError(CHEM_ESTATE, "fused rebuild: grid barrier timed out (device shared with another job?); set option fused_rebuild=0");
      if (h.bucket_overflow) {
        // a cell more crowded than a bucket row: widen the rows up to what one wave sorts (64), beyond that this
        // system needs the unfused chain (its crowded-cell path ranks through global memory)
        set_ctl_field(&DevCtl::bucket_overflow, 0);
        set_ctl_field(&DevCtl::halt, 0);       // (the launch that met the full row also stopped the run: this IS the recovery)
        if (h.bucket_overflow <= 64 && bcap < 64) {
          bcap = 64; bucket.alloc((size_t)box.ncell * bcap);
          HIPCHK(hipMemsetAsync(bucket.p, 0, sizeof(int) * (size_t)box.ncell * bcap, stream));   // (the sort phase reads whole rows: entries must always be valid indices)
        }
        else { use_fused = false; if (g_trace) fprintf(stderr, "[chem trace] cell with %d particles: fused rebuild off\n", h.bucket_overflow); }
        continue;
      }
      if (h.stage_overflow) {
        // denser than the mean-occupancy estimate (clusters, chains): grow the staged-tile capacity while it
        // fits the LDS, then build again; beyond that the system is too crowded for tiles
        const int want = (h.stage_overflow + h.stage_overflow / 8 + 255) / 256 * 256;
        const int old_cap = tile_cap;
        if (use_tiles && want > tile_cap) {
          tile_cap = want;
          if (tile_lds_need() <= kTileLdsBudget) {
            set_tile_lds_attr(); setup_fused();
            set_ctl_field(&DevCtl::stage_overflow, 0);
            if (g_trace) fprintf(stderr, "[chem trace] staged-tile capacity %d -> %d slots\n", old_cap, tile_cap);
            continue;
          }
          tile_cap = old_cap;
          if (!dd_on) {
            // the stencil no longer fits the LDS next to the list build's image: this system continues on the per-cell
            // kernels (int32 rows, positions through L1/L2) instead of failing
            use_tiles = false; use_fused = false; ntiles = 0;
            alloc_lists();
            set_ctl_field(&DevCtl::stage_overflow, 0);
            if (g_trace) fprintf(stderr, "[chem trace] stencil of %d particles does not fit the LDS: per-cell kernels\n", h.stage_overflow);
            continue;
          }
        }
        throw ChemError(CHEM_ENOSPC, "cell stencil holds " + std::to_string(h.stage_overflow) + " particles, LDS tile capacity " + std::to_string(use_tiles ? tile_cap : 1536) + " (local density too high for the LDS-staged tiles; set option tiles=0)");
      }
      if (!h.nl_overflow) { resort = false; tm.rebuild_wall_s += now_s() - t0; return; }
      int newS = ((int)(h.nl_overflow * 1.25) + 31) / 16 * 16;
      if (nl_capacity_user > 0) throw ChemError(CHEM_ENOSPC, "neighbour capacity " + std::to_string(S) + " too small, need " + std::to_string(h.nl_overflow));
      S = std::min(newS, std::max(((dd_on ? nglob : n) + 15) / 16 * 16, 16));
      alloc_lists();
      set_ctl_field(&DevCtl::nl_overflow, 0);
    }
    throw ChemError(CHEM_ENOSPC, "neighbour list capacity could not be satisfied");
  }

  // every rank must take the same branch after a collective rebuild: maxima of the flag words
  void agree_flags(DevCtl& h) {
    double v[4] = {(double)h.nl_overflow, (double)h.stage_overflow, (double)h.mig_error, (double)h.skin_violation};
    HIPCHK(hipMemcpyAsync(redbuf.p, v, sizeof(v), hipMemcpyHostToDevice, stream));
    tr->allreduce_max_f64(redbuf.p, 4, stream);
    HIPCHK(hipMemcpyAsync(v, redbuf.p, sizeof(v), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    h.nl_overflow = (int)v[0]; h.stage_overflow = (int)v[1]; h.mig_error = (int)v[2]; h.skin_violation = (int)v[3];
  }

  // ---- forces -------------------------------------------------------------------------
  int pick_tpp() const {
    if (opt_tpp > 0) return use_tiles ? std::min(opt_tpp, 8) : opt_tpp;
    if (use_tiles) return 1;   // one lane per home particle at every size (measured 10k..1M particles after the ds_read_b128 fix: 125k particles 17.4 us against 20.6 with two lanes)
    return n >= 100000 ? 4 : (n >= 8000 ? 8 : 16);
  }

  template <bool ENERGY> int launch_pair(V4* fdst, int tpp) {
    const double hs = 0.5 * skin_eff();
    const bool inline_now = bonds_inline();
    int bond_mode = inline_now ? (bond_by_pass() ? 2 : 1) : 0;      // (what the LAST rebuild did: both only change where a rebuild is forced)
    if (bond_mode == 2 && dd_on) {      // slabs: the host knows which launch follows a rebuild (mode 3 = mode 2 + record the partner slots now)
      if (dd_record_bonds) bond_mode = 3;
      if (pair_subset != 1) dd_record_bonds = false;      // (a launch of the interior tiles alone is followed by the boundary launch of the same step)
    }
    const BondRec<R>* brec = nullptr;
    if (bond_mode >= 2) {      // device copy of what the force launch behind a rebuild needs to record the partner slots (refreshed when a pointer or the box changed)
      BondRec<R> now;
      std::memset(&now, 0, sizeof(now));      // (compared bytewise below: no indeterminate padding)
      now.tag = tag.p; now.excl_start = excl_start.p; now.excl_list = excl_list.p; now.rtag = rtag.p;
      now.gtag = dd_on ? gtag.p : nullptr; now.real0 = dd_on ? G : 0; now.real1 = dd_on ? G + n : 0x7fffffff;
      std::memcpy(&now.box, &box, sizeof(box));
      if (!brec_dev.p) brec_dev.alloc(1);
      if (!brec_valid || std::memcmp(&now, &brec_host, sizeof(now)) != 0) {
        brec_host = now; brec_valid = true;
        HIPCHK(hipMemcpyAsync(brec_dev.p, &brec_host, sizeof(now), hipMemcpyHostToDevice, stream));
      }
      brec = brec_dev.p;
    }
    if (use_tiles) {
      // which tiles: all (default), or the interior / boundary subset of a slab (see TileSub)
      const int ntxy = tile_ntx(box.nc[0], box.xs_nb, box.xs_w) * ((box.nc[1] + HY - 1) / HY);
      TileSub ts{0, ntiles, 0};
      int nsub = ntiles;
      if (pair_subset == 1) { ts = TileSub{ntxy, ntiles - 2 * ntxy, 0}; nsub = ntiles - 2 * ntxy; }
      else if (pair_subset == 2) { ts = TileSub{0, ntxy, ntiles - ntxy}; nsub = 2 * ntxy; }
      if (nsub <= 0) return 0;
#define LTD(T, M, B, D) hipLaunchKernelGGL((k_pair_tiles<R, T, ENERGY, B, M, D>), dim3(nsub), dim3(B), pair_lds_bytes(), stream, nsub, tile_cap, x4.p, fdst, tdesc.p, \
                                 nl16.p, nnh.p, S, pcore.p, pext.p, ntypes, tab.p, uni, eout.p, hs, ctl.p, pair_guard, opt_ablate, dbg_on ? dbgbuf.p : (long long*)nullptr, ts, pair_da, \
                                 (inline_now && (!ENERGY || bond_mode >= 2)) ? bslots.p : (uint4*)nullptr, inline_K, inline_r0, bond_mode, act, brec)
#define LT(T, M, B) LTD(T, M, B, false)
      // diagnostics (options debug_stamps / ablate): one instantiation, one lane per particle, 512 threads
#define LTB(T, M) do { if ((dbg_on || opt_ablate) && !ENERGY) { if (T != 1 || pair_bs != 512) throw ChemError(CHEM_EINVAL, "debug_stamps / ablate need tpp=1 and pair_block=512"); LTD(1, M, 512, true); } \
                       else if (pair_bs == 256) LT(T, M, 256); else if (pair_bs == 512) LT(T, M, 512); else LT(T, M, 1024); } while (0)
#define LTT(M) do { switch (tpp) { case 1: LTB(1, M); break; case 2: LTB(2, M); break; case 8: LTB(8, M); break; default: LTB(4, M); break; } } while (0)
      const int mode = ENERGY ? 0 : (uniform_lj ? 2 : (lj_only ? 1 : 0));
      if (mode == 2) LTT(2); else if (mode == 1) LTT(1); else LTT(0);
#undef LTT
#undef LTB
#undef LT
#undef LTD
      return ntiles;
    }
    const int nb = cdiv((long long)n * tpp, 256);
#define LP(T) hipLaunchKernelGGL((k_pair_force<R, T, ENERGY>), dim3(nb), dim3(256), 0, stream, n, x4.p, fdst, nlist.p, nn.p, S, box, \
                                 pcore.p, pext.p, ntypes, tab.p, eout.p, hs, ctl.p)
    switch (tpp) {
      case 1: LP(1); break; case 2: LP(2); break; case 4: LP(4); break; case 8: LP(8); break;
      case 16: LP(16); break; case 32: LP(32); break; default: LP(64); break;
    }
#undef LP
    return nb;
  }

  void compute_forces(bool speculative = false, int subset = 0) {
    pair_guard = speculative ? (pair_da.gathered ? 2 : 1) : 0;
    pair_subset = subset;
    const int tpp = pick_tpp();
    const bool timed = timed_step && subset == 0;
    if (timed) tbeg(0);
    launch_pair<false>(f4.p, tpp);
    if (timed) tend();
    pair_subset = 0;
    if (subset == 1) { pair_guard = 0; return; }   // interior tiles only: bonded terms follow with the boundary launch
    const bool inl = nbent > 0 && bonds_inline();
    if (inl && excl_over == 0) {
      // harmonic bonds were evaluated by the force kernel (inline bonds): no bonded launch
    } else if (nbent > 0 && (use_fused || dd_on)) {
      // work-list kernel: owners only, partner indices resolved at the last rebuild
      if (bwork_dirty) {   // bonded lists changed without a rebuild since
        HIPCHK(hipMemsetAsync(&ctl.p->bw64, 0, sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(k_bonded_prep, dim3(std::max(1, std::min(cdiv(n, 256), 1024))), dim3(256), 0, stream, G, n, tag.p, rtag.p, bstart.p, bent.p, bwork.p, bj.p, ctl.p,
                           inl ? (const int*)excl_start.p : (const int*)nullptr);
        bwork_dirty = false;
      }
      if (timed) tbeg(3);
      const int nown = inl ? (int)std::min<int64_t>(nb_owner, excl_over) : nb_owner;   // (inline bonds: the list holds the owners with > kBondSlots exclusions only)
      if (bonds_only) hipLaunchKernelGGL((k_bonded_work<R, true>), dim3(cdiv(nown, 256)), dim3(256), 0, stream, x4.p, f4.p, bwork.p, bj.p, bent.p, bpar.p, boxd, ctl.p, speculative ? 1 : 0, btab_view());
      else hipLaunchKernelGGL((k_bonded_work<R, false>), dim3(cdiv(nown, 256)), dim3(256), 0, stream, x4.p, f4.p, bwork.p, bj.p, bent.p, bpar.p, boxd, ctl.p, speculative ? 1 : 0, btab_view());
      if (timed) tend();
    } else if (nbent > 0)
      hipLaunchKernelGGL((k_bonded<R, false>), dim3(cdiv(n, 256)), dim3(256), 0, stream, G, n, x4.p, f4.p, tag.p, rtag.p, bstart.p, bent.p,
                         bpar.p, boxd, elist.p, ctl.p, btab_view());
    pair_guard = 0;
  }

  LangevinP<R> lang_params(int64_t istep, int phase) const {
    LangevinP<R> lp{};
    lp.on = lang ? 1 : 0; lp.kT = kT; lp.gamma = gamma; lp.dt = dt; lp.seed = lang_seed; lp.tmask = lang_tmask; lp.step = (uint64_t)istep; lp.phase = (uint32_t)phase;
    return lp;
  }

  template <int MODE> void launch_integrate(bool with_lang, bool storef, int64_t istep, int phase) {
    const int nb = cdiv(n, kIntPerBlock);
    LangevinP<R> lp = lang_params(istep, phase);
    // CapForce acts on the freshly evaluated conservative force: every launch that applies the thermostat
    // consumes exactly that; without a thermostat f4 is never overwritten, so every launch does.
    const R cap = (cap_force > 0 && (with_lang || !lang)) ? (R)cap_force : (R)0;
    if (with_lang && storef)
      hipLaunchKernelGGL((k_integrate<R, MODE, true, true>), dim3(nb), dim3(256), 0, stream, n, x4.p + G, v4.p + G, f4.p + G, tag.p + G, (R)dt, lp, blockmax.p, opt_criterion ? x0.p + G : (const V4*)nullptr, cap, pos_scale(), (const DevCtl*)ctl.p, fold_arg());
    else if (with_lang)
      hipLaunchKernelGGL((k_integrate<R, MODE, true, false>), dim3(nb), dim3(256), 0, stream, n, x4.p + G, v4.p + G, f4.p + G, tag.p + G, (R)dt, lp, blockmax.p, opt_criterion ? x0.p + G : (const V4*)nullptr, cap, pos_scale(), (const DevCtl*)ctl.p, fold_arg());
    else
      hipLaunchKernelGGL((k_integrate<R, MODE, false, false>), dim3(nb), dim3(256), 0, stream, n, x4.p + G, v4.p + G, f4.p + G, tag.p + G, (R)dt, lp, blockmax.p, opt_criterion ? x0.p + G : (const V4*)nullptr, cap, pos_scale(), (const DevCtl*)ctl.p, fold_arg());
  }

  void check_flags() {
    DevCtl h = read_ctl();
    if (use_fused && getenv("CHEM_FUSED_STAMPS")) {   // phase boundaries of the last rebuild, 100 MHz ticks
      GridBar g; HIPCHK(hipMemcpy(&g, gbar.p, sizeof(GridBar), hipMemcpyDeviceToHost));
      fprintf(stderr, "[fused] bin %.1f scan %.1f place %.1f sort %.1f desc %.1f list %.1f us\n", (g.stamp[1] - g.stamp[0]) * 0.01, (g.stamp[2] - g.stamp[1]) * 0.01,
              (g.stamp[3] - g.stamp[2]) * 0.01, (g.stamp[4] - g.stamp[3]) * 0.01, (g.stamp[5] - g.stamp[4]) * 0.01, (g.stamp[6] - g.stamp[5]) * 0.01);
    }
    if (dd_on) agree_flags(h);
    tm.rebuilds = h.ref_rebuilds + (dd_on ? dd_direct_rebuilds : 0);   // the reference rule's count (== the list builds unless a wider list skin is in use)
    tm.list_rebuilds = dd_on ? dd_rebuilds : h.rebuild_count;
    if (h.mig_error) throw ChemError(CHEM_ESTATE, "domain decomposition: particle migration error " + std::to_string(h.mig_error));
    if (h.bonded_missing) throw ChemError(CHEM_ESTATE, "domain decomposition: a bonded partner is farther than the ghost layer (rc+skin)");
    if (h.stage_overflow) throw ChemError(CHEM_ENOSPC, "cell stencil exceeded the LDS tile capacity (" + std::to_string(h.stage_overflow) + " particles)");
    if (h.nl_overflow) throw ChemError(CHEM_ENOSPC, "neighbour row overflow during run: needed " + std::to_string(h.nl_overflow) + ", capacity " + std::to_string(S) + " (chem_set_nlist_capacity)");
    if (h.skin_violation) throw ChemError(CHEM_ESTATE, "internal: neighbour list used past skin/2");
    if (h.excl_slot_error) throw ChemError(CHEM_ESTATE, "internal: list build could not locate an excluded partner in its cell");
    if (h.bond_slot_miss) throw ChemError(CHEM_ESTATE, "internal: inline bonds enabled for a system with a bonded partner outside the located exclusions (set option bonds_inline=0)");
    if (h.barrier_timeout) throw ChemError(CHEM_ESTATE, "fused rebuild: grid barrier timed out (device shared with another job?); set option fused_rebuild=0");
    if (h.bucket_overflow) throw ChemError(CHEM_ENOSPC, "a cell filled up with " + std::to_string(h.bucket_overflow) + " particles during the run (fused rebuild bucket rows hold " + std::to_string(bcap) + "); set option fused_rebuild=0");
    if (h.cand_overflow) throw ChemError(CHEM_ENOSPC, "reaction candidate buffer overflow");
  }

  // one step's neighbour bookkeeping in slab mode: cross-rank max of the displacement, the ghost
  // position update (posted before the decision is known: it is needed unless we rebuild), and the
  // collective rebuild when the trigger fired.  One host synchronisation per step.
  bool dd_overlap() const {
    const int ntxy = tile_ntx(box.nc[0], box.xs_nb, box.xs_w) * ((box.nc[1] + HY - 1) / HY);
    const bool want_overlap = opt_overlap > 0 || (opt_overlap < 0 && ntiles >= 8192);
    return want_overlap && use_tiles && ntiles > 2 * ntxy && !getenv("CHEM_DD_NOPOLL");
  }
  // decision by the force kernel's workgroups instead of a one-block launch between the halo exchange and the forces
  bool dd_merged() const { return dd_on && opt_dd_merge && use_tiles && !dd_overlap() && !getenv("CHEM_DD_NOPOLL"); }
  void dd_step_sync() {
    ensure_hflag();
    if (dd_merged()) {
      const bool fold = fold_on();
      if (fold) {
        // the integrate kernel has folded the block maxima into foldmax with atomics: the words travel with the halo, no fold launch
        tr->exchange_with_scalar(x4.p + halo_dn_off, halo_dn_cnt * sizeof(V4), x4.p + halo_up_off, halo_up_cnt * sizeof(V4), x4.p + G + n, ngup * sizeof(V4),
                                 x4.p + G - nglo, nglo * sizeof(V4), lower, upper, reinterpret_cast<const double*>(foldmax.p), dd_vals.p, stream, kFoldSlots);
      } else {
        hipLaunchKernelGGL(k_rebuild_decide<R>, dim3(1), dim3(1024), 0, stream, ctl.p, blockmax.p, cdiv(acap(), kIntPerBlock), 0.5 * skin_eff(), opt_criterion, 1,
                           (const double*)nullptr, 0, (volatile int*)nullptr, 0, 0.5 * skin);
        tr->exchange_with_scalar(x4.p + halo_dn_off, halo_dn_cnt * sizeof(V4), x4.p + halo_up_off, halo_up_cnt * sizeof(V4), x4.p + G + n, ngup * sizeof(V4),
                                 x4.p + G - nglo, nglo * sizeof(V4), lower, upper, &ctl.p->step_m2, dd_vals.p, stream);
      }
      const int ticket = ++hticket;
      pair_da = DecideArgs{dd_vals.p, P, (volatile int*)hflag_dev, ticket, dd_par, opt_criterion, 0.5 * skin, fold ? foldmax.p : nullptr};
      dd_par ^= 1;
      compute_forces(true, 0);     // (guard 2: decision in the prologue of the force kernel)
      pair_da = DecideArgs{};
      volatile int* hf = hflag;
      poll_ticket(hflag + 1, ticket, "rebuild decision");
      if (hf[0]) { rebuild_dd(); compute_forces(); }
      return;
    }
    // local fold -> ctl->step_m2; its all-to-all rides in the halo exchange group; decision from the
    // P gathered values, mirrored into pinned host memory so that the host learns it by polling one
    // word (no memcpy, no stream synchronisation call)
    hipLaunchKernelGGL(k_rebuild_decide<R>, dim3(1), dim3(1024), 0, stream, ctl.p, blockmax.p, cdiv(acap(), kIntPerBlock), 0.5 * skin_eff(), opt_criterion, 1,
                       (const double*)nullptr, 0, (volatile int*)nullptr, 0, 0.5 * skin);
    // Overlap: the halo exchange runs on the communication stream while the tiles that need no ghost
    // (every tile layer but the lowest and the highest of the slab) already compute their forces.
    // Those launches cannot know the decision yet; if it is "rebuild", everything is recomputed below.
    const int ntxy = tile_ntx(box.nc[0], box.xs_nb, box.xs_w) * ((box.nc[1] + HY - 1) / HY);
    // Measured with one rank (1M particles, RCCL to self): the cross-stream hand-over costs ~15 us and the
    // boundary launch cannot fill the chip, so the overlap only pays once the interior force kernel is much
    // longer than that -- automatic for slabs of >= 8192 tiles (~4M particles per GPU), option overlap_halo.
    const bool want_overlap = opt_overlap > 0 || (opt_overlap < 0 && ntiles >= 8192);
    const bool overlap = want_overlap && use_tiles && ntiles > 2 * ntxy && !getenv("CHEM_DD_NOPOLL");
    hipStream_t xs = stream;
    if (overlap) {
      if (!cstream) {
        HIPCHK(hipStreamCreateWithFlags(&cstream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ev_fold, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ev_halo, hipEventDisableTiming));
      }
      HIPCHK(hipEventRecord(ev_fold, stream));
      HIPCHK(hipStreamWaitEvent(cstream, ev_fold, 0));
      xs = cstream;
    }
    tr->exchange_with_scalar(x4.p + halo_dn_off, halo_dn_cnt * sizeof(V4), x4.p + halo_up_off, halo_up_cnt * sizeof(V4), x4.p + G + n, ngup * sizeof(V4),
                             x4.p + G - nglo, nglo * sizeof(V4), lower, upper, &ctl.p->step_m2, dd_vals.p, xs);
    if (overlap) {
      HIPCHK(hipEventRecord(ev_halo, cstream));
      compute_forces(false, 1);                    // interior tiles, main stream, concurrent with the exchange
      HIPCHK(hipStreamWaitEvent(stream, ev_halo, 0));
    }
    const int ticket = ++hticket;
    hipLaunchKernelGGL(k_rebuild_decide<R>, dim3(1), dim3(1024), 0, stream, ctl.p, blockmax.p, 0, 0.5 * skin_eff(), opt_criterion, 2,
                       dd_vals.p, P, (volatile int*)hflag_dev, ticket, 0.5 * skin);
    // The force kernels are enqueued BEFORE the host looks at the decision: they leave at once when a
    // rebuild is pending (then the host rebuilds and launches them again), otherwise the device never
    // waits for the host's poll + launch latency.
    compute_forces(true, overlap ? 2 : 0);
    if (getenv("CHEM_DD_NOPOLL")) { DevCtl hc = read_ctl(); if (hc.need_rebuild) { rebuild_dd(); compute_forces(); } return; }
    volatile int* hf = hflag;
    poll_ticket(hflag + 1, ticket, "rebuild decision");
    if (hf[0]) { rebuild_dd(); compute_forces(); }
  }

  // Berendsen / Isokinetic: scale factor from the global kinetic energy, then one streaming pass over v
  DBuf<double> resc_buf;   // [0] Ekin (local, then global), [1] lambda
  void rescale_velocities() {
    const int nkb = cdiv(n, 256);
    resc_buf.alloc(2);
    hipLaunchKernelGGL(k_kinetic<R>, dim3(nkb), dim3(256), 0, stream, G, n, v4.p, ekout.p);
    const bool multi = dd_on && P > 1;
    hipLaunchKernelGGL(k_rescale_lambda, dim3(1), dim3(256), 0, stream, ekout.p, nkb, resc_buf.p, multi ? (double*)nullptr : resc_buf.p + 1,
                       resc_kind, resc_kT, dt / resc_param, (double)nglob, svr_seed, (uint64_t)step);
    if (multi) {
      tr->allreduce_sum_f64(resc_buf.p, 1, stream);
      hipLaunchKernelGGL(k_rescale_lambda, dim3(1), dim3(256), 0, stream, ekout.p, 0, resc_buf.p, resc_buf.p + 1, resc_kind, resc_kT, dt / resc_param, (double)nglob, svr_seed, (uint64_t)step);
    }
    hipLaunchKernelGGL(k_scale_v<R>, dim3(nkb), dim3(256), 0, stream, G, n, v4.p, resc_buf.p + 1, (const DevCtl*)ctl.p);
  }

  // ---- the hot call -------------------------------------------------------------------
  void run(int64_t nsteps) override {
    if (!(dt > 0)) throw ChemError(CHEM_ESTATE, "dt not set");
    flush_host_state();
    const double t0 = now_s();
    if (opt_time_pair) {
      const size_t want = (size_t)std::min<int64_t>(8 * (nsteps / opt_time_pair + 2), 16384);
      while (ev.size() < want) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); ev.push_back(e); }
      ev_used = 0; ev_kind.clear();
    }
    timed_step = false;
    if (react_on && !reactions.empty() && pin_ev_cap == 0) {
      // pinned staging for the event download: at most one event per two particles; allocated here so
      // that the first reaction step does not pay ~10 ms for it
      pin_ev_cap = (size_t)top.n / 2 + 1024;
      HIPCHK(hipHostMalloc((void**)&pin_ev, pin_ev_cap * sizeof(Candidate), hipHostMallocDefault));
      {  // the first copy-engine transfer out of a fresh device allocation into a fresh pinned range costs ~7 ms
         // (mappings): pay it here, not in the first reaction step
        alloc_reaction_buffers();
        const size_t bytes = std::min(pin_ev_cap, (size_t)cand_cap) * sizeof(Candidate);
        HIPCHK(hipMemsetAsync(evout.p, 0, bytes, stream));
        HIPCHK(hipMemcpyAsync(pin_ev, evout.p, bytes, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
      }
    }
    if (react_on && !reactions.empty()) reserve_reaction_tables();
    if (resort) rebuild_now();
    compute_forces();
    if (lang) launch_integrate<0>(true, true, step, 0);  // thermalize: f += friction + noise, stored
    bool need_int1 = true;
    const int64_t step0 = step;
    bool resume = false;       // re-entering the step at which the device stopped the run (DevCtl::halt)
    // The device may stop an asynchronous run (fused rebuild: a cell outgrew its bucket row).  Every launch behind that
    // point has left at once; the host learns of it at its next synchronisation -- a reaction / ATRP step or the end of
    // the call --, repairs the set-up (wider rows or the unfused chain: rebuild_now) and re-enters the loop at that step.
    auto halted = [&](int64_t& s) -> bool {
      if (!use_fused) return false;
      HIPCHK(hipStreamSynchronize(stream));
      const DevCtl h = read_ctl();
      if (!h.halt) return false;
      s = h.halt_step - step0; step = h.halt_step;
      set_ctl_field(&DevCtl::halt, 0);
      resort = true; resume = true; need_int1 = false;
      ++halts_recovered;
      if (g_trace) fprintf(stderr, "[chem trace] run stopped by the device at step %lld (bucket rows of %d, tile capacity %d, list rows of %d): recovering\n", (long long)h.halt_step, bcap, tile_cap, S);
      return true;
    };
    for (int64_t s = 0; s < nsteps; ++s) {
      if (need_int1) { launch_integrate<2>(false, false, step, 1); need_int1 = false; }
      timed_step = opt_time_pair && (pair_launch_no++ % opt_time_pair) == 0;
      if (timed_step) timed_extra = 1 + (int)((pair_launch_no / opt_time_pair) % 3);
      const bool react_due = react_on && interval > 0 && ((step + 1) % interval == 0);
      const bool atrp_due = atrp_on && ((step + 1) % atrp.interval == 0);
      const bool last = (s == nsteps - 1);
      if (resume) { resume = false; rebuild_now(); compute_forces(); }   // (positions are drifted, forces of this step were never evaluated)
      else if (dd_on) dd_step_sync();   // decision, (rebuild,) forces
      else { decide_and_rebuild(); compute_forces(); }
      resort = false;   // a rebuild requested by the last reaction step (force_rebuild on the device) has happened by now

      if (last || react_due || atrp_due || !opt_fuse || resc_kind) {
        launch_integrate<1>(lang, lang, step, 1);
        ++step;
        if (resc_kind == 1 || resc_kind == 3 || (resc_kind == 2 && step % (int64_t)resc_param == 0)) rescale_velocities();
        if ((react_due || atrp_due || last) && halted(s)) { --s; continue; }   // (the loop's ++s lands on the stopped step)
        if (react_due) react_step();
        if (atrp_due) atrp_step();      // behind the reaction step: the driver adds the extension after `ar` (start_simulation.py:737-740)
        need_int1 = true;
      } else {
        tbeg(2);
        launch_integrate<3>(lang, false, step, 1);
        tend();
        ++step;
      }
      timed_step = false;
    }
    check_flags();
    tm.run_wall_s += now_s() - t0;
    tm.steps += nsteps;
    if (opt_time_pair && ev_used) {
      // Decomposed path: speculative pair launches that met a pending rebuild leave at once; they are not
      // force evaluations, so samples far below the median are dropped from the average.
      std::vector<float> d(ev_used / 2);
      for (size_t k = 0; k < ev_used / 2; ++k) HIPCHK(hipEventElapsedTime(&d[k], ev[2 * k], ev[2 * k + 1]));
      std::vector<float> pd;
      for (size_t k = 0; k < d.size(); ++k) if (ev_kind[k] == 0) pd.push_back(d[k]);
      std::sort(pd.begin(), pd.end());
      const float cut = (dd_on && !pd.empty()) ? 0.5f * pd[pd.size() / 2] : 0.f;
      tm.pair_kernel_ms = tm.rebuild_kernel_ms = tm.decide_kernel_ms = tm.integrate_kernel_ms = tm.bonded_kernel_ms = 0;
      tm.pair_kernel_launches = tm.rebuild_kernel_launches = tm.decide_kernel_launches = tm.integrate_kernel_launches = tm.bonded_kernel_launches = 0;
      for (size_t k = 0; k < d.size(); ++k) {
        const float t = d[k];
        switch (ev_kind[k]) {
          case 0: if (t >= cut) { tm.pair_kernel_ms += t; tm.pair_kernel_launches++; } break;
          // the neighbour kernel either only takes the decision (a few us) or rebuilds cells, tiles and lists
          // (hundreds of us at any size that uses the fused path): told apart by duration
          case 1: if (t > 0.03f) { tm.rebuild_kernel_ms += t; tm.rebuild_kernel_launches++; } else { tm.decide_kernel_ms += t; tm.decide_kernel_launches++; } break;
          case 2: tm.integrate_kernel_ms += t; tm.integrate_kernel_launches++; break;
          default: tm.bonded_kernel_ms += t; tm.bonded_kernel_launches++; break;
        }
      }
    }
  }

  // The int32 Verlet list (all pairs within rc+skin) is only needed by the reaction scan and by
  // diagnostics: it is produced by a forced rebuild right there instead of at every rebuild.
  void ensure_list32() {
    if (!use_tiles) { if (resort) rebuild_now(); return; }
    want32 = true;
    rebuild_now();
    want32 = false;
    if (bonds_inline() && (sizeof(R) == 8 || bond_by_pass())) rebuild_now();   // same order, same rows; the regular build: partner slots again (fp64), excluded pairs back in the force list (bond pass)
  }

  // ---- reactions ----------------------------------------------------------------------
  void alloc_reaction_buffers() {
    if (cand_cap != 0) return;
    cand_cap = std::max(8 * nglob, 1024);
    cand.alloc(cand_cap); evout.alloc(cand_cap); st0.alloc(cand_cap); st1.alloc(cand_cap);
    asA.alloc(nglob); asB.alloc(nglob); best1.alloc(nglob); best2.alloc(nglob); evcount.alloc(1); rs_dev.alloc(1);
    if (dd_on) { cand_loc.alloc((size_t)cand_cap); cnt_all.alloc(64); }
  }
  void react_step() {
    tm.reaction_steps++;
    if (reactions.empty()) return;
    // the host runs ahead of the device by up to `interval` steps: drain that backlog first so that
    // reaction_wall_s measures the reaction step, not the MD steps queued before it
    HIPCHK(hipStreamSynchronize(stream));
    const double t0 = now_s();
    join_thread();   // labels of the previous reaction step must be on the device before the scan
    alloc_reaction_buffers();
    ReactSet rs{};
    rs.n = (int)reactions.size(); rs.seed = react_seed; rs.step = (uint64_t)step; rs.nearest = nearest;
    ReactApplySet ras{};
    for (int q = 0; q < rs.n; ++q) {
      const chem_reaction_desc& d = reactions[q];
      ReactionDev& r = rs.r[q];
      r.type_1 = d.type_1; r.type_2 = d.type_2; r.delta_1 = d.delta_1; r.delta_2 = d.delta_2;
      r.min1 = d.min_state_1; r.max1 = d.max_state_1; r.min2 = d.min_state_2; r.max2 = d.max_state_2;
      r.intramolecular = d.intramolecular; r.intraresidual = d.intraresidual; r.active = d.active;
      r.restricted = (restricted_mask >> q) & 1u;
      r.cons_role = q < (int)constraints.size() ? constraints[q].role : 0; r.pad_ = 0;
      r.cut2 = d.cutoff * d.cutoff; r.mincut2 = d.min_cutoff * d.min_cutoff; r.prob = d.rate * dt * (double)interval;
      ras.r[q] = ReactApply{d.delta_1, d.delta_2, d.new_type_1, d.new_type_2, d.new_mass_1, d.new_mass_2};
    }
    HIPCHK(hipMemcpyAsync(rs_dev.p, &rs, sizeof(ReactSet), hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));
    Trace trc("react");
    // No rebuild here: the particle order must not change between the force evaluation of this step and
    // the first kick of the next one (forces are not re-sorted), and on the decomposed path a rebuild
    // would migrate particles away from their forces.  Tiles: scan the staged stencils of the last
    // rebuild (see k_react_scan_tiles).  Per-cell / brute-force lists: the int32 list is always current.
    Candidate* cdst = dd_on ? cand_loc.p : cand.p;
    if (restrict_dirty) {      // per-tag CSR of the allowed partners (both directions), built on the host: the map is set-up data
      std::vector<int> hs((size_t)nglob + 1, 0), hp; std::vector<unsigned int> hm;
      for (auto& kv : restrict_map) { hs[kv.first.first + 1]++; hs[kv.first.second + 1]++; }
      for (int t = 0; t < nglob; ++t) hs[t + 1] += hs[t];
      hp.resize(hs[nglob] ? hs[nglob] : 1); hm.resize(hp.size());
      std::vector<int> cur(hs.begin(), hs.end() - 1);
      for (auto& kv : restrict_map) {
        hp[cur[kv.first.first]] = kv.first.second; hm[cur[kv.first.first]++] = kv.second;
        hp[cur[kv.first.second]] = kv.first.first; hm[cur[kv.first.second]++] = kv.second;
      }
      conn_start.upload(hs, stream); conn_partner.upload(hp, stream); conn_mask.upload(hm, stream);
      restrict_dirty = false;
    }
    bool any_cons = false;
    for (auto& cs : constraints) any_cons |= cs.role != 0;
    if (any_cons) {
      // neighbour-state constraints: evaluated on the host per particle from the bond graph and the current states and
      // types (the graph lives here; candidates carrying a constraint are rare), one bit per reaction
      if (state_mirror_stale) { std::vector<int> hs; state.download(hs, nglob, stream); top.state.assign(hs.begin(), hs.end()); state_mirror_stale = false; }
      sync_type_mirrors();
      std::vector<unsigned int> ok((size_t)nglob, 0u);
      for (size_t q = 0; q < constraints.size(); ++q) {
        const NbCons& cs = constraints[q];
        if (!cs.role) continue;
        const int own_type = cs.role == 1 ? reactions[q].type_1 : reactions[q].type_2;
        for (int32_t t = 0; t < (int32_t)nglob; ++t) {
          if (top.type[t] != own_type) continue;
          for (int32_t nb : top.graph[t]) if (top.type[nb] == cs.nb_type && top.state[nb] >= cs.min_state && top.state[nb] < cs.max_state) { ok[t] |= 1u << q; break; }
        }
      }
      cons_ok.upload(ok, stream);
    }
    const ConnTable conn{restricted_mask ? conn_start.p : nullptr, conn_partner.p, conn_mask.p, any_cons ? cons_ok.p : nullptr};
    if (use_tiles) {
      const int region_cap = std::max(1, cand_cap / std::max(ntiles, 1));
      tile_cnt.alloc(ntiles + 1); tile_off.alloc(ntiles + 1);
      if (scan_role.n < (size_t)acap()) scan_role.alloc((size_t)acap() + 1024);
      hipLaunchKernelGGL(k_react_roles<R>, dim3(std::min(cdiv(acap(), 256), 4096)), dim3(256), 0, stream, acap(), nglob, x4.p, tag.p, state.p, rs_dev.p, scan_role.p);
      hipLaunchKernelGGL((k_react_scan_tiles<R, 512>), dim3(ntiles), dim3(512), scan_roles_staged() ? scan_lds_bytes() : tile_lds_bytes(), stream, ntiles, tile_cap, x4.p, tag.p, tdesc.p, state.p,
                         res_id.p, mol_id.p, boxd, rs_dev.p, evout.p, region_cap, tile_cnt.p, ctl.p, (R)(0.5 * skin_eff()),
                         has_excl ? (const int*)excl_start.p : (const int*)nullptr, (const int*)excl_list.p, conn, (const unsigned int*)scan_role.p, scan_roles_staged() ? 1 : 0);
      hipLaunchKernelGGL(k_cand_offsets, dim3(1), dim3(1024), 0, stream, ntiles, tile_cnt.p, tile_off.p, ctl.p);
      hipLaunchKernelGGL(k_cand_gather, dim3(std::min(ntiles, 2048)), dim3(256), 0, stream, ntiles, evout.p, region_cap, tile_cnt.p, tile_off.p, cdst, cand_cap, ctl.p);
    } else {
      HIPCHK(hipMemsetAsync(&ctl.p->cand_count, 0, sizeof(int), stream));
      hipLaunchKernelGGL(k_react_scan<R>, dim3(cdiv((long long)n * 8, 256)), dim3(256), 0, stream, G, n, x4.p, tag.p, nlist.p, nn.p, S, state.p,
                         res_id.p, mol_id.p, boxd, rs_dev.p, cdst, cand_cap, ctl.p, conn);
    }
    DevCtl h = read_ctl();
    trc.lap("scan");
    if (h.cand_overflow) throw ChemError(CHEM_ENOSPC, "reaction candidate buffer overflow");
    int nc = h.cand_count;
    if (dd_on) {
      // A boundary pair is seen by two ranks (real-ghost on each); the scan emits it only where the
      // particle with the lower tag is owned, so the union over ranks holds every candidate once.
      // All ranks then run the same deterministic resolve on the gathered set (one exchange).
      std::vector<int> cnts(P, 0);
      HIPCHK(hipMemcpyAsync(cnt_all.p + rk, &nc, sizeof(int), hipMemcpyHostToDevice, stream));
      tr->allgather(cnt_all.p + rk, cnt_all.p, sizeof(int), stream);
      HIPCHK(hipMemcpyAsync(cnts.data(), cnt_all.p, sizeof(int) * P, hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
      int mx = 0, tot = 0;
      for (int v : cnts) { mx = std::max(mx, v); tot += v; }
      if ((long long)mx * P > cand_cap || tot > cand_cap) throw ChemError(CHEM_ENOSPC, "reaction candidate buffer overflow (gathered)");
      if (mx > 0) {
        // padded all-gather into evout (scratch here), then compaction into cand in rank order
        HIPCHK(hipMemcpyAsync(evout.p + (size_t)rk * mx, cand_loc.p, sizeof(Candidate) * nc, hipMemcpyDeviceToDevice, stream));
        tr->allgather(evout.p + (size_t)rk * mx, evout.p, sizeof(Candidate) * mx, stream);
        int off = 0;
        for (int r = 0; r < P; ++r) {
          if (cnts[r]) HIPCHK(hipMemcpyAsync(cand.p + off, evout.p + (size_t)r * mx, sizeof(Candidate) * cnts[r], hipMemcpyDeviceToDevice, stream));
          off += cnts[r];
        }
      }
      nc = tot;
    }
    if (nc == 0) { tm.reaction_wall_s += now_s() - t0; return; }
    const int ncb = cdiv(nc, 256), npb = cdiv(nglob, 256);
    hipLaunchKernelGGL(k_fill<int>, dim3(ncb), dim3(256), 0, stream, st0.p, 1, (size_t)nc);
    for (int side = 0; side < 2; ++side) {
      hipLaunchKernelGGL(k_fill<unsigned long long>, dim3(npb), dim3(256), 0, stream, best1.p, ~0ull, (size_t)nglob);
      hipLaunchKernelGGL(k_fill<unsigned long long>, dim3(npb), dim3(256), 0, stream, best2.p, ~0ull, (size_t)nglob);
      hipLaunchKernelGGL(k_res_min1, dim3(ncb), dim3(256), 0, stream, nc, cand.p, st0.p, side, nearest, best1.p);
      hipLaunchKernelGGL(k_res_min2, dim3(ncb), dim3(256), 0, stream, nc, cand.p, st0.p, side, nearest, best1.p, best2.p);
      hipLaunchKernelGGL(k_res_keep, dim3(ncb), dim3(256), 0, stream, nc, cand.p, st0.p, side, nearest, best1.p, best2.p);
    }
    hipLaunchKernelGGL(k_fill<int>, dim3(npb), dim3(256), 0, stream, asA.p, -1, (size_t)nglob);
    hipLaunchKernelGGL(k_fill<int>, dim3(npb), dim3(256), 0, stream, asB.p, -1, (size_t)nglob);
    hipLaunchKernelGGL(k_res_index, dim3(ncb), dim3(256), 0, stream, nc, cand.p, st0.p, asA.p, asB.p);
    if (g_trace) { HIPCHK(hipStreamSynchronize(stream)); trc.lap("uniqueAB"); }
    int* sin = st0.p; int* sout = st1.p;
    for (int batch = 0; batch < 20000; ++batch) {
      // `alive` is only meaningful for the last round of a batch: extra rounds on a finished
      // matching are no-ops, so rounds are enqueued 8 at a time between host checks
      for (int rr = 0; rr < 8; ++rr) {
        if (rr == 7) HIPCHK(hipMemsetAsync(&ctl.p->alive, 0, sizeof(int), stream));
        hipLaunchKernelGGL(k_res_round, dim3(ncb), dim3(256), 0, stream, nc, cand.p, sin, sout, asA.p, asB.p, nearest, ctl.p);
        std::swap(sin, sout);
      }
      if (read_ctl().alive == 0) break;
    }
    if (g_trace) { fprintf(stderr, "[chem trace] candidates %d\n", nc); trc.lap("rounds"); }
    if (max_per_interval > 0) {
      // ChemicalReaction.max_per_interval (reaction_setup.py:426-427): keep the max_per_interval accepted events of
      // highest priority (nearest: r^2 then A's tag; random: pair hash then A's tag), as the oracle does.  Rare
      // option: selection on the host from the downloaded candidate records.
      std::vector<Candidate> hc; std::vector<int> hs;
      cand.download(hc, nc, stream);
      { hs.resize(nc); HIPCHK(hipMemcpyAsync(hs.data(), sin, sizeof(int) * nc, hipMemcpyDeviceToHost, stream)); HIPCHK(hipStreamSynchronize(stream)); }
      std::vector<int> accd;
      for (int k = 0; k < nc; ++k) if (hs[k] == 2) accd.push_back(k);
      if ((int64_t)accd.size() > max_per_interval) {
        auto key = [&](int k) { return std::make_pair(nearest ? (unsigned long long)reinterpret_cast<const long long&>(hc[k].d2) : (unsigned long long)hc[k].h, hc[k].a); };
        std::sort(accd.begin(), accd.end(), [&](int p, int q) { return key(p) < key(q); });
        for (size_t k = (size_t)max_per_interval; k < accd.size(); ++k) hs[accd[k]] = 0;
        HIPCHK(hipMemcpyAsync(sin, hs.data(), sizeof(int) * nc, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
      }
    }
    HIPCHK(hipMemsetAsync(evcount.p, 0, sizeof(int), stream));
    hipLaunchKernelGGL(k_react_apply<R>, dim3(ncb), dim3(256), 0, stream, nc, cand.p, sin, ras, state.p, rtag.p, x4.p, v4.p, evout.p, evcount.p);
    int nev = 0;
    HIPCHK(hipMemcpyAsync(&nev, evcount.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    trc.lap("resolve+apply");
    if ((size_t)nev > pin_ev_cap) {
      if (pin_ev) (void)hipHostFree(pin_ev);
      pin_ev_cap = std::max<size_t>((size_t)nev * 2, 1 << 16);
      HIPCHK(hipHostMalloc((void**)&pin_ev, pin_ev_cap * sizeof(Candidate), hipHostMallocDefault));
    }
    if (nev) {
      const size_t n16 = ((size_t)nev * sizeof(Candidate) + 15) / 16;   // evout/pin_ev are sized in whole candidates (24 B): rounding up stays inside
      hipLaunchKernelGGL(k_copy16, dim3(512), dim3(256), 0, stream, reinterpret_cast<const uint4*>(evout.p), reinterpret_cast<uint4*>(pin_ev), n16);
    }
    HIPCHK(hipStreamSynchronize(stream));
    trc.lap("download");
    // the event records are processed where the copy kernel put them (pinned, host-cached memory): no second copy
    struct Span { Candidate* p; size_t n; Candidate* begin() const { return p; } Candidate* end() const { return p + n; }
                  Candidate* data() const { return p; } size_t size() const { return n; } Candidate& operator[](size_t k) const { return p[k]; } };
    const Span hev{pin_ev, (size_t)nev};
    // Host mirrors + topology.  Only bond-forming events need the canonical order now (it fixes
    // the order of the bond lists); the event log itself is put in canonical order lazily
    // by chem_get_events.
    auto ekey = [](const Candidate& p) { return ((uint64_t)(uint32_t)std::min(p.a, p.b) << 32) | (uint32_t)std::max(p.a, p.b); };
    // bond-forming events in canonical order, extracted into a heap buffer; the pinned event array stays as the device wrote it.
    // (0.6-0.8 ms at 2.5e5 events, as partitioning and sorting in place was: ONE pass over 6 MB the GPU has just written
    //  through PCIe is most of it -- a device-side compaction of the bond-forming events would be the next step)
    std::vector<Candidate>& bev = bond_ev;
    bev.clear();
    for (const Candidate& e : hev) if (!reactions[e.r].is_virtual) bev.push_back(e);
    {
      // A particle takes part in at most one event per reaction step, so min(a,b) alone is a unique
      // key: LSD radix sort (3 x 11 bits) instead of a comparison sort of 10^5 events.
      const size_t m = bev.size();
      if (radix_tmp.size() < m) radix_tmp.resize(m);
      Candidate* src = bev.data(); Candidate* dst = radix_tmp.data();
      for (int pass = 0; pass < 3; ++pass) {
        size_t cnt[2049] = {0};
        const int sh = 11 * pass;
        for (size_t k = 0; k < m; ++k) ++cnt[(((uint32_t)std::min(src[k].a, src[k].b) >> sh) & 2047u) + 1];
        for (int d = 0; d < 2048; ++d) cnt[d + 1] += cnt[d];
        for (size_t k = 0; k < m; ++k) dst[cnt[((uint32_t)std::min(src[k].a, src[k].b) >> sh) & 2047u]++] = src[k];
        std::swap(src, dst);
      }
      if (src != bev.data()) std::copy(src, src + m, bev.data());
      bool unique = true;
      for (size_t k = 1; k < m && unique; ++k) unique = std::min(bev[k - 1].a, bev[k - 1].b) != std::min(bev[k].a, bev[k].b);
      if (!unique) std::sort(bev.begin(), bev.end(), [&](const Candidate& p, const Candidate& q) { return ekey(p) < ekey(q); });
    }
    trc.lap("bond events");
    std::vector<std::pair<int32_t, int32_t>> newbonds;
    newbonds.reserve(hev.size());
    bool types_changed = false;
    // event log + host mirrors of the new types (only what the host itself needs later: types for
    // bonded-slot resolution and the topology manager; chemical states live on the device).  Events
    // touch disjoint particles, nothing below reads the mirrors unless a list is typed or tuples are
    // spawned, so this runs beside the table builds and is joined at the end of the step.
    // intra-/inter-cluster flag of every event from the cluster labels as they were BEFORE this step's bonds
    // (the label thread of the previous reaction step was joined above, this step's has not started yet)
    std::vector<int8_t> intra_flags;
    if (opt_intra_inter) {
      intra_flags.resize(hev.size());
      for (size_t k = 0; k < hev.size(); ++k) intra_flags[k] = top.mol_id[hev[k].a] == top.mol_id[hev[k].b] ? 1 : 0;
    }
    // event log: appended to the arena in device order (canonical order is established lazily by chem_get_events)
    for (size_t k = 0; k < hev.size() && !types_changed; ++k) { const chem_reaction_desc& d = reactions[hev[k].r]; types_changed = d.new_type_1 >= 0 || d.new_type_2 >= 0; }   // (conservative: the force list depends on the types)
    // the host's type mirrors are needed right now only where the host itself resolves something by type
    bool mirrors_first = top.spawns_tuples() || !nb_rules.empty();
    for (auto& l : top.lists) mirrors_first |= l.by_types != 0;
    // (otherwise the log is appended further down, while the device rebuilds its tables)
    if (mirrors_first) { append_events(hev.data(), hev.size(), step, intra_flags); sync_type_mirrors(); }
    const Candidate* log_ev = hev.data(); const size_t log_n = mirrors_first ? 0 : hev.size(); const int64_t log_step = step;   // what the thread still has to log
    // (pin_ev is not touched again before the next reaction step, which joins the thread first)
    std::thread mirror_thr;
    std::exception_ptr mirror_err;
    // (measured, round 3: inserting by hash shards on helper threads -- a persistent pool of 1..3 -- made this loop SLOWER,
    //  1.0-1.7 ms -> 2.0-3.9 ms for 2.4-4.3e4 new bonds on the 2-socket host: the table lives on the caller's NUMA node)
    // Where every bond is an excluded pair (bonds_excluded: checked at the last full upload, kept by construction since) and the
    // scan honours the exclusions, a candidate pair cannot be bonded already, and a particle takes part in one event per step:
    // every bond-forming event is a new bond.  The insertion into the list's de-duplication set -- one cache miss per bond in a
    // table of 10^6, 0.2 -> 1.7 ms of this step as the conversion grows -- then moves to the bookkeeping thread.
    const bool defer_seen = bonds_excluded && has_excl && !top.spawns_tuples() && nb_rules.empty();
    label_seen_lists.clear();
    for (size_t k = 0; defer_seen && k < bev.size(); ++k) {
      const Candidate& e = bev[k];
      const chem_reaction_desc& d = reactions[e.r];
      HostList& l = top.lists[d.bond_list];
      const int32_t t[2] = {e.a, e.b};
      l.ent.insert(l.ent.end(), t, t + 2);
      newbonds.emplace_back(e.a, e.b);
      label_seen_lists.push_back(d.bond_list);
    }
    for (size_t k = 0; !defer_seen && k < bev.size(); ++k) {
      const Candidate& e = bev[k];
      const chem_reaction_desc& d = reactions[e.r];
      if (k + 24 < bev.size()) {   // the de-duplication set is a 10^6-entry hash table: hide its misses
        int32_t tp[2] = {bev[k + 24].a, bev[k + 24].b};
        top.lists[reactions[bev[k + 24].r].bond_list].seen.prefetch(tuple_key(tp, 2));
      }
      int32_t t[2] = {e.a, e.b};
      if (top.list_insert(top.lists[d.bond_list], t)) newbonds.emplace_back(e.a, e.b);
    }
    state_mirror_stale = true;
    if (g_trace) fprintf(stderr, "[chem trace] events %zu new bonds %zu\n", hev.size(), newbonds.size());
    trc.lap("host events");
    // PostProcessChangeNeighboursProperty: needs the bond graph with this step's bonds and the mirrors of the new
    // types; events in canonical order, role 1 before role 2, rules in insertion order (as the oracle)
    auto neighbour_changes = [&] {
      if (nb_rules.empty()) return;
      std::vector<Candidate> ord;
      for (auto& e : hev) for (auto& rl : nb_rules) if (rl.reaction == e.r) { ord.push_back(e); break; }
      std::sort(ord.begin(), ord.end(), [&](const Candidate& p, const Candidate& q) { return ekey(p) < ekey(q); });
      bool reads_state = false;
      for (auto& rl : nb_rules) reads_state |= rl.set_state == 2 || rl.min_state < rl.max_state;
      if (reads_state) {      // states as the events of this step left them (k_react_apply ran on the device)
        std::vector<int> hs; state.download(hs, nglob, stream); top.state.assign(hs.begin(), hs.end()); state_mirror_stale = false;
      }
      std::vector<HostTopology::PropChange> chg;
      for (auto& e : ord)
        for (int role = 1; role <= 2; ++role)
          for (auto& rl : nb_rules) if (rl.reaction == e.r && (rl.invoke_on & role)) top.neighbour_change(role == 1 ? e.a : e.b, rl, chg);
      if (chg.empty()) return;
      static_assert(sizeof(HostTopology::PropChange) == sizeof(PropChangeDev), "layout");
      DBuf<PropChangeDev> dchg; dchg.alloc(chg.size());
      HIPCHK(hipMemcpyAsync(dchg.p, chg.data(), chg.size() * sizeof(PropChangeDev), hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL((k_apply_props<R>), dim3(cdiv((long long)chg.size(), 256)), dim3(256), 0, stream, (int)chg.size(), dchg.p, state.p, rtag.p, x4.p, v4.p);
      HIPCHK(hipStreamSynchronize(stream));
      types_changed = true;
      if (!reads_state) state_mirror_stale = true;      // (rules that set a state moved the device copy, not a refreshed mirror)
    };
    if (!newbonds.empty() && !nb_rules.empty()) {
      // same dependency order as below, fully synchronous: graph -> (labels on the thread) -> spawned tuples (types as
      // they are right after the events) -> neighbour property changes -> tables
      label_bonds = newbonds; label_touched.clear(); labels_pending = true;
      top.link_new_bonds(newbonds);
      label_thr = std::thread([this] {
        try { top.merge_new_bonds(label_bonds, label_touched); } catch (...) { label_err = std::current_exception(); }
      });
      top.spawn_for_new_bonds(newbonds);
      if (label_thr.joinable()) label_thr.join();   // the flood fill reads the graph only; types are not touched by it, but keep it simple
      neighbour_changes();
      upload_excl(false);
      upload_bonded(false);
      resort = true;
      set_ctl_field(&DevCtl::force_rebuild, 1);
    } else if (!newbonds.empty()) {
      // Host work of a bond-forming step, arranged by what depends on what:
      //   bonded CSR  <- lists            exclusion CSR <- exclusions <- (spawned tuples <- graph)
      //   cluster labels <- graph, read again only by the NEXT reaction scan -> host thread, joined lazily
      label_bonds = newbonds; label_touched.clear(); labels_pending = true;
      if (!top.spawns_tuples()) {
        // the device tables only need the new tuples / pairs: the bond graph, the cluster labels AND the host's own
        // exclusion rows (read by chem_get_exclusions only) are brought up to date on the thread, beside the MD steps
        top.excl_log.reserve(top.excl_log.size() + newbonds.size());
        const size_t log0 = top.excl_log.size();
        for (auto& e : newbonds) top.excl_log.emplace_back(e.first, e.second);   // (a bond's pair is always new: list_insert deduplicated)
        label_thr = std::thread([this, log0, log_ev, log_n, log_step, intra_flags] {
          try {
            if (log_n) append_events(log_ev, log_n, log_step, intra_flags);
            for (size_t k = 0; k < label_seen_lists.size(); ++k) {      // (deferred de-duplication keys of this step's bonds)
              const int32_t t[2] = {label_bonds[k].first, label_bonds[k].second};
              top.lists[label_seen_lists[k]].seen.insert(tuple_key(t, 2));
            }
            top.link_new_bonds(label_bonds); top.merge_new_bonds(label_bonds, label_touched);
            // rows + pair count; the log already holds these pairs
            for (auto& e : label_bonds) { if (HostTopology::sorted_insert(top.excl[e.first], e.second)) { HostTopology::sorted_insert(top.excl[e.second], e.first); ++top.n_excl_pairs; } }
            (void)log0;
          } catch (...) { label_err = std::current_exception(); }
        });
        trc.lap("threads started");
      } else {
        top.link_new_bonds(newbonds);
        label_thr = std::thread([this] {
          try { top.merge_new_bonds(label_bonds, label_touched); } catch (...) { label_err = std::current_exception(); }
        });
        top.spawn_for_new_bonds(newbonds);
        trc.lap("link+spawn");
      }
      upload_excl(false, false);
      upload_bonded(false, false);
      trc.lap("table builds enqueued");
      HIPCHK(hipStreamSynchronize(stream));
      trc.lap("uploads");
      resort = true;
      set_ctl_field(&DevCtl::force_rebuild, 1);
    }
    if (mirror_thr.joinable()) mirror_thr.join();
    if (mirror_err) std::rethrow_exception(mirror_err);
    if (log_n && newbonds.empty()) {     // no bonds, no label work: the thread only writes the log
      label_thr = std::thread([this, log_ev, log_n, log_step, intra_flags] {
        try { append_events(log_ev, log_n, log_step, intra_flags); } catch (...) { label_err = std::current_exception(); }
      });
    }
    if (newbonds.empty()) neighbour_changes();   // rules on reactions that form no bond
    if (types_changed) { resort = true; set_ctl_field(&DevCtl::force_rebuild, 1); }   // force list depends on types
    if (newbonds.empty() && types_changed) {
      bool any_typed = false;
      for (auto& l : top.lists) any_typed |= l.by_types != 0;
      if (any_typed) upload_bonded(false);     // same tuples, slots re-resolved against the new types
    }
    trc.lap("end");
    tm.reaction_wall_s += now_s() - t0;
  }

  // ---- ATRPActivator: every `interval` steps, on the host (a few thousand centres at reaction cadence) -----------
  // Rule set: include/chem_mi355.h (chem_atrp_desc).  Types and states live on the device: the state array is read
  // back, the type mirrors are replayed from the event arena, the flips go back through k_apply_props.
  void atrp_step() {
    HIPCHK(hipStreamSynchronize(stream));
    join_async();
    if (state_mirror_stale) { std::vector<int> hs; state.download(hs, nglob, stream); top.state.assign(hs.begin(), hs.end()); state_mirror_stale = false; }
    struct Sel { uint32_t key; int32_t tag; uint32_t u; int center; };
    auto center_of = [&](int32_t t) {
      for (size_t c = 0; c < atrp_centers.size(); ++c) if (atrp_centers[c].type == top.type[t] && atrp_centers[c].state == top.state[t]) return (int)c;
      return -1;
    };
    std::vector<Sel> pool;
    int64_t ncand = 0;
    for (int32_t t = 0; t < (int32_t)top.n; ++t) {
      const int c = center_of(t);
      if (c >= 0) ++ncand;
      if (c < 0 && !atrp.select_from_all) continue;
      uint32_t r[4];
      chem_philox::atrp_draw(atrp.seed, (uint64_t)step, (uint32_t)t, r);
      pool.push_back(Sel{r[0], t, r[1], c});
    }
    auto less = [](const Sel& a, const Sel& b) { return a.key != b.key ? a.key < b.key : a.tag < b.tag; };
    if ((int64_t)pool.size() > atrp.num_particles) { std::nth_element(pool.begin(), pool.begin() + atrp.num_particles, pool.end(), less); pool.resize((size_t)atrp.num_particles); }
    std::sort(pool.begin(), pool.end(), less);
    const double dc = atrp.delta_catalyst / (double)atrp.num_particles;
    chem_atrp_stats st{}; st.step = step; st.candidates = ncand; st.selected = (int64_t)pool.size();
    std::vector<PropChangeDev> chg;
    bool types_changed = false;
    for (auto& sl : pool) {
      if (sl.center < 0) continue;
      const AtrpCenter& c = atrp_centers[sl.center];
      const double p = c.is_activator ? atrp.k_deactivate * atrp.ratio_deactivator : atrp.k_activate * atrp.ratio_activator;
      if (!(chem_philox::u01(sl.u) < p)) continue;
      const int32_t t = sl.tag;
      if (c.new_type >= 0 && c.new_type != top.type[t]) { top.type[t] = c.new_type; top.mass[t] = c.new_mass; top.q[t] = c.new_q; types_changed = true; }
      top.state[t] += c.delta_state;
      chg.push_back(PropChangeDev{t, top.type[t], 1, top.state[t], top.mass[t], top.q[t]});
      if (c.is_activator) { const double m = std::min(dc, atrp.ratio_deactivator); atrp.ratio_deactivator -= m; atrp.ratio_activator += m; st.deactivated++; }
      else { const double m = std::min(dc, atrp.ratio_activator); atrp.ratio_activator -= m; atrp.ratio_deactivator += m; st.activated++; }
    }
    st.ratio_activator = atrp.ratio_activator; st.ratio_deactivator = atrp.ratio_deactivator;
    atrp_stats.push_back(st);
    if (chg.empty()) return;
    DBuf<PropChangeDev> dchg; dchg.alloc(chg.size());
    HIPCHK(hipMemcpyAsync(dchg.p, chg.data(), chg.size() * sizeof(PropChangeDev), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL((k_apply_props<R>), dim3(cdiv((long long)chg.size(), 256)), dim3(256), 0, stream, (int)chg.size(), dchg.p, state.p, rtag.p, x4.p, v4.p);
    HIPCHK(hipStreamSynchronize(stream));
    if (types_changed) {     // the force list and the typed bonded slots depend on the types
      bool any_typed = false;
      for (auto& l : top.lists) any_typed |= l.by_types != 0;
      if (any_typed) upload_bonded(false);
      resort = true;
      set_ctl_field(&DevCtl::force_rebuild, 1);
    }
  }

  // ---- read-back ------------------------------------------------------------------------
  // ---- cross-rank gather of fixed-size host records (read-back paths only) ---------------
  DBuf<unsigned char> gbuf;
  std::vector<unsigned char> gather_records(const std::vector<unsigned char>& loc, size_t rec) {
    if (!dd_on || P == 1) return loc;
    if (!cnt_all.p) cnt_all.alloc(64);
    int nloc = (int)(loc.size() / rec);
    std::vector<int> cnts(P, 0);
    HIPCHK(hipMemcpyAsync(cnt_all.p + rk, &nloc, sizeof(int), hipMemcpyHostToDevice, stream));
    tr->allgather(cnt_all.p + rk, cnt_all.p, sizeof(int), stream);
    HIPCHK(hipMemcpyAsync(cnts.data(), cnt_all.p, sizeof(int) * P, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    size_t mx = 0, tot = 0;
    for (int v : cnts) { mx = std::max<size_t>(mx, v); tot += v; }
    std::vector<unsigned char> all(tot * rec);
    if (mx == 0) return all;
    // sized once for any record type and a slab 50 % over its share (it only grows beyond that): the inter-process transport
    // maps peer allocations by handle and caches the mappings -- a buffer that is freed and allocated again at another size
    // between two gathers handed the peers a stale mapping ("copy: invalid argument" in the multi-process driver run)
    if (gbuf.n < mx * rec * P) gbuf.alloc(std::max(mx * rec * P, ((size_t)nglob * 3 / (2 * (size_t)P) + 1024) * 64 * (size_t)P));
    if (nloc) HIPCHK(hipMemcpyAsync(gbuf.p + (size_t)rk * mx * rec, loc.data(), loc.size(), hipMemcpyHostToDevice, stream));
    tr->allgather(gbuf.p + (size_t)rk * mx * rec, gbuf.p, mx * rec, stream);
    std::vector<unsigned char> raw(mx * rec * P);
    HIPCHK(hipMemcpyAsync(raw.data(), gbuf.p, raw.size(), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    size_t off = 0;
    for (int r = 0; r < P; ++r) { std::memcpy(all.data() + off, raw.data() + (size_t)r * mx * rec, (size_t)cnts[r] * rec); off += (size_t)cnts[r] * rec; }
    return all;
  }

  struct StateRec { int tag, pad; double v[3]; };

  // ---- read-back ------------------------------------------------------------------------
  int64_t get_state(int what, void* out, int64_t cap) override {
    flush_host_state();
    const int per = (what == CHEM_STATE_POS || what == CHEM_STATE_VEL || what == CHEM_STATE_FORCE || what == CHEM_STATE_IMAGE || what == CHEM_STATE_POS_UNFOLDED) ? 3 : 1;
    if (cap < (int64_t)nglob * per) throw ChemError(CHEM_ENOSPC, "get_state: capacity");
    double* d = (double*)out; int32_t* i32 = (int32_t*)out; int64_t* i64 = (int64_t*)out;
    switch (what) {   // replicated by-tag data: no gather
      case CHEM_STATE_STATE: { std::vector<int> h; state.download(h, nglob, stream); std::copy(h.begin(), h.end(), i32); return nglob; }
      case CHEM_STATE_RESID: { std::vector<int> h; res_id.download(h, nglob, stream); std::copy(h.begin(), h.end(), i32); return nglob; }
      case CHEM_STATE_MOLID: { std::vector<int> h; mol_id.download(h, nglob, stream); for (int t = 0; t < nglob; ++t) i32[t] = (int32_t)top.id[h[t]]; return nglob; }
      case CHEM_STATE_ID: for (int t = 0; t < nglob; ++t) i64[t] = top.id[t]; return nglob;
      default: break;
    }
    // per-particle data of the owned range [G, G+n)
    std::vector<int> ht(n); std::vector<V4> hv(n); std::vector<int4> hi;
    if (n) HIPCHK(hipMemcpyAsync(ht.data(), tag.p + G, sizeof(int) * n, hipMemcpyDeviceToHost, stream));
    const V4* src = (what == CHEM_STATE_VEL || what == CHEM_STATE_MASS) ? v4.p : (what == CHEM_STATE_FORCE ? f4.p : x4.p);
    if (n) HIPCHK(hipMemcpyAsync(hv.data(), src + G, sizeof(V4) * n, hipMemcpyDeviceToHost, stream));
    if (what == CHEM_STATE_POS_UNFOLDED || what == CHEM_STATE_IMAGE) { hi.resize(n); if (n) HIPCHK(hipMemcpyAsync(hi.data(), img4.p + G, sizeof(int4) * n, hipMemcpyDeviceToHost, stream)); }
    HIPCHK(hipStreamSynchronize(stream));
    std::vector<unsigned char> loc((size_t)n * sizeof(StateRec));
    StateRec* rec = reinterpret_cast<StateRec*>(loc.data());
    for (int i = 0; i < n; ++i) {
      StateRec r{ht[i], 0, {(double)hv[i].x, (double)hv[i].y, (double)hv[i].z}};
      if (src == x4.p) { r.v[0] = dec_pos(hv[i].x, 0); r.v[1] = dec_pos(hv[i].y, 1); r.v[2] = dec_pos(hv[i].z, 2); }
      if (what == CHEM_STATE_POS) { for (int k = 0; k < 3; ++k) { double s = std::floor(r.v[k] / L[k]); r.v[k] -= s * L[k]; if (r.v[k] >= L[k]) r.v[k] -= L[k]; } }
      else if (what == CHEM_STATE_POS_UNFOLDED) { r.v[0] += hi[i].x * L[0]; r.v[1] += hi[i].y * L[1]; r.v[2] += hi[i].z * L[2]; }
      else if (what == CHEM_STATE_IMAGE) { r.v[0] = hi[i].x; r.v[1] = hi[i].y; r.v[2] = hi[i].z; }
      else if (what == CHEM_STATE_MASS || what == CHEM_STATE_TYPE) r.v[0] = (double)hv[i].w;
      rec[i] = r;
    }
    const std::vector<unsigned char> all = gather_records(loc, sizeof(StateRec));
    const StateRec* ar = reinterpret_cast<const StateRec*>(all.data());
    const size_t na = all.size() / sizeof(StateRec);
    if ((int64_t)na != nglob) throw ChemError(CHEM_ESTATE, "get_state: owned particles do not add up to the global count (" + std::to_string(na) + " vs " + std::to_string(nglob) + ")");
    for (size_t k = 0; k < na; ++k) {
      const int t = ar[k].tag;
      switch (what) {
        case CHEM_STATE_MASS: d[t] = ar[k].v[0]; break;
        case CHEM_STATE_TYPE: i32[t] = (int32_t)ar[k].v[0]; break;
        case CHEM_STATE_IMAGE: for (int c = 0; c < 3; ++c) i32[3 * t + c] = (int32_t)ar[k].v[c]; break;
        default: for (int c = 0; c < 3; ++c) d[3 * t + c] = ar[k].v[c]; break;
      }
    }
    if (what != CHEM_STATE_POS && what != CHEM_STATE_POS_UNFOLDED && what != CHEM_STATE_VEL && what != CHEM_STATE_FORCE && what != CHEM_STATE_MASS &&
        what != CHEM_STATE_TYPE && what != CHEM_STATE_IMAGE) throw ChemError(CHEM_EINVAL, "get_state: unknown selector");
    return nglob;
  }

  void observe(chem_obs* out) override {
    flush_host_state();
    if (resort) { rebuild_now(); compute_forces(); }   // a rebuild re-sorts the particles but not f4: keep it aligned for get_state(FORCE)
    std::memset(out, 0, sizeof(*out));
    const int tpp = pick_tpp();
    const int nb = launch_pair<true>(x4o.p, tpp);  // scratch force buffer: leaves f4 untouched
    HIPCHK(hipMemsetAsync(elist.p, 0, sizeof(double) * CHEM_MAX_LISTS, stream));
    if (nbent > 0)
      hipLaunchKernelGGL((k_bonded<R, true>), dim3(cdiv(n, 256)), dim3(256), 0, stream, G, n, x4.p, x4o.p, tag.p, rtag.p, bstart.p, bent.p,
                         bpar.p, boxd, elist.p, ctl.p, btab_view());
    const int nkb = cdiv(n, 256);
    hipLaunchKernelGGL(k_kinetic<R>, dim3(nkb), dim3(256), 0, stream, G, n, v4.p, ekout.p);
    std::vector<double> he, hk, hl;
    eout.download(he, 3 * (size_t)nb, stream); ekout.download(hk, 4 * (size_t)nkb, stream); elist.download(hl, CHEM_MAX_LISTS, stream);
    double acc[8 + CHEM_MAX_LISTS] = {0};   // elj, etab, vir, ek, px, py, pz, -, lists...
    for (int b = 0; b < nb; ++b) { acc[0] += he[3 * b]; acc[1] += he[3 * b + 1]; acc[2] += he[3 * b + 2]; }
    for (int b = 0; b < nkb; ++b) { acc[3] += hk[4 * b]; acc[4] += hk[4 * b + 1]; acc[5] += hk[4 * b + 2]; acc[6] += hk[4 * b + 3]; }
    for (int l = 0; l < CHEM_MAX_LISTS; ++l) acc[8 + l] = hl[l];
    if (dd_on && P > 1) {
      HIPCHK(hipMemcpyAsync(redbuf.p, acc, sizeof(acc), hipMemcpyHostToDevice, stream));
      tr->allreduce_sum_f64(redbuf.p, 8 + CHEM_MAX_LISTS, stream);
      HIPCHK(hipMemcpyAsync(acc, redbuf.p, sizeof(acc), hipMemcpyDeviceToHost, stream));
      HIPCHK(hipStreamSynchronize(stream));
    }
    out->step = step; out->npart = nglob; out->ekin = acc[3]; out->temperature = 2.0 * acc[3] / (3.0 * nglob);
    out->epot_lj = acc[0]; out->epot_tab = acc[1]; out->virial_nb = acc[2];
    for (int k = 0; k < 3; ++k) out->momentum[k] = acc[4 + k];
    for (size_t l = 0; l < top.lists.size(); ++l) { out->epot_list[l] = acc[8 + l]; out->list_size[l] = top.lists[l].size(); }
  }

  int64_t verlet_pairs(int64_t* out, int64_t cap) override {
    flush_host_state();
    ensure_list32();
    compute_forces();   // the rebuild re-sorted the particles but not f4: keep it aligned for get_state(FORCE)
    const int ac = acap();
    std::vector<int> hn, ht, hl;
    nn.download(hn, ac, stream); tag.download(ht, ac, stream); nlist.download(hl, (size_t)ac * S, stream);
    std::vector<unsigned char> loc;
    for (int i = G; i < G + n; ++i)
      for (int k = 0; k < hn[i]; ++k) {
        const int a = ht[i], b = ht[hl[(size_t)i * S + k]];
        const int pr2[2] = {std::min(a, b), std::max(a, b)};
        const unsigned char* pb = reinterpret_cast<const unsigned char*>(pr2);
        loc.insert(loc.end(), pb, pb + 8);
      }
    const std::vector<unsigned char> all = gather_records(loc, 8);
    std::vector<std::pair<int64_t, int64_t>> pr;
    pr.reserve(all.size() / 8);
    for (size_t k = 0; k + 8 <= all.size(); k += 8) { int ab[2]; std::memcpy(ab, all.data() + k, 8); pr.emplace_back(top.id[ab[0]], top.id[ab[1]]); }
    std::sort(pr.begin(), pr.end());
    pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
    const int64_t m = (int64_t)pr.size();
    if (!out) return m;
    if (cap < m) throw ChemError(CHEM_ENOSPC, "verlet_pairs: capacity");
    for (int64_t k = 0; k < m; ++k) { out[2 * k] = pr[k].first; out[2 * k + 1] = pr[k].second; }
    return m;
  }

  void set_pair_bs(int v) override { pair_bs = v; }
  void debug_enable(int on) override {
    dbg_on = on != 0;
    if (dbg_on) { dbgbuf.alloc(6 * (size_t)std::max(ntiles, 1)); wgst.alloc(8 * (size_t)std::max(fused_grid, 1)); HIPCHK(hipMemsetAsync(wgst.p, 0, 8 * sizeof(long long) * (size_t)std::max(fused_grid, 1), stream)); }
  }
  // diagnostic / tests: the 16-bit force list of one particle as partner tags, in list order (padding dropped)
  int64_t debug_force_list(int tg, int32_t* out, int64_t cap) override {
    if (!use_tiles || !device_ready) return -1;
    HIPCHK(hipStreamSynchronize(stream));
    int p = -1; HIPCHK(hipMemcpy(&p, rtag.p + tg, sizeof(int), hipMemcpyDeviceToHost));
    std::vector<TileLDS<R>> hd(ntiles);
    HIPCHK(hipMemcpy(hd.data(), tdesc.p, sizeof(TileLDS<R>) * (size_t)ntiles, hipMemcpyDeviceToHost));
    std::vector<int> htag; tag.download(htag, acap(), stream);
    for (int t = 0; t < ntiles; ++t) {
      const TileLDS<R>& T = hd[t];
      for (int sg = 0; sg < NHSEG; ++sg) {
        const int cnt = T.hoff[sg + 1] - T.hoff[sg];
        if (p < T.hstart[sg] || p >= T.hstart[sg] + cnt) continue;
        const int q = T.hoff[sg] + (p - T.hstart[sg]), nhome = T.geom[4], hbase = T.geom[5];
        int n16 = 0; HIPCHK(hipMemcpy(&n16, nnh.p + hbase + q, sizeof(int), hipMemcpyDeviceToHost));
        int64_t m = 0;
        for (int c = 0; c * 8 < n16; ++c) {
          unsigned short ch[8];
          HIPCHK(hipMemcpy(ch, nl16.p + (size_t)hbase * S + ((size_t)c * nhome + q) * 8, 16, hipMemcpyDeviceToHost));
          for (int k = 0; k < 8 && c * 8 + k < n16; ++k) {
            const int sl = ch[k];
            int r = 0; while (r + 1 < NROW && sl >= T.rowoff[r + 1]) ++r;
            const int e = sl - T.rowoff[r];
            int kc = 0; for (int qq = 1; qq < SX; ++qq) kc += e >= T.celloff[r][qq] ? 1 : 0;
            const int j = T.cellg[r][kc] + (e - T.celloff[r][kc]);
            if (m < cap) out[m] = htag[j];
            ++m;
          }
        }
        return m;
      }
    }
    return -1;
  }
  int64_t debug_dump_rebuild(long long* out, int64_t cap) override {
    if (!dbg_on || !use_fused) return 0;
    std::vector<long long> h; wgst.download(h, 8 * (size_t)fused_grid, stream);
    const int64_t m = std::min<int64_t>(cap, (int64_t)h.size());
    std::copy(h.begin(), h.begin() + m, out);
    return m;
  }
  int64_t debug_dump(long long* out, int64_t cap) override {
    if (!dbg_on) return 0;
    std::vector<long long> h; dbgbuf.download(h, 6 * (size_t)ntiles, stream);
    const int64_t m = std::min<int64_t>(cap, (int64_t)h.size());
    std::copy(h.begin(), h.begin() + m, out);
    return m;
  }

  void refresh_timers() override {
    if (!device_ready || particles_dirty) return;
    std::vector<int> hn; (use_tiles ? nnh : nn).download(hn, acap(), stream);
    long long tot = 0;
    if (use_tiles) { for (int k = 0; k < n; ++k) tot += hn[k]; } else { for (int k = G; k < G + n; ++k) tot += hn[k]; }
    tm.nlist_entries = tot; tm.nlist_capacity = S;
  }

  void modify_particle(int t, int what, double value) override {
    if (!device_ready || particles_dirty) return;  // host mirror only; uploaded later
    int idx = 0;
    HIPCHK(hipMemcpyAsync(&idx, rtag.p + t, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (what == CHEM_STATE_TYPE) { R w = (R)value; HIPCHK(hipMemcpyAsync(&x4.p[idx].w, &w, sizeof(R), hipMemcpyHostToDevice, stream)); }
    else if (what == CHEM_STATE_MASS) { R w = (R)value; HIPCHK(hipMemcpyAsync(&v4.p[idx].w, &w, sizeof(R), hipMemcpyHostToDevice, stream)); }
    else if (what == CHEM_STATE_STATE) { int w = (int)value; HIPCHK(hipMemcpyAsync(state.p + t, &w, sizeof(int), hipMemcpyHostToDevice, stream)); }
    else if (what == CHEM_STATE_RESID) { int w = (int)value; HIPCHK(hipMemcpyAsync(res_id.p + t, &w, sizeof(int), hipMemcpyHostToDevice, stream)); }
    HIPCHK(hipStreamSynchronize(stream));
  }
};

static std::string g_create_error;

}  // namespace chem

// =========================================================================================
// C ABI
// =========================================================================================
using namespace chem;

struct chem_ctx { std::unique_ptr<Ctx> c; };

#define CTX (*ctx->c)
#define API_BEGIN try { if (ctx && ctx->c) ctx->c->join_async();
#define API_BEGIN_NOJOIN try {
#define API_END(ctxp)                                                                  \
  } catch (const ChemError& e) { (ctxp)->c->err = e.what(); return e.code; }           \
    catch (const std::exception& e) { (ctxp)->c->err = e.what(); return CHEM_EINVAL; }
#define REQUIRE(cond, code, msg) do { if (!(cond)) throw ChemError((code), (msg)); } while (0)

extern "C" {

int chem_abi_version(void) { return CHEM_ABI_VERSION; }

chem_ctx* chem_create(int device_id, int precision) {
  // an exception that escapes a helper thread or a destructor ends the process: say what it was before it does
  static std::once_flag term_once;
  std::call_once(term_once, [] {
    std::set_terminate([] {
      try { if (auto e = std::current_exception()) std::rethrow_exception(e); fprintf(stderr, "[libchem_mi355] std::terminate without an active exception (a joinable std::thread destroyed?)\n"); }
      catch (const std::exception& ex) { fprintf(stderr, "[libchem_mi355] std::terminate: uncaught exception: %s\n", ex.what()); }
      catch (...) { fprintf(stderr, "[libchem_mi355] std::terminate: uncaught exception of unknown type\n"); }
      std::abort();
    });
  });
  try {
    if (precision != CHEM_PREC_F32 && precision != CHEM_PREC_F64) throw ChemError(CHEM_EINVAL, "precision must be 32 or 64");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw ChemError(CHEM_EDEVICE, "no HIP device available (libchem_mi355 has no CPU fall-back)");
    if (device_id < 0 || device_id >= ndev) throw ChemError(CHEM_EINVAL, "device id out of range");
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
      throw ChemError(CHEM_EDEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    auto* h = new chem_ctx();
    if (precision == CHEM_PREC_F32) h->c.reset(new CtxT<float>()); else h->c.reset(new CtxT<double>());
    h->c->device = device_id; h->c->prec = precision;
    return h;
  } catch (const std::exception& e) { g_create_error = e.what(); return nullptr; }
}

void chem_destroy(chem_ctx* ctx) { delete ctx; }
const char* chem_last_error(chem_ctx* ctx) { return ctx ? ctx->c->err.c_str() : g_create_error.c_str(); }

int chem_set_box(chem_ctx* ctx, const double L[3]) {
  API_BEGIN
  for (int d = 0; d < 3; ++d) { REQUIRE(L[d] > 0, CHEM_EINVAL, "box edge must be positive"); CTX.L[d] = L[d]; }
  CTX.geom_dirty = true; CTX.resort = true;
  return 0;
  API_END(ctx)
}

int chem_set_cutoff(chem_ctx* ctx, double max_cutoff, double skin) {
  API_BEGIN
  REQUIRE(max_cutoff > 0 && skin >= 0, CHEM_EINVAL, "cutoff must be > 0 and skin >= 0");
  CTX.rc = max_cutoff; CTX.skin = skin; CTX.geom_dirty = true; CTX.resort = true;
  return 0;
  API_END(ctx)
}

int chem_set_dt(chem_ctx* ctx, double dt) { API_BEGIN REQUIRE(dt > 0, CHEM_EINVAL, "dt"); CTX.dt = dt; return 0; API_END(ctx) }

int chem_set_particles(chem_ctx* ctx, int64_t n, const int64_t* id, const int32_t* type, const double* pos, const double* vel,
                       const double* mass, const double* q, const int32_t* state, const int32_t* res_id) {
  API_BEGIN
  REQUIRE(n > 0 && id && type && pos && mass, CHEM_EINVAL, "set_particles: null or empty input");
  REQUIRE(n < (1ll << 27), CHEM_EINVAL, "set_particles: too many particles for one context");
  Ctx& c = CTX; HostTopology& t = c.top;
  std::vector<int64_t> order(n);
  for (int64_t i = 0; i < n; ++i) order[i] = i;
  bool sorted = true;
  for (int64_t i = 1; i < n; ++i) if (id[i] <= id[i - 1]) { sorted = false; break; }
  if (!sorted) std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return id[a] < id[b]; });
  t.n = n; t.id.resize(n); t.type.resize(n); t.state.resize(n); t.res_id.resize(n); t.mol_id.resize(n); t.mass.resize(n); t.q.resize(n);
  t.graph.assign(n, TagRow()); t.excl.assign(n, TagRow()); t.n_excl_pairs = 0; t.excl_log.clear(); t.id2tag.clear();
  for (auto& l : t.lists) { l.ent.clear(); l.seen.clear(); }
  c.pos0.resize(3 * n); c.vel0.assign(3 * n, 0.0);
  t.contiguous = true; t.id0 = id[order[0]];
  for (int64_t k = 0; k < n; ++k) {
    const int64_t s = order[k];
    REQUIRE(k == 0 || id[s] != t.id[k - 1], CHEM_EINVAL, "duplicate particle id");
    REQUIRE(type[s] >= 0 && type[s] < CHEM_MAX_TYPES, CHEM_EINVAL, "particle type out of range [0,16)");
    REQUIRE(mass[s] > 0, CHEM_EINVAL, "particle mass must be positive");
    t.id[k] = id[s]; if (id[s] != t.id0 + k) t.contiguous = false;
    t.type[k] = type[s]; t.mass[k] = mass[s]; t.q[k] = q ? q[s] : 0.0;
    t.state[k] = state ? state[s] : 0; t.res_id[k] = res_id ? res_id[s] : (int32_t)id[s]; t.mol_id[k] = (int32_t)k;
    for (int d = 0; d < 3; ++d) { c.pos0[3 * k + d] = pos[3 * s + d]; if (vel) c.vel0[3 * k + d] = vel[3 * s + d]; }
  }
  if (!t.contiguous) for (int64_t k = 0; k < n; ++k) t.id2tag[t.id[k]] = (int32_t)k;
  c.particles_dirty = c.pair_dirty = c.bonded_dirty = c.excl_dirty = c.labels_dirty = true; c.resort = true;
  c.step = 0; c.events.clear(); c.arena.clear();
  return 0;
  API_END(ctx)
}

int chem_modify_particle(chem_ctx* ctx, int64_t id, int what, double value) {
  API_BEGIN
  Ctx& c = CTX; const int t = c.top.tag_of(id);
  REQUIRE(t >= 0, CHEM_EINVAL, "modify_particle: unknown id");
  if (what == CHEM_STATE_TYPE) { REQUIRE(value >= 0 && value < CHEM_MAX_TYPES, CHEM_EINVAL, "type"); c.top.type[t] = (int)value; c.pair_dirty = true; c.bonded_dirty = true; c.resort = true; }
  else if (what == CHEM_STATE_STATE) c.top.state[t] = (int)value;
  else if (what == CHEM_STATE_MASS) { REQUIRE(value > 0, CHEM_EINVAL, "mass"); c.top.mass[t] = value; }
  else if (what == CHEM_STATE_RESID) c.top.res_id[t] = (int)value;
  else throw ChemError(CHEM_EINVAL, "modify_particle: selector");
  c.modify_particle(t, what, value);
  return 0;
  API_END(ctx)
}

int chem_set_exclusions(chem_ctx* ctx, int64_t n, const int64_t* p) {
  API_BEGIN
  HostTopology& t = CTX.top;
  REQUIRE(t.n > 0, CHEM_ESTATE, "set particles before exclusions");
  for (auto& r : t.excl) r.clear();
  t.n_excl_pairs = 0; t.excl_log.clear();
  for (int64_t k = 0; k < n; ++k) {
    const int a = t.tag_of(p[2 * k]), b = t.tag_of(p[2 * k + 1]);
    REQUIRE(a >= 0 && b >= 0, CHEM_EINVAL, "exclusion: unknown particle id");
    t.exclude(a, b);
  }
  CTX.excl_dirty = true; CTX.resort = true;
  return 0;
  API_END(ctx)
}

int chem_nb_lj(chem_ctx* ctx, int t1, int t2, double eps, double sig, double rc, int shift_auto) {
  API_BEGIN
  REQUIRE(t1 >= 0 && t2 >= 0 && t1 < CHEM_MAX_TYPES && t2 < CHEM_MAX_TYPES, CHEM_EINVAL, "nb_lj: type out of range");
  HostPairPot p; p.kind = (sig > 0 && rc > 0) ? 1 : 0; p.eps = eps; p.sig = sig; p.rc = rc;
  if (p.kind && shift_auto) { const double s2 = sig * sig / (rc * rc), s6 = s2 * s2 * s2; p.shift = -4.0 * eps * (s6 * s6 - s6); }
  CTX.pp[t1][t2] = p; CTX.pp[t2][t1] = p; CTX.pair_dirty = true;
  return 0;
  API_END(ctx)
}

int chem_nb_table(chem_ctx* ctx, int t1, int t2, int64_t nrow, double r0, double dr, const double* e, const double* f, double rc) {
  API_BEGIN
  REQUIRE(t1 >= 0 && t2 >= 0 && t1 < CHEM_MAX_TYPES && t2 < CHEM_MAX_TYPES, CHEM_EINVAL, "nb_table: type out of range");
  REQUIRE(nrow >= 2 && dr > 0 && e && f && rc > 0, CHEM_EINVAL, "nb_table: need >=2 rows, dr>0");
  HostPairPot p; p.kind = 2; p.rc = rc; p.r0 = r0; p.dr = dr; p.e.assign(e, e + nrow); p.f.assign(f, f + nrow);
  CTX.pp[t1][t2] = p; CTX.pp[t2][t1] = p; CTX.pair_dirty = true;
  return 0;
  API_END(ctx)
}

int chem_list_create(chem_ctx* ctx, int arity, int kind, int by_types) {
  API_BEGIN
  REQUIRE(arity >= 2 && arity <= 4, CHEM_EINVAL, "list arity must be 2, 3 or 4");
  const bool ok = (arity == 2 && (kind == CHEM_POT_HARMONIC || kind == CHEM_POT_FENE || kind == CHEM_POT_TABULATED || kind == CHEM_POT_FENE_LJ || kind == CHEM_POT_LJ_BOND)) ||
                  (arity == 3 && (kind == CHEM_POT_ANG_HARMONIC || kind == CHEM_POT_ANG_COSINE || kind == CHEM_POT_ANG_TABULATED)) ||
                  (arity == 4 && (kind == CHEM_POT_DIH_NCOS || kind == CHEM_POT_DIH_RB || kind == CHEM_POT_DIH_TABULATED || kind == CHEM_POT_DIH_HARMONIC));
  REQUIRE(ok, CHEM_ENOTIMPL, "potential kind not supported for this arity");
  REQUIRE((int)CTX.top.lists.size() < CHEM_MAX_LISTS, CHEM_ENOSPC, "too many lists");
  HostList l; l.arity = arity; l.kind = kind; l.by_types = by_types ? 1 : 0;
  CTX.top.lists.push_back(std::move(l));
  return (int)CTX.top.lists.size() - 1;
  API_END(ctx)
}

int chem_table_create(chem_ctx* ctx, int64_t nrow, double r0, double dr, const double* e, const double* f) {
  API_BEGIN
  REQUIRE(nrow >= 2 && dr > 0 && e && f, CHEM_EINVAL, "table_create: need >= 2 rows, dr > 0");
  HostBondTable t; t.r0 = r0; t.dr = dr; t.e.assign(e, e + nrow); t.f.assign(f, f + nrow);
  CTX.top.btables.push_back(std::move(t));
  CTX.bonded_dirty = true;
  return (int)CTX.top.btables.size() - 1;
  API_END(ctx)
}

int chem_list_add(chem_ctx* ctx, int list, int64_t n, const int64_t* ids) {
  API_BEGIN
  HostTopology& t = CTX.top;
  REQUIRE(list >= 0 && list < (int)t.lists.size(), CHEM_EINVAL, "list handle");
  REQUIRE(t.n > 0, CHEM_ESTATE, "set particles before list entries");
  HostList& l = t.lists[list];
  std::vector<std::pair<int32_t, int32_t>> nb;
  for (int64_t e = 0; e < n; ++e) {
    int32_t tg[4];
    for (int k = 0; k < l.arity; ++k) { tg[k] = t.tag_of(ids[e * l.arity + k]); REQUIRE(tg[k] >= 0, CHEM_EINVAL, "list_add: unknown particle id"); }
    if (t.list_insert(l, tg) && l.arity == 2) nb.emplace_back(tg[0], tg[1]);
  }
  for (auto& e : nb) t.graph_add(e.first, e.second);
  std::vector<int32_t> touched;
  for (auto& e : nb) if (t.mol_id[e.first] != t.mol_id[e.second]) t.merge_cluster(e.first, e.second, false, touched);
  CTX.bonded_dirty = true; CTX.labels_dirty = true;
  return 0;
  API_END(ctx)
}

int chem_list_set_params(chem_ctx* ctx, int list, int t1, int t2, int t3, int t4, const double* p, int np) {
  API_BEGIN
  HostTopology& t = CTX.top;
  REQUIRE(list >= 0 && list < (int)t.lists.size() && p && np >= 1 && np <= CHEM_MAX_POT_PARAMS, CHEM_EINVAL, "list_set_params");
  if (t.lists[list].kind == CHEM_POT_TABULATED || t.lists[list].kind == CHEM_POT_ANG_TABULATED || t.lists[list].kind == CHEM_POT_DIH_TABULATED)
    REQUIRE(p[0] >= 0 && p[0] < (double)t.btables.size() && p[0] == (double)(int)p[0], CHEM_EINVAL, "tabulated bonded terms: parameter must be a handle from chem_table_create");
  HostList& l = t.lists[list];
  std::array<double, CHEM_MAX_POT_PARAMS> v{};
  std::copy(p, p + np, v.begin());
  if (!l.by_types) { l.plain = v; l.has_plain = true; }
  else {
    int tt[4] = {t1, t2, t3, t4};
    std::array<int, 4> key{-1, -1, -1, -1};
    for (int k = 0; k < l.arity; ++k) { REQUIRE(tt[k] >= 0 && tt[k] < CHEM_MAX_TYPES, CHEM_EINVAL, "type tuple"); key[k] = tt[k]; }
    l.typed[key] = v;
  }
  CTX.bonded_dirty = true;
  return 0;
  API_END(ctx)
}

int64_t chem_get_list(chem_ctx* ctx, int list, int64_t* out, int64_t cap) {
  API_BEGIN
  HostTopology& t = CTX.top;
  REQUIRE(list >= 0 && list < (int)t.lists.size(), CHEM_EINVAL, "list handle");
  const HostList& l = t.lists[list];
  if (!out) return l.size();
  REQUIRE(cap >= l.size(), CHEM_ENOSPC, "get_list: capacity");
  for (size_t k = 0; k < l.ent.size(); ++k) out[k] = t.id[l.ent[k]];
  return l.size();
  API_END(ctx)
}

int chem_thermostat_langevin(chem_ctx* ctx, double kT, double gamma, uint64_t seed) {
  API_BEGIN
  CTX.lang = gamma > 0 && kT >= 0; CTX.kT = kT; CTX.gamma = gamma; CTX.lang_seed = seed;
  return 0;
  API_END(ctx)
}

int chem_thermostat_langevin_types(chem_ctx* ctx, int n, const int32_t* types) {
  API_BEGIN
  REQUIRE(n >= 0 && (n == 0 || types), CHEM_EINVAL, "thermal groups");
  uint32_t m = 0;
  for (int k = 0; k < n; ++k) { REQUIRE(types[k] >= 0 && types[k] < CHEM_MAX_TYPES, CHEM_EINVAL, "thermal group type id"); m |= 1u << types[k]; }
  CTX.lang_tmask = m;
  return 0;
  API_END(ctx)
}

int chem_thermostat_rescale(chem_ctx* ctx, int kind, double kT, double param) {
  API_BEGIN
  REQUIRE(kind >= 0 && kind <= 2, CHEM_EINVAL, "thermostat_rescale: kind must be 0 (off), 1 (Berendsen) or 2 (Isokinetic)");
  if (kind) REQUIRE(kT > 0, CHEM_EINVAL, "thermostat_rescale: temperature");
  if (kind == 1) REQUIRE(param > 0, CHEM_EINVAL, "Berendsen: tau must be > 0");
  if (kind == 2) REQUIRE(param >= 1, CHEM_EINVAL, "Isokinetic: coupling must be >= 1 step");
  CTX.resc_kind = kind; CTX.resc_kT = kT; CTX.resc_param = kind == 2 ? std::floor(param) : param;
  return 0;
  API_END(ctx)
}

int chem_thermostat_svr(chem_ctx* ctx, double kT, double coupling, uint64_t seed) {
  API_BEGIN
  if (coupling > 0) REQUIRE(kT > 0, CHEM_EINVAL, "thermostat_svr: temperature");
  CTX.resc_kind = coupling > 0 ? 3 : 0; CTX.resc_kT = kT; CTX.resc_param = coupling; CTX.svr_seed = seed;
  return 0;
  API_END(ctx)
}

int chem_cap_force(chem_ctx* ctx, double max_force) { API_BEGIN CTX.cap_force = max_force > 0 ? max_force : 0; return 0; API_END(ctx) }

int chem_reaction_init(chem_ctx* ctx, int interval, int nearest, int max_per_interval, uint64_t seed) {
  API_BEGIN
  REQUIRE(interval > 0, CHEM_EINVAL, "reaction interval must be positive");
  CTX.react_init = true; CTX.interval = interval; CTX.nearest = nearest ? 1 : 0; CTX.react_seed = seed;
  CTX.max_per_interval = max_per_interval > 0 ? max_per_interval : 0;
  return 0;
  API_END(ctx)
}

int chem_reaction_add(chem_ctx* ctx, const chem_reaction_desc* d) {
  API_BEGIN
  Ctx& c = CTX;
  REQUIRE(c.react_init, CHEM_ESTATE, "chem_reaction_init first");
  REQUIRE(d, CHEM_EINVAL, "null descriptor");
  REQUIRE((int)c.reactions.size() < CHEM_MAX_REACTIONS, CHEM_ENOSPC, "too many reactions");
  REQUIRE(d->type_1 >= 0 && d->type_1 < CHEM_MAX_TYPES && d->type_2 >= 0 && d->type_2 < CHEM_MAX_TYPES, CHEM_EINVAL, "reaction types");
  REQUIRE(d->new_type_1 < CHEM_MAX_TYPES && d->new_type_2 < CHEM_MAX_TYPES, CHEM_EINVAL, "reaction new types");
  REQUIRE(d->cutoff > 0 && d->cutoff <= c.rc + 1e-12, CHEM_EINVAL, "reaction cutoff must be in (0, max_cutoff]: candidates come from the Verlet list");
  if (!d->is_virtual) REQUIRE(d->bond_list >= 0 && d->bond_list < (int)c.top.lists.size() && c.top.lists[d->bond_list].arity == 2, CHEM_EINVAL, "reaction bond_list must be an arity-2 list");
  if (d->new_type_1 >= 0) REQUIRE(d->new_mass_1 > 0, CHEM_EINVAL, "new_mass_1");
  if (d->new_type_2 >= 0) REQUIRE(d->new_mass_2 > 0, CHEM_EINVAL, "new_mass_2");
  c.reactions.push_back(*d); c.pair_dirty = true;
  return (int)c.reactions.size() - 1;
  API_END(ctx)
}

int chem_reaction_neighbour_change(chem_ctx* ctx, const chem_nb_change* r) {
  API_BEGIN
  Ctx& c = CTX;
  REQUIRE(r, CHEM_EINVAL, "null rule");
  REQUIRE(r->reaction >= 0 && r->reaction < (int)c.reactions.size(), CHEM_EINVAL, "neighbour_change: reaction index");
  REQUIRE(r->invoke_on >= 1 && r->invoke_on <= 3 && r->nb_level >= 1, CHEM_EINVAL, "neighbour_change: invoke_on must be 1, 2 or 3 and nb_level >= 1");
  REQUIRE(r->old_type >= 0 && r->old_type < CHEM_MAX_TYPES && r->new_type >= 0 && r->new_type < CHEM_MAX_TYPES, CHEM_EINVAL, "neighbour_change: types");
  REQUIRE(r->new_mass > 0, CHEM_EINVAL, "neighbour_change: new_mass");
  c.nb_rules.push_back(*r); c.pair_dirty = true;
  return 0;
  API_END(ctx)
}

int chem_reaction_constraint(chem_ctx* ctx, int reaction, int role, int nb_type, int min_state, int max_state) {
  API_BEGIN
  Ctx& c = CTX;
  REQUIRE(reaction >= 0 && reaction < (int)c.reactions.size() && reaction < 32, CHEM_EINVAL, "reaction_constraint: reaction index");
  REQUIRE((role == 1 || role == 2) && nb_type >= 0 && nb_type < CHEM_MAX_TYPES, CHEM_EINVAL, "reaction_constraint: role must be 1 or 2, nb_type a type id");
  if (c.constraints.size() < c.reactions.size()) c.constraints.resize(c.reactions.size());
  c.constraints[reaction] = Ctx::NbCons{role, nb_type, min_state, max_state};
  return 0;
  API_END(ctx)
}

int chem_reaction_restrict(chem_ctx* ctx, int reaction, int64_t n, const int64_t* p) {
  API_BEGIN
  Ctx& c = CTX;
  REQUIRE(reaction >= 0 && reaction < (int)c.reactions.size() && reaction < 32, CHEM_EINVAL, "reaction_restrict: reaction index");
  REQUIRE(n >= 0 && (n == 0 || p), CHEM_EINVAL, "reaction_restrict: pairs");
  for (int64_t k = 0; k < n; ++k) {
    const int a = c.top.tag_of(p[2 * k]), b = c.top.tag_of(p[2 * k + 1]);
    REQUIRE(a >= 0 && b >= 0 && a != b, CHEM_EINVAL, "reaction_restrict: unknown id or self pair");
    c.restrict_map[{std::min(a, b), std::max(a, b)}] |= 1u << reaction;
  }
  c.restricted_mask |= 1u << reaction; c.restrict_dirty = true;
  return 0;
  API_END(ctx)
}

int chem_atrp_init(chem_ctx* ctx, const chem_atrp_desc* d) {
  API_BEGIN
  Ctx& c = CTX;
  if (!d) { c.atrp_on = false; return 0; }
  REQUIRE(d->interval > 0 && d->num_particles > 0, CHEM_EINVAL, "atrp_init: interval and num_particles must be positive");
  REQUIRE(d->ratio_activator >= 0 && d->ratio_deactivator >= 0 && d->delta_catalyst >= 0 && d->k_activate >= 0 && d->k_deactivate >= 0, CHEM_EINVAL, "atrp_init: negative rate or ratio");
  c.atrp = *d; c.atrp_on = true;
  return 0;
  API_END(ctx)
}

int chem_atrp_add_center(chem_ctx* ctx, int type, int state, int is_activator, int new_type, double new_mass, double new_q, int delta_state) {
  API_BEGIN
  Ctx& c = CTX;
  REQUIRE(type >= 0 && type < CHEM_MAX_TYPES && new_type < CHEM_MAX_TYPES, CHEM_EINVAL, "atrp_add_center: types");
  REQUIRE(new_type < 0 || new_mass > 0, CHEM_EINVAL, "atrp_add_center: new_mass");
  c.atrp_centers.push_back(Ctx::AtrpCenter{type, state, is_activator ? 1 : 0, new_type, delta_state, new_mass, new_q});
  c.pair_dirty = true;    // (the type-pair tables must cover the new type)
  return 0;
  API_END(ctx)
}

int64_t chem_atrp_get_stats(chem_ctx* ctx, chem_atrp_stats* out, int64_t cap) {
  API_BEGIN
  Ctx& c = CTX;
  const int64_t n = (int64_t)c.atrp_stats.size();
  if (!out) return n;
  REQUIRE(cap >= n, CHEM_ENOSPC, "atrp stats capacity");
  std::copy(c.atrp_stats.begin(), c.atrp_stats.end(), out);
  return n;
  API_END(ctx)
}

int chem_topology_register(chem_ctx* ctx, int arity, int list, const int32_t* types) {
  API_BEGIN
  HostTopology& t = CTX.top;
  REQUIRE(list >= 0 && list < (int)t.lists.size() && t.lists[list].arity == arity && types, CHEM_EINVAL, "topology_register");
  std::array<int, 4> key{-1, -1, -1, -1};
  for (int k = 0; k < arity; ++k) key[k] = types[k];
  t.lists[list].registered.push_back(key);
  return 0;
  API_END(ctx)
}

int chem_reactions_enable(chem_ctx* ctx, int on) { API_BEGIN CTX.react_on = on != 0; return 0; API_END(ctx) }

int chem_reaction_set_rate(chem_ctx* ctx, int r, double rate) {
  API_BEGIN
  REQUIRE(r >= 0 && r < (int)CTX.reactions.size(), CHEM_EINVAL, "reaction index");
  CTX.reactions[r].rate = rate;
  return 0;
  API_END(ctx)
}

int chem_run(chem_ctx* ctx, int64_t nsteps) {
  API_BEGIN_NOJOIN   // a pending label merge keeps running beside the MD steps; react_step joins it
  REQUIRE(nsteps >= 0, CHEM_EINVAL, "nsteps");
  CTX.run(nsteps);
  return 0;
  API_END(ctx)
}

int64_t chem_num_particles(chem_ctx* ctx) { return ctx->c->top.n; }
int64_t chem_get_step(chem_ctx* ctx) { return ctx->c->step; }

int64_t chem_get_state(chem_ctx* ctx, int what, void* out, int64_t cap) {
  API_BEGIN
  REQUIRE(out, CHEM_EINVAL, "null output");
  return CTX.get_state(what, out, cap);
  API_END(ctx)
}

int64_t chem_get_events(chem_ctx* ctx, chem_event* out, int64_t cap) {
  API_BEGIN
  Ctx& c = CTX;
  const int64_t n = (int64_t)c.arena.size();
  if (!out) return n;
  REQUIRE(cap >= n, CHEM_ENOSPC, "get_events: capacity");
  // expand the blocks that are new since the last call, each into canonical order: (step, min id, max id)
  auto& A = c.arena;
  for (size_t bi = A.expanded_blocks; bi < A.blocks.size(); ++bi) {
    const size_t k0 = A.blocks[bi].second, k1 = bi + 1 < A.blocks.size() ? A.blocks[bi + 1].second : A.size(), first = c.events.size();
    for (size_t k = k0; k < k1; ++k)
      c.events.push_back(chem_event{A.blocks[bi].first, c.top.id[A.a[k]], c.top.id[A.b[k]], A.r[k], A.intra.size() > k ? (int32_t)A.intra[k] : 0, A.d2[k]});
    std::sort(c.events.begin() + first, c.events.end(), [](const chem_event& p, const chem_event& q) {
      return std::make_pair(std::min(p.id_a, p.id_b), std::max(p.id_a, p.id_b)) < std::make_pair(std::min(q.id_a, q.id_b), std::max(q.id_a, q.id_b));
    });
  }
  A.expanded_blocks = A.blocks.size();
  std::copy(c.events.begin(), c.events.end(), out);
  return n;
  API_END(ctx)
}

int64_t chem_get_exclusions(chem_ctx* ctx, int64_t* out, int64_t cap) {
  API_BEGIN
  HostTopology& t = CTX.top;
  if (!out) return t.n_excl_pairs;
  REQUIRE(cap >= t.n_excl_pairs, CHEM_ENOSPC, "get_exclusions: capacity");
  int64_t k = 0;
  for (int64_t a = 0; a < t.n; ++a) for (int32_t b : t.excl[a]) if (b > a) { out[2 * k] = t.id[a]; out[2 * k + 1] = t.id[b]; ++k; }
  return k;
  API_END(ctx)
}

int64_t chem_get_verlet_pairs(chem_ctx* ctx, int64_t* out, int64_t cap) { API_BEGIN return CTX.verlet_pairs(out, cap); API_END(ctx) }

int chem_observe(chem_ctx* ctx, chem_obs* out) { API_BEGIN REQUIRE(out, CHEM_EINVAL, "null output"); CTX.observe(out); return 0; API_END(ctx) }

int chem_get_timers(chem_ctx* ctx, chem_timers* out) {
  API_BEGIN
  REQUIRE(out, CHEM_EINVAL, "null output");
  CTX.refresh_timers();
  *out = CTX.tm;
  return 0;
  API_END(ctx)
}

int chem_device_sync(chem_ctx* ctx) { API_BEGIN CTX.sync(); return 0; API_END(ctx) }

int chem_set_nlist_capacity(chem_ctx* ctx, int m) { API_BEGIN REQUIRE(m >= 0, CHEM_EINVAL, "capacity"); CTX.nl_capacity_user = m; CTX.geom_dirty = true; return 0; API_END(ctx) }

int chem_set_option(chem_ctx* ctx, const char* name, double value) {
  API_BEGIN
  const std::string k = name ? name : "";
  if (k == "tpp") { const int v = (int)value; REQUIRE(v == 0 || v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64, CHEM_EINVAL, "tpp must be a power of two <= 64"); CTX.opt_tpp = v; }
  else if (k == "time_pair_kernel") CTX.opt_time_pair = value > 0 ? (int)value : 0;
  else if (k == "fuse_integrate") CTX.opt_fuse = value != 0;
  else if (k == "ablate_list") CTX.opt_ablate_list = (int)value;
  else if (k == "dd_merge") CTX.opt_dd_merge = value != 0;
  else if (k == "dd_fold") CTX.opt_dd_fold = value != 0;
  else if (k == "dd_fastx") { CTX.opt_dd_fastx = value != 0; CTX.resort = true; }
  else if (k == "lpt_tiles") { CTX.opt_lpt = value != 0; CTX.resort = true; }
  else if (k == "tiles") { CTX.opt_tiles = value != 0; CTX.geom_dirty = true; }
  else if (k == "fused_rebuild") { CTX.opt_fused = value != 0; CTX.geom_dirty = true; }
  else if (k == "overlap_halo") CTX.opt_overlap = value < 0 ? -1 : (value != 0 ? 1 : 0);
  else if (k == "pair_block") { const int v = (int)value; REQUIRE(v == 256 || v == 512 || v == 1024, CHEM_EINVAL, "pair_block must be 256, 512 or 1024"); CTX.set_pair_bs(v); }
  else if (k == "ablate") CTX.opt_ablate = (int)value;
  else if (k == "rebuild_criterion") { CTX.opt_criterion = value != 0 ? 1 : 0; CTX.resort = true; CTX.geom_dirty = true; }
  else if (k == "bonds_inline") { CTX.opt_bonds_inline = value != 0; CTX.resort = true; }
  else if (k == "bond_pass") { CTX.set_bond_pass(value != 0); CTX.resort = true; }
  else if (k == "bucket_cap") { CTX.opt_bucket_cap = (int)value; CTX.geom_dirty = true; }
  else if (k == "tile_split") { CTX.opt_tile_split = (int)value; CTX.geom_dirty = true; CTX.resort = true; }
  else if (k == "list_skin") { CTX.opt_list_skin = value; CTX.geom_dirty = true; CTX.resort = true; }
  else if (k == "debug_stamps") CTX.debug_enable((int)value);
  else if (k == "dd_self") {   // testing: one rank, ghost layers in z exchanged with itself by device copies
    REQUIRE(CTX.particles_dirty && !CTX.dd_on, CHEM_ESTATE, "dd_self must be set before the first run");
    if (value != 0) { CTX.tr.reset(new SelfTransport()); CTX.dd_on = true; CTX.P = 1; CTX.rk = 0; CTX.geom_dirty = true; }
  }
  else if (k == "dd_self_rccl") {   // testing: the same through RCCL (one-rank communicator, send/recv to self)
    REQUIRE(CTX.particles_dirty && !CTX.dd_on, CHEM_ESTATE, "dd_self_rccl must be set before the first run");
    char uid[128];
    REQUIRE(chem_comm_unique_id(uid) == 0, CHEM_ECOMM, "cannot create an RCCL unique id");
    CTX.tr.reset(new RcclTransport(1, 0, uid)); CTX.dd_on = true; CTX.P = 1; CTX.rk = 0; CTX.geom_dirty = true;
  }
  else if (k == "count_intra_inter") CTX.opt_intra_inter = value != 0;
  else if (k == "skip_inactive_pairs") { CTX.opt_skip_inactive = value != 0; CTX.pair_dirty = true; }
  else throw ChemError(CHEM_EINVAL, "unknown option " + k);
  return 0;
  API_END(ctx)
}

// diagnostic: per-tile phase stamps of the last pair-force launch (6 int64 per tile); not part of the public header
int64_t chem_debug_dump(chem_ctx* ctx, long long* out, int64_t cap) { return ctx->c->debug_dump(out, cap); }
// diagnostic: 8 int64 per workgroup of the last rebuilding k_rebuild_fused launch (phase stamps, tiles done)
int64_t chem_debug_dump_rebuild(chem_ctx* ctx, long long* out, int64_t cap) { return ctx->c->debug_dump_rebuild(out, cap); }
// diagnostic / tests: partner tags of the force list of particle `tag`, in list order; -1 without tiles
// diagnostics / tests (unlisted, like chem_debug_force_list): how many times a run stopped by the device was resumed
// ... and the tile geometry in use: out[0..5] = tiles, cells along x, wide tiles per row, width of the narrow ones, tile rows, LDS slots per tile
int64_t chem_debug_tiles(chem_ctx* ctx, int32_t* out) { try { ctx->c->debug_tiles(out); return 0; } catch (...) { return -2; } }
int64_t chem_debug_halts(chem_ctx* ctx) { return ctx && ctx->c ? ctx->c->halts_recovered : -1; }
int64_t chem_debug_force_list(chem_ctx* ctx, int32_t tag, int32_t* out, int64_t cap) { try { return ctx->c->debug_force_list(tag, out, cap); } catch (...) { return -2; } }

int chem_comm_unique_id(char uid[128]) {
  RcclApi& a = rccl_api();
  if (!a.load()) { g_create_error = a.err; return CHEM_ECOMM; }
  RcclApi::UniqueId id;
  if (a.GetUniqueId(&id) != 0) { g_create_error = "ncclGetUniqueId failed"; return CHEM_ECOMM; }
  std::memcpy(uid, id.internal, 128);
  return 0;
}

int chem_comm_init(chem_ctx* ctx, int nranks, int rank, const int node_grid[3], const char uid[128]) {
  API_BEGIN
  Ctx& c = CTX;
  REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, CHEM_EINVAL, "comm_init: rank/nranks");
  REQUIRE(!c.dd_on, CHEM_ESTATE, "comm_init: already initialised");
  if (nranks == 1 && !uid) return 0;   // single domain, nothing to do
  REQUIRE(node_grid && node_grid[0] == 1 && node_grid[1] == 1 && node_grid[2] == nranks, CHEM_ENOTIMPL,
          "comm_init: this build decomposes along z only, node grid must be (1,1,nranks)");
  REQUIRE(uid, CHEM_EINVAL, "comm_init: unique id");
  REQUIRE(c.particles_dirty, CHEM_ESTATE, "comm_init must precede the first run (particles are already on the device)");
  HIPCHK(hipSetDevice(c.device));   // the communicator binds to the calling thread's current device
  c.tr.reset(new RcclTransport(nranks, rank, uid));
  c.dd_on = true; c.P = nranks; c.rk = rank; c.geom_dirty = true;
  return 0;
  API_END(ctx)
}

// In-process ranks (several contexts of one process, one host thread each) exchanging through a
// shared hub: validates the multi-rank decomposition on a single GPU.  Same contract as
// chem_comm_init, `hub_id` names the group.
namespace chem {
IpcTransport::IpcTransport(int nr, int rk, const char* shm_name) {
  nranks = nr; rank = rk; name = shm_name ? shm_name : "";
  if (nr > IpcShm::kMaxRanks || name.empty()) throw ChemError(CHEM_EINVAL, "ipc transport: 1..16 ranks and a segment name");
  // rank 0 creates the segment, the others wait for it (the launcher hands every rank the same fresh name)
  int fd = -1;
  if (rk == 0) {
    shm_unlink(name.c_str());
    fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(IpcShm)) != 0) throw ChemError(CHEM_ECOMM, "ipc transport: cannot create shared memory " + name);
    owner = true;
  } else {
    const double t0 = now();
    struct stat st;
    while ((fd = shm_open(name.c_str(), O_RDWR, 0600)) < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(IpcShm)) {
      if (fd >= 0) { close(fd); fd = -1; }
      if (now() - t0 > 120.0) throw ChemError(CHEM_ECOMM, "ipc transport: shared memory " + name + " did not appear");
      usleep(1000);
    }
  }
  void* p = mmap(nullptr, sizeof(IpcShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) throw ChemError(CHEM_ECOMM, "ipc transport: mmap");
  shm = (IpcShm*)p;
  if (rk == 0) { shm->P = (unsigned)nr; shm->arrived = 0; shm->gen = 0; shm->attached = 0; __atomic_store_n(&shm->magic, 0xC4E31234u, __ATOMIC_RELEASE); }
  else {
    const double t0 = now();
    while (__atomic_load_n(&shm->magic, __ATOMIC_ACQUIRE) != 0xC4E31234u) { if (now() - t0 > 120.0) throw ChemError(CHEM_ECOMM, "ipc transport: segment never initialised"); usleep(1000); }
    if (shm->P != (unsigned)nr) throw ChemError(CHEM_ECOMM, "ipc transport: rank count mismatch");
  }
  __atomic_add_fetch(&shm->attached, 1u, __ATOMIC_ACQ_REL);
  barrier();
}
IpcTransport::~IpcTransport() {
  for (auto& kv : opened) (void)hipIpcCloseMemHandle(kv.second);
  if (shm) {
    const unsigned left = __atomic_sub_fetch(&shm->attached, 1u, __ATOMIC_ACQ_REL);
    munmap(shm, sizeof(IpcShm));
    if (owner || left == 0) shm_unlink(name.c_str());
  }
}
}  // namespace chem

/* One process per rank, several ranks per device allowed: the slab decomposition of chem_comm_init over hipIpc
 * memory handles with a POSIX shared-memory rendezvous named `shm_name` (same fresh name on every rank). */
int chem_comm_init_ipc(chem_ctx* ctx, int nranks, int rank, const char* shm_name) {
  API_BEGIN
  Ctx& c = CTX;
  REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks && shm_name, CHEM_EINVAL, "comm_init_ipc: rank/nranks/name");
  REQUIRE(!c.dd_on && c.particles_dirty, CHEM_ESTATE, "comm_init_ipc must precede the first run");
  HIPCHK(hipSetDevice(c.device));
  c.tr.reset(new IpcTransport(nranks, rank, shm_name));
  c.dd_on = true; c.P = nranks; c.rk = rank; c.geom_dirty = true;
  return 0;
  API_END(ctx)
}

int chem_comm_init_local(chem_ctx* ctx, int nranks, int rank, int hub_id) {
  API_BEGIN
  Ctx& c = CTX;
  REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, CHEM_EINVAL, "comm_init_local: rank/nranks");
  REQUIRE(!c.dd_on && c.particles_dirty, CHEM_ESTATE, "comm_init_local must precede the first run");
  c.tr.reset(new LocalTransport(nranks, rank, hub_id));
  c.dd_on = true; c.P = nranks; c.rk = rank; c.geom_dirty = true;
  return 0;
  API_END(ctx)
}

}  // extern "C"
