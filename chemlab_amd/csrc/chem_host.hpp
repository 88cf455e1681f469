// chem_host.hpp -- host-side state of one context: particle mirrors, bonded lists, the
// topology manager (bond graph, angle/dihedral spawning, cluster labels) and the dynamic
// exclusion list.  Pure C++, no device code; the device back end consumes the CSR tables
// built here.  Events (new bonds) are rare compared to MD steps, so this irregular graph
// work stays on the host (SURVEY.md 7 "Hard parts").
//
// Reference call sites: TopologyManager / DynamicExcludeList wiring,
// src/start_simulation.py:189,211-212,378-441; FixedPair/Triple/QuadrupleList construction,
// src/chemlab/gromacs_topology.py:949-961,1086-1096,1206-1224.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/chem_mi355.h"

namespace chem {

struct ChemError : std::runtime_error {
  int code;
  ChemError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct TupleKey {
  uint64_t a, b;
  bool operator==(const TupleKey& o) const { return a == o.a && b == o.b; }
};
struct TupleKeyHash {
  size_t operator()(const TupleKey& k) const { return (size_t)(k.a * 0x9E3779B97F4A7C15ull ^ (k.b + 0x7F4A7C15ull + (k.a << 6))); }
};

// orientation-free key of a tuple of tags
inline TupleKey tuple_key(const int32_t* t, int arity) {
  int32_t f[4] = {-1, -1, -1, -1}, r[4] = {-1, -1, -1, -1};
  for (int k = 0; k < arity; ++k) { f[k] = t[k]; r[arity - 1 - k] = t[k]; }
  bool use_f = true;
  for (int k = 0; k < arity; ++k) { if (f[k] != r[k]) { use_f = f[k] < r[k]; break; } }
  const int32_t* c = use_f ? f : r;
  return {((uint64_t)(uint32_t)c[0] << 32) | (uint32_t)c[1], ((uint64_t)(uint32_t)c[2] << 32) | (uint32_t)c[3]};
}

// open-addressing set of tuple keys (no allocation per insert)
struct TupleSet {
  std::vector<TupleKey> slot; size_t used = 0;
  static bool empty_key(const TupleKey& k) { return k.a == ~0ull && k.b == ~0ull; }
  void clear() { slot.clear(); used = 0; }
  void grow() {
    std::vector<TupleKey> old; old.swap(slot);
    slot.assign(old.empty() ? 1024 : old.size() * 2, TupleKey{~0ull, ~0ull}); used = 0;
    for (auto& k : old) if (!empty_key(k)) insert(k);
  }
  void reserve(size_t want) {   // room for `want` keys without a re-hash
    size_t cap = slot.empty() ? 1024 : slot.size();
    while ((want + 1) * 10 >= cap * 7) cap *= 2;
    if (cap == slot.size()) return;
    std::vector<TupleKey> old; old.swap(slot);
    slot.assign(cap, TupleKey{~0ull, ~0ull}); used = 0;
    for (auto& k : old) if (!empty_key(k)) insert(k);
  }
  void prefetch(const TupleKey& k) const { if (!slot.empty()) __builtin_prefetch(&slot[TupleKeyHash()(k) & (slot.size() - 1)]); }
  bool insert(const TupleKey& k) {
    if ((used + 1) * 10 >= slot.size() * 7) grow();
    size_t m = slot.size() - 1, i = TupleKeyHash()(k) & m;
    while (!empty_key(slot[i])) { if (slot[i] == k) return false; i = (i + 1) & m; }
    slot[i] = k; ++used; return true;
  }
};

// sorted small set of tags with inline storage for the common case (<= 6 partners): the bond
// graph and the exclusion rows of 10^6 particles would otherwise cost one heap block each
struct TagRow {
  int32_t inl[6]; int32_t n = 0; std::vector<int32_t>* big = nullptr;
  TagRow() {}
  TagRow(const TagRow& o) : n(o.n), big(o.big ? new std::vector<int32_t>(*o.big) : nullptr) { std::copy(o.inl, o.inl + 6, inl); }
  TagRow& operator=(const TagRow& o) { if (this != &o) { delete big; n = o.n; big = o.big ? new std::vector<int32_t>(*o.big) : nullptr; std::copy(o.inl, o.inl + 6, inl); } return *this; }
  ~TagRow() { delete big; }
  const int32_t* begin() const { return big ? big->data() : inl; }
  const int32_t* end() const { return begin() + n; }
  size_t size() const { return (size_t)n; }
  void clear() { delete big; big = nullptr; n = 0; }
  bool insert(int32_t x) {
    int32_t* b = big ? big->data() : inl;
    int32_t* it = std::lower_bound(b, b + n, x);
    if (it != b + n && *it == x) return false;
    if (!big && n < 6) { std::copy_backward(it, b + n, b + n + 1); *it = x; ++n; return true; }
    if (!big) { big = new std::vector<int32_t>(inl, inl + n); }
    big->insert(big->begin() + (it - b), x); ++n; return true;
  }
};

struct HostList {
  int arity = 2, kind = 0, by_types = 0;
  std::vector<int32_t> ent;  // arity tags per entry
  TupleSet seen;
  bool has_plain = false;
  std::array<double, CHEM_MAX_POT_PARAMS> plain{};
  std::map<std::array<int, 4>, std::array<double, CHEM_MAX_POT_PARAMS>> typed;
  std::vector<std::array<int, 4>> registered;
  int64_t size() const { return (int64_t)ent.size() / arity; }
};

struct HostPairPot {
  int kind = 0;  // 0 none, 1 LJ, 2 table
  double eps = 0, sig = 0, rc = 0, shift = 0, r0 = 0, dr = 0;
  std::vector<double> e, f;
};

// Device-facing bonded parameter slot and CSR entry (mirrors md_kernels.hpp)
struct HBondedParam { int kind, list, arity, pad; double p[CHEM_MAX_POT_PARAMS]; };
struct HBondedEntry { int t0, t1, t2, meta; };

struct HostBondTable { double r0 = 0, dr = 1; std::vector<double> e, f; };

struct HostTopology {
  int64_t n = 0;
  std::vector<HostBondTable> btables;   // chem_table_create registry (tabulated bonds)
  std::vector<int64_t> id;  // tag -> external id (ascending)
  bool contiguous = true;
  int64_t id0 = 0;
  std::unordered_map<int64_t, int32_t> id2tag;
  std::vector<int32_t> type, state, res_id, mol_id;
  std::vector<double> mass, q;
  std::vector<TagRow> graph;  // sorted adjacency (bond graph)
  std::vector<TagRow> excl;   // sorted, symmetric
  std::vector<HostList> lists;
  int64_t n_excl_pairs = 0;

  int tag_of(int64_t pid) const {
    if (contiguous) { int64_t t = pid - id0; return (t >= 0 && t < n) ? (int)t : -1; }
    auto it = id2tag.find(pid);
    return it == id2tag.end() ? -1 : it->second;
  }

  static bool sorted_insert(TagRow& v, int32_t x) { return v.insert(x); }

  // every accepted pair is also appended to excl_log: the device builds its exclusion CSR from that flat,
  // append-only array (only the tail added since the last upload travels)
  std::vector<std::pair<int32_t, int32_t>> excl_log;
  bool exclude(int32_t a, int32_t b) {
    if (a == b) return false;
    bool ins = sorted_insert(excl[a], b);
    sorted_insert(excl[b], a);
    if (ins) { ++n_excl_pairs; excl_log.emplace_back(a, b); }
    return ins;
  }

  bool list_insert(HostList& l, const int32_t* t) {
    if (!l.seen.insert(tuple_key(t, l.arity))) return false;
    l.ent.insert(l.ent.end(), t, t + l.arity);
    return true;
  }

  void graph_add(int32_t a, int32_t b) { sorted_insert(graph[a], b); sorted_insert(graph[b], a); }

  // relabel the bonded cluster containing a (a-b already linked).  Labels: res_id takes
  // min(res_id[a],res_id[b]) (reaction.cfg: "the residue id will be transfered"), mol_id the
  // lowest tag of the cluster.  `touched` collects tags whose labels changed.
  std::vector<uint32_t> visit_stamp;   // flood-fill marks (epoch stamped, no per-call allocation)
  uint32_t visit_epoch = 0;
  std::vector<int32_t> flood_stack;

  void merge_cluster(int32_t a, int32_t b, bool relabel_res, std::vector<int32_t>& touched) {
    const int32_t new_res = std::min(res_id[a], res_id[b]);
    const int32_t new_mol = std::min(mol_id[a], mol_id[b]);
    if (visit_stamp.size() != (size_t)n) { visit_stamp.assign((size_t)n, 0); visit_epoch = 0; }
    if (++visit_epoch == 0) { std::fill(visit_stamp.begin(), visit_stamp.end(), 0); visit_epoch = 1; }
    flood_stack.clear(); flood_stack.push_back(a);
    visit_stamp[a] = visit_epoch;
    while (!flood_stack.empty()) {
      int32_t p = flood_stack.back(); flood_stack.pop_back();
      bool ch = false;
      if (relabel_res && res_id[p] != new_res) { res_id[p] = new_res; ch = true; }
      if (mol_id[p] != new_mol) { mol_id[p] = new_mol; ch = true; }
      if (ch) touched.push_back(p);
      for (int32_t nb : graph[p]) if (visit_stamp[nb] != visit_epoch) { visit_stamp[nb] = visit_epoch; flood_stack.push_back(nb); }
    }
  }

  // place tuple t into the first list of that arity with a matching registered type tuple
  bool spawn_tuple(int arity, const int32_t* t) {
    for (auto& l : lists) {
      if (l.arity != arity) continue;
      for (auto& reg : l.registered) {
        bool fwd = true, rev = true;
        for (int k = 0; k < arity; ++k) {
          if (type[t[k]] != reg[k]) fwd = false;
          if (type[t[arity - 1 - k]] != reg[k]) rev = false;
        }
        if (!fwd && !rev) continue;
        int32_t tt[4];
        for (int k = 0; k < arity; ++k) tt[k] = fwd ? t[k] : t[arity - 1 - k];
        if (list_insert(l, tt)) exclude(tt[0], tt[arity - 1]);
        return true;
      }
    }
    return false;
  }

  // TopologyManager reaction to freshly created bonds (canonical order = given order)
  // The reaction to new bonds has two independent halves: the cluster labels (res_id / mol_id, read
  // again only by the NEXT reaction scan) and the lists/exclusions (needed by the very next force
  // evaluation).  on_new_bonds = both, in the reference's order; the engine calls the halves
  // separately so that the label flood fills can run beside the following MD steps.
  void on_new_bonds(const std::vector<std::pair<int32_t, int32_t>>& nb, std::vector<int32_t>& touched) {
    link_new_bonds(nb);
    merge_new_bonds(nb, touched);
    spawn_for_new_bonds(nb);
  }
  void link_new_bonds(const std::vector<std::pair<int32_t, int32_t>>& nb) {
    for (size_t k = 0; k < nb.size(); ++k) {
      if (k + 8 < nb.size()) { __builtin_prefetch(&graph[nb[k + 8].first]); __builtin_prefetch(&graph[nb[k + 8].second]); }
      graph_add(nb[k].first, nb[k].second);
    }
  }
  bool spawns_tuples() const { for (auto& l : lists) if (!l.registered.empty()) return true; return false; }
  // exclusions of the new bonds only (no registered angle/dihedral types: nothing else to spawn)
  void exclude_new_bonds(const std::vector<std::pair<int32_t, int32_t>>& nb) {
    for (size_t k = 0; k < nb.size(); ++k) {
      if (k + 8 < nb.size()) { __builtin_prefetch(&excl[nb[k + 8].first]); __builtin_prefetch(&excl[nb[k + 8].second]); }
      exclude(nb[k].first, nb[k].second);
    }
  }
  void merge_new_bonds(const std::vector<std::pair<int32_t, int32_t>>& nb, std::vector<int32_t>& touched) {
    for (auto& e : nb) merge_cluster(e.first, e.second, true, touched);
  }
  void spawn_for_new_bonds(const std::vector<std::pair<int32_t, int32_t>>& nb) {
    bool any3 = false, any4 = false;
    for (auto& l : lists) { if (!l.registered.empty()) { any3 |= l.arity == 3; any4 |= l.arity == 4; } }
    for (auto& e : nb) {
      const int32_t a = e.first, b = e.second;
      exclude(a, b);
      if (any3) {
        for (int32_t n1 : graph[a]) if (n1 != b) { int32_t t[3] = {n1, a, b}; spawn_tuple(3, t); }
        for (int32_t m : graph[b]) if (m != a) { int32_t t[3] = {a, b, m}; spawn_tuple(3, t); }
      }
      if (any4) {
        for (int32_t n1 : graph[a]) if (n1 != b) {
          for (int32_t n2 : graph[n1]) if (n2 != a && n2 != b) { int32_t t[4] = {n2, n1, a, b}; spawn_tuple(4, t); }
          for (int32_t m : graph[b]) if (m != a && m != n1) { int32_t t[4] = {n1, a, b, m}; spawn_tuple(4, t); }
        }
        for (int32_t m : graph[b]) if (m != a)
          for (int32_t m2 : graph[m]) if (m2 != b && m2 != a) { int32_t t[4] = {a, b, m, m2}; spawn_tuple(4, t); }
      }
    }
  }

  // PostProcessChangeNeighboursProperty (reaction_post_process.py:76-115): particles exactly nb_level bonds away
  // from `root` whose type is old_type take the new properties.  Mirrors are updated here; the caller pushes the
  // returned changes to the device arrays.
  struct PropChange { int32_t tag, type, set_state, state; double mass, q; };
  std::vector<int32_t> nb_frontier, nb_next;
  void neighbour_change(int32_t root, const chem_nb_change& rl, std::vector<PropChange>& out) {
    if (visit_stamp.size() != (size_t)n) { visit_stamp.assign((size_t)n, 0); visit_epoch = 0; }
    if (++visit_epoch == 0) { std::fill(visit_stamp.begin(), visit_stamp.end(), 0); visit_epoch = 1; }
    nb_frontier.assign(1, root); visit_stamp[root] = visit_epoch;
    for (int lvl = 0; lvl < rl.nb_level; ++lvl) {
      nb_next.clear();
      for (int32_t p : nb_frontier) for (int32_t nb : graph[p]) if (visit_stamp[nb] != visit_epoch) { visit_stamp[nb] = visit_epoch; nb_next.push_back(nb); }
      nb_frontier.swap(nb_next);
    }
    std::sort(nb_frontier.begin(), nb_frontier.end());
    for (int32_t p : nb_frontier) {
      if (type[p] != rl.old_type) continue;
      // (rules with a state window or an increment read the state mirror: the caller refreshes it from the device first)
      if (rl.min_state < rl.max_state && !(state[p] >= rl.min_state && state[p] < rl.max_state)) continue;
      type[p] = rl.new_type; mass[p] = rl.new_mass; q[p] = rl.new_q;
      int32_t st_abs = rl.new_state;
      if (rl.set_state == 2) st_abs = state[p] + rl.new_state;
      if (rl.set_state) state[p] = st_abs;
      out.push_back(PropChange{p, rl.new_type, rl.set_state ? 1 : 0, st_abs, rl.new_mass, rl.new_q});
    }
  }

  // parameter slots of the bonded lists and the keys the device resolves them by (md_kernels.hpp SlotKey):
  // one slot per plain list, one per (typed list, type tuple), in list order
  struct HSlotKey { int list, t0, t1, t2, t3, by_types, arity, pad; };
  void build_params(std::vector<HBondedParam>& bpar, std::vector<HSlotKey>& keys) const {
    bpar.clear(); keys.clear();
    for (size_t li = 0; li < lists.size(); ++li) {
      const HostList& l = lists[li];
      if (!l.by_types) {
        if (!l.has_plain) continue;
        HBondedParam bp{l.kind, (int)li, l.arity, 0, {0}};
        std::copy(l.plain.begin(), l.plain.end(), bp.p);
        bpar.push_back(bp); keys.push_back(HSlotKey{(int)li, -1, -1, -1, -1, 0, l.arity, 0});
      } else {
        for (auto& kv : l.typed) {
          HBondedParam bp{l.kind, (int)li, l.arity, 0, {0}};
          std::copy(kv.second.begin(), kv.second.end(), bp.p);
          bpar.push_back(bp); keys.push_back(HSlotKey{(int)li, kv.first[0], kv.first[1], kv.first[2], kv.first[3], 1, l.arity, 0});
        }
      }
    }
  }

  // ---- device tables ------------------------------------------------------------------
  // Parameter slots: one per plain list, one per (typed list, type tuple).
  mutable std::vector<int32_t> fill_scratch;   // reused between calls (fresh 4 MB vectors cost their page faults every time)
  template <class VI, class VE>
  void build_bonded(VI& bstart, VE& bent, std::vector<HBondedParam>& bpar) const {
    bpar.clear();
    std::vector<int> plain_slot(lists.size(), -1);
    std::vector<std::map<std::array<int, 4>, int>> typed_slot(lists.size());
    for (size_t li = 0; li < lists.size(); ++li) {
      const HostList& l = lists[li];
      if (!l.by_types) {
        if (l.has_plain) {
          HBondedParam bp{l.kind, (int)li, l.arity, 0, {0}};
          std::copy(l.plain.begin(), l.plain.end(), bp.p);
          plain_slot[li] = (int)bpar.size(); bpar.push_back(bp);
        }
      } else {
        for (auto& kv : l.typed) {
          HBondedParam bp{l.kind, (int)li, l.arity, 0, {0}};
          std::copy(kv.second.begin(), kv.second.end(), bp.p);
          typed_slot[li][kv.first] = (int)bpar.size(); bpar.push_back(bp);
        }
      }
    }
    auto slot_of = [&](size_t li, const int32_t* t) -> int {
      const HostList& l = lists[li];
      if (!l.by_types) return plain_slot[li];
      std::array<int, 4> key{-1, -1, -1, -1}, rkey{-1, -1, -1, -1};
      for (int k = 0; k < l.arity; ++k) { key[k] = type[t[k]]; rkey[l.arity - 1 - k] = type[t[k]]; }
      auto it = typed_slot[li].find(key);
      if (it == typed_slot[li].end()) it = typed_slot[li].find(rkey);
      return it == typed_slot[li].end() ? -1 : it->second;
    };
    bstart.assign((size_t)n + 1, 0);
    // count: pairs/triples one entry per member, quadruples two
    for (size_t li = 0; li < lists.size(); ++li) {
      const HostList& l = lists[li];
      const int w = l.arity == 4 ? 2 : 1;
      for (size_t e = 0; e + l.arity <= l.ent.size(); e += l.arity) {
        if (slot_of(li, &l.ent[e]) < 0) continue;
        for (int k = 0; k < l.arity; ++k) bstart[(size_t)l.ent[e + k] + 1] += w;
      }
    }
    for (int64_t t = 0; t < n; ++t) bstart[t + 1] += bstart[t];
    if ((size_t)bstart[n] > bent.capacity()) bent.reserve((size_t)bstart[n] * 2);   // (pinned storage: reallocation is expensive)
    bent.resize((size_t)bstart[n]);
    fill_scratch.assign(bstart.begin(), bstart.end() - 1);
    std::vector<int32_t>& fill = fill_scratch;
    for (size_t li = 0; li < lists.size(); ++li) {
      const HostList& l = lists[li];
      for (size_t e = 0; e + l.arity <= l.ent.size(); e += l.arity) {
        const int32_t* tt = &l.ent[e];
        const int slot = slot_of(li, tt);
        if (slot < 0) continue;
        for (int k = 0; k < l.arity; ++k) {
          int32_t& pos = fill[tt[k]];
          bent[pos++] = HBondedEntry{tt[0], tt[1], l.arity > 2 ? tt[2] : 0, slot | (k << 28)};
          if (l.arity == 4) bent[pos++] = HBondedEntry{tt[3], 0, 0, 0};
        }
      }
    }
  }

  template <class VI>
  void build_excl(VI& estart, VI& elist) const {
    // one pass over the rows (40 B each, 10^6 of them: the pass itself is the cost); the symmetric
    // table has exactly 2 entries per excluded pair
    estart.resize((size_t)n + 1);
    if ((size_t)(2 * n_excl_pairs) > elist.capacity()) elist.reserve((size_t)(4 * n_excl_pairs));
    elist.resize((size_t)(2 * n_excl_pairs));
    int32_t pos = 0;
    for (int64_t t = 0; t < n; ++t) {
      estart[t] = pos;
      const TagRow& r = excl[t];
      if (r.n) { if (pos + r.n > (int64_t)elist.size()) elist.resize((size_t)pos + r.n + 1024); std::copy(r.begin(), r.end(), elist.begin() + pos); pos += r.n; }
    }
    estart[n] = pos;
    elist.resize((size_t)pos);
  }
};

}  // namespace chem
