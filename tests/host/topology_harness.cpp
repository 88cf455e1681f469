// Test harness (CPU only): drives chemlab_amd/csrc/chem_host.hpp -- the host topology manager the HIP
// back end consumes -- from a plain-text script on stdin and prints its state, so that a pytest can
// compare it with an independent Python model.  Not part of the product library.
//   n <N>; type <tag> <t>; res <tag> <r>; list <arity> [typed]; reg <list> <t...>; bond <list> <a> <b> (initial, graph only)
//   newbonds <k> <a b>...  -> list_insert into list 0 + on_new_bonds;  reserve <list> <keys> (hash set sized up front);  dump
#include <cstdio>
#include <iostream>
#include <sstream>
#include "../../chemlab_amd/csrc/chem_host.hpp"
using namespace chem;
int main() {
  HostTopology t;
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream is(line);
    std::string cmd; is >> cmd;
    if (cmd == "n") { is >> t.n; t.type.assign(t.n, 0); t.res_id.resize(t.n); t.mol_id.resize(t.n); t.mass.assign(t.n, 1.0); t.q.assign(t.n, 0.0);
      t.graph.assign(t.n, TagRow()); t.excl.assign(t.n, TagRow());
      for (int64_t i = 0; i < t.n; ++i) { t.res_id[i] = (int32_t)i + 1; t.mol_id[i] = (int32_t)i; } }
    else if (cmd == "type") { int a, b; is >> a >> b; t.type[a] = b; }
    else if (cmd == "res") { int a, b; is >> a >> b; t.res_id[a] = b; }
    else if (cmd == "list") { int ar; is >> ar; HostList l; l.arity = ar; l.kind = 1; l.has_plain = true; t.lists.push_back(l); }
    else if (cmd == "reg") { int li; is >> li; std::array<int, 4> r{-1, -1, -1, -1}; for (int k = 0; k < t.lists[li].arity; ++k) is >> r[k]; t.lists[li].registered.push_back(r); }
    else if (cmd == "reserve") { int li; long long k; is >> li >> k; t.lists[li].seen.reserve((size_t)k); t.lists[li].ent.reserve(t.lists[li].ent.size() + 2 * (size_t)k); }   // what a run with reactions does up front
    else if (cmd == "bond") { int li; int32_t p[2]; is >> li >> p[0] >> p[1]; if (t.list_insert(t.lists[li], p)) { t.graph_add(p[0], p[1]); t.exclude(p[0], p[1]); } }
    else if (cmd == "newbonds") {
      int k; is >> k; std::vector<std::pair<int32_t, int32_t>> nb;
      for (int i = 0; i < k; ++i) { int32_t p[2]; is >> p[0] >> p[1]; if (t.list_insert(t.lists[0], p)) nb.emplace_back(p[0], p[1]); }
      std::vector<int32_t> touched; t.on_new_bonds(nb, touched);
    } else if (cmd == "dump") {
      for (size_t li = 0; li < t.lists.size(); ++li) {
        const HostList& l = t.lists[li];
        printf("list %zu %d %lld\n", li, l.arity, (long long)l.size());
        for (size_t e = 0; e < l.ent.size(); e += l.arity) { for (int k = 0; k < l.arity; ++k) printf("%d ", l.ent[e + k]); printf("\n"); }
      }
      std::vector<int32_t> es, el; t.build_excl(es, el);
      printf("excl %lld\n", (long long)t.n_excl_pairs);
      for (int64_t i = 0; i < t.n; ++i) { printf("%lld:", (long long)i); for (int e = es[i]; e < es[i + 1]; ++e) printf(" %d", el[e]); printf("\n"); }
      printf("labels\n");
      for (int64_t i = 0; i < t.n; ++i) printf("%d %d\n", t.res_id[i], t.mol_id[i]);
      std::vector<int32_t> bs; std::vector<HBondedEntry> be; std::vector<HBondedParam> bp; t.build_bonded(bs, be, bp);
      printf("csr %zu %zu\n", be.size(), bp.size());
      for (int64_t i = 0; i < t.n; ++i) { printf("%lld:", (long long)i); for (int e = bs[i]; e < bs[i + 1]; ++e) printf(" (%d %d %d s%d m%d)", be[e].t0, be[e].t1, be[e].t2, be[e].meta & 0x0fffffff, (be[e].meta >> 28) & 3); printf("\n"); }
    }
  }
  return 0;
}
