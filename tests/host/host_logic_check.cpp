// Unit checks of the host-side logic of libpyfem_hip.so (csrc/pf_hostlogic.h), built WITHOUT HIP by
// tests/test_host_logic.py with  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all  and run on
// the CPU: the greedy aggregation, the coarse graphs, the host fallback of the smoothed-prolongator patterns and the
// tables of the fused halo exchange.  Any sanitizer report or failed check ends the process with a non-zero status.
// Test infrastructure, not product code.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>

#include "pf_hostlogic.h"

#define CHECK(c)                                                          \
  do {                                                                    \
    if (!(c)) {                                                           \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); \
      std::exit(2);                                                       \
    }                                                                     \
  } while (0)

// node graph of an nx x ny grid of quads (9-point stencil) or of the same grid split into triangles (7-point)
static void grid_graph(int nx, int ny, bool tri, std::vector<int32_t>& ptr, std::vector<int32_t>& idx, std::vector<double>& xy) {
  const int NX = nx + 1, NY = ny + 1;
  ptr.assign(1, 0);
  idx.clear();
  xy.clear();
  for (int j = 0; j < NY; ++j)
    for (int i = 0; i < NX; ++i) {
      for (int dj = -1; dj <= 1; ++dj)
        for (int di = -1; di <= 1; ++di) {
          if (tri && di * dj == -1) continue;  // split along one diagonal: the anti-diagonal neighbours are not coupled
          const int a = i + di, b = j + dj;
          if (a >= 0 && a < NX && b >= 0 && b < NY) idx.push_back(b * NX + a);
        }
      ptr.push_back((int32_t)idx.size());
      xy.push_back((double)i / nx);
      xy.push_back((double)j / ny);
    }
}

static void check_aggregation(int nx, int ny, bool tri, int isolated) {
  std::vector<int32_t> ptr, idx;
  std::vector<double> xy;
  grid_graph(nx, ny, tri, ptr, idx, xy);
  int64_t n = (int64_t)ptr.size() - 1;
  for (int k = 0; k < isolated; ++k) {  // nodes no element uses: empty rows (pyfem.py:683-684 allows them)
    ptr.push_back(ptr.back());
    xy.push_back(2.0);
    xy.push_back(2.0);
    ++n;
  }
  std::vector<int32_t> agg;
  int64_t nc = 0;
  aggregate_host(n, ptr, idx, xy, 2, agg, nc);
  CHECK((int64_t)agg.size() == n && nc > 0 && nc < n);
  std::vector<int32_t> size(nc, 0);
  for (int64_t i = 0; i < n; ++i) {
    CHECK(agg[i] >= 0 && agg[i] < nc);  // every node, isolated ones included, belongs to exactly one aggregate
    ++size[agg[i]];
  }
  for (int64_t a = 0; a < nc; ++a) CHECK(size[a] >= 1);
  // an aggregate of coupled nodes is connected through its root's neighbourhood: no member is farther than two edges
  // from another (sanity bound on the shape: 3 x 3 patches on quads)
  if (!isolated) CHECK(n / nc >= (tri ? 4 : 6) && n / nc <= 12);
  std::vector<int32_t> aptr, amem, cptr, cidx;
  coarse_graph_host(n, ptr, idx, agg, nc, aptr, amem, cptr, cidx, false);
  CHECK((int64_t)aptr.size() == nc + 1 && aptr[nc] == n && (int64_t)amem.size() == n);
  std::set<int32_t> seen(amem.begin(), amem.end());
  CHECK((int64_t)seen.size() == n);
  CHECK((int64_t)cptr.size() == nc + 1);
  for (int64_t a = 0; a < nc; ++a) {  // rows sorted, unique, contain the diagonal, symmetric graph
    bool diag = false;
    for (int32_t g = cptr[a]; g < cptr[a + 1]; ++g) {
      CHECK(cidx[g] >= 0 && cidx[g] < nc);
      if (g > cptr[a]) CHECK(cidx[g] > cidx[g - 1]);
      diag |= cidx[g] == a;
      const int32_t b = cidx[g];
      CHECK(std::binary_search(cidx.begin() + cptr[b], cidx.begin() + cptr[b + 1], (int32_t)a));
    }
    CHECK(diag);
  }
  // host fallback of the smoothed-prolongator / Galerkin patterns: P rows sorted and containing the node's own
  // aggregate, the transpose a permutation of the entries, P^T A P's pattern containing the tentative coarse graph
  std::vector<int32_t> pptr, pidx, tptr, tnode, tslot, c2ptr, c2idx;
  t_host_threads = 3;
  sa_pattern_host(n, ptr, idx, agg, nc, pptr, pidx, tptr, tnode, tslot, c2ptr, c2idx);
  CHECK((int64_t)pptr.size() == n + 1 && (int64_t)tptr.size() == nc + 1 && tptr[nc] == (int32_t)pidx.size());
  for (int64_t i = 0; i < n; ++i) {
    CHECK(std::binary_search(pidx.begin() + pptr[i], pidx.begin() + pptr[i + 1], agg[i]));
    for (int32_t s = pptr[i] + 1; s < pptr[i + 1]; ++s) CHECK(pidx[s] > pidx[s - 1]);
  }
  std::vector<uint8_t> hit(pidx.size(), 0);
  for (int64_t I = 0; I < nc; ++I)
    for (int32_t q = tptr[I]; q < tptr[I + 1]; ++q) {
      CHECK(tslot[q] >= pptr[tnode[q]] && tslot[q] < pptr[tnode[q] + 1] && pidx[tslot[q]] == I && !hit[tslot[q]]);
      hit[tslot[q]] = 1;
      if (q > tptr[I]) CHECK(tnode[q] > tnode[q - 1]);
    }
  for (int64_t a = 0; a < nc; ++a)
    for (int32_t g = cptr[a]; g < cptr[a + 1]; ++g)
      CHECK(std::binary_search(c2idx.begin() + c2ptr[a], c2idx.begin() + c2ptr[a + 1], cidx[g]));
}

// exchange tables of a strip / scattered partition of an nx x ny node grid over `world` ranks, as dist.py builds them,
// then the fused tables of every rank and a replay of one exchange: what every rank "stores" through (sh_peer, sh_pos)
// must be exactly what the packed message to that peer would carry, position by position
static void check_fuse_tables(int NX, int NY, int world, int d, int tile_nodes, bool scattered, unsigned seed) {
  const int64_t nn = (int64_t)NX * NY;
  std::mt19937 rng(seed);
  // owner sets: rank of every ELEMENT (quad); a node belongs to every rank that has one of its elements
  const int ex = NX - 1, ey = NY - 1;
  std::vector<int> erank((size_t)ex * ey);
  for (int e = 0; e < ex * ey; ++e) erank[e] = scattered ? (int)(rng() % world) : (int)((int64_t)e * world / (ex * ey));
  std::vector<std::set<int>> ranks_of(nn);
  for (int j = 0; j < ey; ++j)
    for (int i = 0; i < ex; ++i)
      for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a) ranks_of[(int64_t)(j + b) * NX + (i + a)].insert(erank[(size_t)j * ex + i]);
  for (int r = 0; r < world; ++r) {
    std::vector<int64_t> nodes;  // ascending global ids of rank r: local id = position
    for (int64_t g = 0; g < nn; ++g)
      if (ranks_of[g].count(r)) nodes.push_back(g);
    if (nodes.empty()) continue;
    const int64_t nl = (int64_t)nodes.size();
    std::map<int, std::vector<int32_t>> shared;  // neighbour -> local nodes shared with it, ascending global id
    std::vector<int32_t> if_idx;
    std::vector<double> own_w((size_t)nl * d, 0.0);
    for (int64_t l = 0; l < nl; ++l) {
      const auto& rs = ranks_of[nodes[l]];
      if (*rs.begin() == r)
        for (int k = 0; k < d; ++k) own_w[l * d + k] = 1.0;
      if (rs.size() > 1) {
        for (int k = 0; k < d; ++k) if_idx.push_back((int32_t)(l * d + k));
        for (int q : rs)
          if (q != r) shared[q].push_back((int32_t)l);
      }
    }
    std::vector<int32_t> nbr, nbr_ptr(1, 0), send_idx;
    for (auto& kv : shared) {
      nbr.push_back(kv.first);
      for (int32_t l : kv.second)
        for (int k = 0; k < d; ++k) send_idx.push_back(l * d + k);
      nbr_ptr.push_back((int32_t)send_idx.size());
    }
    const int64_t nt = (nl + tile_nodes - 1) / tile_nodes;
    std::vector<int32_t> base((size_t)nt);
    for (int64_t t = 0; t < nt; ++t) base[t] = (int32_t)t;
    if (scattered) std::shuffle(base.begin(), base.end(), rng);  // a Morton-like walk to refine
    std::vector<int32_t> ifslot, sh_ptr, sh_peer, sh_pos, tperm;
    std::vector<uint8_t> flags;
    int32_t nift = -1;
    const std::string why = pf_host_fuse_tables(nl, d, tile_nodes, (int)nbr.size(), nbr.data(), nbr_ptr.data(), send_idx.data(),
                                                (int64_t)if_idx.size(), if_idx.data(), scattered ? base.data() : nullptr,
                                                own_w.data(), ifslot, sh_ptr, sh_peer, sh_pos, tperm, nift, flags);
    CHECK(why.empty());
    CHECK((int64_t)tperm.size() == nt && nift >= 0 && nift <= nt);
    std::set<int32_t> tiles(tperm.begin(), tperm.end());
    CHECK((int64_t)tiles.size() == nt);
    std::set<int32_t> first(tperm.begin(), tperm.begin() + nift);
    int64_t nifn = 0;
    for (int64_t l = 0; l < nl; ++l) {
      const bool isif = ranks_of[nodes[l]].size() > 1;
      CHECK((ifslot[l] >= 0) == isif);
      if (isif) {
        CHECK(ifslot[l] == nifn++);
        CHECK(first.count((int32_t)(l / tile_nodes)));  // its tile is walked in the interface part
        // the sharers: exactly the other ranks of the node, ascending, each with the node's position in that message
        const auto& rs = ranks_of[nodes[l]];
        int32_t k = sh_ptr[ifslot[l]];
        for (int q : rs) {
          if (q == r) continue;
          CHECK(k < sh_ptr[ifslot[l] + 1] && sh_peer[k] == q);
          const auto& msg = shared[q];
          const int64_t pos = std::lower_bound(msg.begin(), msg.end(), (int32_t)l) - msg.begin();
          CHECK(sh_pos[k] == pos * d);
          ++k;
        }
        CHECK(k == sh_ptr[ifslot[l] + 1]);
      }
      for (int kk = 0; kk < d; ++kk) CHECK(flags[l * d + kk] == ((own_w[l * d + kk] != 0.0 ? 1 : 0) | (isif ? 2 : 0)));
    }
    for (int32_t t : first) {  // and no tile without an interface node got in front
      bool any = false;
      for (int64_t l = (int64_t)t * tile_nodes; l < std::min<int64_t>(nl, (int64_t)(t + 1) * tile_nodes); ++l) any |= ifslot[l] >= 0;
      CHECK(any);
    }
    if (scattered) {  // both parts keep the order of the walk they refine
      std::vector<int32_t> posn((size_t)nt);
      for (int64_t t = 0; t < nt; ++t) posn[base[t]] = (int32_t)t;
      for (int64_t t = 1; t < nt; ++t)
        if (t != nift) CHECK(posn[tperm[t]] > posn[tperm[t - 1]]);
    }
  }
  // malformed plans are refused with a message, not walked out of bounds
  std::vector<int32_t> a, b, c, e, t;
  std::vector<uint8_t> f;
  int32_t nift = 0;
  const int32_t nbr1[1] = {1}, ptr1[2] = {0, 2}, send_bad[2] = {1, 2}, ifx[2] = {0, 1};
  CHECK(!pf_host_fuse_tables(4, 2, 64, 1, nbr1, ptr1, send_bad, 2, ifx, nullptr, nullptr, a, b, c, e, t, nift, f).empty());
  const int32_t ifbad[2] = {0, 99};
  CHECK(!pf_host_fuse_tables(4, 2, 64, 0, nullptr, ptr1, nullptr, 2, ifbad, nullptr, nullptr, a, b, c, e, t, nift, f).empty());
}

int main() {
  check_aggregation(24, 17, false, 0);
  check_aggregation(24, 17, true, 0);
  check_aggregation(7, 5, false, 70);   // more isolated nodes than one bundle of 64 holds
  check_aggregation(3, 3, true, 1);
  check_fuse_tables(41, 29, 2, 2, 64, false, 1);
  check_fuse_tables(41, 29, 3, 2, 64, false, 2);
  check_fuse_tables(41, 29, 3, 1, 128, false, 3);
  check_fuse_tables(23, 31, 4, 2, 64, true, 4);   // scattered blocks: nodes shared by up to four ranks
  check_fuse_tables(23, 31, 3, 1, 128, true, 5);
  check_fuse_tables(9, 9, 1, 2, 64, false, 6);    // one rank: no interface at all
  std::puts("host logic ok");
  return 0;
}
