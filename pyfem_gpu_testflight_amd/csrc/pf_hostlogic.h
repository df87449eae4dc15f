// Host-side logic of libpyfem_hip.so that decides correctness WITHOUT a GPU, as plain C++ (no HIP headers): the greedy
// aggregation of the AMG hierarchy and its host pattern fallbacks (pf_amg.hip), and the tables of the fused halo
// exchange (pf_halo.hip).  tests/test_host_logic.py compiles tests/host/host_logic_check.cpp against this header with
// g++ -fsanitize=address,undefined and runs it (a sanitizer build of the host side); the gloo tests
// replay the exchange tables through pf_host_fuse_tables.  No counterpart in the reference (pyamg builds its own
// aggregates, pyfem.py:2397; the reference is single process).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------
// host: greedy aggregation + coarse graph
// ------------------------------------------------------------------------------------------
static void aggregate_host(int64_t n, const std::vector<int32_t>& ptr, const std::vector<int32_t>& idx,
                           const std::vector<double>& xy, int xdim, std::vector<int32_t>& agg, int64_t& nc) {
  agg.assign(n, -1);
  nc = 0;
  std::vector<int32_t> root;
  // pass 0: nodes coupled to nothing but themselves -- grid nodes no element uses (the 20-node brick generator leaves
  // half of its grid empty) and, one level down, their aggregates -- are bundled 64 at a time, so that they vanish
  // from the hierarchy geometrically instead of surviving as one-node aggregates on every level
  int32_t bundle = -1;
  int bundled = 0;
  for (int64_t i = 0; i < n; ++i) {
    bool lonely = true;
    for (int32_t g = ptr[i]; g < ptr[i + 1] && lonely; ++g) lonely = idx[g] == (int32_t)i;
    if (!lonely) continue;
    if (bundled % 64 == 0) {
      bundle = (int32_t)nc++;
      root.push_back((int32_t)i);
    }
    agg[i] = bundle;
    ++bundled;
  }
  // pass 1: a node whose whole neighbourhood is free becomes the root of a new aggregate
  for (int64_t i = 0; i < n; ++i) {
    if (agg[i] != -1 || ptr[i + 1] == ptr[i]) continue;
    bool free_nb = true;
    for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g)
      if (agg[idx[g]] != -1) { free_nb = false; break; }
    if (!free_nb) continue;
    for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g) agg[idx[g]] = (int32_t)nc;
    root.push_back((int32_t)i);
    ++nc;
  }
  // pass 2: a leftover joins the neighbouring pass-1 aggregate it has the most edges into; ties go to the nearest root.
  // (On a triangulated grid the two corners a 7-node hexagon leaves out are tied three ways; "first found" strings
  // them onto the diagonal and the aggregates come out twice as long as wide -- 41 PCG iterations against 17.)
  std::vector<int32_t> pass1(agg);
  for (int64_t i = 0; i < n; ++i) {
    if (agg[i] != -1) continue;
    int32_t best = -1, best_cnt = 0;
    double best_d = 0.0;
    for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g) {
      const int32_t a = pass1[idx[g]];
      if (a == -1 || a == best) continue;
      bool seen = false;
      for (int32_t q = ptr[i]; q < g && !seen; ++q) seen = pass1[idx[q]] == a;
      if (seen) continue;
      int32_t cnt = 0;
      for (int32_t q = g; q < ptr[i + 1]; ++q) cnt += pass1[idx[q]] == a;
      double d = 0.0;
      for (int c = 0; c < xdim; ++c) {
        const double t = xy[(size_t)xdim * i + c] - xy[(size_t)xdim * root[a] + c];
        d += t * t;
      }
      if (cnt > best_cnt || (cnt == best_cnt && d < best_d)) { best = a; best_cnt = cnt; best_d = d; }
    }
    agg[i] = best;
  }
  // pass 3: whatever is still free (and isolated nodes) forms new aggregates with its free neighbours
  for (int64_t i = 0; i < n; ++i) {
    if (agg[i] != -1) continue;
    agg[i] = (int32_t)nc;
    for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g)
      if (agg[idx[g]] == -1) agg[idx[g]] = (int32_t)nc;
    ++nc;
  }
}

static void coarse_graph_host(int64_t n, const std::vector<int32_t>& ptr, const std::vector<int32_t>& idx,
                              const std::vector<int32_t>& agg, int64_t nc, std::vector<int32_t>& aptr,
                              std::vector<int32_t>& amem, std::vector<int32_t>& cptr, std::vector<int32_t>& cidx,
                              bool members_only) {
  aptr.assign(nc + 1, 0);
  for (int64_t i = 0; i < n; ++i) aptr[agg[i] + 1]++;
  for (int64_t a = 0; a < nc; ++a) aptr[a + 1] += aptr[a];
  amem.resize(n);
  std::vector<int32_t> cur(aptr.begin(), aptr.end() - 1);
  for (int64_t i = 0; i < n; ++i) amem[cur[agg[i]]++] = (int32_t)i;
  if (members_only) return;  // the caller builds the graph of P^T A P with the smoothed P instead
  cptr.assign(nc + 1, 0);
  cidx.clear();
  std::vector<int32_t> row;
  for (int64_t a = 0; a < nc; ++a) {
    row.clear();
    row.push_back((int32_t)a);
    for (int32_t k = aptr[a]; k < aptr[a + 1]; ++k) {
      const int32_t i = amem[k];
      for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g) row.push_back(agg[idx[g]]);
    }
    std::sort(row.begin(), row.end());
    row.erase(std::unique(row.begin(), row.end()), row.end());
    cidx.insert(cidx.end(), row.begin(), row.end());
    cptr[a + 1] = (int32_t)cidx.size();
  }
}

// Pattern of the smoothed prolongator and of the Galerkin product built with it (graph only, host):
//   row n of P couples to the aggregates of n and of its neighbours;  P^T A P couples I and J when some n in a
//   column of I has a neighbour m whose row of P contains J.
static thread_local int t_host_threads = 0;  // pf_ctx::host_threads of the hierarchy being built (PF_HOST_THREADS at pf_ctx_create)
static int host_threads() {
  if (t_host_threads > 0) return t_host_threads;
  return (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));  // GPU boxes grant ~16 CPUs per GPU
}

// run fn(chunk, begin, end) over [0, n) in contiguous chunks on host threads
template <class F>
static void parallel_chunks(int64_t n, F fn) {
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / 4096));
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back(fn, t, n * t / T, n * (t + 1) / T);
  for (auto& x : th) x.join();
}
static int parallel_chunk_count(int64_t n) {
  return (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n / 4096));
}

// concatenate per-chunk row lists into (ptr, idx); rowlen[c] holds the row lengths of chunk c in order
static void concat_rows(int64_t nrows, const std::vector<std::vector<int32_t>>& rowlen,
                        const std::vector<std::vector<int32_t>>& part, std::vector<int32_t>& ptr,
                        std::vector<int32_t>& idx) {
  ptr.assign(nrows + 1, 0);
  int64_t r = 0;
  size_t total = 0;
  for (const auto& p : part) total += p.size();
  idx.clear();
  idx.reserve(total);
  for (size_t c = 0; c < part.size(); ++c) {
    for (int32_t len : rowlen[c]) { ptr[r + 1] = ptr[r] + len; ++r; }
    idx.insert(idx.end(), part[c].begin(), part[c].end());
  }
}

static void sa_pattern_host(int64_t n, const std::vector<int32_t>& ptr, const std::vector<int32_t>& idx,
                            const std::vector<int32_t>& agg, int64_t nc, std::vector<int32_t>& pptr,
                            std::vector<int32_t>& pidx, std::vector<int32_t>& tptr, std::vector<int32_t>& tnode,
                            std::vector<int32_t>& tslot, std::vector<int32_t>& cptr, std::vector<int32_t>& cidx) {
  {  // rows of P: the aggregates of i and of its neighbours, ascending
    const int C = parallel_chunk_count(n);
    std::vector<std::vector<int32_t>> part(C), rowlen(C);
    parallel_chunks(n, [&](int c, int64_t b, int64_t e) {
      std::vector<int32_t> row;
      part[c].reserve(3 * (e - b));
      rowlen[c].reserve(e - b);
      for (int64_t i = b; i < e; ++i) {
        row.clear();
        row.push_back(agg[i]);
        for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g) row.push_back(agg[idx[g]]);
        std::sort(row.begin(), row.end());
        row.erase(std::unique(row.begin(), row.end()), row.end());
        part[c].insert(part[c].end(), row.begin(), row.end());
        rowlen[c].push_back((int32_t)row.size());
      }
    });
    concat_rows(n, rowlen, part, pptr, pidx);
  }
  const int64_t np = (int64_t)pidx.size();
  tptr.assign(nc + 1, 0);
  for (int64_t s = 0; s < np; ++s) tptr[pidx[s] + 1]++;
  for (int64_t I = 0; I < nc; ++I) tptr[I + 1] += tptr[I];
  tnode.resize(np);
  tslot.resize(np);
  std::vector<int32_t> cur(tptr.begin(), tptr.end() - 1);
  for (int64_t i = 0; i < n; ++i)
    for (int32_t s = pptr[i]; s < pptr[i + 1]; ++s) {
      const int32_t at = cur[pidx[s]]++;
      tnode[at] = (int32_t)i;
      tslot[at] = s;
    }
  {  // rows of P^T A P
    const int C = parallel_chunk_count(nc);
    std::vector<std::vector<int32_t>> part(C), rowlen(C);
    parallel_chunks(nc, [&](int c, int64_t b, int64_t e) {
      std::vector<int32_t> mark(nc, -1), row;
      rowlen[c].reserve(e - b);
      for (int64_t I = b; I < e; ++I) {
        row.clear();
        for (int32_t q = tptr[I]; q < tptr[I + 1]; ++q) {
          const int32_t i = tnode[q];
          for (int32_t g = ptr[i]; g < ptr[i + 1]; ++g) {
            const int32_t m = idx[g];
            for (int32_t s = pptr[m]; s < pptr[m + 1]; ++s) {
              const int32_t J = pidx[s];
              if (mark[J] != (int32_t)I) { mark[J] = (int32_t)I; row.push_back(J); }
            }
          }
        }
        if (mark[I] != (int32_t)I) row.push_back((int32_t)I);
        std::sort(row.begin(), row.end());
        part[c].insert(part[c].end(), row.begin(), row.end());
        rowlen[c].push_back((int32_t)row.size());
      }
    });
    concat_rows(nc, rowlen, part, cptr, cidx);
  }
}


// ------------------------------------------------------------------------------------------
// Tables of the FUSED halo exchange (pf_halo_plan_fuse), from the packed-message tables of a plan:
//   ifslot[node]            index of the node among the interface nodes (ascending node id) or -1
//   sh_ptr / sh_peer / sh_pos   per interface node: the peers that share it, ascending rank, and the offset (doubles) of
//                           the node's d values inside the message to that peer
//   tperm                   the product's tile walk: tiles (tile_nodes nodes each) that hold an interface node first,
//                           the rest after them, both in the order of base_perm (identity when NULL)
//   flags8[dof]             bit 0: own_w[dof] != 0, bit 1: the dof lies on an interface
// Returns "" or what is wrong with the plan.
// ------------------------------------------------------------------------------------------
static inline std::string pf_host_fuse_tables(int64_t nnodes, int d, int tile_nodes, int nnbr, const int32_t* nbr,
                                              const int32_t* nbr_ptr, const int32_t* send_idx, int64_t n_if,
                                              const int32_t* if_idx, const int32_t* base_perm, const double* own_w,
                                              std::vector<int32_t>& ifslot, std::vector<int32_t>& sh_ptr,
                                              std::vector<int32_t>& sh_peer, std::vector<int32_t>& sh_pos,
                                              std::vector<int32_t>& tperm, int32_t& n_iftiles, std::vector<uint8_t>& flags8) {
  if (nnodes < 0 || d < 1 || tile_nodes < 1 || nnbr < 0 || n_if < 0) return "bad sizes";
  ifslot.assign((size_t)nnodes, -1);
  std::vector<int32_t> ifnodes;
  for (int64_t i = 0; i < n_if; ++i) {
    if (if_idx[i] < 0 || if_idx[i] / d >= nnodes) return "interface entry outside the context's vectors";
    const int32_t node = if_idx[i] / d;
    if (ifslot[node] < 0) { ifslot[node] = 0; ifnodes.push_back(node); }
  }
  if ((int64_t)ifnodes.size() * d != n_if)
    return "the plan does not move d entries per node (a nodal plan on a dof context cannot be fused)";
  std::sort(ifnodes.begin(), ifnodes.end());
  for (size_t s = 0; s < ifnodes.size(); ++s) ifslot[ifnodes[s]] = (int32_t)s;
  sh_ptr.assign(ifnodes.size() + 1, 0);
  for (int k = 0; k < nnbr; ++k) {
    const int32_t b = nbr_ptr[k], e = nbr_ptr[k + 1];
    if (e < b || (e - b) % d != 0) return "a message is not a whole number of nodes";
    if (k && nbr[k] <= nbr[k - 1]) return "neighbour ranks must be ascending";
    for (int32_t i = b; i < e; i += d) {
      const int32_t v = send_idx[i];
      if (v < 0 || v % d != 0 || v / d >= nnodes || ifslot[v / d] < 0) return "packed entry is not a node of the interface";
      for (int j = 1; j < d; ++j)
        if (send_idx[i + j] != v + j) return "the entries of a node are not consecutive";
      ++sh_ptr[ifslot[v / d] + 1];
    }
  }
  for (size_t s = 0; s < ifnodes.size(); ++s) sh_ptr[s + 1] += sh_ptr[s];
  sh_peer.assign((size_t)sh_ptr.back(), 0);
  sh_pos.assign((size_t)sh_ptr.back(), 0);
  std::vector<int32_t> fill(sh_ptr.begin(), sh_ptr.end() - 1);
  for (int k = 0; k < nnbr; ++k)  // neighbours ascending => every node's sharers come out in ascending rank order
    for (int32_t i = nbr_ptr[k]; i < nbr_ptr[k + 1]; i += d) {
      const int32_t slot = fill[ifslot[send_idx[i] / d]]++;
      sh_peer[slot] = nbr[k];
      sh_pos[slot] = i - nbr_ptr[k];
    }
  const int64_t nt = (nnodes + tile_nodes - 1) / tile_nodes;
  std::vector<uint8_t> has((size_t)nt, 0), seen((size_t)nt, 0);
  for (int32_t node : ifnodes) has[node / tile_nodes] = 1;
  tperm.clear();
  tperm.reserve((size_t)nt);
  for (int pass = 0; pass < 2; ++pass) {
    for (int64_t t = 0; t < nt; ++t) {
      const int32_t tile = base_perm ? base_perm[t] : (int32_t)t;
      if (tile < 0 || tile >= nt) return "tile walk out of range";
      if (pass == 0 && seen[tile]++) return "tile walk is not a permutation";
      if ((has[tile] != 0) == (pass == 0)) tperm.push_back(tile);
    }
    if (pass == 0) n_iftiles = (int32_t)tperm.size();
  }
  flags8.assign((size_t)(nnodes * d) + 8, 0);
  if (own_w)
    for (int64_t i = 0; i < nnodes * d; ++i) flags8[i] = own_w[i] != 0.0 ? 1 : 0;
  for (int64_t i = 0; i < n_if; ++i) flags8[if_idx[i]] |= 2;
  return "";
}
