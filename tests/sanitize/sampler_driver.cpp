// Driver of the host-side neighbourhood sampler (csrc/sampler.hip: pure host C++) for the AddressSanitizer /
// UndefinedBehaviorSanitizer build of tests/test_sanitizers.py (the race / memory checks of the host shim;
// GPU sanitizers are not available on the pool).  A ring-with-chords graph plus isolated vertices and self-loops; samples of
// every size from 0 to all edges, repeated (the sampler restores its state through an undo log between samples), the
// error paths (sample_size > n, NULL arguments), create / destroy cycles.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "../../include/rgcn.h"

static int fail(const char* what) {
  std::fprintf(stderr, "sampler_driver: %s\n", what);
  return 1;
}

int main() {
  const int V = 257, R = 5;
  std::vector<int32_t> tri;
  for (int v = 0; v < 200; ++v) {                       // ring
    tri.push_back(v); tri.push_back(v % R); tri.push_back((v + 1) % 200);
  }
  for (int v = 0; v < 200; v += 7) {                    // chords, a hub at vertex 3
    tri.push_back(3); tri.push_back(1); tri.push_back(v);
  }
  for (int v = 210; v < 220; ++v) {                     // self-loops and a second component
    tri.push_back(v); tri.push_back(2); tri.push_back(v);
    tri.push_back(v); tri.push_back(3); tri.push_back(v + 1);
  }
  const int64_t n = (int64_t)tri.size() / 3;
  for (int cycle = 0; cycle < 3; ++cycle) {
    rgcn_sampler* s = nullptr;
    if (rgcn_sampler_create(tri.data(), n, V, &s) != RGCN_OK || !s) return fail("create");
    std::vector<int32_t> out((size_t)n + 1);
    for (int64_t k = 0; k <= n; k += (k < 8 ? 1 : 13)) {
      for (uint64_t seed = 0; seed < 3; ++seed) {
        if (rgcn_sampler_edge_neighborhood(s, k, seed + 17 * (uint64_t)cycle, out.data()) != RGCN_OK) return fail("sample");
        std::set<int32_t> seen(out.begin(), out.begin() + k);
        if ((int64_t)seen.size() != k) return fail("repeated edge id in a sample");
        for (int32_t e : seen)
          if (e < 0 || e >= n) return fail("edge id out of range");
      }
    }
    if (rgcn_sampler_edge_neighborhood(s, n, 99, out.data()) != RGCN_OK) return fail("full sample");
    if (rgcn_sampler_edge_neighborhood(s, n + 1, 1, out.data()) == RGCN_OK) return fail("sample_size > n accepted");
    if (rgcn_sampler_edge_neighborhood(s, -1, 1, out.data()) == RGCN_OK) return fail("negative sample_size accepted");
    rgcn_sampler_destroy(s);
  }
  {
    rgcn_sampler* s = nullptr;
    const int32_t bad[3] = {0, 0, V};                   // object id out of range
    if (rgcn_sampler_create(bad, 1, V, &s) == RGCN_OK) return fail("out-of-range id accepted");
    if (rgcn_sampler_create(nullptr, 1, V, &s) == RGCN_OK) return fail("NULL triples accepted");
    if (rgcn_sampler_create(tri.data(), 0, V, &s) == RGCN_OK && s) rgcn_sampler_destroy(s);   // an empty graph may be legal
    rgcn_sampler_destroy(nullptr);
  }
  std::puts("sampler_driver: ok");
  return 0;
}
