// What rgcn_api.hip (the C ABI), rgcn_schedule.hip (the per-layer orchestration) and rgcn_devtools.hip (the devtools
// build's stand-alone entry points) share.  Internal to the library.
#ifndef RGCN_API_INTERNAL_H_
#define RGCN_API_INTERNAL_H_

#include <cstdlib>
#include <string>

#include "rgcn_internal.h"
#ifdef RGCN_DEVTOOLS
#include "../../include/rgcn_devtools.h"
#endif

#define RGCN_NEED(c)                \
  do {                              \
    if (!(c)) return RGCN_ERR_INVALID; \
    (void)hipSetDevice((c)->cfg.device); \
  } while (0)

namespace rgcn {

// ---- rgcn_api.hip
rgcn_status stream_join_both(rgcn_ctx* c);
rgcn_status join_abandoned_side_work(rgcn_ctx* c);
int auto_split_k(int M, int N, int K, bool narrow = false);
rgcn_status to_host(rgcn_ctx* c, void* host, const void* dev, size_t bytes);
rgcn_status to_dev(rgcn_ctx* c, void* dev, const void* host, size_t bytes);

template <class T>
inline rgcn_status dmalloc(rgcn_ctx* c, T** p, size_t n, bool zero = true) {
  *p = nullptr;
  hipError_t e = hipMalloc((void**)p, (n ? n : 1) * sizeof(T));
  if (e != hipSuccess) {
    c->err = std::string("hipMalloc of ") + std::to_string(n * sizeof(T)) + " bytes: " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? RGCN_ERR_NOMEM : RGCN_ERR_HIP;
  }
  if (zero) RGCN_HIP(c, hipMemsetAsync(*p, 0, (n ? n : 1) * sizeof(T), c->stream));
  return RGCN_OK;
}

// ---- rgcn_schedule.hip
rgcn_status fwd_begin(rgcn_ctx* c, int train, uint64_t seed, const uint8_t* masks_host);
rgcn_status fwd_layer_partial(rgcn_ctx* c, int l);
rgcn_status fwd_layer_finish(rgcn_ctx* c, int l);
// ds_ready: dcodes * dropout of the top layer, already written by the producer of dcodes (the device decoder of a train step)
rgcn_status bwd_begin(rgcn_ctx* c, const float* dcodes_dev, const float* ds_ready = nullptr);
rgcn_status bwd_layer_partial(rgcn_ctx* c, int l);
rgcn_status bwd_layer_finish(rgcn_ctx* c, int l);
rgcn_status bwd_end(rgcn_ctx* c);
rgcn_status forward_all(rgcn_ctx* c, int train, uint64_t seed, const uint8_t* masks);
rgcn_status backward_all(rgcn_ctx* c, const float* dcodes_dev, const float* ds_ready = nullptr);

}  // namespace rgcn
#endif  // RGCN_API_INTERNAL_H_
