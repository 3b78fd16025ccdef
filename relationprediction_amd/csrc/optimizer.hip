// Optimizer step on the device: global-norm gradient clipping + Adam over every trainable tensor
// ("next" row f2 of SURVEY.md 8f).  Reference chain (Appendix B): GradientClipping(max_norm) =
// tf.clip_by_global_norm (code/optimization/tensorflow_backend/algorithms.py:58-68) wrapped around
// Adam(learning_rate) = tf.train.AdamOptimizer(lr, beta1=0.9, beta2=0.999), epsilon 1e-8 (:27-42).
//
//   scale = max_norm / max(||g||_2, max_norm)                 (clip_by_global_norm)
//   lr_t  = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
//   w    -= lr_t * m / (sqrt(v) + eps)                         (TF's AdamOptimizer update)
// One table-driven launch covers all tensors (device layouts: Adam is elementwise, so the private
// block / basis weight layouts need no conversion); the norm is a two-stage fixed-order reduction whose
// result stays on the device -- no host synchronisation, the whole train step is one stream of launches.
// The unused per-layer bias `b` gets no gradient in TF (None) and is skipped; W_relation is updated on
// its first RelationCount rows only (the other rows never receive a gradient).
// Deviation kept on purpose: TF computes the global norm over UN-aggregated IndexedSlices for gathered
// weights; here gradients are already aggregated (trajectory parity with TF is unpinned either way).
#include "rgcn_internal.h"

namespace rgcn {

namespace {

constexpr int kOptThreads = 256;
constexpr int kOptItems = 8;          // elements per thread
constexpr int kOptMaxTensors = 40;

struct OptTensor {
  float* w;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  int32_t block0;     // first block of this tensor
};
struct OptTable {
  OptTensor t[kOptMaxTensors];
  int32_t count;
  int32_t nblocks;
};

__device__ __forceinline__ int find_tensor(const OptTable& tab, int b) {
  int lo = 0, hi = tab.count;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (tab.t[mid].block0 <= b) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(kOptThreads) k_sqnorm_part(OptTable tab, float* __restrict__ part) {
  __shared__ float red[kOptThreads / 64];
  const int ti = find_tensor(tab, blockIdx.x);
  const OptTensor t = tab.t[ti];
  const int64_t base = (int64_t)(blockIdx.x - t.block0) * kOptThreads * kOptItems;
  // 16 B per lane and load where the tensor starts on a 16-byte boundary (every hipMalloc'ed gradient does; a slice of
  // the contiguous replicated-gradient buffer of a sharded run need not); a tensor's last few elements one by one
  const bool aligned = (reinterpret_cast<uintptr_t>(t.g) & 15) == 0;
  float acc = 0.f;
  float4 q[kOptItems / 4];
#pragma unroll
  for (int k = 0; k < kOptItems / 4; ++k) {
    const int64_t i = base + ((int64_t)k * kOptThreads + threadIdx.x) * 4;
    q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (aligned && i + 3 < t.n) {
      q[k] = *reinterpret_cast<const float4*>(t.g + i);
    } else if (i < t.n) {
      q[k].x = t.g[i];
      if (i + 1 < t.n) q[k].y = t.g[i + 1];
      if (i + 2 < t.n) q[k].z = t.g[i + 2];
      if (i + 3 < t.n) q[k].w = t.g[i + 3];
    }
  }
#pragma unroll
  for (int k = 0; k < kOptItems / 4; ++k) {
    acc = fmaf(q[k].x, q[k].x, acc);
    acc = fmaf(q[k].y, q[k].y, acc);
    acc = fmaf(q[k].z, q[k].z, acc);
    acc = fmaf(q[k].w, q[k].w, acc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// state[0] = clip scale, state[1] = global norm (for inspection), state[2] = step count t, state[3] = the
// bias-corrected rate lr * sqrt(1 - beta2^t) / (1 - beta1^t).  t advances HERE, on the device, so that a
// captured hipGraph of the train step (rgcn_capture_*) keeps counting when it is replayed.
// Sum of a range of block partials, as one float: the squared norm of this rank's relation-sharded gradients,
// which a multi-GPU run sum-all-reduces before k_clip_scale adds it to the replicated tensors' partials.
__global__ void __launch_bounds__(256) k_sum_range(const float* __restrict__ part, int nparts, float* __restrict__ out) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)red[0];
}

__global__ void __launch_bounds__(256) k_clip_scale(const float* __restrict__ part, int nparts,
                                                    const float* __restrict__ exchanged, float max_norm,
                                                    float lr, float b1, float b2, float* __restrict__ state) {
  __shared__ double red[256];
  double a = (exchanged && threadIdx.x == 0) ? (double)exchanged[0] : 0.0;
  // a thread's partials part[tid], part[tid + 256], ... added in that order -- but fetched eight at a time: one after the
  // other they are twenty dependent trips to the L2 (5,000 partials at FB15k-237 size), most of this kernel's time
  for (int i0 = threadIdx.x; i0 < nparts; i0 += 8 * 256) {
    float p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = i0 + k * 256 < nparts ? part[i0 + k * 256] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (i0 + k * 256 < nparts) a += p[k];
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    state[1] = norm;
    state[0] = max_norm > 0.f ? max_norm / fmaxf(norm, max_norm) : 1.0f;
    const double t = (double)state[2] + 1.0;
    state[2] = (float)t;
    state[3] = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
}

__global__ void __launch_bounds__(kOptThreads) k_adam(OptTable tab, const float* __restrict__ state,
                                                      float b1, float b2, float eps) {
  const int ti = find_tensor(tab, blockIdx.x);
  const OptTensor t = tab.t[ti];
  const float scale = state[0], lr_t = state[3];
  const int64_t base = (int64_t)(blockIdx.x - t.block0) * kOptThreads * kOptItems;
#pragma unroll
  for (int k = 0; k < kOptItems; ++k) {
    const int64_t i = base + (int64_t)k * kOptThreads + threadIdx.x;
    if (i < t.n) {
      const float g = t.g[i] * scale;
      const float m = fmaf(b1, t.m[i], (1.0f - b1) * g);
      const float v = fmaf(b2, t.v[i], (1.0f - b2) * g * g);
      t.m[i] = m;
      t.v[i] = v;
      t.w[i] -= lr_t * m / (sqrtf(v) + eps);
    }
  }
}

}  // namespace

namespace {

// Relation-sharded tensors: block W_forward / W_backward, basis C_forward / C_backward.  On a
// world > 1 context each rank holds the gradient of the relations it owns (zeros elsewhere), so their squared
// norm is a sum over ranks; everything else is replicated and identical on every rank.
bool is_sharded_param(const rgcn_ctx* c, const Param& p) {
  if (c->world <= 1) return false;
  const char* pre = c->kind == RGCN_KIND_BLOCK ? "W_" : "C_";
  return p.name.compare(0, 2, pre) == 0 && (p.name[2] == 'f' || p.name[2] == 'b') && p.name != "W_relation";
}

// the launch table: replicated tensors first (blocks [0, *nrep)), sharded ones behind them
rgcn_status build_table(rgcn_ctx* c, OptTable& tab, int* nrep, double* total) {
  OptimizerState& o = c->opt;
  tab.count = 0;
  int nblocks = 0;
  *total = 0;
  if (o.m.empty()) {
    o.m.assign(c->params.size(), nullptr);
    o.v.assign(c->params.size(), nullptr);
  }
  for (int pass = 0; pass < 2; ++pass) {
    for (size_t i = 0; i < c->params.size(); ++i) {
      const Param& p = c->params[i];
      if (p.no_grad) continue;                       // the never-used layer bias: TF returns None for it
      if ((int)is_sharded_param(c, p) != pass) continue;
      int64_t n = p.count;
      if (p.name == "W_relation") n = (int64_t)c->R * c->d;     // rows >= RelationCount never get a gradient
      if (!o.m[i]) {
        RGCN_HIP(c, hipMalloc((void**)&o.m[i], sizeof(float) * (size_t)n));
        RGCN_HIP(c, hipMalloc((void**)&o.v[i], sizeof(float) * (size_t)n));
        RGCN_HIP(c, hipMemsetAsync(o.m[i], 0, sizeof(float) * (size_t)n, c->stream));
        RGCN_HIP(c, hipMemsetAsync(o.v[i], 0, sizeof(float) * (size_t)n, c->stream));
      }
      if (tab.count >= kOptMaxTensors) RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "too many tensors for one optimizer launch");
      OptTensor& t = tab.t[tab.count++];
      t.w = p.val; t.g = p.grad; t.m = o.m[i]; t.v = o.v[i]; t.n = n; t.block0 = nblocks;
      nblocks += (int)((n + kOptThreads * kOptItems - 1) / (kOptThreads * kOptItems));
      *total += (double)n;
    }
    if (pass == 0) *nrep = nblocks;
  }
  tab.nblocks = nblocks;
  if ((size_t)nblocks > o.part_cap) {
    if (o.part) (void)hipFree(o.part);
    RGCN_HIP(c, hipMalloc((void**)&o.part, sizeof(float) * (size_t)nblocks));
    o.part_cap = (size_t)nblocks;
  }
  if (!o.state) {
    RGCN_HIP(c, hipMalloc((void**)&o.state, 4 * sizeof(float)));
    RGCN_HIP(c, hipMemsetAsync(o.state, 0, 4 * sizeof(float), c->stream));
  }
  if (c->world > 1 && !o.shard_sq) {
    RGCN_HIP(c, hipMalloc((void**)&o.shard_sq, sizeof(float)));
    RGCN_HIP(c, hipMemsetAsync(o.shard_sq, 0, sizeof(float), c->stream));
  }
  return RGCN_OK;
}

}  // namespace

// Phase 1: squared-norm partials of every gradient; on a sharded context the relation-sharded tensors' share
// ends in opt.shard_sq (RGCN_BUF_NORM_EXCHANGE), which the caller sum-all-reduces before phase 2.
rgcn_status optimizer_norm_partial(rgcn_ctx* c) {
  OptimizerState& o = c->opt;
  if (!o.configured) RGCN_FAIL(c, RGCN_ERR_STATE, "rgcn_optimizer_config was not called");
  OptTable tab;
  int nrep = 0;
  double total = 0;
  RGCN_TRY(build_table(c, tab, &nrep, &total));
  {
    ProfScope ps(c, "opt_grad_norm", 4.0 * total, 2.0 * total);
    hipLaunchKernelGGL(k_sqnorm_part, dim3(tab.nblocks), dim3(kOptThreads), 0, c->stream, tab, o.part);
    if (c->world > 1)
      hipLaunchKernelGGL(k_sum_range, dim3(1), dim3(256), 0, c->stream, o.part + nrep, tab.nblocks - nrep, o.shard_sq);
  }
  RGCN_HIP(c, hipGetLastError());
  o.norm_pending = true;
  return RGCN_OK;
}

// Phase 2: clip scale from the (exchanged) norm, then Adam on every tensor.
rgcn_status optimizer_apply(rgcn_ctx* c) {
  OptimizerState& o = c->opt;
  if (!o.norm_pending) RGCN_FAIL(c, RGCN_ERR_STATE, "rgcn_optimizer_apply without rgcn_optimizer_norm_partial");
  o.norm_pending = false;
  OptTable tab;
  int nrep = 0;
  double total = 0;
  RGCN_TRY(build_table(c, tab, &nrep, &total));
  o.t += 1;        // host-side mirror (exact only while no captured graph is replayed)
  c->weights_version += 1;   // derived copies of the weights (block-major tables) are stale after this step
  {
    ProfScope ps(c, "opt_clip_scale", 4.0 * tab.nblocks, 0);
    hipLaunchKernelGGL(k_clip_scale, dim3(1), dim3(256), 0, c->stream, o.part, nrep,
                       c->world > 1 ? (const float*)o.shard_sq : (const float*)nullptr, o.max_norm, o.lr, o.beta1,
                       o.beta2, o.state);
  }
  {
    ProfScope ps(c, "opt_adam", 28.0 * total, 10.0 * total);
    hipLaunchKernelGGL(k_adam, dim3(tab.nblocks), dim3(kOptThreads), 0, c->stream, tab, o.state, o.beta1, o.beta2,
                       o.eps);
  }
  RGCN_HIP(c, hipGetLastError());
  return RGCN_OK;
}

rgcn_status optimizer_step(rgcn_ctx* c) {
  RGCN_TRY(optimizer_norm_partial(c));
  if (c->world > 1) RGCN_TRY(comm_allreduce(c, c->opt.shard_sq, 1));
  return optimizer_apply(c);
}

void optimizer_free(rgcn_ctx* c) {
  OptimizerState& o = c->opt;
  for (float* p : o.m) if (p) (void)hipFree(p);
  for (float* p : o.v) if (p) (void)hipFree(p);
  if (o.part) (void)hipFree(o.part);
  if (o.state) (void)hipFree(o.state);
  if (o.shard_sq) (void)hipFree(o.shard_sq);
  o = OptimizerState();
}

}  // namespace rgcn
