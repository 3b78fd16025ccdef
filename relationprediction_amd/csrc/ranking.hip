// Link-prediction ranking on the device: the evaluation half of the reference's scoring path.
//
// Reference: BilinearDiag.predict_all_subject_scores / predict_all_object_scores
// (code/decoders/bilinear_diag.py:51-61) give, for every query triple, sigmoid(energy) against EVERY entity;
// Scorer.evaluate_mrr + MrrScore.append_line (code/common/evaluation.py:148-153, 349-389) turn each row into
//     raw rank      = #{e : score[e] >= score[gold]}
//     filtered rank = raw rank - #{e in known : score[e] >= score[gold]} + 1
// where `known` are the entities that complete a triple seen in train / valid / test for the same
// (entity, relation) pair (the gold entity among them).  The reference encodes the full graph again for
// every chunk of 1000 triples and ranks in numpy; here the codes of the last rgcn_forward (test mode on the
// full graph) are reused, the [queries, V] energies come from one NT GEMM per chunk, and one workgroup per
// query counts both ranks.
//
// Ties: the comparison is made on fp32 sigmoid values, as in the reference, so that energies that
// saturate to the same float tie (and count against the gold entity).  The sigmoid is evaluated in double
// and rounded once to float — reproducible on the host (oracle.distmult_ranks does the same).
// Round 4: that sigmoid is monotone in the energy, so  score[e] >= score[gold]  <=>  energy[e] >= t  with t the
// SMALLEST float whose score reaches the gold score; t is found once per query by bisection over the ordered float
// bit patterns (32 sigmoids per query) and the [queries, V] pass compares energies -- 64 thousand double-precision
// exponentials per 2,000 queries instead of 58 million, the same counts bit for bit.
#include "rgcn_internal.h"

namespace rgcn {

namespace {

__device__ __forceinline__ float sigmoid_f32(float x) {
  return (float)(1.0 / (1.0 + exp(-(double)x)));
}

// Q[n,:] = codes[s_n] * W_rel[r_n]  (object side)   or   W_rel[r_n] * codes[o_n]  (subject side)
__global__ void k_rank_query(const float* __restrict__ codes, const float* __restrict__ wrel,
                             const int32_t* __restrict__ X, int n, int d, int predict_object,
                             float* __restrict__ Q, int V, int R) {
  const int row = blockIdx.x;
  if (row >= n) return;
  int ent = X[3 * row + (predict_object ? 0 : 2)], rel = X[3 * row + 1];
  // (an id out of range makes the whole call fail -- k_rank_check, read back at the end; until then nothing may fault)
  if ((unsigned)ent >= (unsigned)V) ent = 0;
  if ((unsigned)rel >= (unsigned)R) rel = 0;
  const float* e = codes + (size_t)ent * d;
  const float* r = wrel + (size_t)rel * d;
  for (int k = threadIdx.x; k < d; k += blockDim.x) Q[(size_t)row * d + k] = e[k] * r[k];
}

// floats in their numeric order as unsigned integers (and back)
__device__ __forceinline__ uint32_t float_key(float x) {
  const uint32_t b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// thr[row] = the smallest float energy whose fp32 sigmoid is >= the gold entity's: one thread per query
__global__ void k_rank_threshold(const float* __restrict__ S, int V, const int32_t* __restrict__ X, int n,
                                 int predict_object, float* __restrict__ thr) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  int gold = X[3 * row + (predict_object ? 2 : 0)];
  if ((unsigned)gold >= (unsigned)V) gold = 0;
  const float xg = S[(size_t)row * V + gold];
  const float g = sigmoid_f32(xg);
  uint32_t lo = float_key(-INFINITY), hi = float_key(xg);      // sigmoid(hi) >= g always; sigmoid(-inf) = 0
  if (sigmoid_f32(-INFINITY) >= g) {
    thr[row] = -INFINITY;                                        // the gold score is 0: everything ties with it
    return;
  }
  while (hi - lo > 1u) {                                         // invariant: score(lo) < g <= score(hi)
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (sigmoid_f32(key_float(mid)) >= g) hi = mid; else lo = mid;
  }
  thr[row] = key_float(hi);
}

__global__ void __launch_bounds__(256) k_rank_rows(const float* __restrict__ S, int V, const int32_t* __restrict__ X,
                                                   int n, int predict_object, const int64_t* __restrict__ filt_ptr,
                                                   const int32_t* __restrict__ filt_idx, const float* __restrict__ thr,
                                                   int32_t* __restrict__ raw_rank, int32_t* __restrict__ filt_rank,
                                                   int32_t* __restrict__ bad, const int64_t* __restrict__ filt_end) {
  const int row = blockIdx.x;
  if (row >= n) return;
  __shared__ int red[2][256];
  const float* s = S + (size_t)row * V;
  const float t = thr[row];
  int cnt = 0, fcnt = 0;
  for (int e = threadIdx.x; e < V; e += 256) cnt += s[e] >= t ? 1 : 0;
  int64_t fb = filt_ptr[row], fe = filt_ptr[row + 1];
  // a negative, decreasing or overlong range (k_rank_check has flagged it: the call fails) is not walked at all --
  // nothing may fault before the verdict is read back
  if (fb < 0 || fe < fb || fe > *filt_end) fb = fe = 0;
  bool oob = false;          // a filter entry out of range fails the call (k_rank_check holds the other checks)
  for (int64_t j = fb + threadIdx.x; j < fe; j += 256) {
    const int fi = filt_idx[j];
    oob = oob || (unsigned)fi >= (unsigned)V;
    fcnt += ((unsigned)fi < (unsigned)V && s[fi] >= t) ? 1 : 0;
  }
  if (oob) atomicAdd(bad, 1);
  red[0][threadIdx.x] = cnt;
  red[1][threadIdx.x] = fcnt;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    raw_rank[row] = red[0][0];
    filt_rank[row] = red[0][0] - red[1][0] + 1;
  }
}

// (the last entry of filt_ptr is the list's declared length: no range may end beyond it)
__global__ void k_rank_check(const int32_t* __restrict__ X, int n, int V, int R, const int64_t* __restrict__ filt_ptr,
                             const int32_t* __restrict__ filt_idx, int32_t* __restrict__ bad) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const int s = X[3 * row], r = X[3 * row + 1], o = X[3 * row + 2];
  // (the filter entries themselves are range-checked where they are read, 256 lanes wide, in k_rank_rows: one thread
  // walking a list of 1,700 entries here was most of a call's time on the subject side of FB15k-237)
  const bool ok = s >= 0 && s < V && o >= 0 && o < V && r >= 0 && r < R && filt_ptr[row] >= 0 &&
                  filt_ptr[row] <= filt_ptr[row + 1] && filt_ptr[row + 1] <= filt_ptr[n];
  if (!ok) atomicAdd(bad, 1);
}

}  // namespace

void rank_free(rgcn_ctx* c) {
  if (c->rank_q) (void)hipFree(c->rank_q);
  if (c->rank_s) (void)hipFree(c->rank_s);
  if (c->rank_bad) (void)hipFree(c->rank_bad);
  if (c->rank_thr) (void)hipFree(c->rank_thr);
  c->rank_thr = nullptr;
  c->rank_q = c->rank_s = nullptr;
  c->rank_bad = nullptr;
  c->rank_max = 0;
}

rgcn_status rank_reserve(rgcn_ctx* c, int64_t max_queries) {
  if (max_queries <= c->rank_max) return RGCN_OK;
  rank_free(c);
  RGCN_HIP(c, hipMalloc((void**)&c->rank_q, sizeof(float) * (size_t)max_queries * c->d));
  RGCN_HIP(c, hipMalloc((void**)&c->rank_s, sizeof(float) * (size_t)max_queries * c->V));
  RGCN_HIP(c, hipMalloc((void**)&c->rank_bad, sizeof(int32_t)));
  RGCN_HIP(c, hipMalloc((void**)&c->rank_thr, sizeof(float) * (size_t)max_queries));
  c->rank_max = max_queries;
  return RGCN_OK;
}

rgcn_status rank_compute(rgcn_ctx* c, const int32_t* X_dev, int64_t N, int predict_object, const int64_t* filt_ptr,
                         const int32_t* filt_idx, int32_t* raw_out, int32_t* filt_out) {
  const float* codes = c->H[c->L];
  // ids are validated on the device (out-of-range ids are rejected, never clamped): the verdict is read back at the END
  // of the call, with no host wait in the middle -- until then the kernels below substitute id 0 for a bad id so that
  // nothing faults, and the (meaningless) ranks of a rejected call are never returned
  RGCN_HIP(c, hipMemsetAsync(c->rank_bad, 0, sizeof(int32_t), c->stream));
  hipLaunchKernelGGL(k_rank_check, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream, X_dev, (int)N, c->V,
                     c->R, filt_ptr, filt_idx, c->rank_bad);
  RGCN_HIP(c, hipGetLastError());
  for (int64_t b = 0; b < N; b += c->rank_max) {
    const int n = (int)std::min<int64_t>(c->rank_max, N - b);
    const int32_t* X = X_dev + 3 * b;
    {
      ProfScope ps(c, "rank_query", 4.0 * 3 * n * c->d, 0);
      hipLaunchKernelGGL(k_rank_query, dim3((unsigned)n), dim3(128), 0, c->stream, codes, c->w_rel, X, n, c->d,
                         predict_object, c->rank_q, c->V, c->R);
      RGCN_HIP(c, hipGetLastError());
    }
    RGCN_TRY(gemm_f32(c, "rank_scores", true, true, n, c->V, c->d, c->rank_q, c->d, codes, c->d, c->rank_s, c->V, 1));
    {
      ProfScope ps(c, "rank_rows", 4.0 * n * c->V, 0);
      hipLaunchKernelGGL(k_rank_threshold, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, c->rank_s, c->V, X, n,
                         predict_object, c->rank_thr);
      hipLaunchKernelGGL(k_rank_rows, dim3((unsigned)n), dim3(256), 0, c->stream, c->rank_s, c->V, X, n,
                         predict_object, filt_ptr + b, filt_idx, c->rank_thr, raw_out + b, filt_out + b, c->rank_bad,
                         filt_ptr + N);
      RGCN_HIP(c, hipGetLastError());
    }
  }
  int32_t bad = 0;
  RGCN_HIP(c, hipMemcpyAsync(&bad, c->rank_bad, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  RGCN_HIP(c, hipStreamSynchronize(c->stream));
  if (bad) RGCN_FAIL(c, RGCN_ERR_INVALID, "rank: entity / relation / filter index out of range");
  return RGCN_OK;
}

}  // namespace rgcn
