// Entry points that exist in librgcn_devtools.so only (include/rgcn_devtools.h): the dense contractions on their own, the
// XCD placement probe.  The product library compiles this file to nothing.
#include <cstdio>
#include <vector>

#include "rgcn_api_internal.h"

using namespace rgcn;

extern "C" {

#ifdef RGCN_DEVTOOLS
namespace {
// the XCD every workgroup of a plain 1-D launch lands on (HW_REG_XCC_ID, bits 3:0)
__global__ void k_xcd_of_block(int32_t* out) {
  uint32_t id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) out[blockIdx.x] = (int32_t)(id & 0xf);
}
}  // namespace

rgcn_status rgcn_debug_xcd_map(rgcn_ctx* c, int32_t n_blocks, int32_t* out_host) {
  RGCN_NEED(c);
  if (n_blocks <= 0 || !out_host) RGCN_FAIL(c, RGCN_ERR_INVALID, "bad arguments");
  int32_t* dev = nullptr;
  RGCN_TRY(dmalloc(c, &dev, (size_t)n_blocks));
  hipLaunchKernelGGL(k_xcd_of_block, dim3((unsigned)n_blocks), dim3(256), 0, c->stream, dev);
  rgcn_status s = hipGetLastError() == hipSuccess ? RGCN_OK : RGCN_ERR_HIP;
  if (s == RGCN_OK) s = to_host(c, out_host, dev, sizeof(int32_t) * (size_t)n_blocks);
  (void)hipFree(dev);
  return s;
}

rgcn_status rgcn_debug_gemm(rgcn_ctx* c, int32_t ta, int32_t tb, int32_t M, int32_t N, int32_t K,
                            int32_t split_k, const float* a_host, const float* b_host, float* c_host) {
  RGCN_NEED(c);
  if (M <= 0 || N <= 0 || K <= 0 || !a_host || !b_host || !c_host) RGCN_FAIL(c, RGCN_ERR_INVALID, "bad arguments");
  if (ta && tb) RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "TT form not instantiated");
  float *A = nullptr, *B = nullptr, *C = nullptr;
  rgcn_status s = RGCN_OK;
  do {
    if ((s = dmalloc(c, &A, (size_t)M * K, false)) != RGCN_OK) break;
    if ((s = dmalloc(c, &B, (size_t)K * N, false)) != RGCN_OK) break;
    if ((s = dmalloc(c, &C, (size_t)M * N)) != RGCN_OK) break;
    if ((s = to_dev(c, A, a_host, sizeof(float) * (size_t)M * K)) != RGCN_OK) break;
    if ((s = to_dev(c, B, b_host, sizeof(float) * (size_t)K * N)) != RGCN_OK) break;
    int sk = split_k > 0 ? split_k : auto_split_k(M, N, K);
    if ((size_t)sk * M * N > c->slab_floats) sk = 1;
    // trans_a: A given as [K,M] (row-contiguous operand); trans_b: B given as [N,K] (k-contiguous)
    s = gemm_f32(c, "debug_gemm", ta == 0, tb != 0, M, N, K, A, ta ? M : K, B, tb ? K : N, C, N, sk);
    if (s != RGCN_OK) break;
    s = to_host(c, c_host, C, sizeof(float) * (size_t)M * N);
  } while (0);
  if (A) (void)hipFree(A);
  if (B) (void)hipFree(B);
  if (C) (void)hipFree(C);
  return s;
}

rgcn_status rgcn_debug_gemm_presplit(rgcn_ctx* c, int32_t tb, int32_t M, int32_t N, int32_t K, int32_t iters,
                                     const float* a_host, const float* b_host, float* c_host, float* avg_ms) {
  RGCN_NEED(c);
  if (M <= 0 || N <= 0 || K <= 0 || !a_host || !b_host || !c_host) RGCN_FAIL(c, RGCN_ERR_INVALID, "bad arguments");
  float *A = nullptr, *B = nullptr, *C = nullptr;
  void* F = nullptr;
  rgcn_status s = RGCN_OK;
  do {
    if ((s = dmalloc(c, &A, (size_t)M * K, false)) != RGCN_OK) break;
    if ((s = dmalloc(c, &B, (size_t)K * N, false)) != RGCN_OK) break;
    if ((s = dmalloc(c, &C, (size_t)M * N)) != RGCN_OK) break;
    if (hipMalloc(&F, 16 * gemm_bfrag_words(K, N)) != hipSuccess) { s = RGCN_ERR_NOMEM; break; }
    if ((s = to_dev(c, A, a_host, sizeof(float) * (size_t)M * K)) != RGCN_OK) break;
    if ((s = to_dev(c, B, b_host, sizeof(float) * (size_t)K * N)) != RGCN_OK) break;
    const PresplitJob pj{B, F, tb ? K : N, K, N, tb ? 1 : 0};
    if ((s = gemm_presplit_b(c, &pj, 1)) != RGCN_OK) break;
    GemmBatch gb;
    gb.bfrag = F;
    gb.wide = 1;
    s = gemm_f32(c, "debug_gemm", true, tb != 0, M, N, K, A, K, B, tb ? K : N, C, N, 1, &gb);
    if (s != RGCN_OK) break;
    if ((s = to_host(c, c_host, C, sizeof(float) * (size_t)M * N)) != RGCN_OK) break;
    // RGCN_GEMM_TL_FILE: the per-wavefront stamps of ONE more launch (k_gemm_w8<.., DBG_TIMELINE / DBG_FINE>), raw uint64
    if (const char* tlf = getenv("RGCN_GEMM_TL_FILE")) {
      const size_t tb_bytes = gemm_w8_timeline_bytes(M, N, 1);
      if (hipMalloc(&c->debug_buf, tb_bytes) != hipSuccess) { s = RGCN_ERR_NOMEM; break; }
      (void)hipMemsetAsync(c->debug_buf, 0, tb_bytes, c->stream);
      s = gemm_f32(c, "debug_gemm", true, tb != 0, M, N, K, A, K, B, tb ? K : N, C, N, 1, &gb);
      std::vector<char> hb(tb_bytes);
      if (s == RGCN_OK) s = to_host(c, hb.data(), c->debug_buf, tb_bytes);
      (void)hipFree(c->debug_buf);
      c->debug_buf = nullptr;
      if (s != RGCN_OK) break;
      if (FILE* f = fopen(tlf, "wb")) { fwrite(hb.data(), 1, tb_bytes, f); fclose(f); }
    }
    if (iters > 0 && avg_ms) {
      if ((s = rgcn_timer_start(c)) != RGCN_OK) break;
      for (int it = 0; it < iters && s == RGCN_OK; ++it)
        s = gemm_f32(c, "debug_gemm", true, tb != 0, M, N, K, A, K, B, tb ? K : N, C, N, 1, &gb);
      if (s != RGCN_OK) break;
      float ms = 0.f;
      if ((s = rgcn_timer_stop(c, &ms)) != RGCN_OK) break;
      *avg_ms = ms / iters;
    }
  } while (0);
  if (A) (void)hipFree(A);
  if (B) (void)hipFree(B);
  if (C) (void)hipFree(C);
  if (F) (void)hipFree(F);
  return s;
}

rgcn_status rgcn_debug_gemm_time(rgcn_ctx* c, int32_t ta, int32_t tb, int32_t M, int32_t N, int32_t K,
                                 int32_t split_k, int32_t iters, const float* a_host,
                                 const float* b_host, float* avg_ms) {
  RGCN_NEED(c);
  if (M <= 0 || N <= 0 || K <= 0 || iters <= 0 || !a_host || !b_host || !avg_ms)
    RGCN_FAIL(c, RGCN_ERR_INVALID, "bad arguments");
  if (ta && tb) RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "TT form not instantiated");
  float *A = nullptr, *B = nullptr, *C = nullptr;
  rgcn_status s = RGCN_OK;
  do {
    if ((s = dmalloc(c, &A, (size_t)M * K, false)) != RGCN_OK) break;
    if ((s = dmalloc(c, &B, (size_t)K * N, false)) != RGCN_OK) break;
    if ((s = dmalloc(c, &C, (size_t)M * N)) != RGCN_OK) break;
    if ((s = to_dev(c, A, a_host, sizeof(float) * (size_t)M * K)) != RGCN_OK) break;
    if ((s = to_dev(c, B, b_host, sizeof(float) * (size_t)K * N)) != RGCN_OK) break;
    int sk = split_k > 0 ? split_k : auto_split_k(M, N, K);
    if ((size_t)sk * M * N > c->slab_floats) sk = 1;
    for (int it = 0; it < 3 && s == RGCN_OK; ++it)
      s = gemm_f32(c, "debug_gemm", ta == 0, tb != 0, M, N, K, A, ta ? M : K, B, tb ? K : N, C, N, sk);
    if (s != RGCN_OK) break;
    if ((s = rgcn_timer_start(c)) != RGCN_OK) break;
    for (int it = 0; it < iters && s == RGCN_OK; ++it)
      s = gemm_f32(c, "debug_gemm", ta == 0, tb != 0, M, N, K, A, ta ? M : K, B, tb ? K : N, C, N, sk);
    if (s != RGCN_OK) break;
    float ms = 0.f;
    if ((s = rgcn_timer_stop(c, &ms)) != RGCN_OK) break;
    *avg_ms = ms / iters;
  } while (0);
  if (A) (void)hipFree(A);
  if (B) (void)hipFree(B);
  if (C) (void)hipFree(C);
  return s;
}

#endif  // RGCN_DEVTOOLS

}  // extern "C"
