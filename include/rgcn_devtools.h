/*
 * rgcn_devtools.h -- entry points that exist ONLY in librgcn_devtools.so, the -DRGCN_DEVTOOLS build of the same
 * sources that tools/ and the dense-contraction tests load.  The product library (librgcn.so, include/rgcn.h) does
 * not export them and carries no experiment hooks.
 */
#ifndef RGCN_DEVTOOLS_H_
#define RGCN_DEVTOOLS_H_

#include "rgcn.h"

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: these declarations are its whole dynamic symbol table */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ---- the dense contractions on their own (GPU parity tests of the GEMM kernels, tools/gemm_*.py) ---- */
/* C[M,N] = op(A) . op(B) through the library's fp32-MFMA GEMM.  trans_a == 0: A is [M,K] row-major,
 * else A is [K,M] row-major (used transposed); trans_b == 0: B is [K,N], else [N,K].
 * split_k == 0 picks the split the encoder would use; > 1 forces that many K slabs. */
rgcn_status rgcn_debug_gemm(rgcn_ctx* ctx, int32_t trans_a, int32_t trans_b, int32_t M, int32_t N,
                            int32_t K, int32_t split_k, const float* a_host, const float* b_host,
                            float* c_host);

/* Same contraction on device copies of the operands, `iters` back-to-back launches timed with HIP
 * events on the context's stream; *avg_ms = mean time of one product (incl. the split-K reduce). */
rgcn_status rgcn_debug_gemm_time(rgcn_ctx* ctx, int32_t trans_a, int32_t trans_b, int32_t M, int32_t N,
                                 int32_t K, int32_t split_k, int32_t iters, const float* a_host,
                                 const float* b_host, float* avg_ms);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* RGCN_DEVTOOLS_H_ */
