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

/* C[M,N] = A . op(B) with B handed to the kernel PRE-SPLIT into bf16 planes in MFMA fragment order -- the way the encoder
 * passes its weights (W_self, the basis tensors) since round 5, csrc/gemm_bf16x3.hip B_PRE.  A is [M,K]; trans_b == 0: B
 * is [K,N], else [N,K].  Bitwise the result of rgcn_debug_gemm on the same operands (split arithmetic, modes 6 / 9;
 * mode 0 ignores the table).  iters > 0: also the mean time of one product over that many launches. */
rgcn_status rgcn_debug_gemm_presplit(rgcn_ctx* ctx, int32_t trans_b, int32_t M, int32_t N, int32_t K, int32_t iters,
                                     const float* a_host, const float* b_host, float* c_host, float* avg_ms);

/* ---- placement self-check ---- */
/* out_host[b] = the XCD (HW_REG_XCC_ID) workgroup b of a plain 1-D launch of n_blocks workgroups ran on.  The
 * destination-major block layer (csrc/block_rows.hip) and the decoder's line kernel give column band x to the workgroups
 * with blockIdx % 8 == x and count on them sharing one L2: a performance assumption HIP does not promise
 * (MI355X_MICROARCH.md: "observed, for speed only: block b runs on XCD b % 8").  tests/test_gpu_parity.py::
 * test_workgroups_of_a_band_share_an_xcd re-checks it on the box the suite runs on. */
rgcn_status rgcn_debug_xcd_map(rgcn_ctx* ctx, int32_t n_blocks, int32_t* out_host);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* RGCN_DEVTOOLS_H_ */
