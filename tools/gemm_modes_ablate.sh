#!/bin/bash
# Where the split-GEMM time goes (NT form, 6 terms) at one and two workgroups per CU.
# RGCN_GEMM_ABLATE bits: 1 no in-loop global loads, 2 no split/store, 4 no epilogue, 8 no fragment reads,
# 16 no barrier, 32 no MFMA, 128 split with shift/mask/subtract instead of v_dot2c
for ab in ${ABLATIONS:-0 1 2 3 4 8 11 16 27 32 59 63 128 160}; do
  RGCN_PROBE_ONLY=1 RGCN_GEMM_ABLATE=$ab python tools/gemm_modes.py 50 2>&1 | grep -E "^ablate"
done
