bash tools/gpu_spmm_ab.sh r03spmm3 8 32 64
RGCN_FUSE=2 bash tools/gpu_profile.sh r03spmm3 fb237_block serial
