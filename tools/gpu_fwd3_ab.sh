#!/bin/bash
# forward: three row tiles per workgroup (rr_mlp_forward3_kernel) against one (RR_MLP_FWD3=0).   bash tools/gpu_fwd3_ab.sh
for rep in 1 2; do for p in 1 0; do RR_MLP_FWD3=$p timeout -k 10 200 python3 tools/bench_learner_kernels.py 2>/dev/null | tail -n 2 | head -n 1 | cut -c1-110 | sed "s/^/fwd3=$p /"; done; done
