#!/bin/bash
# forward kernel with / without 16-byte aligned first-layer weight rows (RR_MLP_PAD).   bash tools/gpu_fwd_pad_ab.sh
for rep in 1 2; do for p in 1 0; do RR_MLP_PAD=$p timeout -k 10 200 python3 tools/bench_learner_kernels.py 2>/dev/null | tail -n 2 | head -n 1 | cut -c1-120 | sed "s/^/pad=$p /"; done; done
