#!/bin/bash
# Round-3 closing pass on the GPU box: every config's bench line, the 2-rank rehearsal of config 3 on one GPU (gloo), the profile pass.
tag=${1:-r3k}
o=gpurun_out/final_$tag; mkdir -p $o
export OMP_NUM_THREADS=16
echo "== cfg1"; timeout -k 10 200 python3 bench.py --config 1 --steps 200 --warmup 20 --no-cpu-baseline > $o/bench_cfg1.json 2> $o/bench_cfg1.err; cut -c1-300 $o/bench_cfg1.json
echo "== cfg5"; timeout -k 10 200 python3 bench.py --config 5 --steps 100 --warmup 20 --no-cpu-baseline > $o/bench_cfg5.json 2> $o/bench_cfg5.err; cut -c1-300 $o/bench_cfg5.json
echo "== cfg3"; timeout -k 10 300 python3 bench.py --config 3 --steps 3 --warmup 1 > $o/bench_cfg3.json 2> $o/bench_cfg3.err; tail -c 900 $o/bench_cfg3.json
echo "== cfg3, 2 ranks sharing the GPU (gloo rehearsal; not a scaling number)"; RR_BENCH_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --config 3 --steps 2 --warmup 1 > $o/bench_cfg3_2ranks_shared_gpu.json 2> $o/bench_cfg3_2ranks.err; tail -c 900 $o/bench_cfg3_2ranks_shared_gpu.json; tail -n 3 $o/bench_cfg3_2ranks.err
echo "== cfg2 default protocol"; timeout -k 10 300 python3 bench.py --no-cpu-baseline > $o/bench_cfg2_survey_protocol.json 2> $o/bench_cfg2_survey.err; cut -c1-330 $o/bench_cfg2_survey_protocol.json
bash tools/profile_round3.sh $tag
