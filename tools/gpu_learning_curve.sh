#!/bin/bash
# The launcher configuration for a few dozen training steps: does the whole system LEARN (eval reward over training)?   bash tools/gpu_learning_curve.sh [steps]
n=${1:-40}
timeout -k 10 800 python3 examples/rodent_run_ppo.py --max-training-steps $n --envs-per-gpu 2048 --xml ./models/rodent_optimized.xml > gpurun_out/example_ppo_$n.log 2>&1; echo rc $?
grep "eval/episode_reward" gpurun_out/example_ppo_$n.log | python3 -c "
import sys, json
print('env_steps eval_episode_reward eval_pos_reward avg_episode_length training_sps total_loss')
for l in sys.stdin:
    d = json.loads(l)
    print(int(d['num_steps']), round(d['eval/episode_reward'], 3), round(d['eval/episode_pos_reward'], 3), round(d['eval/avg_episode_length'], 2), round(d.get('training/sps', 0)), round(d.get('training/total_loss', 0), 4))"
