#!/bin/bash
source tools/gpu_call.sh
mkdir -p gpurun_out/train_r03
# same command as profiles/r01_train_curve_bf16_8192.tsv (round 1), on the round-3 simulator (edge contacts, stiffer stick friction)
step 1000 train_r03.log python scripts/run_tracker.py --mode train --num_envs 8192 --device cuda:0 --visualize false --rand_seed 1 \
  --env_config data/configs/tracker_config/dm_env_default.yaml --agent_config data/configs/tracker_config/dm_agent_bf16.yaml \
  --max_samples 600000000 --out_model_file gpurun_out/train_r03/model.pt --log_file gpurun_out/train_r03/log.txt
tail -5 gpurun_out/train_r03/log.txt | cut -c1-300
rm -f gpurun_out/train_r03/model.pt
