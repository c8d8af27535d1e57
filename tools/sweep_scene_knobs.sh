#!/bin/bash
# Sweep of the two-level kernel's run-time knobs (environment variables read by launch_scene_persistent) over the render
# legs, on the GPU box:  bash tools/sweep_scene_knobs.sh > gpurun_out/sweep_knobs.log
cd $GRAFT_REPO_ROOT
for kv in "X=0" "PTX_SCENE_REFILL_MIN_IDLE=4" "PTX_SCENE_REFILL_MIN_IDLE=16" "PTX_SCENE_REFILL_MIN_IDLE=24" "PTX_SCENE_WALK_STEPS=4" "PTX_SCENE_WALK_STEPS=5" "PTX_SCENE_WALK_STEPS=8" "PTX_SCENE_WAVE_CHUNK=256" "PTX_SCENE_WAVE_CHUNK=64"; do
  echo "== $kv"
  env $kv timeout -k 10 200 python3 tools/time_render.py 2 5 5f 2>&1 | grep config | cut -c1-100 || exit 1
done
