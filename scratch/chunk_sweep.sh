#!/bin/bash
R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/chunk.log; : > $L
for c in 64 128 256 512 1024; do
  echo "mesh chunk $c" >> $L
  PTX_WAVE_CHUNK=$c timeout -k 10 100 python $R/scratch/sweep.py 2>&1 | grep Mrays >> $L || exit 1
done
for c in 64 128 256 512; do
  echo "scene chunk $c" >> $L
  PTX_SCENE_WAVE_CHUNK=$c timeout -k 10 100 python $R/scratch/prof_render.py 256 2>&1 | grep Mspp >> $L || exit 1
done
cat $L
