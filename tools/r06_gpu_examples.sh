#!/bin/bash
# every example once (smoke): exit codes and last lines
OUT=gpurun_out/r06_examples; mkdir -p $OUT
for ex in example1_dubins_time_optimal.py "example2_swarm_3d.py 5" "example3_sequential_swarm.py" "example4_complex_obstacles.py" example5_fd_step_one_launch.py example6_min_dist_curves.py "example7_dubins_degree8.py example2 10" "example8_driving_on_a_track.py --raw"; do
  name=$(echo $ex | tr ' ./' '___')
  timeout -k 5 200 python examples/$ex > $OUT/$name.log 2>&1; rc=$?
  echo "== $ex rc=$rc"; grep -v "amdgpu.ids" $OUT/$name.log | tail -3
done
