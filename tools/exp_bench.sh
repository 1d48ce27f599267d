#!/bin/bash
# scratch tool: run bench.py with an experimental build of the library swapped in
cp optimalbeziertrajectorygeneration_amd/libobtg_hip.so /tmp/libobtg_hip.so.bak
cp "$1" optimalbeziertrajectorygeneration_amd/libobtg_hip.so
python bench.py --steps 20 --warmup 3 --no-cpu | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(k['kernel'],k['avg_ms']) for k in d['kernels']]"
cp /tmp/libobtg_hip.so.bak optimalbeziertrajectorygeneration_amd/libobtg_hip.so
