#!/bin/bash
# the whole GPU suite, then the quick numbers
OUT=gpurun_out/${1:-full}; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -q -x -m gpu > $OUT/pytest.log 2>&1 || { tail -25 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
bash tools/r02_quick.sh $1
