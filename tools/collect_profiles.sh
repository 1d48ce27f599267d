#!/bin/bash
# Collect the judged evidence for one round on the GPU box (run through gpurun):
#   tools/collect_profiles.sh r02_a
# Writes gpurun_out/<tag>/{stats,fetch,write}/ + bench lines; tools/parse_profiles.py then copies
# the summaries into profiles/ (run that in the container after gpurun merged gpurun_out/ back).
set -e -o pipefail
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench done" > $OUT/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 bench.py --no-cpu --no-variants > $OUT/bench_stats.json 2> $OUT/stats.err
echo "kernel stats done" >> $OUT/progress.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dedup -o run -- python3 bench.py --no-cpu --no-variants --fd-dedup > $OUT/bench_stats_dedup.json 2> $OUT/stats_dedup.err
echo "dedup stats done" >> $OUT/progress.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-variants > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch pmc done" >> $OUT/progress.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-variants > $OUT/bench_write.json 2> $OUT/write.err
echo "write pmc done" >> $OUT/progress.log
# C5: its own stats + traffic passes (different dominant kernel)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -o run -- python3 bench.py --no-cpu --no-variants --workload C5 --steps 50 --warmup 5 > $OUT/bench_stats_c5.json 2> $OUT/stats_c5.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_c5 -o run -- python3 bench.py --workload C5 --steps 3 --warmup 1 --no-cpu --no-variants > $OUT/bench_fetch_c5.json 2> $OUT/fetch_c5.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_c5 -o run -- python3 bench.py --workload C5 --steps 3 --warmup 1 --no-cpu --no-variants > $OUT/bench_write_c5.json 2> $OUT/write_c5.err
echo "c5 done" >> $OUT/progress.log
# the structured finite-difference step (variants.fd_structured): its own kernel, 600 launches
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_structured -o run -- python3 tools/timeline_structured_run.py > $OUT/structured.log 2> $OUT/structured.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_structured -o run -- python3 tools/timeline_structured_run.py > $OUT/structured_w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_structured -o run -- python3 tools/timeline_structured_run.py > $OUT/structured_f.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_structured_c5 -o run -- python3 tools/timeline_structured_run.py C5 > $OUT/structured_c5.log 2> $OUT/structured_c5.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_structured_c5 -o run -- python3 tools/timeline_structured_run.py C5 > $OUT/structured_c5_w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_structured_c5 -o run -- python3 tools/timeline_structured_run.py C5 > $OUT/structured_c5_f.log 2>&1
echo "structured done" >> $OUT/progress.log
# C4 (256 vehicles, degree 15, B = 7169): the brute-force one-launch step and the structured step, stats + traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c4 -o run -- python3 bench.py --no-cpu --no-variants --workload C4 --steps 10 --warmup 2 > $OUT/bench_stats_c4.json 2> $OUT/stats_c4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_structured_c4 -o run -- python3 tools/timeline_structured_run.py C4 > $OUT/structured_c4.log 2> $OUT/structured_c4.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_structured_c4 -o run -- python3 tools/timeline_structured_run.py C4 > $OUT/structured_c4_w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_structured_c4 -o run -- python3 tools/timeline_structured_run.py C4 > $OUT/structured_c4_f.log 2>&1
echo "c4 done" >> $OUT/progress.log
cat $OUT/progress.log
