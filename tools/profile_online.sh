#!/bin/bash
# Kernel trace of the streaming mode (one frame pair per push, HIP-graph steady state):   bash tools/profile_online.sh <tag>
TAG=${1:-online}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $R/bench.py --online --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_prof.log
DB=$(find /tmp/prof_$TAG -name "*_results.db" | head -1)
python $R/tools/prof_summary.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --online --steps 3 --warmup 1" > $OUT/${TAG}_kernel_stats.txt
head -60 $OUT/${TAG}_kernel_stats.txt
tail -1 $OUT/${TAG}_bench.json | head -c 600
