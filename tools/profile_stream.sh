#!/bin/bash
# Kernel trace + per-push timeline of the steady-state stream:   bash tools/profile_stream.sh <tag> [bench_stream.py args]
TAG=${1:-stream}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $R/tools/bench_stream.py --pushes 60 "$@" > $OUT/${TAG}_run.txt 2> $OUT/${TAG}_prof.log
DB=$(find /tmp/prof_$TAG -name "*_results.db" | head -1)
python $R/tools/prof_summary.py $DB "rocprofv3 --kernel-trace --stats -- python tools/bench_stream.py --pushes 60 $*" > $OUT/${TAG}_kernel_stats.txt
python $R/tools/push_timeline.py $DB stem_pool 3 > $OUT/${TAG}_timeline.txt
tail -2 $OUT/${TAG}_run.txt
tail -40 $OUT/${TAG}_timeline.txt
