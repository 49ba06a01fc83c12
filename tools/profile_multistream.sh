#!/bin/bash
# Kernel trace + per-push timeline of the batch-of-streams steady state:   bash tools/profile_multistream.sh <tag> [S] [extra run_multistream args]
TAG=${1:-multi}
S=${2:-8}
shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $R/tools/run_multistream.py $S 30 "$@" > $OUT/${TAG}_run.txt 2> $OUT/${TAG}_prof.log
DB=$(find /tmp/prof_$TAG -name "*_results.db" | head -1)
python $R/tools/push_timeline.py $DB stem_pool 3 > $OUT/${TAG}_timeline.txt
tail -1 $OUT/${TAG}_run.txt
