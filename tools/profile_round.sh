#!/bin/bash
# Profile passes of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02      -> gpurun_out/<tag>_kernel_stats.txt, <tag>_pmc_hbm.json, <tag>_pmc_mfma.json
# kernel trace and every PMC group are separate runs (counter multiplexing; gpurun refuses --pmc with sys traces).
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
rm -rf /tmp/prof_$TAG /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG /tmp/pmc_m_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- $CMD > $OUT/${TAG}_prof.log 2>&1
DB=$(find /tmp/prof_$TAG -name "*_results.db" | head -1)
python $R/tools/prof_summary.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs" > $OUT/${TAG}_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$TAG -- $CMD > $OUT/${TAG}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$TAG -- $CMD > $OUT/${TAG}_pmc_w.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG > $OUT/${TAG}_pmc_hbm.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_m_$TAG -- $CMD > $OUT/${TAG}_pmc_m.log 2>&1
python $R/tools/pmc_kernels.py /tmp/pmc_m_$TAG conv_wino conv_igemm render_average cost_volume maxpool > $OUT/${TAG}_pmc_mfma.json
ls -la $OUT/${TAG}_*
