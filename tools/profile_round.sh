#!/bin/bash
# Profile passes of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r03      -> gpurun_out/<tag>_kernel_stats.txt, <tag>_bench_unprofiled.json, <tag>_bench_profiled.json,
#                                           <tag>_pmc_hbm.json, <tag>_pmc_mfma.json, <tag>_pmc_lds.json
# The kernel trace is taken with the DRIVER'S OWN command (bench.py --steps 20 --warmup 5; CPU baseline / other configs off so that
# only headline steps are in the trace), and the same command is run un-profiled right before it: the two JSON lines give the
# per-kernel HIP-event averages with and without the profiler attached (profiled passes clock lower; never mix the arms).
# Every PMC group is a separate run (counter multiplexing; gpurun refuses --pmc together with sys traces).
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
CMD="python $R/bench.py $ARGS"
# round 4: the PMC passes run the DRIVER'S command too (25 clips), so traffic, clocks and the trace come from
# the same run shape
PSTEPS=25
PCMD="$CMD"
rm -rf /tmp/prof_$TAG /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG /tmp/pmc_m_$TAG /tmp/pmc_v_$TAG /tmp/pmc_l_$TAG
$CMD > $OUT/${TAG}_bench_unprofiled.json 2> $OUT/${TAG}_bench_unprofiled.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- $CMD > $OUT/${TAG}_bench_profiled.json 2> $OUT/${TAG}_prof.log
DB=$(find /tmp/prof_$TAG -name "*_results.db" | head -1)
python $R/tools/prof_summary.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py $ARGS   (25 clips: 5 warm-up + 20 timed)" > $OUT/${TAG}_kernel_stats.txt
python $R/tools/prof_vs_events.py $OUT/${TAG}_kernel_stats.txt $OUT/${TAG}_bench_profiled.json $OUT/${TAG}_bench_unprofiled.json >> $OUT/${TAG}_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$TAG -- $PCMD > $OUT/${TAG}_pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$TAG -- $PCMD > $OUT/${TAG}_pmc_w.log 2>&1
python $R/tools/pmc_summary.py /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG $PSTEPS "python bench.py $ARGS" > $OUT/${TAG}_pmc_hbm.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_m_$TAG -- $PCMD > $OUT/${TAG}_pmc_m.log 2>&1
python $R/tools/pmc_kernels.py /tmp/pmc_m_$TAG conv_wino43 conv_wino_kernel conv_igemm stem_pool_kernel render_average cost_volume maxpool homo_warp > $OUT/${TAG}_pmc_mfma.json
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_l_$TAG -- $PCMD > $OUT/${TAG}_pmc_l.log 2>&1
python $R/tools/pmc_kernels.py /tmp/pmc_l_$TAG conv_wino43 conv_wino_kernel conv_igemm stem_pool_kernel render_average cost_volume ccl_softmax > $OUT/${TAG}_pmc_lds.json
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d /tmp/pmc_v_$TAG -- $PCMD > $OUT/${TAG}_pmc_v.log 2>&1
python $R/tools/pmc_render_summary.py /tmp/pmc_v_$TAG > $OUT/${TAG}_pmc_render.json
ls -la $OUT/${TAG}_*
