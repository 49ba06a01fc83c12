#!/bin/bash
# LDS bank-conflict counters per kernel family for a short run of the headline clip:   bash tools/pmc_lds.sh <tag>
TAG=${1:-lds}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_l_$TAG
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_l_$TAG -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/${TAG}_pmc_l.log 2>&1
python $R/tools/pmc_kernels.py /tmp/pmc_l_$TAG conv_wino conv_igemm stem_pool_kernel render_average cost_volume ccl_softmax > $OUT/${TAG}_pmc_lds.json
python - <<PY
import json
d = json.load(open('$OUT/${TAG}_pmc_lds.json'))
for k, v in d.items():
    print('%-18s launches %4d  %9.1f us  LDS bank-conflict cycles / LDS active %.3f   LDS active / kernel cycles %.3f' % (
        k, v['launches'], v['sum_us'], v.get('lds_bank_conflict_frac', 0), v.get('lds_busy_frac', 0)))
PY
