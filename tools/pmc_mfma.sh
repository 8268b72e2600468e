#!/bin/bash
# MFMA-pipe busy cycles and GPU-active cycles per kernel (one rocprofv3 counter pass).  gpurun -- 'bash tools/pmc_mfma.sh <tag>'
set -e -o pipefail
TAG=${1:-mfma}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc -o ${TAG} -- python $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-roofline --no-variants > /dev/null 2> $OUT/${TAG}_pmc.err
python $R/tools/pmc_summary.py $OUT/${TAG}_pmc | tee $OUT/${TAG}_summary.txt
