#!/bin/bash
# Round profile on the GPU box: the default bench line, its rocprofv3 kernel stats, the two HBM counter passes, and kernel stats
# of the flagged split-bf16 mode, the CTC lattice workload and the training step.
#   gpurun -- 'bash tools/profile_round.sh <tag>'   ->  gpurun_out/<tag>_*  (copy what should be judged into profiles/)
set -e -o pipefail
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LEAN="--no-variants --no-cpu-baseline"
python3 $R/bench.py 2> $OUT/${TAG}_bench.err | tee $OUT/${TAG}_bench.json
stats() {   # stats <name> <bench args...>
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${name}_stats -o ${TAG} -- python3 $R/bench.py "$@" > $OUT/${TAG}_${name}_profiled.json 2> $OUT/${TAG}_${name}_stats.err
  find $OUT/${TAG}_${name}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_${name}_kernel_stats.csv
}
stats joint $LEAN
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o ${TAG} -- python3 $R/bench.py --steps 16 --warmup 8 $LEAN --no-roofline > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o ${TAG} -- python3 $R/bench.py --steps 16 --warmup 8 $LEAN --no-roofline > /dev/null 2> $OUT/${TAG}_pmc_write.err
python3 $R/tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write "bench.py --steps 16 --warmup 8 (B=512 per pass)" > $OUT/${TAG}_pmc_traffic.json
stats bf16x3 --precision bf16x3 --steps 64 --warmup 16 $LEAN
stats f32mfma --precision f32 --steps 32 --warmup 8 $LEAN
stats ctc --workload ctc256 --steps 50 --warmup 5
stats train --workload train32 --steps 5 --warmup 2
stats trainx3 --workload train32 --train-precision bf16x3 --steps 5 --warmup 2
rm -rf $OUT/${TAG}_*_stats $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
ls $OUT | grep ${TAG}
