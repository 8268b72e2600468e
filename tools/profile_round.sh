#!/bin/bash
# Round profile on the GPU box: the default bench line, its rocprofv3 kernel stats, and the two HBM counter passes.
#   gpurun -- 'bash tools/profile_round.sh <tag>'   ->  gpurun_out/<tag>_*  (copy what should be judged into profiles/)
set -e -o pipefail
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2> $OUT/${TAG}_bench.err | tee $OUT/${TAG}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o ${TAG} -- python $R/bench.py > $OUT/${TAG}_bench_profiled.json 2> $OUT/${TAG}_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o ${TAG} -- python $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-roofline > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o ${TAG} -- python $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-roofline > /dev/null 2> $OUT/${TAG}_pmc_write.err
python $R/tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write "bench.py --steps 16 --warmup 8 (B=256 per pass)" > $OUT/${TAG}_pmc_traffic.json
find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
ls $OUT | grep ${TAG}
