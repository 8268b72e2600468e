set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -s -k "split_bf16_variant" > gpurun_out/t_train.log 2>&1 || { tail -40 gpurun_out/t_train.log; exit 1; }
tail -3 gpurun_out/t_train.log
timeout -k 10 200 python bench.py --workload train32 --steps 20 --warmup 3 --train-precision bf16x3 > gpurun_out/train_x3.json 2> gpurun_out/train_x3.err
cat gpurun_out/train_x3.json | cut -c1-900
