set -e
mkdir -p gpurun_out
( time timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_style.json 2> gpurun_out/bench_driver_style.err ) 2> gpurun_out/bench_time.txt
cat gpurun_out/bench_time.txt | tail -4
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_driver_style.json'))
print({k:d[k] for k in ("value","fuse1","ragged","f32_mode","greedy32_h256","train32_f32","train32_bf16x3")})
print(d["variants"]["train32_bf16x3"])
PY
