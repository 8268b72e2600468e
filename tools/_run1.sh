set -e
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --hidden 256 --steps 64 --warmup 16 --no-variants --no-cpu-baseline > gpurun_out/bench_kh2.json 2> gpurun_out/bench_kh2.err
MDD_LSTM_WAVES=4 timeout -k 10 300 python bench.py --hidden 256 --steps 64 --warmup 16 --no-variants --no-cpu-baseline > gpurun_out/bench_kh1.json 2> gpurun_out/bench_kh1.err
python - <<'PY'
import json
for n in ("kh2","kh1"):
    d=json.load(open('gpurun_out/bench_%s.json'%n))
    print(n, d["value"], d["ms_per_step"], d["roofline"].get("kernel_classes_ms"))
PY
