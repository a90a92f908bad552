set -o pipefail
mkdir -p gpurun_out/q40
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_planar.py tests/test_gpu_chain.py tests/test_gpu_full_size.py -x -q -m gpu -k "q4_0 or planar or chain or mixed or Q4_0" > gpurun_out/q40/tests.log 2>&1; echo tests rc=$?; tail -n 5 gpurun_out/q40/tests.log
timeout -k 10 500 python bench.py --workload llama2-7b-q4_0 --no-cpu-baseline > gpurun_out/q40/b.json 2> gpurun_out/q40/b.err; echo rc=$?
python - <<PY
import json
r=json.loads(open("gpurun_out/q40/b.json").read().strip().splitlines()[-1])
print(r.get("value"), r.get("tg_ms_per_token"), r.get("pp512_tok_s"), r["roofline"]["frac"], r.get("e2e"))
PY
