set -o pipefail
mkdir -p gpurun_out/q23
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu -k "q2_K or q3_K or q2_k or q3_k" > gpurun_out/q23/tests.log 2>&1; echo tests rc=$? && tail -n 3 gpurun_out/q23/tests.log &&
for w in llama3-8b-q3_k_m llama3-8b-q2_k; do
  for v in 1 0; do
    GGML_MI355X_REGB_Q23=$v timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --no-e2e > gpurun_out/q23/$w.$v.json 2> gpurun_out/q23/$w.$v.err; echo $w $v rc=$?
    python - <<PY
import json
r=json.loads(open("gpurun_out/q23/$w.$v.json").read().strip().splitlines()[-1])
print("$w", $v, r.get("value"), r.get("pp512_tok_s"), {k:r[k] for k in r if k.startswith("pp") or k.startswith("tg")})
PY
  done
done
