#!/bin/bash
# One bench line per BASELINE config on one GPU (from the repo root, on the MI355X box):  bash profiles/tools/configs_pass.sh
# Writes gpurun_out/configs/<workload>.json; copy into profiles/rNN_configs/.
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/configs
mkdir -p $OUT
cd $ROOT
for w in llama2-7b-q4_0 synth-7b-q4_k mixtral-8x7b-q4_k_m; do
  echo "== $w"; timeout -k 10 900 python3 bench.py --workload $w --no-cpu-baseline > $OUT/$w.json 2> $OUT/$w.err; echo rc=$?
done
echo "== llama3-70b-q4_k_m (one GPU, hot path only)"; timeout -k 10 900 python3 bench.py --workload llama3-70b-q4_k_m --no-cpu-baseline --no-e2e --steps 2 > $OUT/llama3-70b-q4_k_m.json 2> $OUT/llama3-70b-q4_k_m.err; echo rc=$?
echo done
