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
for w in llama3-8b-q3_k_m llama3-8b-q2_k llama3-8b-iq4_xs; do      # not BASELINE configs: the other weight formats on the llama3-8b shapes
  echo "== $w"; timeout -k 10 600 python3 bench.py --workload $w --no-cpu-baseline > $OUT/$w.json 2> $OUT/$w.err; echo rc=$?
done
echo "== llama3-70b-q4_k_m (one GPU, hot path only)"; timeout -k 10 900 python3 bench.py --workload llama3-70b-q4_k_m --no-cpu-baseline --no-e2e --steps 2 > $OUT/llama3-70b-q4_k_m.json 2> $OUT/llama3-70b-q4_k_m.err; echo rc=$?
echo done
echo "== rocprofv3 kernel trace of Mixtral token generation end to end"
export GGML_BACKEND_PATH=$ROOT/ggml-hexagon_amd/libggml-mi355x.so LD_LIBRARY_PATH=$ROOT/oracle/_ref:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
(cd $ROOT/oracle/_ref && ./llama-e2e write --config mixtral-8x7b-q4_k_m --gguf /tmp/mx.gguf > /dev/null 2>&1)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mixtral_tg -o tg -- $ROOT/oracle/_ref/llama-e2e bench --gguf /tmp/mx.gguf --ngl 99 -p 0 -n 64 -r 1 -t 16 > $OUT/prof_mixtral_tg.log 2>&1; echo rc=$?
find $OUT -name "*kernel_trace*" -delete
echo done
