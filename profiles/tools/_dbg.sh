set -o pipefail
ROOT=$(pwd)
mkdir -p gpurun_out/mx
timeout -k 10 900 python -m pytest tests/test_gpu_llama_e2e.py tests/test_gpu_split_buffer.py -x -q -m gpu > gpurun_out/mx/tests.log 2>&1; echo tests rc=$?; tail -n 4 gpurun_out/mx/tests.log
export GGML_BACKEND_PATH=$ROOT/ggml-hexagon_amd/libggml-mi355x.so LD_LIBRARY_PATH=$ROOT/oracle/_ref:$LD_LIBRARY_PATH
(cd $ROOT/oracle/_ref && ./llama-e2e write --config mixtral-8x7b-q4_k_m --gguf /tmp/mx.gguf > /dev/null 2>&1)
for v in off on off on; do
  if [ $v = off ]; then export GGML_MI355X_ROUTER_NORM_OFF=1; else unset GGML_MI355X_ROUTER_NORM_OFF; fi
  GGML_MI355X_TIMING=1 timeout -k 10 300 $ROOT/oracle/_ref/llama-e2e bench --gguf /tmp/mx.gguf --ngl 99 -p 0 -n 128 -r 2 -t 16 > gpurun_out/mx/b.$v.log 2> gpurun_out/mx/b.$v.err; echo $v rc=$?
  grep -h -o '"tg_tok_s": [0-9.]*' gpurun_out/mx/b.$v.log | tail -n 1
  grep -h -o "tg graphs [0-9]* stream_ms [0-9.]*" gpurun_out/mx/b.$v.err | tail -n 1
done
