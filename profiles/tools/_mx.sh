set -o pipefail
ROOT=$(pwd)
mkdir -p gpurun_out/mx
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_llama_e2e.py tests/test_gpu_chain.py -x -q -m gpu -k "mixed or group or e2e or llama or chain" > gpurun_out/mx/tests.log 2>&1; echo tests rc=$?; tail -n 5 gpurun_out/mx/tests.log
export GGML_BACKEND_PATH=$ROOT/ggml-hexagon_amd/libggml-mi355x.so LD_LIBRARY_PATH=$ROOT/oracle/_ref:$LD_LIBRARY_PATH
(cd $ROOT/oracle/_ref && ./llama-e2e write --config mixtral-8x7b-q4_k_m --gguf /tmp/mx.gguf > /dev/null 2>&1)
for v in 1 2 1 2; do
  GGML_MI355X_MV_KMIX=$v GGML_MI355X_TIMING=1 timeout -k 10 300 $ROOT/oracle/_ref/llama-e2e bench --gguf /tmp/mx.gguf --ngl 99 -p 0 -n 128 -r 2 -t 16 > gpurun_out/mx/k.$v.log 2> gpurun_out/mx/k.$v.err; echo kmix=$v rc=$?
  grep -h -o '"tg_tok_s": [0-9.]*' gpurun_out/mx/k.$v.log | tail -n 1
  grep -h -o "tg graphs [0-9]* stream_ms [0-9.]*" gpurun_out/mx/k.$v.err | tail -n 1
done
