set -o pipefail
ROOT=$(pwd)
mkdir -p gpurun_out/rope
timeout -k 10 900 python -m pytest tests/test_gpu_layer_ops.py tests/test_gpu_llama_e2e.py tests/test_gpu_backend_ops.py -x -q -m gpu > gpurun_out/rope/tests.log 2>&1; echo tests rc=$?; tail -n 4 gpurun_out/rope/tests.log
export GGML_BACKEND_PATH=$ROOT/ggml-hexagon_amd/libggml-mi355x.so LD_LIBRARY_PATH=$ROOT/oracle/_ref:$LD_LIBRARY_PATH
(cd $ROOT/oracle/_ref && ./llama-e2e write --config llama3-8b-q4_k_m --gguf /tmp/l3.gguf > /dev/null 2>&1)
for i in 1 2; do
GGML_MI355X_TIMING=1 timeout -k 10 300 $ROOT/oracle/_ref/llama-e2e bench --gguf /tmp/l3.gguf --ngl 99 -p 512 -n 0 -r 4 -t 16 > gpurun_out/rope/pp.$i.log 2> gpurun_out/rope/pp.$i.err; echo rc=$?
grep -h -o '"pp_tok_s": [0-9.]*' gpurun_out/rope/pp.$i.log | tail -n 1
grep -h -o "pp graphs [0-9]* tokens [0-9]* stream_ms [0-9.]* min_ms [0-9.]*" gpurun_out/rope/pp.$i.err | tail -n 1
done
