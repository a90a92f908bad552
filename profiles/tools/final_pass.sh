#!/bin/bash
# The round's profile set in one call (on the GPU box, from the repo root):  bash profiles/tools/final_pass.sh
# Writes under gpurun_out/final/; copy what is to be judged into profiles/ (README.md there says which file is which).
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== bench (plain, with cpu_baseline and e2e)"
timeout -k 10 600 python3 $ROOT/bench.py --steps 3 > $OUT/bench.json 2> $OUT/bench.err; echo rc=$?
echo "== rocprofv3 kernel trace of the hot-path bench"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o b -- python3 $ROOT/bench.py --no-e2e --no-cpu-baseline --steps 2 > $OUT/prof_bench.log 2>&1; echo rc=$?
python3 $ROOT/profiles/kernel_shapes.py $OUT/prof_bench/b_kernel_trace.csv > $OUT/kernel_shapes.csv 2>/dev/null
echo "== rocprofv3 kernel trace of end-to-end token generation (libllama + plugin)"
export GGML_BACKEND_PATH=$ROOT/ggml-hexagon_amd/libggml-mi355x.so LD_LIBRARY_PATH=$ROOT/oracle/_ref:$LD_LIBRARY_PATH
(cd $ROOT/oracle/_ref && ./llama-e2e write --config llama3-8b-q4_k_m --gguf /tmp/l3.gguf > /dev/null 2>&1)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_e2e -o tg -- $ROOT/oracle/_ref/llama-e2e bench --gguf /tmp/l3.gguf --ngl 99 -p 0 -n 64 -r 2 > $OUT/prof_e2e.log 2>&1; echo rc=$?
echo "== prefill shapes per kernel choice"
for v in 0 1 2; do echo "GGML_MI355X_R64=$v"; GGML_MI355X_R64=$v timeout -k 10 120 python3 $ROOT/profiles/tools/shape_times.py 30; done > $OUT/shape_times.txt 2>&1
for v in 0 2; do echo "GGML_MI355X_R64=$v"; GGML_MI355X_R64=$v timeout -k 10 200 python3 $ROOT/profiles/tools/kloop_times.py; done > $OUT/kloop_times.txt 2>&1
echo "== SQ counters: 256 x 256 kernel and the 64-rows-per-wave kernel, 28672 x 4096 x 512"
cd $ROOT
GGML_MI355X_R64=0 bash profiles/tools/pmc_prefill.sh 12 28672 4096 512 > $OUT/pmc_wide.txt 2>&1
GGML_MI355X_R64=2 bash profiles/tools/pmc_prefill.sh 12 28672 4096 512 > $OUT/pmc_r64.txt 2>&1
echo done
