#!/bin/bash
# The round's profile set in one call (on the GPU box, from the repo root):  bash profiles/tools/final_pass.sh [round-tag]
# Writes under gpurun_out/final/; copy what is to be judged into profiles/ (README.md there says which file is which).
# The development harnesses (r64s_dev, attn_dev, nb4) are built beforehand into profiles/tools/_bin/ (hipcc line at the top of each .hip).
TAG=${1:-r03}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== bench (plain, with cpu_baseline and e2e)"
timeout -k 10 600 python3 $ROOT/bench.py --steps 3 > $OUT/bench.json 2> $OUT/bench.err; echo rc=$?
echo "== rocprofv3 kernel trace of the hot-path bench"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o b -- python3 $ROOT/bench.py --no-e2e --no-cpu-baseline --steps 2 > $OUT/prof_bench.log 2>&1; echo rc=$?
python3 $ROOT/profiles/kernel_shapes.py $OUT/prof_bench/b_kernel_trace.csv > $OUT/kernel_shapes.csv 2>/dev/null
echo "== rocprofv3 kernel traces of end-to-end token generation and prompt processing (libllama + plugin), llama-bench's two tests"
export GGML_BACKEND_PATH=$ROOT/ggml-hexagon_amd/libggml-mi355x.so LD_LIBRARY_PATH=$ROOT/oracle/_ref:$LD_LIBRARY_PATH
(cd $ROOT/oracle/_ref && ./llama-e2e write --config llama3-8b-q4_k_m --gguf /tmp/l3.gguf > /dev/null 2>&1)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_e2e_tg -o tg -- $ROOT/oracle/_ref/llama-e2e bench --gguf /tmp/l3.gguf --ngl 99 -p 0 -n 128 -r 2 > $OUT/prof_e2e_tg.log 2>&1; echo rc=$?
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_e2e_pp -o pp -- $ROOT/oracle/_ref/llama-e2e bench --gguf /tmp/l3.gguf --ngl 99 -p 512 -n 0 -r 4 > $OUT/prof_e2e_pp.log 2>&1; echo rc=$?
echo "== prefill shapes, hand-placed K-step on / off"
for v in 0 1; do echo "GGML_MI355X_R64S=$v"; GGML_MI355X_R64S=$v timeout -k 10 120 python3 $ROOT/profiles/tools/shape_times.py 30; done > $OUT/shape_times.txt 2>&1
for v in 0 1; do echo "GGML_MI355X_R64S=$v"; GGML_MI355X_R64S=$v timeout -k 10 200 python3 $ROOT/profiles/tools/kloop_times.py; done > $OUT/kloop_times.txt 2>&1
echo "== in-kernel stamps: the Q4_K prefill kernel (three shapes), the decode attention (general and short kernel), the nb4 micro-benchmark"
B=$ROOT/profiles/tools/_bin
if [ -x $B/r64s_dev ]; then
  for a in "28672 4096 512 1" "4096 14336 512 8" "4096 4096 512 8" "6144 4096 512 4"; do echo "== r64s_dev $a"; R64S_VARIANT=2 timeout -k 10 120 $B/r64s_dev $a 3; done > $OUT/r64s_stamps.txt 2>&1
  for n in 256 512 64 640 768 1016; do echo "== attn_dev $n"; timeout -k 10 120 $B/attn_dev $n; done > $OUT/attn_decode_stamps.txt 2>&1
  timeout -k 10 200 $B/nb4 > $OUT/nb4.log 2>&1
fi
echo "== SQ counters: the 64-rows-per-wave kernel compiler-scheduled and hand-placed, 28672 x 4096 x 512"
cd $ROOT
GGML_MI355X_R64S=0 bash profiles/tools/pmc_prefill.sh 12 28672 4096 512 > $OUT/pmc_r64.txt 2>&1
GGML_MI355X_R64S=1 bash profiles/tools/pmc_prefill.sh 12 28672 4096 512 > $OUT/pmc_r64s.txt 2>&1
echo "== HBM traffic of the token-generation kernels (FETCH_SIZE / WRITE_SIZE, separate passes)"
bash profiles/tools/pmc_traffic.sh $TAG > $OUT/pmc_traffic.log 2>&1; echo rc=$?
cp profiles/pmc_traffic.json profiles/${TAG}_pmc_hbm_traffic.csv $OUT/ 2>/dev/null
echo "== full-size parity report"
timeout -k 10 600 python3 -m pytest tests/test_gpu_full_size.py -q -s -k "config_shape or zz_print" > $OUT/full_size_parity.txt 2>&1; echo rc=$?
echo done
