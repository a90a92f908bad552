set -e
mkdir -p gpurun_out/r29
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r29/gputests.log 2>&1
timeout -k 10 300 python bench.py --no-e2e --no-cpu-baseline --steps 3 > gpurun_out/r29/hot.json 2>/dev/null
