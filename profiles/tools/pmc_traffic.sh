#!/bin/bash
# HBM bytes per launch of the token-generation kernels from the PMC counters, as MI355X_MICROARCH.md (HBM, rocprofv3 PMC slots)
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes, no tracing flags next to --pmc; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB
# (the gfx950 x2 correction for 16-byte-per-lane streaming reads).  The profiled program is tg_population.py: token-generation
# passes only, whose algorithmic bytes per kernel it writes beside the counters.  Writes profiles/pmc_traffic.json + a per-kernel CSV.
#   bash profiles/tools/pmc_traffic.sh [round-tag]          (on the MI355X box, from the repo root)
set -e
TAG=${1:-r03}
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o t --output-format csv -- python3 $ROOT/profiles/tools/tg_population.py /tmp/pop_f.json 8 > /tmp/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o t --output-format csv -- python3 $ROOT/profiles/tools/tg_population.py /tmp/pop_w.json 8 > /tmp/pmc_w.log 2>&1
cd $ROOT
python3 profiles/tools/pmc_traffic.py $(find /tmp/pmc_f -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) /tmp/pop_f.json $TAG
