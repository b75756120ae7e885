#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats of the headline bench command, and the PMC passes (separate runs,
# counters only: no trace domains besides --kernel-trace) for the dominant kernel.  Output under gpurun_out/prof_$1/.
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --headline-only > "$OUT/bench_under_rocprof.log" 2>&1
python3 bench.py 2> /dev/null | tail -1 > "$OUT/bench.json"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --headline-only --steps 300 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --headline-only --steps 300 > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py --headline-only --steps 300 > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_lds" -- python3 bench.py --headline-only --steps 300 > "$OUT/pmc_lds.log" 2>&1
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts read 8 B and 16 B per lane
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch" -- ./tools/pmc_calib > "$OUT/calib.log" 2>&1
ls -R "$OUT" | head -60
