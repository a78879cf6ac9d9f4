#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_run.sh <tag>
# rocprofv3 passes behind the summaries under profiles/: kernel trace + stats; then counters ONLY (no tracing beside --pmc), one pass per counter
# group: FETCH_SIZE, WRITE_SIZE (cannot share a pass), the SQ matrix-core group, and the FETCH_SIZE calibration micro-benchmark.
tag=${1:-round2}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
BENCH="bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_stats_serial gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_mfma gpurun_out/${tag}_calib
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o run -- python3 $BENCH > gpurun_out/${tag}_stats.log 2>&1 && echo "stats ok" &&
VG_SIDE_STREAM=0 VG_OVERLAP_GAINS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats_serial -o run -- python3 $BENCH > gpurun_out/${tag}_stats_serial.log 2>&1 && echo "serial stats ok" &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -o run -- python3 $BENCH > gpurun_out/${tag}_fetch.log 2>&1 && echo "fetch ok" &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -o run -- python3 $BENCH > gpurun_out/${tag}_write.log 2>&1 && echo "write ok" &&
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/${tag}_mfma -o run -- python3 $BENCH > gpurun_out/${tag}_mfma.log 2>&1 && echo "mfma ok" &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_calib -o run -- tools/micro/fetch_calib > gpurun_out/${tag}_calib.log 2>&1 && echo "calib ok"
# the raw traces are large: keep only what profile_summary.py reads
find gpurun_out/${tag}_stats gpurun_out/${tag}_stats_serial -name '*kernel_trace.csv' -size +20M -delete 2>/dev/null
ls gpurun_out/${tag}_*/ 2>/dev/null | head -30
