#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <tag> "<COUNTER COUNTER ...>" <python script + args>
# one rocprofv3 --pmc pass (counters only: no tracing beside it), csv under gpurun_out/<tag>/
tag=$1; shift; ctrs=$1; shift
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rm -rf gpurun_out/$tag
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/$tag -o run -- python3 "$@" > gpurun_out/$tag.log 2>&1
python3 tools/pmc_table.py gpurun_out/$tag
