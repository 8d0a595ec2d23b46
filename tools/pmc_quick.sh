#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the dominant kernels on tools/pmc_traffic.py's shapes: two separate PMC passes (counters are
# collected in their own runs, with --kernel-trace only), summarised into gpurun_out/<tag>/traffic.json by tools/pmc_summary.py.
# usage (GPU box, repo root): bash tools/pmc_quick.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-pmc}
mkdir -p $R/gpurun_out/$T
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$T/fetch -- python3 $R/tools/pmc_traffic.py > $R/gpurun_out/$T/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$T/write -- python3 $R/tools/pmc_traffic.py > $R/gpurun_out/$T/write.log 2>&1
cd $R
cp $(ls gpurun_out/$T/fetch/*/*counter_collection.csv | head -1) gpurun_out/$T/fetch_size_counter_collection.csv
cp $(ls gpurun_out/$T/write/*/*counter_collection.csv | head -1) gpurun_out/$T/write_size_counter_collection.csv
rm -rf gpurun_out/$T/fetch gpurun_out/$T/write
python3 tools/pmc_summary.py gpurun_out/$T
