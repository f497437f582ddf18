#!/bin/bash
# The profile set committed under profiles/ for one round (run on the GPU box through gpurun; TAG = e.g. r01e):
#   kernel stats with and without the side stream, bench JSON lines, and the two PMC passes behind pmc_latest.json.
set -u
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/overlap -o k --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof_overlap.json 2>> $O/bench.err
export VDQN_NO_OVERLAP=1
rocprofv3 --kernel-trace --stats -d $O/serial -o k --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof_serial.json 2>> $O/bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --ramp-seconds 0 --no-cpu-baseline --no-profile > /dev/null 2>> $O/bench.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --ramp-seconds 0 --no-cpu-baseline --no-profile > /dev/null 2>> $O/bench.err
find $O -name "*.csv" | head -20
