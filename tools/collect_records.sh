#!/bin/bash
# Copy one record set (written under gpurun_out/ by `tools/job.sh TAG bench profile pmc records`) into profiles/ under its tag.
#   tools/collect_records.sh r05m
set -e
TAG=$1; G=gpurun_out; P=profiles
for i in 1 2 3; do [ -f $G/$TAG/bench_$i.json ] && tail -n 1 $G/$TAG/bench_$i.json > $P/${TAG}_bench_run$i.json; done
D=$G/prof_$TAG
[ -f $D/${TAG}_bench.json ] && cp $D/${TAG}_bench.json $P/${TAG}_bench.json
for k in serial overlap; do
  [ -f $D/${TAG}_bench_under_rocprof_$k.json ] && cp $D/${TAG}_bench_under_rocprof_$k.json $P/
  [ -f $D/$k/k_kernel_stats.csv ] && cp $D/$k/k_kernel_stats.csv $P/${TAG}_kernel_stats_$k.csv
done
ls $D/*.json $D/*.txt 2>/dev/null | while read f; do b=$(basename $f); [ -f $P/$b ] || cp $f $P/$b; done
R=$G/records_$TAG
if [ -d $R ]; then cp $R/${TAG}_*.json $R/${TAG}_*.txt $R/${TAG}_*.csv $P/ 2>/dev/null || true; fi
[ -f $P/${TAG}_pmc_traffic.json ] && cp $P/${TAG}_pmc_traffic.json $P/pmc_latest.json
python tools/profiles_index.py > /dev/null
ls $P | grep "^$TAG"
