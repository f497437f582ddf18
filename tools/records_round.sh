#!/bin/bash
# The bench / profile records of one round beyond the headline line (run on the GPU box; TAG = e.g. r02):
# other configurations of BASELINE.json (12 views, 4 views, 'basic', f32), the config-4 timing window with two target
# refreshes inside it, deterministic mode, Huber, the self-launched 2-rank run, RCCL with one rank, the convergence record.
set -u
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/records_$TAG
mkdir -p $O
cd $R
export VDQN_BENCH_NO_LIVE_PMC=1  # the PMC child passes belong to the headline run (tools/profile_round.sh), not to every record
B="python3 bench.py --no-cpu-baseline"
$B                                          > $O/${TAG}_bench_c2.json 2> $O/err.log
$B --frames 12 --batch 16                   > $O/${TAG}_bench_f12_b16.json 2>> $O/err.log
$B --frames 12 --batch 128 --steps 10 --warmup 3 --profile-steps 2 > $O/${TAG}_bench_f12_b128.json 2>> $O/err.log
$B --frames 4 --batch 64                    > $O/${TAG}_bench_f4_b64.json 2>> $O/err.log
$B --arch basic                             > $O/${TAG}_bench_basic.json 2>> $O/err.log
$B --dtype f32 --batch 64 --steps 30 --warmup 5 > $O/${TAG}_bench_f32_b64.json 2>> $O/err.log
$B --c4 --warmup 20                         > $O/${TAG}_bench_c4_2200steps.json 2>> $O/err.log
VDQN_DETERMINISTIC=1 $B                     > $O/${TAG}_bench_deterministic.json 2>> $O/err.log
$B --loss-kind huber                        > $O/${TAG}_bench_huber.json 2>> $O/err.log
$B --h2d overlap                            > $O/${TAG}_bench_h2d_overlap.json 2>> $O/err.log
$B --force-dist                             > $O/${TAG}_bench_rccl_1rank.json 2>> $O/err.log
VDQN_BENCH_SINGLE_DEVICE=1 $B --gpus 2 --backend gloo --batch 128 > $O/${TAG}_bench_2ranks_gloo_shared_gpu.json 2>> $O/err.log
# BASELINE config 3's per-rank shape, functionally: four self-launched ranks x batch 256 share this box's GPU (gloo carries the exchange)
VDQN_BENCH_SINGLE_DEVICE=1 $B --gpus 4 --backend gloo --batch 256 --steps 10 --warmup 2 --ramp-seconds 0 --no-profile > $O/${TAG}_bench_c3_4ranks_x256_gloo_shared_gpu.json 2>> $O/err.log
python3 tools/convergence.py --steps 300 --batch 64 --out $O/${TAG}_convergence.json > $O/${TAG}_convergence.txt 2>> $O/err.log
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof_f12 -o k --output-format csv -- python3 $R/bench.py --frames 12 --batch 16 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>> $O/err.log)
find $O/prof_f12 -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats_f12_b16.csv \;
for f in $O/*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    if "value" in d:
        r = d.get("roofline") or {}
        print(f"{sys.argv[1].split('/')[-1]:48s} {d['value']:10.1f} {d['unit']}  {d['ms_per_step']:8.3f} ms/step  n_gpus={d['n_gpus']}  dominant {r.get('kernel')} {r.get('achieved')} {r.get('unit')}")
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -3 $O/${TAG}_convergence.txt
