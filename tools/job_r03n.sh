cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03n
mkdir -p $O
timeout 300 python tools/timeline_live.py --dump > $O/timeline.txt 2> $O/timeline.err; head -8 $O/timeline.txt; tail -3 $O/timeline.err
VDQN_NO_OVERLAP=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_serial.json 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_overlap.json 2>> $O/err.log
python - <<'PY'
import json
for f in ("bench_serial","bench_overlap"):
    d=json.loads(open(f"gpurun_out/r03n/{f}.json").read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], "kernel sum", round(sum(k["ms_per_step"] for k in d["kernels"].values()),3))
PY
