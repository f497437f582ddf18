cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03final_$1
mkdir -p $O
for i in 1 2 3; do timeout 300 python bench.py $( [ $i -gt 1 ] && echo --no-cpu-baseline ) > $O/bench_$i.json 2>> $O/err.log; done
python - "$O" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["achieved"], d["mfma_clock_under_load"]["shader_clock_ghz"])
PY
