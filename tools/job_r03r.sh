cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03r
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_launch.py tests/test_gpu_ddp.py -m gpu -x -q > $O/pytest_ddp.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ddp.log
tail -n 6 $O/pytest_ddp.log | cut -c1-300
timeout 300 python bench.py --force-dist --no-cpu-baseline > $O/bench_rccl_1rank.json 2>> $O/err.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2>> $O/err.log
python - <<'PY'
import json
for f in ("bench_rccl_1rank","bench"):
    d=json.loads(open(f"gpurun_out/r03r/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"])
PY
