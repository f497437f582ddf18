cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03x
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 4 $O/pytest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 3 $O/smoke.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r03x/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['frame_pack'])"
