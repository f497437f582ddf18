cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
./tools/probes/clock_calib > $O/clock_calib.txt 2>&1
cat $O/clock_calib.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/calib_pmc -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/probes/clock_calib > $O/clock_calib_under_pmc.txt 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
