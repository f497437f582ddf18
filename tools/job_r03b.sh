cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03b
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -x -q -k "conv or wgrad or nine_tap or grouped or td_step or deterministic or linear or side_stream" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
python tools/ab_env.py --rounds 3 grouped:VDQN_GROUPED_FWD=1 twopass:VDQN_GROUPED_FWD=0 grouped_win3:VDQN_GROUPED_FWD=1,VDQN_WGRAD_WINDOW=3 > $O/ab.txt 2>&1
cat $O/ab.txt
