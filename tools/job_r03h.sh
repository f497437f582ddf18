cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03h
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -x -q -k "wgrad or deterministic_wgrad or test_td_steps_match or grouped" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log | cut -c1-300
timeout 900 python tools/ab_env.py --rounds 4 trickle: notrickle:VDQN_LIB=notrickle > $O/ab.txt 2>&1
cat $O/ab.txt
