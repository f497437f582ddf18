cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03e
mkdir -p $O
timeout 900 python tools/ab_env.py --rounds 3 base: two_stage:VDQN_WGRAD_TWO_STAGE=1 > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -x -q -k "stem or (non_default and TWO_STAGE) or test_td_steps_match" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log | cut -c1-300
