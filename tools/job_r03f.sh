cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -x -q -k "wgrad or tilings or deterministic or stem or (non_default and (WIN128 or STREAMS or WINDOW))" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log | cut -c1-300
timeout 900 python tools/ab_env.py --rounds 3 base: win128:VDQN_WGRAD_WIN128=1 streams2:VDQN_WGRAD_STREAMS=2 win3:VDQN_WGRAD_WINDOW=3 > $O/ab.txt 2>&1
cat $O/ab.txt
