cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03j
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "stride2 or test_conv_forward or wgrad or linear or fused" > $O/pytest_ops.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ops.log
tail -6 $O/pytest_ops.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -x -q -k "td_step or golden or deterministic_mode or grouped or full_size or (non_default and S2WIN)" > $O/pytest_engine.log 2>&1; echo "pytest rc=$?" >> $O/pytest_engine.log
tail -6 $O/pytest_engine.log | cut -c1-300
timeout 300 python tools/bench_conv.py > $O/bench_conv.txt 2>&1; grep -v amdgpu $O/bench_conv.txt
timeout 900 python tools/ab_env.py --rounds 3 s2win: generic_s2:VDQN_S2WIN=0 > $O/ab.txt 2>&1
cat $O/ab.txt
