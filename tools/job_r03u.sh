cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03u
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv or window or persistent or grouped or linear" > $O/pytest_ops.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ops.log
tail -n 5 $O/pytest_ops.log | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "td_step or golden or deterministic_mode or grouped" > $O/pytest_engine.log 2>&1; echo "pytest rc=$?" >> $O/pytest_engine.log
tail -n 5 $O/pytest_engine.log | cut -c1-300
timeout 300 python tools/bench_conv.py > $O/bench_conv.txt 2>&1; grep -v amdgpu $O/bench_conv.txt | head -4
VDQN_LIB=prev timeout 300 python tools/bench_conv.py > $O/bench_conv_prev.txt 2>&1; grep -v amdgpu $O/bench_conv_prev.txt | head -4
timeout 1200 python tools/ab_env.py --rounds 4 new: prev:VDQN_LIB=prev > $O/ab.txt 2>&1
grep -v '^wgrad\|^pack\|^td_loss\|^colsum\|^stem\|^unfold\|^fold\|^adam\|^igemm' $O/ab.txt
