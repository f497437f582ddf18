cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03l
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "stem" > $O/pytest_ops.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ops.log
tail -4 $O/pytest_ops.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "td_step or golden or deterministic_mode or grouped or (non_default and NOIDX)" > $O/pytest_engine.log 2>&1; echo "pytest rc=$?" >> $O/pytest_engine.log
tail -4 $O/pytest_engine.log | cut -c1-300
timeout 900 python tools/ab_env.py --rounds 3 noidx: idx_all:VDQN_STEM_NOIDX=0 > $O/ab.txt 2>&1
cat $O/ab.txt
