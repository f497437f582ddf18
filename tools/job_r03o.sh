cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03o
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "td_step or golden or deterministic_mode or fold or unfold or pack" > $O/pytest_engine.log 2>&1; echo "pytest rc=$?" >> $O/pytest_engine.log
tail -n 4 $O/pytest_engine.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_basic.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_basic.log 2>&1; echo "pytest rc=$?" >> $O/pytest_basic.log
tail -n 4 $O/pytest_basic.log | cut -c1-300
timeout 900 python tools/ab_env.py --rounds 3 vec: prev:VDQN_LIB=prev > $O/ab.txt 2>&1
grep -v '^igemm\|^wgrad\|^conv64\|^pack\|^td_loss\|^colsum\|^stem' $O/ab.txt
