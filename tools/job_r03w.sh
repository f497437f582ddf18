cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03w
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "packed_one_update or early_adam or td_step or golden or deterministic_mode or target_sync or grouped or (non_default and (SPLIT_ONLINE or GROUPED_FWD))" > $O/pytest_engine.log 2>&1; echo "pytest rc=$?" >> $O/pytest_engine.log
tail -n 5 $O/pytest_engine.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_basic.py tests/test_gpu_launch.py -m gpu -x -q > $O/pytest_other.log 2>&1; echo "pytest rc=$?" >> $O/pytest_other.log
tail -n 4 $O/pytest_other.log | cut -c1-300
timeout 1200 python tools/ab_env.py --rounds 4 ahead: inline:VDQN_BENCH_NO_PACK_AHEAD=1 > $O/ab.txt 2>&1
grep -v '^wgrad\|^td_loss\|^colsum\|^stem\|^unfold\|^fold\|^adam\|^igemm\|^conv64' $O/ab.txt
