cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03t
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "split_k or linear or conv_forward" > $O/pytest_ops.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ops.log
tail -n 12 $O/pytest_ops.log | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "early_adam or td_step or golden or deterministic_mode or target_sync or grouped or (non_default and (KSPLIT or SPLIT_ONLINE or PACK_AFTER))" > $O/pytest_engine.log 2>&1; echo "pytest rc=$?" >> $O/pytest_engine.log
tail -n 6 $O/pytest_engine.log | cut -c1-300
timeout 300 python tools/timeline_live.py --dump > $O/timeline.txt 2> $O/timeline.err; head -5 $O/timeline.txt; tail -n 3 $O/timeline.err
timeout 1200 python tools/ab_env.py --rounds 4 ksplit: nosplit:VDQN_KSPLIT=0 > $O/ab.txt 2>&1
grep -v '^wgrad\|^conv64\|^pack\|^td_loss\|^colsum\|^stem\|^unfold\|^fold\|^adam\|^igemm_\|dgrad_s2' $O/ab.txt
