cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03p
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "early_adam or td_step or golden or deterministic_mode or target_sync or (non_default and (EARLY or WGRAD_MAIN))" > $O/pytest_engine.log 2>&1; echo "pytest rc=$?" >> $O/pytest_engine.log
tail -n 4 $O/pytest_engine.log | cut -c1-300
timeout 300 python tools/timeline_live.py --dump > $O/timeline.txt 2> $O/timeline.err; head -5 $O/timeline.txt; tail -n 3 $O/timeline.err
timeout 1200 python tools/ab_env.py --rounds 3 new: adam_late:VDQN_EARLY_ADAM=0 stemwg_side:VDQN_STEM_WGRAD_MAIN=0 both_off:VDQN_EARLY_ADAM=0,VDQN_STEM_WGRAD_MAIN=0 > $O/ab.txt 2>&1
grep -v '^igemm\|^wgrad\|^conv64\|^pack\|^td_loss\|^colsum\|^stem\|^unfold\|^fold\|^adam' $O/ab.txt
