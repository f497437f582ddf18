cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03s
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "early_adam or td_step or golden or deterministic_mode or target_sync or grouped or (non_default and (PACK_AFTER or SPLIT_ONLINE or FOLD))" > $O/pytest_engine.log 2>&1; echo "pytest rc=$?" >> $O/pytest_engine.log
tail -n 4 $O/pytest_engine.log | cut -c1-300
timeout 600 python tools/ab_env.py --rounds 2 --steps 60 --bench-args "--force-dist --no-profile" rccl_early: rccl_late:VDQN_EARLY_ADAM=0 > $O/ab_rccl.txt 2>&1
grep -v '^igemm\|^wgrad\|^conv64\|^pack\|^td_loss\|^colsum\|^stem\|^unfold\|^fold\|^adam' $O/ab_rccl.txt | tail -8
timeout 1200 python tools/ab_env.py --rounds 4 new: pack_old:VDQN_PACK_AFTER_FIRST=0 > $O/ab.txt 2>&1
grep -v '^igemm\|^wgrad\|^conv64\|^pack\|^td_loss\|^colsum\|^stem\|^unfold\|^fold\|^adam' $O/ab.txt
