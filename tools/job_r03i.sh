cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03i
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log | cut -c1-300
timeout 700 bash tools/profile_round.sh r03i > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
timeout 400 bash tools/pmc_mfma.sh r03i > $O/pmc.log 2>&1; tail -22 $O/pmc.log
VDQN_GROUPED_FWD=1 timeout 200 python bench.py --no-cpu-baseline > $O/r03i_bench_grouped_fwd.json 2>> $O/err.log
timeout 600 python tools/ab_env.py --rounds 3 base: persist2:VDQN_WIN9_PERSIST=2 split_online:VDQN_SPLIT_ONLINE=1 > $O/ab_persist_split.txt 2>&1; tail -8 $O/ab_persist_split.txt | head -5
timeout 1500 bash tools/records_round.sh r03i > $O/records.log 2>&1; tail -16 $O/records.log | cut -c1-220
