cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03v
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 8 $O/pytest.log | cut -c1-300
timeout 700 bash tools/profile_round.sh r03v > $O/profile_round.log 2>&1; tail -n 3 $O/profile_round.log
timeout 400 bash tools/pmc_mfma.sh r03v > $O/pmc.log 2>&1; tail -n 24 $O/pmc.log
timeout 300 python tools/timeline_live.py --dump > $O/timeline.txt 2> $O/timeline.err; head -5 $O/timeline.txt
timeout 1500 bash tools/records_round.sh r03v > $O/records.log 2>&1; tail -n 16 $O/records.log | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 3 $O/smoke.log
