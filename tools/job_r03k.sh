cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03k
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log | cut -c1-300
timeout 700 bash tools/profile_round.sh r03k > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
timeout 400 bash tools/pmc_mfma.sh r03k > $O/pmc.log 2>&1; tail -24 $O/pmc.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
