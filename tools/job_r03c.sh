cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03c
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log | cut -c1-400
grep -n "parity gate" $O/pytest.log | cut -c1-700 | head
