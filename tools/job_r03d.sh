cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -x -q -k "wgrad or tilings or deterministic or (non_default and WIN128)" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log | cut -c1-300
timeout 900 python tools/ab_env.py --rounds 3 win128: win64:VDQN_WGRAD_WIN128=0 win128_b512:VDQN_WGRAD_BLOCKS8=512 > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 600 bash tools/pmc_mfma.sh r03d > $O/pmc.log 2>&1; tail -25 $O/pmc.log
