cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03b
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -x -q -k "conv or wgrad or nine_tap or grouped or td_step or deterministic or linear or side_stream or non_default" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for m in 0 1; do VDQN_WIN9_MFMA32=$m python tools/bench_conv.py > $O/bench_conv_mfma32_$m.txt 2>&1; done
paste -d'|' $O/bench_conv_mfma32_0.txt $O/bench_conv_mfma32_1.txt | cut -c1-230
python tools/ab_env.py --rounds 3 base: mfma32:VDQN_WIN9_MFMA32=1 twopass:VDQN_GROUPED_FWD=0 win3:VDQN_WGRAD_WINDOW=3 > $O/ab.txt 2>&1
cat $O/ab.txt
