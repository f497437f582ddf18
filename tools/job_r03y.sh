cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03y
mkdir -p $O
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_basic.py -m gpu -q -s -k "all_elements_vs_oracle_f32 and 4-3" 2>&1 | grep -a 'worst gradient\|passed\|failed' ; done > $O/basic_f4.txt
for i in 1 2 3; do VDQN_DETERMINISTIC=1 timeout 300 python -m pytest tests/test_gpu_basic.py -m gpu -q -s -k "all_elements_vs_oracle_f32 and 4-3" 2>&1 | grep -a 'worst gradient\|passed\|failed' ; done > $O/basic_f4_det.txt
cat $O/basic_f4.txt; echo ---; cat $O/basic_f4_det.txt
