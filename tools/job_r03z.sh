cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03z
mkdir -p $O
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_basic.py -m gpu -q -x 2>&1 | tail -n 2; done > $O/basic3.txt; cat $O/basic3.txt
