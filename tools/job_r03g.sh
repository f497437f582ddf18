cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03g
mkdir -p $O
VDQN_LIB=stamp timeout 300 python tools/stamp_wgrad.py 256 > $O/wgrad_stamps.txt 2>&1
cat $O/wgrad_stamps.txt
VDQN_LIB=stamp timeout 300 python tools/stamp_win9.py 512 > $O/win9u_stamps.txt 2>&1
cat $O/win9u_stamps.txt
