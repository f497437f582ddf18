set -u
O=gpurun_out/r6k; mkdir -p $O
for n in 192 384; do
  for v in default bm128 bm256; do
    case $v in default) e="";; bm128) e="VDQN_WIN9_BM256=0";; bm256) e="VDQN_WIN9_BM256=2";; esac
    echo "== N_FWD=N_BWD=$n $v" >> $O/bench_conv_c5_rule.txt
    env $e N_FWD=$n N_BWD=$n REPS=30 python tools/bench_conv.py 2>/dev/null | grep -E "layer[234] 3x3 [0-9]+->[0-9]+ @" >> $O/bench_conv_c5_rule.txt
  done
done
cat $O/bench_conv_c5_rule.txt
