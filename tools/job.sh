#!/bin/bash
# The GPU-box jobs of a round as ONE parametrised script (run through tools/gpu_job.sh, which freezes the tree first):
#     tools/gpu_job.sh <timeout_s> 'bash tools/job.sh <TAG> <recipe> [<recipe> ...]'
# Every recipe writes under gpurun_out/<TAG>/; what is to be judged is then copied into profiles/ by hand.
#   suite          the GPU test suite as the driver runs it (plain pytest -m gpu, 1150 s limit, --durations) + smoke(), logs + return codes
#   slow           the full-size CPU-oracle cases (VDQN_TEST_SLOW=1, -m 'gpu and slow_oracle')
#   variants       the switch-variant tests (VDQN_TEST_VARIANTS=1, -m 'gpu and variants')
#   tests:<expr>   pytest -m gpu -k "<expr>" (underscores for spaces: tests:vs_oracle_or_emulating)
#   bench          three default bench.py runs in a row (the first with the CPU baseline)
#   profile        rocprofv3 kernel stats (serial + overlap) and the event-profiled bench line: tools/profile_round.sh
#   pmc            rocprofv3 --pmc passes (MFMA busy, LDS conflicts, HBM traffic): tools/pmc_mfma.sh
#   records        the other configurations of BASELINE.json and the mode records: tools/records_round.sh
#   ab:<a>@<b>     alternating bench runs of two environments, e.g. ab:new:@old:VDQN_SKINNY=0,VDQN_WGRAD_STREAMS=1
#   abd:<a>@<b>    the same with bench.py --force-dist (one rank through RCCL: the exchange's stream ordering on one GPU)
#   e2e            the trainer-loop throughput record next to bench.py: tools/trainer_e2e.py
#   head           per-launch times of the Q-head's layers: tools/bench_head.py
#   diag:<script>  python tools/<script>.py (diag_bf16_emulation, diag_basic_outlier, ...), output captured
#   tool:<out>:<K=V,K=V>:<script>[+arg+arg]   python tools/<script>.py [args] under the given environment, output in <out>.txt
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
for recipe in "$@"; do
  name=${recipe%%:*}; arg=${recipe#*:}; [ "$arg" = "$recipe" ] && arg=""
  case $name in
    suite)   # what the driver runs: plain -m gpu (variants deselected), under the driver's own 1200 s limit, then smoke()
             t0=$(date +%s); timeout 1150 python -m pytest tests -m gpu -q --maxfail=8 --durations=60 --durations-min=1.5 > "$O/pytest_gpu.log" 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - t0 ))s" >> "$O/pytest_gpu.log"
             timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$O/smoke.log" 2>&1; echo "smoke rc=$?" >> "$O/smoke.log"
             tail -4 "$O/pytest_gpu.log" | cut -c1-300; tail -2 "$O/smoke.log" | cut -c1-200 ;;
    slow)    # the full-size CPU-oracle cases the plain suite deselects (@pytest.mark.slow_oracle)
             t0=$(date +%s); VDQN_TEST_SLOW=1 timeout 1500 python -m pytest tests -m "gpu and slow_oracle" -q > "$O/pytest_slow_oracle.log" 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - t0 ))s" >> "$O/pytest_slow_oracle.log"; tail -3 "$O/pytest_slow_oracle.log" | cut -c1-300 ;;
    variants) # the A/B switch variants the plain suite deselects (tests/conftest.py: @pytest.mark.variants)
             t0=$(date +%s); VDQN_TEST_VARIANTS=1 timeout 2400 python -m pytest tests -m "gpu and variants" -q --maxfail=8 > "$O/pytest_variants.log" 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - t0 ))s" >> "$O/pytest_variants.log"; tail -3 "$O/pytest_variants.log" | cut -c1-300 ;;
    tests)   timeout 2400 python -m pytest tests -m gpu -q -k "${arg//_or_/ or }" > "$O/pytest_${arg:0:40}.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest_${arg:0:40}.log"; tail -3 "$O/pytest_${arg:0:40}.log" | cut -c1-300 ;;
    bench)   for i in 1 2 3; do timeout 300 python bench.py $( [ $i -gt 1 ] && echo --no-cpu-baseline ) > "$O/bench_$i.json" 2>> "$O/err.log"; done
             python - "$O" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["achieved"], d["mfma_clock_under_load"]["shader_clock_ghz"])
PY
             ;;
    profile) timeout 900 bash tools/profile_round.sh "$TAG" > "$O/profile_round.log" 2>&1; tail -3 "$O/profile_round.log" ;;
    pmc)     timeout 600 bash tools/pmc_mfma.sh "$TAG" > "$O/pmc.log" 2>&1; tail -24 "$O/pmc.log" ;;
    records) timeout 1800 bash tools/records_round.sh "$TAG" > "$O/records.log" 2>&1; tail -16 "$O/records.log" | cut -c1-220 ;;
    ab)      a=${arg%%@*}; b=${arg#*@}; timeout 900 python tools/ab_env.py --rounds 3 "$a" "$b" > "$O/ab_${a%%:*}_${b%%:*}.txt" 2>&1; grep -A3 "medians" "$O/ab_${a%%:*}_${b%%:*}.txt" ;;
    abd)     a=${arg%%@*}; b=${arg#*@}; timeout 900 python tools/ab_env.py --rounds 3 --bench-args "--force-dist" "$a" "$b" > "$O/abd_${a%%:*}_${b%%:*}.txt" 2>&1; grep -A3 "medians" "$O/abd_${a%%:*}_${b%%:*}.txt" ;;
    e2e)     timeout 1500 python tools/trainer_e2e.py --out "$O/trainer_e2e.json" > "$O/trainer_e2e.log" 2>&1; tail -2 "$O/trainer_e2e.log" | cut -c1-600 ;;
    head)    timeout 300 python tools/bench_head.py >> "$O/bench_head.txt" 2>&1; tail -1 "$O/bench_head.txt" ;;
    diag)    timeout 900 python "tools/${arg%% *}.py" ${arg#* } > "$O/${arg%% *}.txt" 2>&1; tail -3 "$O/${arg%% *}.txt" | cut -c1-300 ;;
    tool)    out=${arg%%:*}; rest=${arg#*:}; envs=${rest%%:*}; cmd=${rest#*:}; cmd=${cmd//+/ }  # ('+' stands for a space in the script's arguments)
             ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; unset IFS; timeout 900 python "tools/${cmd%% *}.py" $( [ "$cmd" != "${cmd#* }" ] && echo ${cmd#* } ) ) > "$O/$out.txt" 2>&1
             tail -4 "$O/$out.txt" | cut -c1-400 ;;
    *)       echo "unknown recipe $recipe"; exit 2 ;;
  esac
done
