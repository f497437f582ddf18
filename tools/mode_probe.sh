#!/bin/bash
# Process-to-process spread of the update time on one box (round 6: 5.70 vs 5.34 ms in consecutive bench.py runs of one job).
# Alternates back-to-back runs with runs behind an idle pause and records the SMI clocks / power / temperature before each:
#   tools/gpu_job.sh 900 'bash tools/mode_probe.sh <TAG>'
TAG=${1:-probe}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
smi() { /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp --showperflevel 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature|Performance" | tr -s ' ' | cut -c1-110; }
for i in 1 2 3 4 5 6 7 8; do
  case $i in 3|6) sleep 45 ;; 4) python -c "
import torch, time
x = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
t = time.time()
while time.time() - t < 40: y = x @ x
torch.cuda.synchronize()" ;; esac
  echo "== run $i ($(case $i in 3|6) echo after 45 s idle;; 4) echo after 40 s of GEMM heat;; *) echo back to back;; esac))" >> "$O/mode_probe.txt"
  smi >> "$O/mode_probe.txt"
  timeout 200 python bench.py --no-cpu-baseline > "$O/probe_$i.json" 2>> "$O/err.log"
  python - "$O/probe_$i.json" >> "$O/mode_probe.txt" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = d["kernels"]
print("   value", d["value"], "ms", d["ms_per_step"], "windows", d["window_ms_per_step"], "kernel sum", round(sum(k["ms_per_step"] for k in ks.values()), 3),
      "clock", d["mfma_clock_under_load"]["shader_clock_ghz"])
PY
done
cat "$O/mode_probe.txt"
