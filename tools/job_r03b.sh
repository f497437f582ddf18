cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03b
mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log | cut -c1-600
for m in 0 1; do VDQN_WIN9_MFMA32=$m timeout 300 python tools/bench_conv.py > $O/bench_conv_mfma32_$m.txt 2>&1; done
paste -d'|' $O/bench_conv_mfma32_0.txt $O/bench_conv_mfma32_1.txt | cut -c1-230
timeout 900 python tools/ab_env.py --rounds 3 base: mfma32:VDQN_WIN9_MFMA32=1 twopass:VDQN_GROUPED_FWD=0 win3:VDQN_WGRAD_WINDOW=3 > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
CALIB_QUICK=1 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/calib_pmc -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/probes/clock_calib > $O/clock_calib_under_pmc.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r03b")
for path in glob.glob(os.path.join(O, "calib_pmc", "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "calib" in r["Kernel_Name"]]
    with open(os.path.join(O, "calib_gui_clock.txt"), "w") as f:
        f.write("dispatch kernel duration_us GRBM_GUI_ACTIVE/8/duration_GHz\n")
        for r in rows:
            d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            f.write(f"{r['Dispatch_Id']} {r['Kernel_Name'][:40]} {d / 1e3:.1f} {float(r['Counter_Value']) / 8 / d:.3f}\n")
print(open(os.path.join(O, "calib_gui_clock.txt")).read() if os.path.exists(os.path.join(O, "calib_gui_clock.txt")) else "no pmc csv")
PY
