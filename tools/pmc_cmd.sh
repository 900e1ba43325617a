# usage: bash tools/pmc_cmd.sh <tag> <kernel-name-substring> <command ...>
# rocprofv3 PMC passes (one counter group per pass, --kernel-trace only) over every launch of the named kernel by the
# given command, e.g.   bash tools/pmc_cmd.sh chain_fwd "conv_chain_bf16_kernel<false>" python tools/chain_time.py
# -> gpurun_out/pmc_<tag>.json (averages over the kernel's last launches + derived fractions)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; SUB=$2; shift; shift
run() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$name -o p -- "${CMD[@]}" > gpurun_out/pmc_${TAG}_$name.log 2>&1; }
CMD=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
SUB="$SUB" TAG="$TAG" python - <<'PY'
import csv, glob, json, os
sub, tag = os.environ["SUB"], os.environ["TAG"]
out = {"_kernel_substring": sub}
for name in ["sq1", "sq2", "fetch", "write", "grbm"]:
    fs = glob.glob("gpurun_out/pmc_%s_%s/*counter_collection.csv" % (tag, name))
    if not fs:
        print(name, "no file"); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if sub in r["Kernel_Name"]]
    if not rows:
        print(name, "no rows"); continue
    ids = sorted(set(int(r["Dispatch_Id"]) for r in rows))[-5:]
    agg = {}
    for r in rows:
        if int(r["Dispatch_Id"]) in ids:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if int(r["Dispatch_Id"]) in ids]
    n_disp = len(ids)
    out[name] = dict({k: sum(v) / n_disp for k, v in agg.items()}, avg_duration_us=sum(dur) / len(dur) / 1e3,
                     kernel=rows[0]["Kernel_Name"][:160], grid=int(rows[0]["Grid_Size"]), workgroup=int(rows[0]["Workgroup_Size"]))
try:
    cu, simd = 256, 1024
    scale = out["sq1"]["avg_duration_us"] / out["grbm"]["avg_duration_us"]
    cyc = out["grbm"]["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
    out["_derived"] = {
        "kernel_cycles_per_xcd": cyc, "effective_clock_ghz": cyc / out["grbm"]["avg_duration_us"] / 1e3,
        "mfma_busy_fraction": out["sq1"]["SQ_VALU_MFMA_BUSY_CYCLES"] / simd / (cyc * scale),
        "lds_array_active_fraction": out["sq1"]["SQ_LDS_IDX_ACTIVE"] / cu / (cyc * scale),
        "waves_waiting_fraction": out["sq1"]["SQ_WAIT_ANY"] / out["sq1"]["SQ_WAVE_CYCLES"],
        "lds_bank_conflict_cycles": out["sq1"]["SQ_LDS_BANK_CONFLICT"],
        "fetch_bytes_corrected": out["fetch"]["FETCH_SIZE"] * 1024 * 2, "write_bytes": out["write"]["WRITE_SIZE"] * 1024,
        "note": "per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); "
                "PMC passes run slower than un-instrumented launches"}
except Exception as e:
    out["_derived_error"] = str(e)
json.dump(out, open("gpurun_out/pmc_%s.json" % tag, "w"), indent=1)
print(tag, json.dumps(out.get("_derived", out.get("_derived_error")), indent=1))
PY
