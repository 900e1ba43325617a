# usage: bash tools/pmc_layer.sh <kind> <layer> <kernel-name-substring>      (kind: fwd | dgrad | wgrad | wgrad_grouped)
# rocprofv3 PMC passes (one counter group per pass) over ONE conv launch; prints per-pass averages and writes
# gpurun_out/pmc_<kind>_<layer>.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KIND=$1; LAYER=$2; SUB=$3
run() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_${KIND}_${LAYER}_$name -o p -- python tools/run_one.py --kind $KIND --layer $LAYER --reps 5 > gpurun_out/pmc_${KIND}_${LAYER}_$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES
run ta TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python - <<PY
import csv,glob,json
out={"_kernel_substring":"$SUB","_launch":"$KIND of $LAYER, BASELINE config 3 shapes (B=32, 500 output frames)"}
for name in ["sq1","sq2","ta","fetch","write","grbm"]:
    fs=glob.glob("gpurun_out/pmc_${KIND}_${LAYER}_%s/*counter_collection.csv"%name)
    if not fs: print(name,"no file"); continue
    rows=[r for r in csv.DictReader(open(fs[0])) if "$SUB" in r["Kernel_Name"]]
    if not rows: print(name,"no rows"); continue
    ids=sorted(set(int(r["Dispatch_Id"]) for r in rows))[-5:]
    agg={}
    for r in rows:
        if int(r["Dispatch_Id"]) in ids: agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    dur=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in rows if int(r["Dispatch_Id"]) in ids]
    print(name,"dur_us",round(sum(dur)/len(dur)/1e3,1),{k:round(sum(v)/len(v)) for k,v in agg.items()},"grid",rows[0]["Grid_Size"],"wg",rows[0]["Workgroup_Size"])
    out[name]=dict({k:sum(v)/len(v) for k,v in agg.items()}, avg_duration_us=sum(dur)/len(dur)/1e3, kernel=rows[0]["Kernel_Name"][:160], grid=int(rows[0]["Grid_Size"]), workgroup=int(rows[0]["Workgroup_Size"]))
try:
    cu=256; simd=1024
    cyc=out["grbm"]["GRBM_GUI_ACTIVE"]/8.0   # summed over the 8 XCDs
    out["_derived"]={"kernel_cycles_per_xcd":cyc,"effective_clock_ghz":cyc/out["grbm"]["avg_duration_us"]/1e3,
        "mfma_busy_fraction":out["sq1"]["SQ_VALU_MFMA_BUSY_CYCLES"]/simd/(out["grbm"]["GRBM_GUI_ACTIVE"]/8.0*out["sq1"]["avg_duration_us"]/out["grbm"]["avg_duration_us"]),
        "lds_array_active_fraction":out["sq1"]["SQ_LDS_IDX_ACTIVE"]/cu/(out["grbm"]["GRBM_GUI_ACTIVE"]/8.0*out["sq1"]["avg_duration_us"]/out["grbm"]["avg_duration_us"]),
        "waves_waiting_fraction":out["sq1"]["SQ_WAIT_ANY"]/out["sq1"]["SQ_WAVE_CYCLES"],
        "lds_bank_conflict_cycles":out["sq1"]["SQ_LDS_BANK_CONFLICT"],
        "fetch_bytes_corrected":out["fetch"]["FETCH_SIZE"]*1024*2,"write_bytes":out["write"]["WRITE_SIZE"]*1024,
        "note":"FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); PMC passes run ~12 % slower than un-instrumented launches"}
except Exception as e:
    out["_derived_error"]=str(e)
json.dump(out,open("gpurun_out/pmc_${KIND}_${LAYER}.json","w"),indent=1)
print(json.dumps(out.get("_derived"),indent=1))
PY
