cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -o p -- python tools/run_one.py --kind fwd --layer big_conv_1 --reps 5 > gpurun_out/pmc_$name.log 2>&1
  tail -1 gpurun_out/pmc_$name.log
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
ls gpurun_out/pmc_sq1
python - <<PY
import csv,glob
for name in ["sq1","sq2","fetch","write","grbm"]:
    files=glob.glob("gpurun_out/pmc_%s/*counter_collection.csv"%name)
    if not files: print(name,"no file"); continue
    rows=[r for r in csv.DictReader(open(files[0])) if "conv_nt_bf16_kernel" in r["Kernel_Name"] and "ELi4ELi4ELi4ELi2ELi2" in r["Kernel_Name"]]
    agg={}
    for r in rows:
        agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    print(name, {k:(len(v), sum(v)/len(v)) for k,v in agg.items()})
    if rows: print("  grid", rows[0].get("Grid_Size"), "wg", rows[0].get("Workgroup_Size"), "vgpr", rows[0].get("VGPR_Count"), "lds", rows[0].get("LDS_Block_Size"))
PY
