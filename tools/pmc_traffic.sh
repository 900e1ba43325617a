# HBM-side traffic (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the launches behind bench.py's roofline
# kernel: wgrad_tn_ilv_kernel = wgrad of big_conv_1, big_conv_2 and the grouped inner_conv_1..7 launch.
# Writes gpurun_out/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_traffic
for L in big_conv_1 big_conv_2 inner_conv_1; do
  K=wgrad; if [ $L = inner_conv_1 ]; then K=wgrad_grouped; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_traffic/${L}_$C -o p -- python tools/run_one.py --kind $K --layer $L --reps 5 > gpurun_out/pmc_traffic/${L}_$C.log 2>&1
  done
done
python - <<PY
import csv, glob, json
out = {"kernel": "wgrad_tn_ilv_kernel", "unit": "bytes per launch", "note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); Infinity-Cache hits are counted, so this is an upper bound on true HBM traffic", "launches": {}}
for L in ["big_conv_1", "big_conv_2", "inner_conv_1"]:
    vals = {}
    for C in ["FETCH_SIZE", "WRITE_SIZE"]:
        f = glob.glob("gpurun_out/pmc_traffic/%s_%s/*counter_collection.csv" % (L, C))[0]
        rows = [r for r in csv.DictReader(open(f)) if "wgrad_tn_ilv_kernel" in r["Kernel_Name"] and r["Counter_Name"] == C]
        ids = sorted(set(int(r["Dispatch_Id"]) for r in rows))[-5:]
        v = [float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) in ids]
        vals[C] = sum(v) / len(v)
    out["launches"]["inner_conv_1..7 (grouped)" if L == "inner_conv_1" else L] = {"FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
                          "traffic_bytes": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024}
out["traffic_bytes_per_launch_avg"] = sum(v["traffic_bytes"] for v in out["launches"].values()) / len(out["launches"])
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
