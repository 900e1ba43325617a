# HBM-side traffic (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the launches behind bench.py's roofline
# kernel: wgrad_tn_ilv_kernel = wgrad of big_conv_1 and big_conv_2; beside them the balanced launch for striding_conv +
# inner_conv_1..7 (wgrad_tn_ilv_multi_kernel) and the one-launch backward of output_conv (conv1x1_bwd_kernel).
# Writes gpurun_out/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_traffic
for L in big_conv_1 big_conv_2 multi output_conv; do
  K=wgrad; if [ $L = multi ]; then K=wgrad_multi; fi; if [ $L = output_conv ]; then K=bwd1x1; fi
  LAYER=$L; if [ $L = multi ]; then LAYER=striding_conv; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_traffic/${L}_$C -o p -- python tools/run_one.py --kind $K --layer $LAYER --reps 5 > gpurun_out/pmc_traffic/${L}_$C.log 2>&1
  done
done
python - <<PY
import csv, glob, json
names = {"big_conv_1": "wgrad_tn_ilv_kernel", "big_conv_2": "wgrad_tn_ilv_kernel", "multi": "wgrad_tn_ilv_multi_kernel",
         "output_conv": "conv1x1_bwd_kernel"}
labels = {"multi": "striding_conv + inner_conv_1..7 (wgrad_tn_ilv_multi_kernel)", "output_conv": "output_conv backward (conv1x1_bwd_kernel)"}
import os, time
out = {"kernel": "wgrad_tn_ilv_kernel", "unit": "bytes per launch",
       "measured": "{} ({})".format(time.strftime("%Y-%m-%d"), os.environ.get("TAG", "untagged")), "note": "FETCH_SIZE/WRITE_SIZE are KiB; FETCH doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); Infinity-Cache hits are counted, so this is an upper bound on true HBM traffic", "launches": {}}
for L in ["big_conv_1", "big_conv_2", "multi", "output_conv"]:
    vals = {}
    for C in ["FETCH_SIZE", "WRITE_SIZE"]:
        f = glob.glob("gpurun_out/pmc_traffic/%s_%s/*counter_collection.csv" % (L, C))[0]
        rows = [r for r in csv.DictReader(open(f)) if names[L] in r["Kernel_Name"] and r["Counter_Name"] == C]
        ids = sorted(set(int(r["Dispatch_Id"]) for r in rows))[-5:]
        v = [float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) in ids]
        vals[C] = sum(v) / len(v)
    out["launches"][labels.get(L, L)] = {"FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
                          "traffic_bytes": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024}
dom = [out["launches"][k]["traffic_bytes"] for k in ("big_conv_1", "big_conv_2")]
out["traffic_bytes_per_launch_avg"] = sum(dom) / len(dom)
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
