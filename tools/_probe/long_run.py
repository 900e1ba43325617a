import sys, time, glob
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import bench
from speechless_amd.engine import Engine, wav2letter_layer_specs
from speechless_amd.net import Wav2Letter
pr = torch.cuda.get_device_properties(0)
bdf = "{:04x}:{:02x}:{:02x}.0".format(pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
hw = glob.glob("/sys/bus/pci/devices/{}/hwmon/hwmon*".format(bdf))[0]
def rd(n):
    try: return float(open(hw + "/" + n).read())
    except Exception: return float("nan")
specs = wav2letter_layer_specs(bench.MEL, bench.K_CLASSES)
x, labels, lab_len, pred_len = bench.synthetic_batch(0, bench.BATCH_PER_GPU)
eng = Engine(specs, bench.K_CLASSES, dtype="bf16")
eng.set_weights(Wav2Letter._glorot_uniform(specs, 2))
eng.load_input(torch.from_numpy(x).cuda())
eng.set_labels(labels, lab_len, pred_len)
for _ in range(5): eng.train_step_resident()
torch.cuda.synchronize()
T0 = time.perf_counter()
mode = sys.argv[1] if len(sys.argv) > 1 else "steady"
for sec in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    n = 0; t0 = time.perf_counter(); ps = []; fs = []
    while time.perf_counter() - t0 < 1.0:
        for _ in range(25): eng.train_step_resident()
        torch.cuda.synchronize(); n += 25
        ps.append(rd("power1_input") / 1e6); fs.append(rd("freq1_input") / 1e6)
    dt = time.perf_counter() - t0
    print("t=%5.1f s  %.4f ms/step  power %.0f W  sclk %.0f MHz  temp %s" % (time.perf_counter() - T0, dt / n * 1e3, np.mean(ps), np.mean(fs),
          [rd(k) / 1e3 for k in ("temp1_input", "temp2_input", "temp3_input")]), flush=True)
    if mode == "pauses" and sec % 4 == 3:
        time.sleep(2.0)
