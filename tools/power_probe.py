#!/usr/bin/env python
"""Socket power and shader clock (amdgpu hwmon / sysfs, sampled every 5 ms) while one phase at a time runs in a loop:
idle, the resident training step with and without the fused inner-layer launch, and single kernels of the step.
Question: is the step power-limited (clock below its maximum under the MFMA load), i.e. is energy per step the invariant?"""
import glob
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def find_sensors():
    import torch
    out = {}
    pr = torch.cuda.get_device_properties(0)
    bdf = "{:04x}:{:02x}:{:02x}.0".format(pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    print("device 0 is PCI", bdf)
    for h in glob.glob("/sys/bus/pci/devices/{}/hwmon/hwmon*".format(bdf)):
        for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
            p = Path(h) / name
            if p.exists():
                out.setdefault(name, str(p))
    for p in glob.glob("/sys/bus/pci/devices/{}/pp_dpm_sclk".format(bdf)):
        out.setdefault("pp_dpm_sclk", p)
    return out


class Sampler(threading.Thread):
    def __init__(self, sensors):
        super().__init__(daemon=True)
        self.sensors, self.rows, self.stop_flag = sensors, [], False

    def run(self):
        while not self.stop_flag:
            row = [time.perf_counter()]
            for k in ("power1_average", "power1_input", "freq1_input", "temp1_input"):
                try:
                    row.append(float(open(self.sensors[k]).read().strip()) if k in self.sensors else float("nan"))
                except Exception:
                    row.append(float("nan"))
            self.rows.append(row)
            time.sleep(0.005)


def main():
    import torch
    import bench
    from speechless_amd.engine import Engine, wav2letter_layer_specs
    from speechless_amd.net import Wav2Letter
    sensors = find_sensors()
    print("sensors:", sensors)
    if "pp_dpm_sclk" in sensors:
        print(open(sensors["pp_dpm_sclk"]).read())
    specs = wav2letter_layer_specs(bench.MEL, bench.K_CLASSES)
    x, labels, lab_len, pred_len = bench.synthetic_batch(0, bench.BATCH_PER_GPU)
    engines = {}
    for chain in (True, False):
        eng = Engine(specs, bench.K_CLASSES, dtype="bf16")
        eng.use_chain = chain
        eng.set_weights(Wav2Letter._glorot_uniform(specs, 2))
        eng.load_input(torch.from_numpy(x).cuda())
        eng.set_labels(labels, lab_len, pred_len)
        for _ in range(3):
            eng.train_step_resident()
        engines[chain] = eng
    torch.cuda.synchronize()
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    sampler = Sampler(sensors)
    sampler.start()
    phases = []

    def phase(name, fn, seconds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            n += 20
        t1 = time.perf_counter()
        phases.append((name, t0, t1, n))

    # single launches of the step on their real operands (energy per launch = power x time; algorithmic GFLOP beside it)
    import ctypes
    from speechless_amd import _lib
    eng = engines[True]
    buf = eng.cur
    st = torch.cuda.current_stream().cuda_stream
    i = [q.index for q in eng.plans if q.spec.name == "big_conv_1"][0]
    p = eng.plans[i]
    _, bias = eng.layer_param_views(eng.params, p)
    dw, _ = eng.layer_param_views(eng.grads, p)

    def fwd_big1():
        eng.lib.call("sl_conv1d_nt", buf.y[i - 1].data_ptr(), eng.w_fwd[i].data_ptr(), bias.data_ptr(), None,
                     buf.y[i].data_ptr(), ctypes.byref(buf.fwd_geom[i]), _lib.EPI_BIAS_RELU, eng.dtype_code, 0, 0,
                     buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)

    def dgrad_big1():
        eng.lib.call("sl_conv1d_nt", buf.g[i].data_ptr(), eng.w_dgrad[i].data_ptr(), None, buf.y[i - 1].data_ptr(),
                     buf.g[i - 1].data_ptr(), ctypes.byref(buf.dgrad_geom[i]), _lib.EPI_RELU_MASK, eng.dtype_code, 0, 0,
                     buf.nt_ws.data_ptr(), buf.nt_ws.numel(), st)

    def wgrad_big1():
        eng.lib.call("sl_conv1d_wgrad", buf.y[i - 1].data_ptr(), buf.g[i].data_ptr(), dw.data_ptr(),
                     ctypes.byref(buf.wgrad_geom[i]), eng.dtype_code, 0, buf.wgrad_ws.data_ptr(), buf.wgrad_ws.numel(), st)
    a1 = torch.randn(16384, 8192, device="cuda", dtype=torch.bfloat16)
    b1 = torch.randn(8192, 2048, device="cuda", dtype=torch.bfloat16)
    phase("idle", lambda: time.sleep(0.01), 1.0)
    phase("big_conv_1 forward alone (512 GF)", fwd_big1, 3.0)
    phase("big_conv_1 input gradient alone (512 GF)", dgrad_big1, 3.0)
    phase("big_conv_1 weight gradient alone (512 GF)", wgrad_big1, 3.0)
    phase("vendor GEMM 16384x8192x2048 (550 GF)", lambda: torch.matmul(a1, b1), 3.0)
    phase("idle", lambda: time.sleep(0.01), 1.0)
    phase("step, fused inner layers", engines[True].train_step_resident, 4.0)
    phase("idle", lambda: time.sleep(0.01), 1.0)
    phase("step, single launches", engines[False].train_step_resident, 4.0)
    phase("idle", lambda: time.sleep(0.01), 1.0)
    phase("forward only, fused", lambda: engines[True].forward(training=True), 3.0)
    phase("torch.matmul 8192^3 bf16 (vendor GEMM)", lambda: torch.matmul(a, a), 3.0)
    sampler.stop_flag = True
    sampler.join()
    rows = np.array(sampler.rows)
    print("{:42s} {:>10s} {:>9s} {:>9s} {:>9s} {:>8s} {:>10s}".format("phase", "ms/iter", "P_avg W", "P_in W", "sclk MHz", "temp C",
                                                                   "J/iter"))
    for name, t0, t1, n in phases:
        m = (rows[:, 0] >= t0 + 0.3 * (t1 - t0)) & (rows[:, 0] <= t1)
        pa, pi, f, tc = (np.nanmean(rows[m, i]) if m.any() else float("nan") for i in (1, 2, 3, 4))
        p = pi if np.isfinite(pi) else pa
        print("{:42s} {:10.3f} {:9.0f} {:9.0f} {:9.0f} {:8.1f} {:10.3f}".format(name, (t1 - t0) / n * 1e3, pa / 1e6, pi / 1e6, f / 1e6,
                                                                          tc / 1e3, p / 1e6 * (t1 - t0) / n))


if __name__ == "__main__":
    main()
