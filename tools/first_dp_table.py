"""Prints the table of tools/first_dp_run.sh from the JSON lines it collected: one row per combination."""
import json
import sys
from pathlib import Path


def last_json_line(path):
    try:
        lines = [l for l in Path(path).read_text().splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def main(out_dir, n):
    out_dir, n = Path(out_dir), int(n)
    rows = []
    for cfg in (3, 5):
        single = last_json_line(out_dir / "c{}_single.json".format(cfg))
        base = single["value"] if single else None
        for path in sorted(out_dir.glob("c{}_*_cus*_split*_*.json".format(cfg))):
            line = last_json_line(path)
            _, exch, cus, split, rccl = path.stem.split("_", 4)
            if line is None:
                rows.append((cfg, exch, cus[3:], split[5:], rccl, "FAILED (see {}.err)".format(path.stem)))
                continue
            dp = line.get("data_parallel", {})
            eff = line["value"] / (n * base) if base else float("nan")
            rows.append((cfg, exch, cus[3:], split[5:], rccl,
                         "{:9.0f} utt/s  {:6.3f} ms  eff {:5.3f}  exposed {:6.3f} ms  allreduce alone {:6.3f} ms "
                         "({:5.1f} GB/s bus)  identical {}".format(
                             line["value"], line["ms_per_step"], eff, dp.get("exposed_communication_ms", float("nan")),
                             dp.get("allreduce_alone_ms", float("nan")), dp.get("allreduce_busbw_GBps", float("nan")),
                             dp.get("reduced_gradients_and_weights_identical_on_all_ranks"))))
        if base:
            print("config {}: single GPU {:.0f} utt/s ({:.3f} ms per step)".format(cfg, base, single["ms_per_step"]))
    print("{:>6} {:>9} {:>8} {:>5} {:>8}  result ({} ranks)".format("config", "exchange", "comm_cus", "split", "rccl", n))
    for r in rows:
        print("{:>6} {:>9} {:>8} {:>5} {:>8}  {}".format(*r))


if __name__ == "__main__":
    main(*sys.argv[1:3])
