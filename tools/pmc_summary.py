#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into per-kernel HBM traffic.
gfx950 correction (MI355X_MICROARCH.md "HBM"): FETCH_SIZE reports half the bytes of a coalesced streaming read
(calibrated here: the pass kernels read exactly 1 GiB per launch and the counter shows 0.5 GiB), WRITE_SIZE is exact.
Counter unit: KiB.   usage: pmc_summary.py <fetch_csv> <write_csv> <workload> <launches_per_step_json> > out.json"""
import collections
import csv
import json
import sys


def per_kernel(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][0] += 1
        acc[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}, {k: v[0] for k, v in acc.items()}


def main():
    fetch, _ = per_kernel(sys.argv[1])
    write, _ = per_kernel(sys.argv[2])
    workload = sys.argv[3]
    launches = json.loads(sys.argv[4])  # {kernel substring: launches per step}
    kernels, total = [], 0.0
    for name in fetch:
        if not any(k in name for k in ("fft_lines_kernel", "fft_xcd_fused_kernel", "fft_xcd_res_kernel", "fft_line32k_kernel", "fft_line_reg_kernel", "stockham", "r2c_post", "c2r_pre", "fft_xcd_r2c", "fft_xcd_c2r", "fft_xcd_rt", "fft_xcd_hx", "fft_xcd_conv")):
            continue
        rd = fetch[name] * 1024 * 2.0      # gfx950 x2 read correction
        wr = write.get(name, 0.0) * 1024
        n = next((v for k, v in launches.items() if k in name), 0)
        kernels.append({"kernel": name, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "launches_per_step": n})
        total += (rd + wr) * n
    print(json.dumps({"workload": workload, "hbm_bytes_per_step": total, "kernels": kernels,
                      "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950), KiB units"}, indent=1))


if __name__ == "__main__":
    main()
