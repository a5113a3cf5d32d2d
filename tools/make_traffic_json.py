#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the default bench workload:

  python3 bench.py --steps S --warmup 0 --no-cpu-baseline --no-extras        (under --pmc FETCH_SIZE, then --pmc WRITE_SIZE)
  python tools/make_traffic_json.py <fetch_dir> <write_dir> <round/tag> <genome_mbp> <steps> [map_range_passes]

Every dispatch of the library's count kernels inside the run is summed (the run has no warm-up, so the dispatches are
exactly the S timed steps + the final table pass) and divided by S: HBM bytes per step = per launch set of one batch.
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (calibrated in round 1 on three known byte counts),
WRITE_SIZE is taken as is; both are KiB.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNT_KERNELS = ("k_p1_hist", "k_p1_scatter", "k_p1_scatter_s", "k_p1_offsets", "k_lv_units", "k_lv_hist", "k_lv_offsets", "k_lv_scatter", "k_lv_scatter_s", "k_count_regions",
                 "k_count_regions_n32", "k_count_regions_q4", "k_scan_sums", "k_exclusive_scan", "k_scan_apply", "k_p3set", "k_set2", "k_count_direct")


def pmc(d):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
            acc[name][0] += float(r["Counter_Value"])
            acc[name][1] += 1
    return acc


def main():
    fe, wr, tag, mbp, steps = pmc(sys.argv[1]), pmc(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    ranges = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    rows, tf, tw = [], 0.0, 0.0
    for k in COUNT_KERNELS:
        if k not in fe and k not in wr:
            continue
        fb, wb = fe[k][0] * 1024 * 2, wr[k][0] * 1024
        rows.append({"kernel": k, "dispatches": fe[k][1], "hbm_read_bytes_total": round(fb), "hbm_write_bytes_total": round(wb),
                     "hbm_read_bytes_per_step": round(fb / steps), "hbm_write_bytes_per_step": round(wb / steps)})
        tf += fb
        tw += wb
    reads = int(mbp * 1e6 * 30 / 150) // steps * steps
    kmers_per_step = reads // steps * 130
    out = {"workload": f"human-{mbp}mbp-{steps}steps", "genome_mbp": mbp, "ranges": ranges, "kmers_per_step": kmers_per_step,
           "hbm_bytes_per_kmer": (tf + tw) / steps / kmers_per_step,
           "what": f"configs[2] shape, {mbp} Mbp genome, 30x 150 bp reads in {steps} batches of {kmers_per_step} k-mers, k=21, counted in {ranges} map-range pass(es) (bench.py default workload)",
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps S --warmup 0 --no-cpu-baseline "
                     "--no-extras`; all dispatches of the count kernels summed and divided by S; FETCH_SIZE x 2 (gfx950), WRITE_SIZE as is",
           "kernels": rows, "hbm_read_bytes_per_step": round(tf / steps), "hbm_write_bytes_per_step": round(tw / steps),
           "hbm_bytes_per_launch": round((tf + tw) / steps), "algorithmic_bytes_per_step": 35 * kmers_per_step,
           "ratio_to_algorithmic": (tf + tw) / steps / (35 * kmers_per_step)}
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    os.makedirs(os.path.join(ROOT, "profiles", os.path.dirname(tag)), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print(tf / steps / 1e9, tw / steps / 1e9, (tf + tw) / steps / 1e9, out["ratio_to_algorithmic"])


if __name__ == "__main__":
    main()
