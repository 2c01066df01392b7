"""Per-launch HBM traffic of the gts kernels from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE collected separately, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).

gfx950 corrections applied (same guide, §HBM): FETCH_SIZE is in KiB and reports exactly half
of the bytes of a wide coalesced read -> bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE (KiB) is
exact for 16-byte-per-lane streaming stores -> bytes = WRITE_SIZE * 1024.
Usage: parse_pmc.py <fetch_dir> <write_dir> <out.json>"""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(directory, counter):
    files = glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def short(name):
    for key, label in (("spmm_max_fwd_kernel<4, 64, 1>", "spmm_max_fwd_f256"),
                       ("spmm_max_bwd_kernel<4, 64, 1>", "spmm_max_bwd_f256"),
                       ("spmm_cluster_stream_kernel<false, 1", "spmm_max_fwd_f256"),      # the clustered forms of K1 / K2
                       ("spmm_cluster_stream_kernel<true, 1", "spmm_max_bwd_f256"),
                       ("spmm_cluster_unit_kernel<false, 1", "spmm_max_fwd_f256"),
                       ("spmm_cluster_unit_kernel<true, 1", "spmm_max_bwd_f256"),
                       # GATConv at D = 256 (C3): the plain kernels, or the clustered form = weight pass + streaming kernel
                       # (both are summed into the label: the bench brackets the whole library call)
                       ("gat_fwd_kernel<4, 64>", "gat_fwd"), ("gat_bwd_edge_kernel<4, 64>", "gat_bwd_edge"),
                       ("gat_bwd_src_kernel<4, 64>", "gat_bwd_src"),
                       ("gat_cluster_stream_kernel<0,", "gat_fwd"), ("gat_weights_one_chunk_kernel<false>", "gat_fwd"),
                       ("gat_cluster_stream_kernel<1,", "gat_bwd_src"), ("gat_weights_one_chunk_kernel<true>", "gat_bwd_src"),
                       ("gat_cluster_stream_kernel<2,", "gat_bwd_edge"), ("gat_edge_finish_kernel", "gat_bwd_edge")):
        if key in name:
            return label
    return None


def main(fetch_dir, write_dir, out):
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    result = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py; "
                        "bytes = 2*FETCH_SIZE*1024 (gfx950 half-count correction) + WRITE_SIZE*1024"}
    for name in sorted(fetch):
        label = short(name)
        if label is None or name not in write:
            continue
        f = sum(fetch[name]) / len(fetch[name])
        w = sum(write[name]) / len(write[name])
        # several kernels of one library call (a pass + the main kernel) add up, launch for launch
        result[label + "_fetch_kib_raw"] = result.get(label + "_fetch_kib_raw", 0.0) + f
        result[label + "_write_kib_raw"] = result.get(label + "_write_kib_raw", 0.0) + w
        result[label + "_launches"] = len(fetch[name])
        result[label + "_bytes_per_launch"] = result.get(label + "_bytes_per_launch", 0) + int(2 * f * 1024 + w * 1024)
        result.setdefault(label + "_kernels", []).append(name.split("(")[0][-60:])
    json.dump(result, open(out, "w"), indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
