"""profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command.

Usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Per kernel and launch: hbm bytes = 2 * FETCH_SIZE_KB * 1024 (gfx950 tallies 128-B read requests at 64 B, see
MI355X_MICROARCH.md "HBM") + WRITE_SIZE_KB * 1024.  Warm-up launches (first half) are dropped."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

SHORT = [  # kernel-name fragment -> name used by bench.py's roofline table
    ("select4_kernel", "select4"), ("s4_regions_kernel", "select_regions"), ("select3_run_kernel", "select3_run"), ("select3_plan_kernel", "select_plan"),
    ("select_run_kernel", "select_run_general"), ("select_plan_kernel", "select_plan"), ("select_export", "select_export"),
    ("pair_flip_kernel", "pair_attention_fused"), ("pair_fused_kernel", "pair_attention_fused_mfma"),
    ("pair_rows_kernel", "pair_attention_rows"),
    ("tail_chain_kernel", "tail_chain"),   # (rows / merge form: told apart below by the last template argument)
    ("dense_chain_kernel<8, 0, 1, 1", "dense_chain_mlp_hidden"), ("dense_chain_kernel<4, 0, 1, 1", "dense_chain_mlp_hidden"),
    ("dense_chain_kernel<16, 0, 1, 1", "dense_chain_mlp_hidden"), ("dense_chain_kernel<2, 0, 1, 1", "dense_chain_mlp_hidden"),
    ("pair_scores_", "pair_scores"), ("pair_softmax_gather_heavy", "pair_softmax_gather_heavy"),
    ("pair_softmax_gather_kernel", "pair_softmax_gather_light"),
    ("gcn_fused_kernel", "gcn_layer_fused"), ("spmm_row_parts", "spmm_row_parts"),
    ("spmm_csr_kernel", "spmm_csr"), ("spmm_long_rows", "spmm_long_rows"),
    ("gemm_f32_kernel<128>", "gemm128"), ("gemm_f32_kernel<64>", "gemm64"), ("layernorm", "layernorm"),
    ("dense_chain_kernel<8, 8", "dense_chain_elementwise"), ("dense_chain_kernel<9, 8", "dense_chain_pairwise"),
    ("pair_gather_kernel", "pair_gather_q"), ("dense_chain_kernel<8, 0", "dense_chain_attn_out"),
    ("dense_chain_kernel<16, 0", "dense_chain_score"),
]


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for frag, short in SHORT:
                if frag in r["Kernel_Name"]:
                    if short == "tail_chain" and re.search(r"tail_chain_kernel<[^>]*true>", r["Kernel_Name"]):
                        short = "tail_chain_rows"   # (the rows form: last template argument)
                    acc[short].append(float(r["Counter_Value"]))
                    break
    return {k: v[len(v) // 2:] for k, v in acc.items()}


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on the " +
                   (sys.argv[5] if len(sys.argv) > 5 else "collab") + "-like bench "
                   "workload, serial single-stream steps; bytes = 2 x FETCH_SIZE_KB x 1024 (gfx950 correction, "
                   "MI355X_MICROARCH.md) + WRITE_SIZE_KB x 1024, mean per launch after warm-up",
           "commit": sys.argv[4] if len(sys.argv) > 4 else "?", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        fk = sum(f) / len(f) if f else 0.0
        wk = sum(w) / len(w) if w else 0.0
        out["kernels"][k] = {"FETCH_SIZE_KB_per_launch_raw": round(fk, 1), "launches_profiled_FETCH_SIZE": len(f),
                             "WRITE_SIZE_KB_per_launch_raw": round(wk, 1), "launches_profiled_WRITE_SIZE": len(w),
                             "hbm_bytes_per_launch_corrected": int(2 * fk * 1024 + wk * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:28s} {v['hbm_bytes_per_launch_corrected'] / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
    main()
