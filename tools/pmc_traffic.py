#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --output-format csv) of
`VIT_AMD_STREAMS=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernels` into the per-launch
averages bench.py reads (profiles/r02d_traffic.json).

  python tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/r02d_traffic.json

Only library kernels are kept, grouped by the bench's kernel names; launches of the same kernel on other shapes
(the pooled-row LayerNorm) are dropped by their grid size."""
import csv, glob, json, os, sys

GROUPS = [  # bench name, substring of the kernel name
    ("gemm_fc1_gelu", "gemm8p_persist_kernel<1, 0, true"),
    ("gemm_qkv", "gemm8p_persist_kernel<0, 0, true"),
    ("gemm_o_proj_res+gemm_fc2_res", "gemm8p_persist_kernel<3, 0, false"),
    ("patch_embed", "gemm8p_persist_kernel<100, 0, false"),
    ("attention", "attention_dh64_kernel"),
    ("layernorm_stats", "ln_stats_kernel"),
    ("layernorm", "layernorm_kernel"),
]


def per_kernel(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dirname}")
    acc = {}
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"]
        for key, sub in GROUPS:
            if sub in name:
                if key == "layernorm" and int(row["Grid_Size"]) < 256 * 256:
                    break  # pooled rows only
                a = acc.setdefault(key, [name, 0.0, 0])
                a[1] += float(row["Counter_Value"])
                a[2] += 1
                break
    return {k: (v[0], v[1] / v[2], v[2]) for k, v in acc.items()}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), VIT_AMD_STREAMS=1 python3 bench.py "
                  "--steps 2 --warmup 1 --no-cpu-baseline --no-kernels (single stream: full-batch launches), MI355X; "
                  "summarised by tools/pmc_traffic.py",
        "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts half the bytes of wide coalesced reads on gfx950, "
                      "so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; Infinity-Cache hits are included, so this is traffic "
                      "beyond L2, an upper bound on HBM bytes",
        "per_launch_kb": {},
    }
    for key, _ in GROUPS:
        if key in fetch and key in write:
            out["per_launch_kb"][key] = {"kernel": fetch[key][0][:120], "launches": fetch[key][2],
                                         "FETCH_SIZE": round(fetch[key][1], 1), "WRITE_SIZE": round(write[key][1], 1)}
    json.dump(out, sys.stdout, indent=2)
    print()


if __name__ == "__main__":
    main()
