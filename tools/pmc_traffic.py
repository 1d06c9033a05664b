#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --output-format csv) of
`VIT_AMD_STREAMS=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernels` into the per-launch
averages bench.py reads (the newest profiles/r*_traffic.json).

  python tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/r03b_traffic.json

Only library kernels are kept, grouped by the bench's kernel names; launches of the same kernel on other shapes
(the pooled-row LayerNorm) are dropped by their grid size. The figures are per operator CALL: the few-tile planner
may run one GEMM as two launches (a head of 256-row tiles, then a tail of shorter tiles: two template instances of
the same kernel), whose counters are added; `launches_per_call` says how many."""
import csv, glob, json, os, sys

GROUPS = [  # bench name, substring of the kernel name
    ("gemm_fc1_gelu", "gemm8p_persist_kernel<1, 0, true"),
    ("gemm_qkv", "gemm8p_persist_kernel<0, 0, true"),
    ("gemm_o_proj_res+gemm_fc2_res", "gemm8p_persist_kernel<3, 0, false"),
    ("patch_embed", "gemm8p_persist_kernel<100, 0, false"),
    ("attention", "attention_dh64_kernel"),
    ("layernorm_stats", "row_stats_kernel"),
    ("row_stats_finalize", "row_stats_finalize_kernel"),
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
                a = acc.setdefault(key, [name, 0.0, {}])
                a[1] += float(row["Counter_Value"])
                a[2][name] = a[2].get(name, 0) + 1
                break
    # calls = launches of the most frequent template instance (every call launches it once; a planned call adds a
    # second instance for its tail)
    return {k: (v[0], v[1] / max(v[2].values()), max(v[2].values()), sum(v[2].values())) for k, v in acc.items()}


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
            out["per_launch_kb"][key] = {"kernel": fetch[key][0][:120], "calls": fetch[key][2],
                                         "launches_per_call": round(fetch[key][3] / fetch[key][2], 2),
                                         "FETCH_SIZE": round(fetch[key][1], 1), "WRITE_SIZE": round(write[key][1], 1)}
    json.dump(out, sys.stdout, indent=2)
    print()


if __name__ == "__main__":
    main()
