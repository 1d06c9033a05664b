#!/usr/bin/env python3
"""Keeps the library's own kernels of a rocprofv3 `*_kernel_stats.csv` (drops torch's synthetic-data / copy kernels):
  python tools/prof_filter.py gpurun_out/prof_r02a/s1/vm/399_kernel_stats.csv > profiles/r02a_bench_single_stream_kernel_stats.csv"""
import csv, sys

DROP = ("at::native", "at::cuda", "rocclr", "rocprim", "hipcub", "__amd_")
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(rows[0])
for r in rows[1:]:
    if r and not any(d in r[0] for d in DROP):
        w.writerow(r)
