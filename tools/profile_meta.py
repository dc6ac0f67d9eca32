#!/usr/bin/env python3
"""Writes <dir>/kernel_source_sha16.json = the hash of the kernel sources the profiled library was built from (bench.kernel_source_sha16),
so that bench.py attaches a committed rocprofv3 summary only to the build it was taken on.  Usage: python tools/profile_meta.py <dir>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
d = sys.argv[1]
os.makedirs(d, exist_ok=True)
json.dump({"sha16": bench.kernel_source_sha16()}, open(os.path.join(d, "kernel_source_sha16.json"), "w"))
print(bench.kernel_source_sha16())
