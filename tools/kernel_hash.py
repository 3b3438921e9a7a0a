#!/usr/bin/env python3
"""SHA-256 over the sources of the SpMV kernels: what a stored PMC traffic number (profiles/traffic_latest.json) was taken on. bench.py refuses a stored number
whose hash differs from the tree it runs from. Usage: python tools/kernel_hash.py"""
import hashlib
import os

FILES = ("spmv.hip", "spmv_pb.hip", "spmv_bcsr.hip", "spmv_pb.hpp", "spmv_bcsr.hpp", "common.hpp")


def spmv_kernel_hash(root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(root, "g4s_amd", "csrc", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(spmv_kernel_hash())
