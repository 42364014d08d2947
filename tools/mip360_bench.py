#!/usr/bin/env python
"""MipNeRF-360 (BASELINE config 5) training-step throughput on one MI355X (mip360.benchmark_step).

    python tools/mip360_bench.py [--rays 4096] [--steps 10] [--warmup 3] [--forward_only]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from outdoor_nerf_depth_amd import mip360 as M                                  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--rays', type=int, default=4096)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--forward_only', action='store_true')
    a = p.parse_args()
    print(json.dumps(M.benchmark_step(torch.device('cuda:0'), a.rays, a.steps, a.warmup, a.forward_only)))


if __name__ == '__main__':
    main()
