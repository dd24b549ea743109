#!/usr/bin/env python3
"""bench.trainer_replay_record on its own: Trainer(graph=True) over freshly shuffled resident batches."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2-gcn-512x84-h64")
ap.add_argument("--steps", type=int, default=64)
a = ap.parse_args()
print(json.dumps(bench.trainer_replay_record(a, a.workload, torch.device("cuda", 0), a.workload)))
