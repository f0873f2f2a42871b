"""Replay the captured C2 distillation step a few times (for `rocprofv3 --kernel-trace`); tools/debug/trace_fold.py folds the CSV."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from cat_amd import _lib as L, synthetic
from cat_amd.graph import GraphedStep
L.load()
args = argparse.Namespace(workload='c2', batch=16, size=256, target_flops=4.6e9)
model, opt = bench.build_model(args, 0)
b = {'A': synthetic.images((16, 3, 256, 256), 1).cuda(), 'B': synthetic.images((16, 3, 256, 256), 2).cuda(), 'A_paths': [], 'B_paths': []}
g = GraphedStep(model, b)
for _ in range(int(os.environ.get('REPS', '3'))):
    g(b)
    torch.cuda.synchronize()
print('done')
