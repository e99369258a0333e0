import os, sys, json
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda:0")
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = int(sys.argv[2]) if len(sys.argv) > 2 else 32
print(json.dumps(bench.roofline_cov_apply(dev, m=m, iters=200, nimg=nimg)))
