import os, sys, json, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from bench import roofline_dense_cov_apply
dev=torch.device('cuda:0')
for d in (4096, 12288):
    print(json.dumps(roofline_dense_cov_apply(dev, d=d)))
