import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from bench import roofline_cov_apply
dev=torch.device('cuda:0')
for m in (2,8,16,32,56,128):
    print(m, roofline_cov_apply(dev, m=m, iters=100))
