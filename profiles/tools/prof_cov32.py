import os, sys, torch
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from bench import roofline_cov_apply
print(roofline_cov_apply(torch.device('cuda:0'), m=32, iters=50))
