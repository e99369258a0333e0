import os, sys, torch, json
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from bench import roofline_conv_mfma
print(json.dumps(roofline_conv_mfma(torch.device('cuda:0'), iters=10)))
