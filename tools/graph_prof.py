#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shacira_amd import harness
r = harness.fit_image(torch.device("cuda:0"), steps=60, graphed=False)
print(r["ms_per_step"])
