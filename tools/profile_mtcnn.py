#!/usr/bin/env python3
"""cProfile of MTCNN.detect on 16 synthetic 512x512 images (host logic vs kernels).  Usage: python tools/profile_mtcnn.py"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from faceposegenerator_amd import mtcnn as M
w = M.synth_weights(5)
det = M.MTCNN(select_largest=True, post_process=False, device="cuda:0", weights=w)
base = torch.rand(16, 3, 64, 64, generator=torch.Generator().manual_seed(3))
imgs = (F.interpolate(base, size=(512, 512), mode="bilinear") * 255).permute(0, 2, 3, 1).to(torch.uint8).contiguous().to("cuda:0")
det.detect(imgs, landmarks=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
det.detect(imgs, landmarks=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
