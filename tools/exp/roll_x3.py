"""The eval rollout of bench.py (rollout_l1: per-pixel L1 and latent flips against the oracle) in the native and in the fp32x3
mode; under rocprofv3 it also shows which kernels the 2-sample rollout runs (the 128-pixel forms)."""
import os, sys, torch
sys.path.insert(0, '.')
import bench
from scm_gan_amd import ops
dev = torch.device('cuda:0')
print('native', bench.rollout_l1(dev))
ops.set_compute_precision('fp32x3')
os.environ['SCM_X3_ANY_BATCH'] = '1'
print('x3', bench.rollout_l1(dev))
