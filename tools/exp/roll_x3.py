import os, sys, torch
sys.path.insert(0, '.')
import bench
from scm_gan_amd import ops
dev = torch.device('cuda:0')
print('native', bench.rollout_l1(dev))
ops.set_compute_precision('fp32x3')
os.environ['SCM_X3_ANY_BATCH'] = '1'
print('x3', bench.rollout_l1(dev))
