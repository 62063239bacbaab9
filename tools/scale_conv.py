import sys, time, torch
sys.path.insert(0, '.')
from scm_gan_amd import ops
dev = 'cuda'
def run(B, cin, cout=128, H=64, W=64, act=1, n=8):
    x = torch.randn(B, cin, H, W, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.zeros(cout, device=dev)
    fp, dp = ops.pack_conv3x3(w, cin, cout)[:2]
    for _ in range(3): ops._conv3x3_raw(x, None, None, None, 0, fp, cout, None, b, None, cout, 1, act)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): ops._conv3x3_raw(x, None, None, None, 0, fp, cout, None, b, None, cout, 1, act)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    fl = 2.0 * B * H * W * cout * cin * 9
    print('B %4d cin %4d cout %3d: %.3f ms  %.1f TF' % (B, cin, cout, dt * 1e3, fl / dt / 1e12)); sys.stdout.flush()
for B in (16, 32, 64, 128, 256): run(B, 128)
for cin in (8, 16, 32, 64, 128, 256, 512): run(32, cin)
for cin in (8, 64, 128): run(64, cin)
print('--- other widths'); sys.stdout.flush()
run(16, 128, H=128, W=128)
run(36, 128, H=84, W=84)
