"""128->128, B=64, 64x64 fwd / gated dgrad / wgrad launches on preallocated buffers (for rocprofv3 --pmc passes)."""
import sys
sys.path.insert(0, '.')
sys.argv = [sys.argv[0], 'fwd,dgrad,wgrad', '3']
import tools.bench_kernels as BK
BK.case(64, 128, 0, 128, n=3)
