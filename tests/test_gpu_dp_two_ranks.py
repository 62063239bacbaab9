"""GPU (-m gpu): data parallelism on the real HIP path, two ranks on ONE GPU.

tests/test_dp_gloo.py drives Trainer.step's host logic on CPU with the networks stubbed; here two fresh processes
(tests/dp_gpu_worker.py) run the real thing -- HIP kernels, the autograd hook on the first latent, the communication
stream, the flat gradient buckets -- on half of a fixed batch each, all-reducing the CUDA buckets over gloo (RCCL does
not accept two ranks on one device).  Their reduced gradients and post-Adam weights must equal one process with the
full batch; the two replicas must stay bit-identical; the early buckets must have been issued from the hook."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("case", ["default", "gan", "no_overlap"])
def test_two_ranks_on_the_hip_path_equal_the_full_batch(case, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_gpu_worker as WK
    world, port = 2, str(_port())
    outs = [str(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_gpu_worker.py"), str(r), str(world), port, case, outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    ref_g, ref_w, fired, enabled = WK.run_steps(case, slice(0, WK.B))          # this process, the full batch
    assert not enabled and fired == 0
    res = [np.load(o) for o in outs]
    for r in res:
        assert int(r["enabled"]) == 1
        assert int(r["fired"]) == (0 if case == "no_overlap" else 2)
        g, w = torch.from_numpy(r["g"]), torch.from_numpy(r["w"])
        assert torch.allclose(g, ref_g, atol=2e-6, rtol=1e-4), (case, float((g - ref_g).abs().max()))
        # Adam moves a weight by <= lr = 1e-4 per step whatever |g|: a ~0 gradient may flip sign
        err = (w - ref_w).abs()
        assert float(err.max()) <= 4.1e-4 and float((err < 2e-6).float().mean()) > 0.995, (case, float(err.max()))
    assert np.array_equal(res[0]["w"], res[1]["w"])                             # replicas stay bit-identical
