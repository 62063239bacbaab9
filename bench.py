#!/usr/bin/env python3
"""Headline benchmark: predicted frames/s of the action-conditional
frame-prediction TRAINING iteration (encoder -> (reward head, decoder + BCE,
transition) x (horizon-2) -> backward -> clip -> Adam; reference main.py:143-297)
on synthetic minipacman-shaped rollouts (64x64x3, horizon 10, A=5, R=2).

  python bench.py --gpus N --steps K --warmup W [--config 1|2]

N > 1: one rank per GPU, RCCL all-reduce of the flat gradient buckets.  Either started by
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (the ranks find
RANK / LOCAL_RANK / WORLD_SIZE in the environment), or as a plain `python bench.py --gpus N`,
which starts those N ranks itself (fresh child processes, before this process touches a GPU).

Per-GPU batch is fixed at 64 (BASELINE.json configs[1]; 256 + the discriminator pass for
--config 2 = configs[2]), so scaling is weak and `value` counts the frames of all ranks.
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16/fp16 MFMA peak (opt-in --precision fp16 only)
H = W = 64
C, A, R, L = 3, 5, 2, 16      # BASELINE configs[1]; --size/--channels/--actions/--rewards select the other shapes


ENC_GFLOP, TRANS_GFLOP, DEC_GFLOP, RP_GFLOP = 2.652, 6.540, 0.302, 0.0385     # forward, per sample


def algorithmic_gflop_per_frame(horizon):
    """SURVEY.md section 8d's formula: 2*MAC, train = 3x forward for every unrolled step, per predicted
    frame (64x64, C=3, A=5, R=2; scaled by the pixel count for other sizes -- the C/A/R dependence is < 1 %)."""
    step = TRANS_GFLOP + DEC_GFLOP + RP_GFLOP
    return 3.0 * (ENC_GFLOP + (horizon - 2) * step) / (horizon - 2) * (H * W) / 4096.0


def executed_gflop_per_frame(horizon):
    """What the iteration really executes (reference and this build alike): the output of the LAST transition
    (t = horizon-2, main.py:207) is never consumed, so that call runs forward only -- (horizon-3) transitions
    at 3x forward + one at 1x."""
    n = horizon - 2
    total = 3.0 * (ENC_GFLOP + n * (DEC_GFLOP + RP_GFLOP) + (n - 1) * TRANS_GFLOP) + TRANS_GFLOP
    return total / n * (H * W) / 4096.0


def build_nets(seed):
    from scm_gan_amd import models
    torch.manual_seed(seed)
    enc = models.Encoder(L, C)
    dec = models.Decoder(L, C)
    rp = models.RewardPredictor(L, R)
    disc = models.Discriminator()
    tr = models.Transition(L, A)
    return enc, dec, rp, disc, tr


def make_batch(ds, batch, horizon, dev):
    from scm_gan_amd import main as M
    states, rewards, dones, actions = ds.get_trajectories(batch, horizon)
    return (torch.as_tensor(states, dtype=torch.float32).to(dev), torch.as_tensor(rewards, dtype=torch.float32).to(dev),
            torch.as_tensor(dones.astype(np.float32)).to(dev), M.onehot(actions, A))


def host_cores():
    """CPU share of this process: affinity mask, capped by the cgroup quota (the
    GPU box exposes every host core but grants 16 per GPU)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("SCM_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(seconds_budget=10.0):
    """Reference CPU path (the oracle: same arithmetic as the reference's torch-CPU
    modules, pinned to them by tests/test_oracle_golden.py) on BASELINE configs[0]:
    64x64, batch 8, horizon 5, timed on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import weights as WS
    from oracle import scm_oracle as O
    from scm_gan_amd.datasource import SyntheticDatasource
    cores = host_cores()
    torch.set_num_threads(cores)
    B, h = 8, 5
    enc, tr, dec, rp = (O.clone_state(s) for s in (WS.encoder_state(0), WS.transition_state(1), WS.decoder_state(2),
                                                   WS.reward_state(3)))
    trainer = O.Trainer(enc, tr, dec, rp, A)
    ds = SyntheticDatasource(seed=0, height=H, width=W, channels=C, num_actions=A, num_rewards=R)
    rng = np.random.default_rng(0)

    def one():
        st, rw, dn, ac = ds.get_trajectories(B, h)
        noises = [torch.from_numpy(rng.random((B, L, H, W), dtype=np.float32)) for _ in range(h - 2)]
        trainer.step(st, rw, dn, ac, 0.5, noises)
    one()
    t0 = time.time()
    n = 0
    while n < 1 or (time.time() - t0 < seconds_budget and n < 50):
        one()
        n += 1
        sys.stderr.write("[bench] cpu_baseline iteration %d, %.1f s\n" % (n, time.time() - t0))
        sys.stderr.flush()
    dt = time.time() - t0
    return {"value": B * (h - 2) * n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d training iterations of configs[0] (64x64x3, batch 8, horizon 5), %.1f s, torch %s CPU ops "
                      "via oracle/scm_oracle.py" % (n, dt, torch.__version__)}


def rollout_l1(dev):
    """Mean per-pixel |sigmoid(decoder(z_t))_hip - _oracle| over a free-running
    eval rollout (main.py:798-835 pattern), B=2, 64x64, 10 steps, same weights."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import weights as WS
    from oracle import scm_oracle as O
    from scm_gan_amd import models, main as M
    from scm_gan_amd.datasource import SyntheticDatasource
    ds = SyntheticDatasource(seed=11, height=H, width=W, channels=C, num_actions=A, num_rewards=R)
    states, rewards, dones, actions = ds.get_trajectories(2, 12, training=False)
    enc, tr, dec = models.Encoder(L, C), models.Transition(L, A), models.Decoder(L, C)
    enc.load_state_dict(WS.to_torch(WS.encoder_state(0)))
    tr.load_state_dict(WS.to_torch(WS.transition_state(1)))
    dec.load_state_dict(WS.to_torch(WS.decoder_state(2)))
    M.test_mode([enc, tr, dec])
    oenc, otr, odec = (O.clone_state(s) for s in (WS.encoder_state(0), WS.transition_state(1), WS.decoder_state(2)))
    orp = O.clone_state(WS.reward_state(3))
    ref = O.measure_prediction_mse(oenc, otr, odec, orp, states, rewards, dones, actions, A, return_frames=True)
    with torch.no_grad():
        st = torch.as_tensor(states, dtype=torch.float32).to(dev)
        oh = M.onehot(actions, A)
        z = enc(st[:, :3])
        z = tr(z, oh[:, 1])
        l1, flips = [], 0
        for i, t in enumerate(range(2, 12)):
            pred = torch.sigmoid(dec(z)).cpu()
            l1.append(float((pred - ref["frames"][i]).abs().mean()))
            flips += int((z.cpu() != ref["z"][i]).sum())
            z = tr(z, oh[:, t])
    return float(np.mean(l1)), flips


def fp32x3_leg(a, opts, batches, dev, headline):
    """The timed iteration once more with the 3x3 convolutions in the split mode: throughput, and for the
    headline workload the eval rollout against the oracle in that mode."""
    from scm_gan_amd import main as M, ops
    ops.set_compute_precision("fp32x3")
    enc2, dec2, rp2, disc2, tr2 = build_nets(0)
    tr2.seed_noise(1234)
    trainer2 = M.Trainer(L, A, R, enc2, dec2, rp2, disc2, tr2, opts)
    for i in range(3):
        trainer2.step(*batches[i % 2], theta=0.5, train_iter=i + 1)
    torch.cuda.synchronize()
    n2 = 10
    t2 = time.perf_counter()
    for i in range(n2):
        trainer2.step(*batches[i % 2], theta=0.5, train_iter=i + 4)
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t2
    x3 = {"value": a.batch * (a.horizon - 2) * n2 / dt2, "unit": "frames/s", "ms_per_step": dt2 / n2 * 1e3,
          "steps": n2, "opt_in": "bench.py --precision fp32x3 / main.py --precision fp32x3",
          "arithmetic": "fp32 operands split exactly into 3 bf16 values, 6 bf16 MFMA products per term, "
                        "fp32 accumulate; error vs float64 within 2.5x of the native f32 MFMA kernels'"}
    del trainer2, enc2, dec2, rp2, disc2, tr2
    if headline:
        # the north star's own criterion in this mode: the eval rollout against the oracle (per-pixel L1 <= 1e-4,
        # binary latents).  SCM_X3_ANY_BATCH sends this 2-sample rollout through the split kernels, which the
        # product would route to the native small tiles.
        os.environ["SCM_X3_ANY_BATCH"] = "1"
        try:
            x3["rollout_l1_vs_ref"], x3["rollout_latent_flips"] = rollout_l1(dev)
        finally:
            del os.environ["SCM_X3_ANY_BATCH"]
    return x3


def decoder_leg(dev, batch):
    """Decoder forward + pixel loss in isolation (north star: achieved HBM GB/s on the decoder): ONE fused launch
    (scm_decoder_fused_fwd; the 64-channel hidden map never leaves the chip) + the per-sample loss reduce.
    Algorithmic bytes per frame = read z (16 planes) + write C logit planes (311 296 B at
    64x64, C=3), the fused minimum; the BCE also reads the C target planes (49 152 B).  `counter_*` = HBM bytes of the
    fused launch from the rocprofv3 --pmc passes under profiles/ (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md),
    null when the kernel source changed since.  The path is NOT HBM-bound: ~290 FLOP per algorithmic byte against a
    machine balance of ~20 FLOP/B, so the fp32 MFMA rate bounds it."""
    from scm_gan_amd import models
    torch.manual_seed(0)
    dec = models.Decoder(L, C)
    z = (torch.rand(batch, L, H, W, device=dev) < 0.5).float()
    target = (torch.rand(batch, C, H, W, device=dev) < 0.1).float()
    with torch.no_grad():
        for _ in range(3):
            dec.forward_loss(z, target)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        s.record()
        for _ in range(n):
            dec.forward_loss(z, target)
        e.record()
        torch.cuda.synchronize()
    sec = s.elapsed_time(e) * 1e-3 / n
    bytes_per_frame = (L + C) * H * W * 4
    flops_per_frame = 2.0 * H * W * 9 * (L * 4 * L + 4 * L * C)        # conv2 with the factor sum folded into its weights
    out = {"fused": bool(dec._fused()), "frames_per_s": batch / sec, "us_per_launch_group": sec * 1e6,
           "algorithmic_bytes_per_frame": bytes_per_frame,
           "achieved_GBps_algorithmic": batch * bytes_per_frame / sec / 1e9, "peak_GBps": 8000.0,
           "tflops_executed_formulation": batch * flops_per_frame / sec / 1e12,
           "bound": "mfma (fp32), not hbm"}
    pmc = measured_pmc("decoder_fused_fwd_kernel", "decoder_fused.hip")
    if pmc is not None and (H, W, C, batch) == (64, 64, 3, 64):
        out.update({"counter_hbm_bytes_per_launch": pmc["hbm_traffic_bytes"],
                    "counter_hbm_bytes_per_frame": pmc["hbm_traffic_bytes"] / 64.0,
                    "counter_to_algorithmic_ratio": pmc["hbm_traffic_bytes"] / 64.0 / bytes_per_frame,
                    "counter_GBps": pmc["hbm_traffic_bytes"] / (pmc["duration_us_median"] * 1e-6) / 1e9,
                    "counter_mfma_utilisation": pmc.get("mfma_utilisation"), "counter_source": pmc["_file"]})
    else:
        out["counter_hbm_bytes_per_launch"] = None
    return out


def git_commit():
    """HEAD of this tree: SCM_GIT_COMMIT, `git rev-parse`, or build/git_commit (written by __graft_entry__.build();
    the GPU box receives no .git)."""
    c = os.environ.get("SCM_GIT_COMMIT")
    if c:
        return c
    try:
        c = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except Exception:
        c = ""
    if c:
        return c
    try:
        return open(os.path.join(ROOT, "build", "git_commit")).read().strip() or None
    except OSError:
        return None


def _sha16(path):
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def measured_pmc(kernel_substr, source_file, need="hbm_traffic_bytes"):
    """Newest committed PMC record (profiles/round*_pmc*.json) of a kernel whose source file is unchanged."""
    import glob
    sha = _sha16(os.path.join(ROOT, "scm_gan_amd", "csrc", source_file))
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc*.json")), reverse=True):
        try:
            pmc = json.load(open(f))
        except Exception:
            continue
        if pmc.get("kernel_source_sha16", {}).get(source_file) != sha:
            continue
        for k, v in pmc.get("kernels", {}).items():
            if kernel_substr in k and need in v:
                return dict(v, _file=os.path.basename(f), _kernel=k, _commit=pmc.get("git_commit"))
    return None


HALF_STORE = False          # set in main(): the 16-bit mode with half-stored activations (its forward is the XH = 1 instantiation)
WINO_T2 = False             # set in main(): the wide Winograd launches run conv3x3_wino_t2.hip (scm_wino_t2_serves' conditions)


def measured_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC pass
    (profiles/round*_pmc*.json, written by tools/pmc_parse.py).  The number is only valid for
    the kernel it was measured on: each file records the sha of the kernel's source file, and a
    file whose sha no longer matches the tree is skipped -> traffic null rather than stale."""
    import glob
    kind = "h16" if "h16" in kernel_name else ("x3" if "x3_kernel" in kernel_name else
                                               ("wino" if "wino" in kernel_name else "mfma"))
    # (the wide Winograd launches of 64-wide frames run the two-tiles-per-workgroup form, conv3x3_wino_t2.hip; a PMC file
    # from before it existed -- or SCM_WINO_T2=0 -- holds the one-tile kernel's row)
    cands = {"h16": [("conv3x3_h16.hip", "conv3x3_h16_kernel<1, 4, 2, 2, 0, 2, 1>" if HALF_STORE else "conv3x3_h16_kernel<1,")],
             "x3": [("conv3x3_x3.hip", "conv3x3_x3_kernel<2, 2, 0,")], "mfma": [("conv3x3_mfma.hip", "conv3x3_mfma_kernel<4, 2, 2, 0,")],
             "wino": ([("conv3x3_wino_t2.hip", "conv3x3_wino_t2_kernel<1>")] if WINO_T2 else [])
                     + [("conv3x3_wino.hip", "conv3x3_wino_kernel<1, 0, 1, 128>")]}[kind]
    for src, want in cands:
        sha = _sha16(os.path.join(ROOT, "scm_gan_amd", "csrc", src))
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc*.json")), reverse=True):
            try:
                pmc = json.load(open(f))
            except Exception:
                continue
            if pmc.get("kernel_source_sha16", {}).get(src) != sha:
                continue
            for k, v in pmc.get("kernels", {}).items():
                if want in k and "hbm_traffic_bytes" in v:
                    return {"bytes": v["hbm_traffic_bytes"], "kernel": k, "file": os.path.basename(f),
                            "kernel_source_sha16": sha, "measured_at_commit": pmc.get("git_commit"),
                            "mfma_busy": v.get("mfma_utilisation"), "duration_us_median": v.get("duration_us_median")}
        if cands[0][0] == "conv3x3_wino_t2.hip" and src == "conv3x3_wino_t2.hip":
            return None         # (the two-tile kernel is the one that ran: do not cite the one-tile kernel's counters for it)
    return None


def encoder_leg(dev, batch):
    """North star: "MFMA utilisation on the encoder convs".  The encoder's forward (its four launches: conv1 9 -> 128,
    conv2 / conv3 128 -> 128, conv4 128 -> 16, zero padding) in isolation, every launch bracketed by HIP events on its
    stream.  Per launch: microseconds, algorithmic TFLOP/s (2 x 9 MACs per output and input channel) and ISSUED
    TFLOP/s on the f32 MFMA pipe (Winograd F(2x2,3x3) launches issue 4/9 of the products) as a fraction of the
    pipe's peak; `mfma_busy_pmc` = SQ_VALU_MFMA_BUSY_CYCLES share of the same instantiations from the newest
    rocprofv3 --pmc pass under profiles/ whose kernel sources are this tree's (null otherwise)."""
    from scm_gan_amd import models, ops
    torch.manual_seed(0)
    enc = models.Encoder(L, C)
    x = (torch.rand(batch, 3, C, H, W, device=dev) < 0.1).float()
    reps = 10
    with torch.no_grad():
        for _ in range(3):
            enc(x)
        timer = ops.KernelTimer()
        ops.kernel_timer = timer
        try:
            for _ in range(reps):
                enc(x)
        finally:
            ops.kernel_timer = None
    torch.cuda.synchronize()
    per = len(timer.records) // reps
    names = ["conv1", "conv2", "conv3", "conv4"] if per == 4 else ["launch%d" % i for i in range(per)]
    peak = PEAK_F32_MFMA_TFLOPS if ops.compute_precision() == "fp32" else PEAK_F16_MFMA_TFLOPS
    layers, tot_us, tot_issued, tot_alg = [], 0.0, 0.0, 0.0
    pmc_names = {"conv1": ("conv3x3_wino_k16_kernel<1, 0, 0", "conv3x3_wino_k16.hip"),
                 "conv2": ("conv3x3_wino_t2_kernel<0>", "conv3x3_wino_t2.hip"),
                 "conv3": ("conv3x3_wino_t2_kernel<0>", "conv3x3_wino_t2.hip"),
                 "conv4": ("conv3x3_wino16_kernel<2, 0", "conv3x3_wino16.hip")}
    for i, nm in enumerate(names):
        recs = timer.records[i::per]
        us = sum(s.elapsed_time(e) for _, _, s, e in recs) * 1e3 / len(recs)
        tag, flops = recs[0][0], recs[0][1]
        issued = flops * (4.0 / 9.0 if "wino" in tag else 1.0)
        e = {"layer": nm, "kernel": tag, "us": us, "algorithmic_tflops": flops / us / 1e6,
             "issued_tflops": issued / us / 1e6, "issued_frac_of_peak": issued / us / 1e6 / peak, "mfma_busy_pmc": None}
        if nm in pmc_names and "wino" in tag and (H, W, batch) == (64, 64, 64):
            pmc = measured_pmc(*pmc_names[nm], need="mfma_utilisation")
            if pmc is not None:
                e["mfma_busy_pmc"] = pmc.get("mfma_utilisation")
                e["mfma_busy_pmc_source"] = pmc["_file"]
        layers.append(e)
        tot_us += us
        tot_issued += issued
        tot_alg += flops
    return {"launches": layers, "us_total": tot_us, "algorithmic_tflops": tot_alg / tot_us / 1e6,
            "issued_tflops": tot_issued / tot_us / 1e6, "issued_frac_of_peak": tot_issued / tot_us / 1e6 / peak,
            "peak_tflops": peak, "target": "north star: >= 0.50 MFMA utilisation on the encoder convs",
            "note": "issued = FLOPs the MFMA pipe executes (4/9 of the algorithmic ones in the Winograd launches)"}


def self_launch(gpus, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes
    (torch.distributed.run), BEFORE this process makes any GPU call (counting devices does not
    initialise the runtime), and exit with their status.  No exec, no fork of a GPU process."""
    have = torch.cuda.device_count()
    if have < gpus:
        raise SystemExit("bench.py --gpus %d needs %d GPUs on this node, found %d" % (gpus, gpus, have))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, choices=(1, 2),
                    help="BASELINE.json configs index: 1 = batch 64, generator only (the headline); 2 = batch 256 + the "
                         "spectral-normalised discriminator's forward / backward / Adam on (B,3,64,64) U{0,1} noise maps "
                         "(SURVEY.md 8(d) cfg3; the reference defines no GAN loss, the critic term is build-defined)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 64 for --config 1, 256 for --config 2)")
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--size", type=int, default=64, help="frame height = width (84: configs[3], 128: configs[4] shapes)")
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--actions", type=int, default=5)
    ap.add_argument("--rewards", type=int, default=2)
    ap.add_argument("--precision", choices=("fp32", "fp32x3", "fp16"), default="fp32",
                    help="MFMA operand precision of the 3x3 convs. fp32 (default) is the reference's arithmetic and the "
                         "only valid headline; fp16 = opt-in fp16 forward / bf16 gradient operands with fp32 accumulation "
                         "(BASELINE configs[4]), reported as a separate, non-headline line")
    ap.add_argument("--conv-algo", choices=("winograd", "direct"), default=None,
                    help="fp32 algorithm of the wide 3x3 convolutions (default: the library's, winograd = F(2x2,3x3) on the "
                         "f32 MFMA pipe; direct = implicit GEMM): both are fp32 operands / fp32 accumulation")
    ap.add_argument("--graph-step", action="store_true",
                    help="replay the iteration as ONE HIP graph (scm_gan_amd.graph_step.GraphedTrainer: two eager iterations, "
                         "capture, then replays; single GPU, default loss): what the launch-bound small-batch shapes gain")
    ap.add_argument("--no-half-store", action="store_true",
                    help="with --precision fp16: keep fp32 tensors between the wide layers (default: the maps only the wide "
                         "16-bit kernels read are stored as IEEE half, ops.set_half_store)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--profile-mode", action="store_true",
                    help="only the warm-up + timed training steps (no rollout check, decoder leg, CPU baseline, event "
                         "brackets): the run to put under rocprofv3 so its per-kernel averages match the bench's")
    a = ap.parse_args()
    global H, W, C, A, R
    if a.batch is None:
        a.batch = 256 if a.config == 2 else 64
    H = W = a.size
    C, A, R = a.channels, a.actions, a.rewards
    shape_ok = (a.size, a.channels, a.actions, a.rewards, a.horizon, a.precision) == (64, 3, 5, 2, 10, "fp32")
    headline = shape_ok and a.config == 1 and a.batch == 64
    cfg2 = shape_ok and a.config == 2 and a.batch == 256
    if a.profile_mode:
        a.no_kernel_timer = a.no_cpu_baseline = True
    if a.graph_step:
        a.no_kernel_timer = True            # (event brackets cannot be recorded inside a captured graph)
        a.warmup = max(a.warmup, 4)         # two eager iterations + the capture + one replay before the timed region

    from scm_gan_amd import dp
    rank, local_rank, world, under_launcher = dp.env_world()
    if not under_launcher and a.gpus > 1:
        self_launch(a.gpus, sys.argv[1:])               # does not return
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d was started under a launcher with WORLD_SIZE=%d" % (a.gpus, world))
    # one process per GPU, RCCL (backend "nccl" on ROCm); also taken for a 1-rank launch under
    # torch.distributed.run so that path is exercised on a single-GPU box
    rank, local_rank, world = dp.init_from_env("nccl" if under_launcher else None)
    dev = torch.device("cuda", local_rank)

    from scm_gan_amd import main as M, ops
    from scm_gan_amd.datasource import SyntheticDatasource
    ops.set_compute_precision(a.precision)
    ops.set_half_store(not a.no_half_store)
    global HALF_STORE, WINO_T2
    HALF_STORE = a.precision == "fp16" and not a.no_half_store
    WINO_T2 = (os.environ.get("SCM_WINO_T2", "1") not in ("", "0") and (H, W) == (64, 64) and a.batch * (H // 4) >= 512
               and a.precision == "fp32" and a.conv_algo != "direct")
    if a.conv_algo:
        ops.set_conv_algo(a.conv_algo)
    opts = M.parser.parse_args(["--train-iters", str(a.steps + a.warmup), "--horizon-min", str(a.horizon),
                                "--horizon-max", str(a.horizon), "--batch-size", str(a.batch * world)]
                               + (["--gan"] if a.config == 2 else []))
    enc, dec, rp, disc, tr = build_nets(0)          # identical replicas on every rank
    tr.seed_noise(1234 + rank)                      # rank-offset Philox stream
    trainer = M.Trainer(L, A, R, enc, dec, rp, disc, tr, opts)
    ds = SyntheticDatasource(seed=rank, height=H, width=W, channels=C, num_actions=A, num_rewards=R)
    batches = [make_batch(ds, a.batch, a.horizon, dev) for _ in range(2)]    # resident in HBM

    def barrier():
        if world > 1 or under_launcher:
            import torch.distributed as dist
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    stepper = trainer
    if a.graph_step:
        from scm_gan_amd.graph_step import GraphedTrainer
        stepper = GraphedTrainer(trainer)
    for i in range(a.warmup):
        stepper.step(*batches[i % 2], theta=0.5, train_iter=i + 1)
    # HIP-event brackets around the MFMA conv launches (roofline): recorded on the launch stream inside the timed
    # region, on every 10th step only -- two events per launch stall the queue between kernels: on all steps they cost
    # ~8 % of the step, on every 5th 1.8 %, on every 10th 0.9 %
    timer = ops.KernelTimer() if (not a.no_kernel_timer and rank == 0) else None
    barrier()
    t0 = time.perf_counter()
    fired0 = trainer.sync.hook_fired
    for i in range(a.steps):
        ops.kernel_timer = timer if i % 10 == 0 else None
        trainer.sync.timing = trainer.sync.enabled and i % 10 == 5       # collective brackets: every 10th step too
        stepper.step(*batches[i % 2], theta=0.5, train_iter=a.warmup + i + 1)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0          # this rank alone, before the closing barrier
    barrier()
    dt = time.perf_counter() - t0
    ops.kernel_timer = None
    trainer.sync.timing = False
    dist_info = None
    if world > 1 or under_launcher:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # what the process group really looked like (for the day the 8-GPU run happens): every rank's device,
        # its own step time, whether its hook issued the early buckets, and the collectives' cost
        mine = torch.tensor([float(rank), float(torch.cuda.current_device()), dt_own / a.steps * 1e3,
                             float(trainer.sync.hook_fired - fired0)], device=dev, dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = [r.tolist() for r in allr]
        dist_info = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(),
                     "devices_seen": sorted(int(r[1]) for r in rows),
                     "rank_ms_per_step": {"min": min(r[2] for r in rows), "max": max(r[2] for r in rows),
                                          "per_rank": [round(r[2], 3) for r in sorted(rows)]},
                     "hook_fired": [int(r[3]) for r in sorted(rows)], "timed_steps": a.steps,
                     "grad_sync_enabled": bool(trainer.sync.enabled),
                     "allreduce_us": trainer.sync.timing_summary() if trainer.sync.enabled else None,
                     "allreduce_buckets": {"early": [f.name for f in trainer.sync.early],
                                           "late": [f.name for f in trainer.sync.late],
                                           "bytes": {f.name: 4 * f.flat.numel() for f in trainer.flats}}}

    if rank == 0:
        frames = a.batch * world * (a.horizon - 2) * a.steps
        value = frames / dt
        out = {
            "metric": "predicted frames/sec/node (minipacman 64x64, horizon=10) + rollout L1 vs ref",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32",
                      "fp32x3": "f32 emulated on the bf16 MFMA pipe (operands split exactly into 3 bf16, 6 products per term, "
                                "f32 accumulate; error vs float64 = the native f32 kernel's); weight gradients and thin layers "
                                "native f32",
                      "fp16": "f16 fwd / bf16 bwd MFMA operands, f32 accumulate + f32 tensors"}[a.precision],
            "data": "synthetic", "git_commit": git_commit(),
            "config": {"workload": "%s: %dx%dx%d frames, per-GPU batch %d, horizon %d, A=%d, R=%d, "
                                   "training iteration fwd+bwd+clip+Adam, reference loss (BCE + reward MSE), %s"
                                   % ("configs[1] minipacman-shaped" if headline else
                                      "configs[2] minipacman-shaped" if cfg2 else
                                      ("non-headline shape" if a.precision == "fp32" else
                                       "NON-HEADLINE opt-in 16-bit MFMA operands"),
                                      H, W, C, a.batch, a.horizon, A, R,
                                      "no discriminator" if a.config == 1 else
                                      "+ spectral-normalised discriminator fwd/bwd/Adam on (B,3,64,64) U{0,1} noise maps "
                                      "(the reference constructs it, models.py:195-223 / main.py:76,128, but defines NO "
                                      "GAN loss: the critic term, the mean logit, is build-defined)"),
                       "global_batch": a.batch * world, "horizon": a.horizon, "parallelism": "dp%d" % world,
                       "graph_step": bool(a.graph_step), "half_store": bool(a.precision == "fp16" and not a.no_half_store),
                       "grad_allreduce": ("none (1 rank)" if world == 1 and not under_launcher else
                                          "RCCL SUM of 4 flat fp32 buckets; decoder/transition/reward-head%s buckets issued "
                                          "from a hook on the first latent's gradient, under the encoder's backward"
                                          % ("/discriminator" if a.config == 2 else ""))},
            # two FLOP counts, both 2*MAC with no credit for padding or recompute:
            #   algorithmic = SURVEY.md 8(d)'s formula (3x forward for every unrolled step);
            #   executed    = the same minus the backward of the last transition, whose output nothing consumes
            #                 (reference main.py:207 at t = horizon-2) -- what the GPU really ran.
            "algorithmic_tflops": value * algorithmic_gflop_per_frame(a.horizon) / 1e3,
            "executed_tflops": value * executed_gflop_per_frame(a.horizon) / 1e3,
            "executed_frac_of_f32_mfma_peak": (value / world * executed_gflop_per_frame(a.horizon) / 1e3 / PEAK_F32_MFMA_TFLOPS
                                               if a.precision == "fp32" else None),
        }
        if dist_info is not None:
            out["distributed"] = dist_info
        if timer is not None:
            summ = timer.summary()
            dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
            name, s = dom
            achieved = s["flops"] / (s["ms"] * 1e-3) / 1e12
            # HBM bytes per launch: a PMC measurement (separate rocprofv3 --pmc passes, tools/prof_hot2.py) cited
            # from profiles/, valid only while the kernel's source is the one it was measured on (else null)
            tr_info = measured_traffic(name)
            # (the split kernels spend six bf16 MFMA products per fp32 product: their roof in fp32 FLOP is the bf16 peak / 6)
            peak = PEAK_F16_MFMA_TFLOPS if "h16" in name else (PEAK_F16_MFMA_TFLOPS / 6.0 if "x3_kernel" in name
                                                             else PEAK_F32_MFMA_TFLOPS)
            wino = "wino" in name
            # `achieved` = FLOPs the MFMA pipe EXECUTES per second in the dominant kernel, so `frac` is a fraction of the
            # pipe (<= 1).  For the direct kernels that is the algorithmic count (2 x 9 MACs per output and input
            # channel); Winograd F(2x2,3x3) issues 16 products per 2x2 output patch and channel pair instead of 36,
            # i.e. 4/9 of it -- the algorithmic (direct-convolution-equivalent) rate is reported beside it.
            issued = achieved * (4.0 / 9.0 if wino else 1.0)
            launch_b, launch_hw = a.batch, H * W
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": issued, "peak": peak,
                               "unit": "TFLOP/s", "frac": issued / peak,
                               "algorithmic_tflops": achieved, "algorithmic_over_issued": achieved / issued,
                               "traffic": tr_info["bytes"] if tr_info else None, "traffic_source": tr_info,
                               "launches": s["launches"], "avg_launch_us": s["ms"] * 1e3 / s["launches"],
                               # brackets are recorded on every 10th timed step: scale to the whole timed region
                               "share_of_step": s["ms"] / (dt * 1e3) * a.steps / len(range(0, a.steps, 10))}
            if tr_info and (H, W, a.batch) == (64, 64, 64):
                # the PMC case is the 128 -> 128 launch at B = 64, 64x64: read x + write y, 4 bytes each
                alg_bytes = 2.0 * launch_b * 128 * launch_hw * 4
                out["roofline"]["traffic_algorithmic_bytes"] = alg_bytes
                out["roofline"]["traffic_ratio"] = tr_info["bytes"] / alg_bytes
                out["roofline"]["mfma_busy_pmc"] = tr_info.get("mfma_busy")
            if wino and WINO_T2:
                out["roofline"]["kernel_form"] = ("conv3x3_wino_t2_kernel: two vertically adjacent 128-channel x 32-patch tiles per "
                                                  "workgroup (the family's launches with <= 16 channels on one side have their own "
                                                  "buckets)")
            if wino:
                out["roofline"]["algorithm"] = ("Winograd F(2x2,3x3), fp32 operands / fp32 accumulate: 16 MFMA products per "
                                                "2x2 output patch and channel pair instead of 36")
            out["kernels"] = {k: {"launches": v["launches"], "ms_total": round(v["ms"], 3),
                                  "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12} for k, v in summ.items()}
            heads = {k: summ[k] for k in ("decoder_fused_fwd", "decoder_fused_bwd", "reward_fused_fwd", "reward_fused_bwd")
                     if k in summ}
            if heads:
                # the reward head and the decoder + pixel loss: launched ONCE per iteration over the (horizon - 2) x batch
                # samples of all unrolled steps (Trainer.step), backward from the hidden maps the forward kept
                nb = len(range(0, a.steps, 10))
                per = {k: v["launches"] / float(nb) for k, v in heads.items()}
                spl = a.batch * (a.horizon - 2) / max(per.values())
                out["heads"] = {"time_batched": all(abs(v - 1.0) < 1e-9 for v in per.values()),
                                "samples_per_launch": spl, "kept_hidden_maps": bool(ops.save_hidden),
                                "us_per_iteration": {k: round(v["ms"] * 1e3 / nb, 1) for k, v in heads.items()},
                                "us_per_64_samples": {k: round(v["ms"] * 1e3 / nb / (a.batch * (a.horizon - 2) / 64.0), 1)
                                                      for k, v in heads.items()}}
        if world == 1 and not a.profile_mode and a.precision == "fp32" and a.size <= 64:
            # The same iteration with the 3x3 convolutions' fp32 arithmetic emulated on the bf16 MFMA pipe (exact 3-way
            # bf16 split of both operands, six partial products per product, fp32 accumulation: DESIGN.md section 4e).
            # NOT `value`: reported beside it, with the evidence that it is fp32-grade (tests/test_gpu_x3_mode.py; the
            # whole -m gpu parity suite passes in this mode at unchanged tolerances).
            try:
                out["fp32x3"] = fp32x3_leg(a, opts, batches, dev, headline)
            except Exception as e:      # an optional leg must not cost the headline line; the failure stays visible
                out["fp32x3"] = {"error": "%s: %s" % (type(e).__name__, e)}
            finally:
                ops.set_compute_precision("fp32")
                os.environ.pop("SCM_X3_ANY_BATCH", None)
        if world == 1 and not a.profile_mode:
            out["decoder"] = decoder_leg(dev, a.batch)
            out["encoder"] = encoder_leg(dev, a.batch)
        if world == 1 and headline and not a.profile_mode:
            l1, flips = rollout_l1(dev)
            out["rollout_l1_vs_ref"] = l1
            out["rollout_latent_flips"] = flips
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    dp.shutdown()


if __name__ == "__main__":
    main()
