"""Fold the rocprofv3 --pmc passes of tools/prof_hot2.py into one JSON (last dispatch of each kernel).
usage: python tools/pmc_parse.py <dir with pass_*/**/*counter_collection.csv and *kernel_trace.csv> > out.json
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B)."""
import csv, glob, hashlib, json, os, subprocess, sys
root = sys.argv[1]
vals = {}      # kernel -> counter -> value of its last dispatch
dur = {}
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    last = {}
    for r in csv.DictReader(open(f)):
        last[(r['Kernel_Name'], r['Counter_Name'])] = (int(r['Dispatch_Id']), float(r['Counter_Value']))
    for (k, c), (_, v) in last.items():
        vals.setdefault(k, {})[c] = v
for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        per.setdefault(r['Kernel_Name'], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in per.items():        # the last launch group of the pass: the shape the counters' (last) dispatch had
        dur.setdefault(k, []).extend(v[-3:])
out = {}
for k, c in vals.items():
    if 'pack_' in k or not any(t in k for t in ('conv3x3_mfma', 'conv3x3_h16', 'conv3x3_x3', 'conv3x3_wgrad', 'conv3x3_wino', 'decoder_fused_fwd', 'decoder_fused_bwd', 'reward_fused', 'sn_power_iter')):
        continue
    e = {}
    if k in dur:
        d = sorted(dur[k]); e['duration_us_median'] = round(d[len(d) // 2], 1)
    if 'FETCH_SIZE' in c: e['hbm_read_bytes'] = c['FETCH_SIZE'] * 1024 * 2
    if 'WRITE_SIZE' in c: e['hbm_write_bytes'] = c['WRITE_SIZE'] * 1024
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c: e['hbm_traffic_bytes'] = e['hbm_read_bytes'] + e['hbm_write_bytes']
    if 'GRBM_GUI_ACTIVE' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c and c['GRBM_GUI_ACTIVE'] > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, MFMA busy cycles over the chip's 1024 SIMDs (256 CUs x 4)
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0
        e['mfma_utilisation'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), 4)
        if 'duration_us_median' in e: e['effective_clock_ghz'] = round(cyc / e['duration_us_median'] / 1e3, 3)
    if 'SQ_LDS_BANK_CONFLICT' in c: e['lds_bank_conflict_cycles'] = c['SQ_LDS_BANK_CONFLICT']
    if 'SQ_LDS_IDX_ACTIVE' in c: e['lds_active_cycles'] = c['SQ_LDS_IDX_ACTIVE']
    out[k] = e
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = {}
for f in ('conv3x3_wino.hip', 'conv3x3_wino_t2.hip', 'conv3x3_wgrad_wino.hip', 'conv3x3_wino16.hip', 'conv3x3_wgrad_wino16.hip', 'conv3x3_wino_k16.hip', 'conv3x3_mfma.hip', 'conv3x3_h16.hip', 'conv3x3_x3.hip', 'conv3x3_wgrad.hip', 'conv3x3_wgrad_h16.hip', 'conv3x3_wgrad_x3.hip',
          'decoder_fused.hip', 'reward_fused.hip', 'sn.hip'):
    try:
        sha[f] = hashlib.sha256(open(os.path.join(here, 'scm_gan_amd', 'csrc', f), 'rb').read()).hexdigest()[:16]
    except OSError:
        pass
try:
    commit = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=here, capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
if not commit:
    try:
        commit = open(os.path.join(here, 'build', 'git_commit')).read().strip() or None     # written by __graft_entry__.build()
    except OSError:
        commit = None
json.dump({'kernel_source_sha16': sha, 'git_commit': os.environ.get('SCM_GIT_COMMIT') or commit, 'source': 'rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE> -- python3 tools/prof_hot2.py (separate passes)',
           'case': '128->128 channels, B=64, 64x64, circular padding, leaky-relu, 1-bit gate mask: forward, gated dgrad, wgrad; fp32 Winograd F(2x2,3x3) (the default), fp32 direct, fp32-on-the-bf16-pipe (x3) and 16-bit-operand kernels; fused decoder + BCE: backward (with the kept hidden map) over the 512 samples of a time-batched iteration, forward as inference launches it at B=64, 64x64x3; fused reward head forward / backward at 512 samples; since round 3 also the producer-gated Winograd forms (ungated weight gradient <0,1,0>) and the 16-channel-side layers of the transition (conv6 forward 256->16, its weight gradient and its input gradient 16->256, conv1 forward and weight gradient 16<->128); last dispatch of each kernel',
           'kernels': out}, sys.stdout, indent=1)
