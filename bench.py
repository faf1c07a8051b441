#!/usr/bin/env python3
"""bench.py -- training volumes/s (2ch x 128^3) of the MI355X-native 3D U-Net+VAE engine.

One "step" = one pass of the hot path over one resident batch: forward (training=True), Dice+0.1*MSE+0.1*KL+L2 loss,
Dice metric, full backward, TF-form Adam (train.py:140-152 of the reference), CLI-default model (base_filters=32,
reduction=8, depth=4, groups=8; args.py:121-143), fp32.  N=1: BASELINE.json configs[1] (batch 1 per GPU).

N>1: one process per GPU, one sample per rank (weak scaling), RCCL all-reduce of the 13 loss sums and of the flat
168.7 MB gradient buffer.  Two ways in:
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the driver's form): this process IS a rank
    (RANK / LOCAL_RANK / WORLD_SIZE in the environment); --gpus must equal WORLD_SIZE or the run is refused;
  * `python bench.py --gpus N` with no WORLD_SIZE: this process only LAUNCHES -- it starts N fresh rank processes before
    anything here touches the GPU (never re-executing a GPU-initialised process), relays rank 0's JSON line and exits with
    the worst child status.  `--share-gpu` lets the ranks share devices when the box has fewer than N (gloo instead of
    RCCL, which refuses two ranks on one device): a functional check of the N>1 path, not a scaling number.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":       dominant kernel (by summed time in the timed region), measured live with HIP events on the launch stream; its
                    `traffic` from the newest committed PMC capture of the same configuration (scripts/round_profile.sh ->
                    profiles/rNN*_pmc_traffic{,_bf16_b8,_infer_f16}.json), read bytes from the request-size counters
  "step_rooflines": SURVEY 8(d)'s algorithmic FLOPs / bytes of the whole step over the measured step time, and the HBM bytes all
                    kernels of a step moved in that capture over the algorithmic bytes (wasted-traffic ratio)
  "cpu_baseline":   the oracle's identical 128^3 step on the host cores (rank 0, N=1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= fp32 vector peak)
PEAK_HBM_TBS = 8.0             # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable)
# matrix instructions issued per algorithmic MAC: F(2x2,3x3) over (z,y) with x direct (conv_wino.hip, the weight-gradient form),
# F(2x2x2,3x3x3) (conv_wino3.hip)
WINOGRAD_EXECUTED = {'wino_kernel': 12.0 / 27.0, 'wgw_kernel': 12.0 / 27.0, 'w3_kernel': 8.0 / 27.0}


def _traffic_table(suffix=''):
    """the newest COMMITTED rocprofv3 PMC capture of one configuration (scripts/round_profile.sh -> profiles/rNN*_pmc_traffic<suffix>.json:
    separate --pmc passes for FETCH_SIZE, WRITE_SIZE and the fabric read-request counters), or (None, None).  suffix '' = the fp32
    step, '_bf16_b8' = BASELINE configs[2], '_infer_f16' = configs[4]."""
    import glob
    import re
    # newest = last by NAME (r01_ < r01c < ... < r04a): modification times mean nothing after a fresh checkout
    files = [f for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic%s.json' % suffix))
             if re.match(r'r\d+[a-z]?_pmc_traffic%s\.json$' % re.escape(suffix), os.path.basename(f))]
    if not files:
        return None, None
    f = sorted(files, key=os.path.basename)[-1]
    return json.load(open(f)), 'profiles/' + os.path.basename(f)


def _kernel_bytes(v):
    """HBM-side bytes of one launch from a kernel's counter means -> (read bytes, write bytes, how the read side was derived).
    MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B whatever the request size, so a kernel whose reads
    are 128-byte requests reports half its bytes.  The read side is therefore priced from the request-SIZE counters of the same capture:
    32 B x TCC_EA0_RDREQ_32B + 64 B x TCC_EA0_RDREQ_64B (or RDREQ - 32B - 128B where that counter is absent) + 128 B x
    TCC_EA0_RDREQ_128B.  Only a capture without those counters falls back to the guide's blanket rule, 2 x FETCH_SIZE (an upper
    bound; FETCH_SIZE itself is the lower one); `read_side` in the output says which of the two was used."""
    fetch = v.get('FETCH_SIZE_KiB_mean')
    write = v.get('WRITE_SIZE_KiB_mean')
    if fetch is None or write is None:
        return None
    rd, r32, r64, r128 = (v.get('TCC_EA0_RDREQ%s_sum_mean' % k) for k in ('', '_32B', '_64B', '_128B'))
    if rd is not None and r32 is not None and r128 is not None:
        if r64 is None:
            r64 = max(rd - r32 - r128, 0.0)
        read = 32.0 * r32 + 64.0 * r64 + 128.0 * r128
        how = 'request counters: 32 B x RDREQ_32B + 64 B x RDREQ_64B + 128 B x RDREQ_128B'
    else:
        read = 2.0 * fetch * 1024.0
        how = '2 x FETCH_SIZE (guide rule for wide reads on gfx950: an upper bound; FETCH_SIZE itself = RDREQ x 64 B is the lower one)'
    return read, write * 1024.0, how


def pmc_traffic(symbol, suffix=''):
    """HBM bytes per launch of `symbol` from the committed capture of this configuration.  The raw counter values and the file they
    come from are reported next to the figure -- it belongs to that capture, not to this run."""
    tab, src = _traffic_table(suffix)
    if tab is None:
        return None
    norm = lambda n: n.replace('void ', '').replace(' ', '').replace('TBF16,', '').replace('TF16,', '')
    key = norm(symbol)
    want_tail = None
    if key.endswith('<3x3x3>') or key.endswith('<1x1x1>'):      # lp_wgrad_kernel<T, NQ, K3>: the kernel size is its last template argument
        want_tail = 'true>' if key.endswith('<3x3x3>') else 'false>'
        key = key.split('<')[0]
    # every capture entry the symbol covers: one template instantiation for a full name, all of them for a bare one (the library's
    # launch records lump e.g. lp_k1_kernel / lp_k1f_kernel<...> under one name) -- launch-weighted means over those
    hits = []
    for k, v in tab.items():
        if k.startswith('_'):
            continue
        nk = norm(k)
        if want_tail is not None and not nk.endswith(want_tail):
            continue
        base = nk.split('<')[0]
        fam = base == key or (key == 'lp_k1_kernel' and base == 'lp_k1f_kernel') or (key == 'lp_conv_gather_kernel' and base == 'lp_conv_gatherq_kernel')
        # (a variant name of the library's launch records -- lp_s1d_kernel<MODE,TXL> -- covers the instantiations that carry further template
        # arguments: lp_s1d_kernel<1,5,false> and <1,5,true>, the plain and the fused-shortcut form)
        pref = ('<' in key) and nk.startswith(key[:-1] + ',')
        if (nk == key or pref or (('<' not in key) and fam)) and _kernel_bytes(v) is not None:
            hits.append(v)
    if not hits:
        return None
    n = float(sum(v['launches'] for v in hits))
    wmean = lambda f: sum(f(v) * v['launches'] for v in hits) / n
    read, write = wmean(lambda v: _kernel_bytes(v)[0]), wmean(lambda v: _kernel_bytes(v)[1])
    how = _kernel_bytes(hits[0])[2]
    opt = lambda name: (wmean(lambda v: v.get(name, 0.0)) if all(name in v for v in hits) else None)
    return {'bytes': read + write, 'read_bytes': read, 'write_bytes': write, 'read_side': how,
            'fetch_size_kib_raw': opt('FETCH_SIZE_KiB_mean'), 'write_size_kib_raw': opt('WRITE_SIZE_KiB_mean'),
            'rdreq_raw': opt('TCC_EA0_RDREQ_sum_mean'), 'rdreq_32b_raw': opt('TCC_EA0_RDREQ_32B_sum_mean'),
            'rdreq_64b_raw': opt('TCC_EA0_RDREQ_64B_sum_mean'), 'rdreq_128b_raw': opt('TCC_EA0_RDREQ_128B_sum_mean'),
            # clock the part held under this kernel in the capture: GRBM_GUI_ACTIVE (summed over the 8 XCDs) over the dispatch duration
            'clock_ghz': (opt('GRBM_GUI_ACTIVE_mean') / 8.0 / opt('duration_ns_mean')) if (opt('GRBM_GUI_ACTIVE_mean') and opt('duration_ns_mean')) else None,
            # the matrix pipe's own counter, same capture: SQ_VALU_MFMA_BUSY_CYCLES / (launch cycles x 1024 SIMDs) -- to hold against the
            # FLOP / time fraction (`frac`): the two agree when the kernel's matrix instructions are the ones its FLOP count assumes
            'mfma_busy': opt('mfma_busy'),
            'launches_in_capture': int(n), 'variants_in_capture': len(hits), 'source': 'committed capture ' + src}


# algorithmic work per unit (CLI-default model).  Training volume 2ch x 128^3: 6.517 TFLOP (fwd 2,172.2 GFLOP, bwd 2x);
# compulsory HBM bytes 32.2 GB fp32 at batch 1, 15.9 GB per volume with 16-bit storage at batch 8
TRAIN_ALGORITHMIC_TFLOP_PER_VOLUME = 6.517
TRAIN_ALGORITHMIC_GB_PER_VOLUME = {'f32': 32.2, 'bf16': 15.9, 'f16': 15.9}


def step_rooflines(alg_tflop, alg_gb, sec, dt, suffix):
    """whole-step rooflines: SURVEY 8(d)'s algorithmic FLOPs and compulsory bytes of one step over the measured step time, against
    the matrix peak of the type and the HBM peak; and, from the committed PMC capture of this configuration, the HBM bytes ALL kernels
    of a step moved (sum over kernels of launches x per-launch bytes / steps in the capture) over the algorithmic bytes = the
    wasted-traffic ratio of the step"""
    peak = PEAK_F32_MFMA_TFLOPS if dt == 'f32' else PEAK_F16_MFMA_TFLOPS
    out = {'algorithmic_tflop_per_step': alg_tflop, 'algorithmic_gb_per_step': alg_gb,
           'mfma_frac': alg_tflop / sec / peak, 'mfma_peak_tflops': peak,
           'hbm_frac': alg_gb / 1e3 / sec / PEAK_HBM_TBS, 'hbm_peak_tbs': PEAK_HBM_TBS}
    if dt == 'f32':
        out['note'] = 'direct-form FLOPs: the fp32 engine\'s Winograd kernels execute 8/27 (12/27) of them, so mfma_frac may exceed 1'
    tab, src = _traffic_table(suffix)
    if tab is not None and tab.get('_meta', {}).get('steps_in_capture'):
        nsteps = float(tab['_meta']['steps_in_capture'])
        rd = wr = 0.0
        hows = set()
        once = 0.0      # bytes of launches that are not per-step work (weight packing at construction, first-use initialisation)
        steady = bool(tab['_meta'].get('steady_step_launches'))
        for k, v in tab.items():
            if k.startswith('_'):
                continue
            b = _kernel_bytes(v)
            if b is None:
                continue
            hows.add(b[2].split(':')[0].split(' (')[0])
            if steady:
                # the capture names its last full step (scripts/pmc_aggregate.py: the dispatches between the last two Adam dispatches):
                # a step's bytes are that step's launches x their own counter means; everything else a kernel moved in the capture
                # beyond (steps x that) ran once (every weight image is packed by a launch of its own at first use)
                sv = v.get('steady')
                sb = _kernel_bytes(sv) if sv else None
                if sb is not None:
                    rd += sb[0] * sv['launches'] * nsteps
                    wr += sb[1] * sv['launches'] * nsteps
                once += max((b[0] + b[1]) * v['launches'] - ((sb[0] + sb[1]) * sv['launches'] * nsteps if sb else 0.0), 0.0)
                continue
            # older captures: a kernel whose launch count is not a multiple of the captured steps ran outside the step loop
            extra = v['launches'] % int(nsteps)
            per_step = v['launches'] - extra
            rd += b[0] * per_step
            wr += b[1] * per_step
            once += (b[0] + b[1]) * extra
        out.update({'measured_hbm_gb_per_step': (rd + wr) / nsteps / 1e9, 'measured_read_gb_per_step': rd / nsteps / 1e9,
                    'measured_write_gb_per_step': wr / nsteps / 1e9, 'wasted_traffic_ratio': (rd + wr) / nsteps / 1e9 / alg_gb,
                    'measured_hbm_frac_of_step': (rd + wr) / nsteps / sec / (PEAK_HBM_TBS * 1e12),
                    'outside_the_step_loop_gb': once / 1e9,
                    'read_side': sorted(hows),
                    'source': 'committed capture ' + src + (' (all kernels of its last full step: %d launches)' % tab['_meta']['steady_step_launches']
                                                            if steady else ' (all kernels, %d steps)' % nsteps)})
    return out


CPU_BASELINE_THREADS = 16   # torch-CPU conv3d stops scaling (and thrashes badly) far below the GPU box's 256 hardware threads


def _cpu_model():
    try:
        for line in subprocess.run(['lscpu'], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if line.lower().startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return None


def _cpu_baseline_worker(crop, agreement_dir=None):
    """child process: oracle (torch-CPU restatement) train step on the host cores; prints one JSON line.  With `agreement_dir` the
    step starts from the PRODUCT's initial weights (weights.npz written by measure_train before its first step, oracle naming) and
    uses the same synthetic volume, dropout mask and eps (seed 1234): its loss / Dice go into the line and its label map and
    updated parameters into oracle_step.npz for the parent's `oracle_agreement`."""
    import numpy as np
    import torch
    from oracle import torch_ref as R
    threads = min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    torch.set_num_threads(threads)
    cfg = R.default_config(base_filters=32, reduction=8)

    def one(c, weights=None):
        x, y, mask, eps = R.synthetic_batch(1, c, latent=128, seed=1234)
        P = R.build_params(cfg, c, seed=0)
        if weights is not None:
            z = np.load(weights)
            assert set(z.files) == set(P.keys()), 'weights.npz does not hold the oracle\'s variables'
            for k in P:
                P[k] = torch.from_numpy(z[k]).reshape(P[k].shape)
        for k in P:
            P[k] = P[k].float()
        t0 = time.time()
        loss, macro, micro, _, outs = R.train_step(P, cfg, x, y, mask, eps, {}, 1e-4, 1)
        t = time.time() - t0
        return t, float(loss), float(macro), float(micro), P, (y, outs[0])

    one((16, 16, 16))                      # warm-up: thread pool, oneDNN primitive caches
    wpath = os.path.join(agreement_dir, 'weights.npz') if agreement_dir else None
    t, loss, macro, micro, P, (y, y_pred) = one((crop,) * 3, wpath)
    rec = {'value': (crop ** 3 / float(128 ** 3)) / t, 'unit': 'volumes/s', 'cores': threads, 'kind': 'port',
           'host_cpus': os.cpu_count(), 'cpu_model': _cpu_model(),
           'sample': ('1 full train step (fwd+bwd+Dice metric+Adam), fp32, CLI-default model, one 2ch x %d^3 '
                      'volume, after a 16^3 warm-up step; oracle/torch_ref.py on %d torch threads' % (crop, threads))
                     + ('' if crop == 128 else '; value scaled to 128^3 volumes')
                     + ('; the product\'s initial weights and the same volume / dropout mask / eps' if wpath else ''),
           'seconds': round(t, 3), 'loss': loss, 'macro_dice': macro, 'micro_dice': micro}
    if agreement_dir:
        _, _, labels = R.dice_coefficient(y, y_pred, cfg['data_format'])
        np.savez(os.path.join(agreement_dir, 'oracle_step.npz'), labels=labels.to(torch.uint8).numpy(), y_pred=y_pred.numpy(),
                 **{'P/' + k: v.numpy() for k, v in P.items()})
    print(json.dumps(rec))


def cpu_baseline(crop=128, timeout_s=240, agreement_dir=None):
    """oracle train step timed on the host cores in a child process; kind 'port' (the reference needs TensorFlow, which
    cannot be installed here, so the restatement is what can be timed).  A 128^3 step is ~20-30 s on 16 threads."""
    fail = {'value': None, 'unit': 'volumes/s', 'cores': min(os.cpu_count() or 1, CPU_BASELINE_THREADS), 'kind': 'port',
            'host_cpus': os.cpu_count(), 'cpu_model': _cpu_model()}
    try:
        cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--crop', str(crop)]
        if agreement_dir:
            cmd += ['--agreement-dir', agreement_dir]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        fail['sample'] = 'worker failed: ' + out.stderr[-200:]
    except subprocess.TimeoutExpired:
        fail['sample'] = 'one %d^3 oracle step did not finish within %d s' % (crop, timeout_s)
    return fail


def oracle_agreement(first, cpu, agreement_dir):
    """north_star: "Dice within 1e-4 of the reference on a fixed seed", argmax label map bit-exact -- the product's FIRST step (before
    the warm-up; initial weights, synthetic volume seed 1234, injected dropout mask and eps) next to the oracle's step from the same
    weights and draws in the cpu_baseline child (fp32 torch-CPU; tests/test_oracle_fullsize_gpu.py holds the engine to the FP64
    oracle with randomised gamma / beta at this size)."""
    import numpy as np
    if first is None or cpu.get('loss') is None:
        return None
    z = np.load(os.path.join(agreement_dir, 'oracle_step.npz'))
    lab_o, yp_o = z['labels'], z['y_pred']
    lab = first['labels'].reshape(lab_o.shape)
    top2 = np.sort(yp_o, axis=-1)[..., -2:]
    # pointwise criterion (tests/test_oracle_fullsize_gpu.py::_label_check): a label may differ only where the oracle's own margin (|p_max -
    # 0.5| or the top-2 gap) is at most twice the engine's |y_pred - oracle| at THAT voxel; and no more voxels than the 1e-5 band holds
    margin = np.minimum(np.abs(top2[..., 1] - 0.5), np.abs(top2[..., 1] - top2[..., 0]))
    dvox = np.abs(first['y_pred'].reshape(yp_o.shape) - yp_o).max(axis=-1)
    near = margin <= 2.0 * dvox
    band5 = int((margin < 1e-5).sum())
    differ = lab != lab_o
    dpar = 0.0
    for name, arr in first['params'].items():
        dpar = max(dpar, float(np.abs(arr.reshape(-1) - z['P/' + name].reshape(-1)).max()))
    return {'dloss_rel': abs(first['loss'] - cpu['loss']) / max(1.0, abs(cpu['loss'])),
            'ddice_macro': abs(first['macro'] - cpu['macro_dice']), 'ddice_micro': abs(first['micro'] - cpu['micro_dice']),
            'label_mismatch': int(differ.sum()), 'label_mismatch_outside_pointwise_criterion': int((differ & ~near).sum()),
            'voxels_inside_1e-5_band': band5, 'voxels': int(lab_o.size),
            'y_pred_max_abs_diff': float(np.abs(first['y_pred'].reshape(yp_o.shape) - yp_o).max()),
            'max_param_diff_after_adam': dpar,
            'engine': {'loss': first['loss'], 'macro_dice': first['macro'], 'micro_dice': first['micro']},
            'oracle': {'loss': cpu['loss'], 'macro_dice': cpu['macro_dice'], 'micro_dice': cpu['micro_dice']},
            'within_north_star': bool(abs(first['macro'] - cpu['macro_dice']) <= 1e-4 and abs(first['micro'] - cpu['micro_dice']) <= 1e-4
                                      and int((differ & ~near).sum()) == 0 and int(differ.sum()) <= band5),
            'what': 'first train step of this run (initial weights, seed-1234 volume, injected dropout mask / eps) vs oracle/torch_ref.py '
                    'in fp32 from the same weights and draws (the cpu_baseline child); tolerance |dDice| <= 1e-4; a label may differ only where '
                    'the oracle margin (|p-0.5| or top-2 gap) <= 2 |dp| at that voxel, and at no more voxels than the 1e-5 band holds'}


def active_overrides():
    """every BTS_* variable in the environment (A/B and test aids; the defaults are the product)"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith('BTS_') and not k.startswith('BTS_BENCH_')}


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Start args.gpus fresh rank processes (this process has not touched the GPU and never will) and relay rank 0's line."""
    import torch
    ndev = torch.cuda.device_count()        # counts devices without initialising the runtime
    n = args.gpus
    if ndev < n and not args.share_gpu:
        raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible (pass --share-gpu for a functional run of the N>1 '
                         'path with ranks sharing devices over gloo)' % (n, ndev))
    if ndev < 1:
        raise SystemExit('bench.py needs an MI355X (no CPU execution path exists for the product)')
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r % ndev), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        if ndev < n:
            env['BTS_BENCH_SHARED_DEVICES'] = str(ndev)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    # poll ALL ranks: when one dies (init failure, out of memory) the others sit in a collective forever -- end them and fail
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            failed = codes
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        raise SystemExit('rank exit codes %s (remaining ranks were terminated)' % failed)
    reader.join(timeout=10)
    sys.stdout.write(buf[0] if buf else '')
    sys.stdout.flush()


PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak (v_mfma_f32_32x32x16_*)
INFER_ALGORITHMIC_GFLOP = 3984.7   # 160x192x160 forward without the VAE branch
INFER_ALGORITHMIC_GB = {'f16': 8.7, 'bf16': 8.7, 'f32': 17.25}   # 8.55 GB of 16-bit activations + 149 MB weights


def measure_infer(args, world, rank, dev, overrides, dtype=None, shape=None, batch=None, steps=None, warmup=None):
    """BASELINE configs[4]: 2ch x 155x190x147 zero-padded to 160x192x160 (test.py:164-178), inference=True (VAE skipped,
    model.py:67-68), CLI-default model, batch 1 per GPU.  A step = one forward.  N > 1: independent replicas (the path has no
    exchange step).  Returns the result dict on rank 0 (None elsewhere)."""
    import torch
    from bts_amd import lowp, ops, parallel
    from bts_amd.model import Model
    dtype = args.dtype if dtype is None else dtype
    batch = args.batch if batch is None else batch
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    dt = {'f32': 'f32', 'fp32': 'f32', 'float32': 'f32', None: 'f16', 'f16': 'f16', 'fp16': 'f16', 'float16': 'f16',
          'bf16': 'bf16', 'bfloat16': 'bf16'}[dtype]
    model = Model(base_filters=32, reduction=8, depth=4, groups=8)
    model.build((1, 128, 128, 128, 2))     # the weights belong to the training crop (the VAE is tied to it, vae.py:101-111)
    g = torch.Generator().manual_seed(1234 + rank)
    shape = tuple(int(v) for v in (args.infer_shape if shape is None else shape).split(','))
    canonical = shape == (160, 192, 160) and batch == 1
    x = torch.randn((batch,) + shape + (2,), generator=g)
    if canonical:        # the zero padding of test.py:164-178
        x[:, 155:] = 0
        x[:, :, 190:] = 0
        x[:, :, :, 147:] = 0
    x = x.to(dev)
    if dt == 'f32':
        fwd = lambda: model(x, training=False, inference=True)[0].t
    else:
        run = lowp.LowPrecisionForward(model, {'f16': 'float16', 'bf16': 'bfloat16'}[dt])
        fwd = lambda: run(x)
    if args.serial_streams:
        ops.enable_side_streams(False)
    for _ in range(warmup):
        y = fwd()
    torch.cuda.synchronize()
    if parallel.active():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = fwd()
    torch.cuda.synchronize()
    if parallel.active():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dts = time.perf_counter() - t0
    ranks_seen = 1
    if parallel.active():
        tt, one = torch.tensor([dts], dtype=torch.float64), torch.ones(1, dtype=torch.float64)
        if torch.distributed.get_backend() != 'gloo':
            tt, one = tt.to(dev), one.to(dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(one, op=torch.distributed.ReduceOp.SUM)
        dts, ranks_seen = float(tt.item()), int(round(float(one.item())))
    # per-kernel pass AFTER the timed region (the HIP-event hooks cost a few percent of a 7-ms forward): `roofline` / `kernel_breakdown`
    prof = None
    prof_steps = min(steps, 5)
    if not args.no_profile and rank == 0:
        ops.enable_side_streams(False)      # (one stream: a launch that shares the chip with the gate stream has no duration of its own)
        y = fwd()
        torch.cuda.synchronize()
        ops.profile_enable(True)
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            y = fwd()
        torch.cuda.synchronize()
        dt_prof = time.perf_counter() - t1
        ops.profile_enable(False)
        prof = ops.profile_records(detail=True)
        ops.enable_side_streams(not args.serial_streams)
    if rank != 0:
        return None
    sec = dts / steps
    out = {
        'metric': 'inference volumes/sec (2ch x 155x190x147 padded to 160x192x160, VAE off)' if canonical else
                  'forward volumes/sec (2ch x %dx%dx%d, batch %d, VAE off)' % (shape + (batch,)),
        'value': ranks_seen * batch / sec,
        'unit': 'volumes/s', 'n_gpus': world, 'ranks_seen': ranks_seen, 'steps': steps, 'warmup': warmup,
        'ms_per_step': 1e3 * sec, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dt, 'data': 'synthetic',
        'config': {'workload': ('BASELINE configs[4]: full-volume inference, 2ch x 155x190x147 zero-padded to 160x192x160, inference=True '
                                '(decoder path, VAE off), batch 1 per GPU, CLI-default model; storage %s, fp32 sums' % dt) if canonical else
                               ('NOT a BASELINE line: forward only (VAE off) at 2ch x %dx%dx%d, batch %d per GPU, CLI-default model; storage %s'
                                % (shape + (batch, dt))),
                   'parallelism': 'replicas%d' % world},
        'y_pred_mean': float(y.mean()),
    }
    if canonical:      # whole-forward rooflines (SURVEY 8d figures): both terms, the larger one bounds the forward
        out['forward_rooflines'] = {
            'algorithmic_gflop': INFER_ALGORITHMIC_GFLOP, 'algorithmic_gb': INFER_ALGORITHMIC_GB[dt],
            'mfma_frac': INFER_ALGORITHMIC_GFLOP / 1e3 / sec / (PEAK_F32_MFMA_TFLOPS if dt == 'f32' else PEAK_F16_MFMA_TFLOPS),
            'hbm_frac': INFER_ALGORITHMIC_GB[dt] / 1e3 / sec / PEAK_HBM_TBS}
        out['step_rooflines'] = step_rooflines(INFER_ALGORITHMIC_GFLOP / 1e3, INFER_ALGORITHMIC_GB[dt], sec, dt,
                                               '_infer_f16' if dt == 'f16' else '_infer_' + dt)
    if overrides:
        out['overrides'] = overrides
    if prof:
        sfx = '_infer_f16' if dt == 'f16' else '_infer_' + dt
        out.update(_roofline_from_records(prof, prof_steps, dt_prof, (lambda sym: pmc_traffic(sym, sfx)) if canonical else None,
                                          'HIP events on the launch stream over %d further one-stream forwards right after the timed region' % prof_steps))
    del model
    return out


def _roofline_from_records(prof, steps_p, dt_p, traffic_of, measured):
    """`roofline` (dominant kernel by summed launch time) + `kernel_breakdown` from the library's HIP-event records"""
    agg = {}
    for sym, flops, ms in prof:
        a = agg.setdefault(sym, [0.0, 0.0, 0])
        a[0] += ms * 1e-3
        a[1] += flops
        a[2] += 1
    sym, (tsec, fl, nl) = max(agg.items(), key=lambda kv: kv[1][0])
    alg = fl / tsec / 1e12                        # ALGORITHMIC (direct-form) FLOP rate
    wino = sym in WINOGRAD_EXECUTED
    ach = alg * WINOGRAD_EXECUTED.get(sym, 1.0)   # what the matrix pipe executes
    traffic = traffic_of(sym) if traffic_of else None
    t_launch = tsec / nl
    peak = PEAK_F16_MFMA_TFLOPS if sym.startswith('lp_') else PEAK_F32_MFMA_TFLOPS   # 16-bit kernels against the 16-bit dense peak
    # which roofline bounds it: time the executed FLOPs need at the matrix peak vs time the measured HBM traffic needs
    t_mfma = (fl / nl) * WINOGRAD_EXECUTED.get(sym, 1.0) / (peak * 1e12)
    t_hbm = (traffic['bytes'] / (PEAK_HBM_TBS * 1e12)) if traffic else 0.0
    roof = {
        'kernel': sym, 'bound': 'mfma' if t_mfma >= t_hbm else 'hbm',
        'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
        'achieved_algorithmic': alg, 'frac_algorithmic': alg / peak,
        'traffic': traffic['bytes'] if traffic else None, 'traffic_detail': traffic,
        'hbm_frac_of_launch': (t_hbm / t_launch) if traffic else None,
        # the matrix pipe's own busy counter for this kernel (committed capture: SQ_VALU_MFMA_BUSY_CYCLES over launch cycles x 1024 SIMDs):
        # cycles, so it is a fraction of the clock the part HELD in the capture, where `frac` prices against the nominal-clock peak
        'mfma_busy_counter': traffic.get('mfma_busy') if traffic else None,
        # clock the part held under this kernel IN THE CAPTURE (GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration).  The counter passes run
        # the kernels one at a time with idle gaps: the sustained step runs hotter and lower-clocked; what the power envelope costs
        # the 16-bit matrix kernels is measured in profiles/r04_power_limit.txt (zero-operand runs, library-GEMM yardstick)
        'clock_ghz_in_capture': traffic.get('clock_ghz') if traffic else None,
        'launches_per_step': nl / steps_p, 'avg_launch_ms': 1e3 * t_launch,
        'algorithmic_gflop_per_launch': fl / nl / 1e9,
        'time_share_of_step': tsec / dt_p,
        'measured': measured,
    }
    if wino:
        roof['note'] = ('Winograd forms issue fewer matrix instructions than algorithmic MACs (F(2x2x2,3x3x3): 8 per 27, '
                        'F(2x2,3x3) x direct: 12 per 27; executed/algorithmic = %.4f here): ' % WINOGRAD_EXECUTED[sym] +
                        '`achieved`/`frac` are the EXECUTED rate (what is left to gain); *_algorithmic is '
                        'direct-form FLOPs / time and may exceed the peak')
    brk = {}
    for k, v in sorted(agg.items()):
        pk = PEAK_F16_MFMA_TFLOPS if k.startswith('lp_') else PEAK_F32_MFMA_TFLOPS
        e = {'ms_per_step': 1e3 * v[0] / steps_p, 'tflops': v[1] / v[0] / 1e12, 'launches_per_step': v[2] / steps_p,
             'mfma_frac': v[1] * WINOGRAD_EXECUTED.get(k, 1.0) / v[0] / 1e12 / pk}
        tr = traffic_of(k) if traffic_of else None
        if tr:        # HBM side of the same kernel (bytes per launch from the committed capture over this run's launch time)
            e['hbm_tbs'] = tr['bytes'] / (v[0] / v[2]) / 1e12
            e['hbm_frac'] = e['hbm_tbs'] / PEAK_HBM_TBS
            e['bound'] = 'hbm' if e['hbm_frac'] > e['mfma_frac'] else 'mfma'
            if tr.get('mfma_busy') is not None:
                e['mfma_busy'] = tr['mfma_busy']
        brk[k] = e
    return {'roofline': roof, 'kernel_breakdown': brk}


def measure_train(args, world, rank, dev, overrides, dtype=None, batch=None, steps=None, warmup=None, shared=None, agreement_dir=None):
    """one training configuration -> result dict on rank 0 (None elsewhere).  dtype f32: BASELINE configs[1] (the headline);
    bf16 / f16: the 16-bit STORAGE step of BASELINE configs[2] (bts_amd.lowp_train)."""
    import torch
    from bts_amd import ops, parallel
    from bts_amd.data import synthetic_batch
    from bts_amd.model import Model
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, train_step
    dtype = args.dtype if dtype is None else dtype
    nb = args.batch if batch is None else batch
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    crop = (args.crop,) * 3
    kw = dict(base_filters=32, reduction=8, depth=4, groups=8)
    model = Model(**kw)
    model.build((nb,) + crop + (2,))
    parallel.broadcast_parameters(model)
    x, y, mask0, eps0 = synthetic_batch(nb, crop, latent=128, seed=1234 + rank)
    x, y = x.to(dev), y.to(dev)
    opt = ScheduledOptim(1e-4)
    opt(epoch=0)
    loss_fn, dice_fn = DiceVAELoss(), DiceCoefficient()
    tdt = {None: 'f32', 'f32': 'f32', 'fp32': 'f32', 'float32': 'f32', 'bf16': 'bf16', 'bfloat16': 'bf16', 'f16': 'f16', 'fp16': 'f16',
           'float16': 'f16'}[dtype]
    if tdt != 'f32':      # BASELINE configs[2]: 16-bit storage, fp32 sums / master weights (bts_amd.lowp_train); its own line, never the headline
        from bts_amd.lowp_train import LowPrecisionTrainer
        trainer = LowPrecisionTrainer(model, {'bf16': 'bfloat16', 'f16': 'float16'}[tdt])
        train_step = lambda model_, opt_, loss_fn_, dice_fn_, x_, y_: trainer.step(opt_, dice_fn_, x_, y_)   # noqa: E731

    if args.serial_streams:
        ops.enable_side_streams(False)
    first = None
    if agreement_dir and tdt == 'f32' and nb == 1 and rank == 0:
        # `oracle_agreement`: one extra step in front of the warm-up, from the initial weights, with the dropout mask and eps of the
        # synthetic batch injected (one-shot: later steps draw their own on the device).  The weights go to the cpu_baseline child
        # (oracle naming), which repeats this very step; nothing of this is inside the timed region.
        import numpy as np
        np.savez(os.path.join(agreement_dir, 'weights.npz'),
                 **{model.oracle_name(p): p.t.detach().cpu().numpy() for p in model.trainable_variables})
        model.encoder.set_dropout_mask(mask0)
        model.vae.set_eps(eps0)
        grab = {}

        def dice_and_grab(y_true, y_pred):
            grab['y_pred'] = y_pred.t.detach().clone()
            return dice_fn(y_true, y_pred)
        l0, ma0, mi0 = train_step(model, opt, loss_fn, dice_and_grab, x, y)
        torch.cuda.synchronize()
        first = {'loss': float(l0), 'macro': float(ma0), 'micro': float(mi0), 'labels': dice_fn.last_labels.cpu().numpy(),
                 'y_pred': grab['y_pred'].cpu().numpy(),
                 'params': {model.oracle_name(p): p.t.detach().cpu().numpy() for p in model.trainable_variables}}
        del grab
    for _ in range(warmup):
        loss, macro, micro = train_step(model, opt, loss_fn, dice_fn, x, y)
    torch.cuda.synchronize()
    if parallel.active():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    gsync = parallel.grad_sync(model) if parallel.active() else None
    if gsync is not None:      # two event records per step around the final waits of the exchange: what the compute stream stood still for
        gsync.timing = True
        gsync.exposed_wait_ms()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]      # per-step GPU time (an event record each: no wait)
    t0 = time.perf_counter()
    marks[0].record()
    for k_ in range(steps):
        loss, macro, micro = train_step(model, opt, loss_fn, dice_fn, x, y)
        marks[k_ + 1].record()
    torch.cuda.synchronize()
    if parallel.active():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[k_].elapsed_time(marks[k_ + 1]) for k_ in range(steps))
    exchange = None
    if gsync is not None:
        waits = gsync.exposed_wait_ms()
        gsync.timing = False
        exchange = _exchange_diagnostics(model, gsync, waits, dev)
    if os.environ.get('BTS_BENCH_MEMSTATS'):       # allocator state of the timed region, to stderr (diagnostics; not part of the line)
        ms_ = torch.cuda.memory_stats()
        print('memstats: reserved peak %.1f GB, allocated peak %.1f GB, hipMalloc retries %d, segments %d' %
              (ms_['reserved_bytes.all.peak'] / 1e9, ms_['allocated_bytes.all.peak'] / 1e9, ms_['num_alloc_retries'],
               ms_['segment.all.current']), file=sys.stderr, flush=True)
    ranks_seen = 1
    if parallel.active():
        tt = torch.tensor([dt], dtype=torch.float64)
        one = torch.ones(1, dtype=torch.float64)
        if torch.distributed.get_backend() != 'gloo':
            tt, one = tt.to(dev), one.to(dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(one, op=torch.distributed.ReduceOp.SUM)
        dt = float(tt.item())
        ranks_seen = int(round(float(one.item())))
    # Per-kernel pass, after the timed region: the step runs its weight gradients and gate branches on side streams, so inside
    # the timed region a kernel shares the chip with other streams' kernels and the HIP events around its launch measure that
    # sharing, not the kernel.  Launch durations (the `roofline` object, `kernel_breakdown`) are therefore taken from PROF_STEPS
    # further steps of the same workload on ONE stream, HIP events on the launch stream around every conv / weight-gradient launch
    # (`bench.py --serial-streams` under rocprofv3 is the capture these averages must agree with).
    do_prof = (not args.no_profile) and rank == 0
    prof = None
    prof_steps = min(steps, 5)
    loss_v, macro_v = float(loss), float(macro)
    if parallel.active():
        torch.distributed.barrier()
    if do_prof and not parallel.active():
        ops.enable_side_streams(False)
        train_step(model, opt, loss_fn, dice_fn, x, y)
        torch.cuda.synchronize()
        ops.profile_enable(True)
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            train_step(model, opt, loss_fn, dice_fn, x, y)
        torch.cuda.synchronize()
        dt_prof = time.perf_counter() - t1
        ops.profile_enable(False)
        prof = ops.profile_records(detail=True)
        ops.enable_side_streams(not args.serial_streams)
    if rank != 0:
        return None
    volumes = ranks_seen * nb * steps
    n_gpus = min(world, int(shared)) if shared else world
    out = {
        'metric': 'training volumes/sec (2ch x %d^3)' % args.crop, 'value': volumes / dt, 'unit': 'volumes/s',
        'n_gpus': n_gpus, 'ranks_seen': ranks_seen, 'steps': steps, 'warmup': warmup,
        'ms_per_step': 1e3 * dt / steps,
        # (diagnostic: the median and the slowest single step by events on the main stream -- `value` stays total work / total time)
        'ms_per_step_median': per_step[len(per_step) // 2], 'ms_per_step_max': per_step[-1],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': tdt, 'data': 'synthetic',
        'config': {'workload': ('BASELINE configs[1]: 2ch x %d^3, batch %d per GPU, fp32, full fwd+bwd with '
                                'Dice+KL+L2 VAE loss + Dice metric + TF-form Adam; CLI-default model '
                                '(base_filters=32, depth=4, groups=8, reduction=8; 42,174,773 params)' % (args.crop, nb)) if tdt == 'f32' else
                               (('BASELINE configs[3] (configs[2]\'s step on each of %d GPUs, RCCL gradient all-reduce launched in buckets from '
                                 'inside the backward): ' % world if (world > 1 and nb == 8 and tdt == 'bf16') else
                                 'BASELINE configs[2] (when batch = 8, bf16): ') +
                                '2ch x %d^3, batch %d per GPU, %s STORAGE of activations / their '
                                'gradients / weight images, fp32 sums, fp32 master weights and Adam; full train step (forward, data, '
                                'weight, GroupNorm and gate gradients on the 16-bit kernels; loss, metric, regulariser, Adam and the '
                                'dense VAE head in fp32); CLI-default model' % (args.crop, nb, tdt)),
                   'parallelism': 'dp%d' % world, 'global_batch': world * nb},
        'loss': loss_v, 'macro_dice': macro_v,
        'streams': 'serial (one HIP stream)' if args.serial_streams else 'main + weight-gradient + gate streams',
    }
    if exchange is not None:
        out['exchange'] = exchange
    if shared:
        out['config']['note'] = '%d ranks sharing %s device(s) over gloo: functional check of the N>1 path' % (world, shared)
    if overrides:
        out['overrides'] = overrides
    sfx = '' if (tdt == 'f32' and nb == 1) else ('_%s_b%d' % (tdt, nb))
    if args.crop == 128:
        out['step_rooflines'] = step_rooflines(TRAIN_ALGORITHMIC_TFLOP_PER_VOLUME * nb, TRAIN_ALGORITHMIC_GB_PER_VOLUME[tdt] * nb,
                                               dt / steps, tdt, sfx)
    if do_prof and prof:
        out.update(_roofline_from_records(
            prof, prof_steps, dt_prof, (lambda sym: pmc_traffic(sym, sfx)) if args.crop == 128 else None,
            'HIP events on the launch stream over %d one-stream steps run right after the timed region (%.2f ms per step that way): in the '
            'timed region the kernel shares the chip with the weight-gradient / gate streams and has no launch duration of its own'
            % (prof_steps, 1e3 * dt_prof / prof_steps)))
    if first is not None:
        out['_first_step'] = first      # (host arrays for main(): popped before the line is printed)
    del model, opt
    return out


def _exchange_diagnostics(model, gsync, waits, dev):
    """N > 1, after the timed region: what the gradient exchange (C1) costs, so that the first multi-GPU run explains itself --
    the buckets GradSync cut (MB, in launch order), each bucket's all-reduce run ALONE (5 repeats, HIP events on the compute stream around
    a blocking all-reduce; bus bandwidth = 2 (N-1)/N x bytes / time, the figure to hold against xGMI's ~153 GB/s per link x 7 links), and
    the time the compute stream waited at the end of the backward for buckets still in flight inside the timed steps (exposed = not
    overlapped).  All ranks run the collectives; the numbers are this rank's."""
    import torch
    from bts_amd import parallel
    d = torch.distributed
    w = parallel.world()
    sizes = [ln * 4 for _, ln, _ in gsync.buckets]
    alone, bw = [], []
    scratch = torch.zeros(max(ln for _, ln, _ in gsync.buckets), dtype=torch.float32, device=dev)
    on_device = d.get_backend() != 'gloo'
    for _, ln, _ in gsync.buckets:
        t = []
        for rep in range(6):
            torch.cuda.synchronize()
            d.barrier()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            parallel._sum_over_ranks(scratch[:ln])
            b.record()
            torch.cuda.synchronize()
            if rep:
                t.append(a.elapsed_time(b))
        ms = sorted(t)[len(t) // 2]
        alone.append(ms)
        bw.append(2.0 * (w - 1) / w * ln * 4 / (ms * 1e-3) / 1e9 if ms > 0 else None)
    return {'buckets': len(sizes), 'bucket_mb': [v / 1e6 for v in sizes], 'bucket_bytes_arg_mb': gsync.bucket_bytes / float(1 << 20),
            'allreduce_alone_ms': alone, 'busbw_gbs': bw, 'backend': d.get_backend(), 'on_device_buffers': on_device,
            'exposed_wait_ms_per_step': (sum(waits) / len(waits)) if waits else None, 'exposed_wait_ms_max': max(waits) if waits else None,
            'algo_env': {k: os.environ[k] for k in ('NCCL_ALGO', 'NCCL_PROTO', 'NCCL_MIN_NCHANNELS', 'NCCL_MAX_NCHANNELS', 'RCCL_MSCCL_ENABLE') if k in os.environ}}


def _sig(v, n=5):
    """floats to n significant digits, recursively (the printed line must fit the 8 KB the driver keeps)"""
    if isinstance(v, float):
        return float('%.*g' % (n, v))
    if isinstance(v, dict):
        return {k: _sig(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, n) for x in v]
    return v


def _compact(out, top):
    """the printed form of one result: the contract keys and `roofline` unchanged in meaning, `kernel_breakdown` cut to the `top`
    kernels by time (ms, executed fraction of the matrix peak, HBM fraction, bound), the counter read-outs behind `traffic`, the
    per-step byte sums and the long notes left to the detail file (gpurun_out/bench_detail.json, same content as round 4's line)"""
    c = {k: v for k, v in out.items() if k not in ('kernel_breakdown', 'also', 'overrides_note')}
    if 'roofline' in c:
        r = dict(c['roofline'])
        td = r.pop('traffic_detail', None)
        if td:
            r['traffic_source'] = td.get('source')
            r['traffic_read_bytes'], r['traffic_write_bytes'] = td.get('read_bytes'), td.get('write_bytes')
        r.pop('note', None)
        r['measured'] = (r.get('measured') or '').split(':')[0]
        c['roofline'] = r
    if 'step_rooflines' in c:
        c['step_rooflines'] = {k: v for k, v in c['step_rooflines'].items()
                               if k in ('algorithmic_tflop_per_step', 'algorithmic_gb_per_step', 'mfma_frac', 'hbm_frac', 'measured_hbm_gb_per_step',
                                        'wasted_traffic_ratio', 'source')}
    if 'forward_rooflines' in c:
        c.pop('forward_rooflines')     # (the same two fractions are in step_rooflines)
    if 'kernel_breakdown' in out:
        rows = sorted(out['kernel_breakdown'].items(), key=lambda kv: -kv[1]['ms_per_step'])
        c['kernel_breakdown'] = {k: {'ms': v['ms_per_step'], 'n': v['launches_per_step'], 'mfma_frac': v['mfma_frac'],
                                     'mfma_busy': v.get('mfma_busy'), 'hbm_frac': v.get('hbm_frac'), 'bound': v.get('bound')} for k, v in rows[:top]}
        c['kernel_breakdown_rest_ms'] = sum(v['ms_per_step'] for _, v in rows[top:])
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--crop', type=int, default=128)
    ap.add_argument('--batch', type=int, default=1, help='samples per GPU')
    ap.add_argument('--share-gpu', action='store_true', help='let ranks share devices (gloo); functional check only')
    ap.add_argument('--bucket-mb', type=float, default=None,
                    help='N > 1: size of the gradient all-reduce buckets in MB (default 64: parallel.BUCKET_BYTES / BTS_DP_BUCKET_MB)')
    ap.add_argument('--allow-overrides', action='store_true', help='run although BTS_* A/B switches are set')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-crop', type=int, default=128)
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--no-also', action='store_true',
                    help='headline only: skip the BASELINE configs[2] / configs[4] (N = 1) or configs[3] (N > 1) measurements nested under "also"')
    ap.add_argument('--serial-streams', action='store_true',
                    help='one HIP stream for the whole step (the per-kernel rocprofv3 capture that backs `roofline` is taken this way)')
    ap.add_argument('--infer', action='store_true', help='BASELINE configs[4]: full-volume inference (VAE off) instead of the train step')
    ap.add_argument('--infer-shape', default='160,192,160', help='D,H,W of the --infer volume (multiples of 8)')
    ap.add_argument('--dtype', default=None, help="storage type: training f32 (default) | bf16 | f16; --infer f16 (default) | bf16 | f32")
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--agreement-dir', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--full-line', action='store_true',
                    help='print the complete record (every kernel, counter read-outs, notes) instead of the tail-safe compact line; '
                         'the complete record is always written to gpurun_out/bench_detail.json')
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        _cpu_baseline_worker(args.crop, args.agreement_dir)
        return
    overrides = active_overrides()
    if overrides and not args.allow_overrides:
        raise SystemExit('bench.py measures the product defaults; unset %s or pass --allow-overrides (they are then listed '
                         'in the JSON line)' % ', '.join(overrides))

    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        launch_ranks(args, sys.argv[1:])
        return
    world = int(env_world) if env_world is not None else 1
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d: refusing to report a %d-rank run as %d GPUs'
                         % (args.gpus, world, world, args.gpus))

    import torch
    import bts_amd  # noqa: F401
    from bts_amd import parallel

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU execution path exists for the product)')
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    shared = os.environ.get('BTS_BENCH_SHARED_DEVICES')
    torch.cuda.set_device(local)
    if args.bucket_mb:
        parallel.set_bucket_bytes(int(args.bucket_mb * (1 << 20)))
    if world > 1 or os.environ.get('BTS_FORCE_PG'):   # BTS_FORCE_PG=1: 1-rank RCCL group, smoke-tests the N>1 code path
        parallel.init_from_env('gloo' if shared else 'nccl')
    dev = torch.device('cuda', local)

    import shutil
    import tempfile
    agreement_dir = None
    if args.infer:
        out = measure_infer(args, world, rank, dev, overrides)
    else:
        headline = args.dtype in (None, 'f32', 'fp32', 'float32') and args.batch == 1 and args.crop == 128
        want_cpu = world == 1 and not parallel.active() and not args.no_cpu_baseline
        if want_cpu and args.dtype in (None, 'f32', 'fp32', 'float32') and args.batch == 1 and args.cpu_baseline_crop == args.crop:
            agreement_dir = tempfile.mkdtemp(prefix='bts_bench_')
        out = measure_train(args, world, rank, dev, overrides, shared=shared, agreement_dir=agreement_dir)
        first = out.pop('_first_step', None) if out is not None else None
        if headline and world == 1 and not parallel.active() and not args.no_also and out is not None:
            # The two secondary BASELINE configurations, measured in this same process right after the (untouched) headline and
            # nested under "also": driver-witnessed numbers for the 16-bit engine.  Each is its own workload with its own timed
            # region (warm-up, then steps bracketed by synchronize); none of it is inside the headline's timed region.
            import gc
            gc.collect()                 # the headline's model and tape are garbage now: collect them here, not inside the next timed region
            torch.cuda.empty_cache()     # (a collection that frees ~20 GB of device memory mid-step stalls that step by tens of ms)
            also = {}
            try:
                # (5 warm-up + 10 timed steps: with 3 + 5 a single host or allocator hiccup inside the 0.4-s window moved the line by 15 %)
                also['configs[2]'] = measure_train(args, world, rank, dev, overrides, dtype='bf16', batch=8, steps=10, warmup=5)
                gc.collect()
                torch.cuda.empty_cache()
                # (a forward is 6.5 ms: 10 warm-up + 30 timed cost 0.3 s and take the first-launch / clock-ramp noise of 3 + 10 out of the line)
                also['configs[4]'] = measure_infer(args, world, rank, dev, overrides, dtype='f16', shape='160,192,160', batch=1,
                                                   steps=30, warmup=10)
            except Exception as e:   # the headline stands on its own; a failure here is reported, not hidden
                also['error'] = '%s: %s' % (type(e).__name__, str(e)[:300])
            out['also'] = also
        if world > 1 and args.dtype in (None, 'f32', 'fp32', 'float32') and args.batch == 1 and not args.no_also:
            # N > 1 (the driver's scaling run: `bench.py --gpus N` with its default arguments): after the fp32 weak-scaling headline, whose
            # N = 1 value is the single-GPU line, BASELINE configs[3] itself -- configs[2]'s step (bf16 storage, batch 8 per GPU) on every
            # rank with the gradient buckets all-reduced from inside the backward -- nested under "also" with its own `ranks_seen`,
            # `exchange` (per-bucket all-reduce alone, exposed wait) and `step_rooflines`.  EVERY rank runs it (collectives); rank 0 reports.
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            also = {}
            try:
                r3 = measure_train(args, world, rank, dev, overrides, dtype='bf16', batch=8, steps=10, warmup=5, shared=shared)
                if r3 is not None:
                    also['configs[3]'] = r3
            except Exception as e:      # (a rank that fails here leaves the others in a collective: the launcher's timeout ends the run)
                also['error'] = '%s: %s' % (type(e).__name__, str(e)[:300])
            if out is not None:
                out['also'] = also
        if world == 1 and not args.no_cpu_baseline and out is not None:
            out['cpu_baseline'] = cpu_baseline(args.cpu_baseline_crop, agreement_dir=agreement_dir)
            if agreement_dir:
                try:
                    out['oracle_agreement'] = oracle_agreement(first, out['cpu_baseline'], agreement_dir)
                except Exception as e:
                    out['oracle_agreement'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        if agreement_dir:
            shutil.rmtree(agreement_dir, ignore_errors=True)
    if out is not None:
        # `summary`: LAST key of the line, so that the three single-GPU BASELINE configurations survive any truncation of the record
        # from the front: [value in volumes/s, ms per step, executed fraction of its roofline for the dominant kernel]
        tri = lambda o: [o['value'], o['ms_per_step'], (o.get('roofline') or {}).get('frac')] if (o and 'value' in o) else None
        summary = {}
        if not args.infer and out.get('dtype') == 'f32':
            summary['configs[1]'] = tri(out)
        for k, v in (out.get('also') or {}).items():
            if isinstance(v, dict):
                summary[k] = tri(v)
        if 'oracle_agreement' in out and out['oracle_agreement'] and 'ddice_macro' in out['oracle_agreement']:
            oa = out['oracle_agreement']
            summary['oracle_agreement'] = {k: oa[k] for k in ('dloss_rel', 'ddice_macro', 'ddice_micro', 'label_mismatch_outside_pointwise_criterion')}
        full = dict(out)
        full['summary'] = summary
        detail = None
        try:       # the complete record (what round 4 printed) goes to a file; gpurun merges gpurun_out/ back
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            detail = os.path.join('gpurun_out', 'bench_detail.json')
            with open(os.path.join(ROOT, detail), 'w') as f:
                json.dump(full, f)
        except OSError:
            detail = None
        if args.full_line:
            line = full
        else:
            line = _compact(out, 8)
            if 'also' in out:
                line['also'] = {k: (_compact(v, 5) if isinstance(v, dict) else v) for k, v in out['also'].items()}
                for name, v in line['also'].items():     # (the nested results repeat the contract boiler-plate: keep what differs)
                    if isinstance(v, dict):
                        for k in ('unit', 'n_gpus', 'ranks_seen', 'higher_is_better', 'scaling', 'vs_baseline', 'data', 'streams'):
                            if not (name == 'configs[3]' and k in ('n_gpus', 'ranks_seen', 'scaling')):
                                v.pop(k, None)
                        for k in ('measured', 'traffic_source', 'clock_ghz_in_capture', 'algorithmic_gflop_per_launch', 'achieved_algorithmic',
                                  'frac_algorithmic'):
                            (v.get('roofline') or {}).pop(k, None)
            if isinstance(line.get('oracle_agreement'), dict):
                line['oracle_agreement'] = {k: v for k, v in line['oracle_agreement'].items() if k != 'what'}
            line['detail'] = detail
            line['summary'] = summary
            line = _sig(line)
        print(json.dumps(line), flush=True)
    if parallel.active():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
