"""CPU: host-side mirror of the reference's call surface (model.py / layers.* / util.py): constructor kwargs and
defaults, lazy build, variable inventory, error behaviour, and that the product fails loudly without a GPU."""
import inspect
import os
import math

import pytest
import torch

import bts_amd  # noqa: F401
from bts_amd.layers import decoder, downsample, encoder, group_norm, resnet, upsample, vae
from bts_amd.model import Model
from bts_amd import parallel, util
from oracle import torch_ref as R


def test_constructor_defaults_match_reference_model_py():
    sig = inspect.signature(Model.__init__)
    exp = dict(data_format='channels_last', groups=8, reduction=2, l2_scale=1e-5, dropout=0.2, downsampling='conv',
               upsampling='conv', base_filters=16, depth=4, in_ch=2, out_ch=3)          # model.py:9-20
    assert {k: v.default for k, v in sig.parameters.items() if k != 'self'} == exp
    assert list(inspect.signature(resnet.ResnetBlock.__init__).parameters)[1:6] == \
        ['filters', 'data_format', 'groups', 'reduction', 'l2_scale']                   # resnet.py:9-13
    gn = inspect.signature(group_norm.GroupNormalization.__init__).parameters
    assert gn['groups'].default == 8 and gn['axis'].default == -1 and gn['epsilon'].default == 1e-5   # group_norm.py:10-13
    assert downsample.get_downsampling('conv') is downsample.ConvDownsample
    assert downsample.get_downsampling('max') is downsample.MaxDownsample
    assert downsample.get_downsampling('avg') is None                                   # downsample.py:7-11
    assert upsample.get_upsampling('conv') is upsample.ConvUpsample
    assert upsample.get_upsampling('linear') is upsample.LinearUpsample
    assert callable(vae.sample) and encoder.Encoder and decoder.Decoder


@pytest.mark.parametrize('kw,count', [(dict(base_filters=32, reduction=8), 42174773), (dict(), 10636061)])
def test_parameter_inventory(kw, count):
    m = Model(**kw)
    m.build((1, 128, 128, 128, 2))
    assert m.n_params == count
    P = R.build_params(R.default_config(**kw), (128, 128, 128))
    names = [m.oracle_name(p) for p in m.trainable_variables]
    assert sorted(names) == sorted(P.keys())
    for p in m.trainable_variables:
        assert tuple(p.t.shape) == tuple(P[m.oracle_name(p)].shape), p.name
        assert p.l2 == pytest.approx(P.l2[m.oracle_name(p)]), p.name
    # flat buffer: one contiguous, 16B-aligned range in backward-completion order (vae, decoder, encoder level 3 .. 0: SURVEY 8e); the
    # regulariser ranges are disjoint, carry the one coefficient of this configuration and cover exactly the regularised variables
    assert m.flat_params.numel() >= count and m.flat_params.numel() % 4 == 0
    rg = sorted(m._l2_ranges)
    assert all(a[0] + a[1] <= b[0] for a, b in zip(rg, rg[1:])) and all(c == 1e-5 for _, _, c in rg) and len(rg) <= 128
    assert sum(ln for _, ln, _ in rg) == sum(p.t.numel() for p in m.trainable_variables if p.l2 > 0)
    base = m.flat_params.data_ptr()
    for p in m.trainable_variables:
        o = (p.t.data_ptr() - base) // 4
        inside = any(a <= o and o + p.t.numel() <= a + ln for a, ln, _ in rg)
        assert inside == (p.l2 > 0), p.name
    first = (m.vae.out_k.t.data_ptr() - base) // 4          # the backward pass finishes the VAE's output conv first ...
    last = max(p.t.data_ptr() for p in m.encoder.levels[0][0][0].trainable_variables)
    assert first == 0 and last == max(p.t.data_ptr() for p in m.trainable_variables)      # ... and the first encoder block last
    # unproj is created last inside the VAE (vae.py:105)
    vnames = [p.name for p in m.vae.trainable_variables]
    assert vnames[-2:] == ['vae/unproj_k', 'vae/unproj_b']


def test_initialisers_follow_keras_variance_scaling():
    m = Model(base_filters=32, reduction=8)
    m.build((1, 128, 128, 128, 2))
    d = {p.name: p for p in m.trainable_variables}
    k = d['encoder/L1/B0/conv1_k'].t            # he_normal: std = sqrt(2/fan_in), fan_in = 27*32
    assert float(k.std()) == pytest.approx(math.sqrt(2.0 / (27 * 32)), rel=0.03)
    assert float(k.abs().max()) <= 2 * math.sqrt(2.0 / (27 * 32)) / 0.87962566103423978 + 1e-6
    u = d['decoder/L0/up/conv_k'].t             # glorot_uniform, (3,3,3,Cout=32,Cin=64)
    lim = math.sqrt(6.0 / (27 * 64 + 27 * 32))
    assert float(u.abs().max()) <= lim and float(u.abs().max()) > 0.95 * lim
    assert float(d['encoder/L0/B0/gn2/gamma'].t.abs().max()) == 0.0        # resnet.py:104-110
    assert float(d['encoder/L0/B0/gn1/gamma'].t.min()) == 1.0
    assert float(d['encoder/L0/B0/conv1_b'].t.abs().max()) == 0.0


def test_error_behaviour_matches_reference():
    with pytest.raises(ValueError, match='Reduction ratio'):
        resnet.ResnetBlock(filters=6, reduction=4)                                      # resnet.py:39-42
    g = group_norm.GroupNormalization(groups=8)
    with pytest.raises(ValueError, match='cannot be'):
        g.build((1, 4, 4, 4, 4))                                                        # group_norm.py:51-54
    g = group_norm.GroupNormalization(groups=3)
    with pytest.raises(ValueError, match='multiple'):
        g.build((1, 4, 4, 4, 8))                                                        # group_norm.py:56-59
    with pytest.raises(ValueError, match='data_format'):
        Model(data_format='channels_middle')
    mcf = Model(data_format='channels_first', base_filters=8, groups=2, depth=2)        # args.py:121-123 default layout
    mcf.build((1, 8, 8, 8, 2))                                                          # internal (NDHWC) build shape
    from bts_amd import ops
    blk = mcf.encoder.levels[0][0][0]
    assert blk.norm1._mode == ops.GN_CHANNEL and blk.norm2._mode == ops.GN_CHANNEL          # true GroupNorm
    assert Model(base_filters=8, groups=2, depth=2).encoder.levels[0][0][0].norm1._semantics == ops.GN_SLAB


def test_scheduled_optim_schedule():
    opt = util.ScheduledOptim(learning_rate=1e-4)
    assert opt.n_epochs == 300.0                                                        # util.py:68 (train.py:105 never overrides)
    for e in (0, 1, 150, 299):
        opt(epoch=e)
        assert opt.learning_rate == pytest.approx(1e-4 * (1.0 - e / 300.0) ** 0.9)
    opt.iterations = 1
    assert opt._lr_t() == pytest.approx(opt.learning_rate * math.sqrt(1 - 0.999) / (1 - 0.9))


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    m = Model(base_filters=4, groups=2, depth=2)
    with pytest.raises(RuntimeError, match='no CPU'):
        m(torch.zeros((1, 8, 8, 8, 2)))
    from bts_amd import ops
    with pytest.raises(RuntimeError, match='GPU'):
        ops.conv_pack(1, 0, torch.zeros((3, 3, 3, 4, 4)), 4, 4)


def test_bucket_ranges_cover_flat_buffer():
    n = 42174776
    r = parallel.bucket_ranges(n, 4, 64 << 20)
    assert r[0] == (0, 16777216) and sum(l for _, l in r) == n and all(a + l == b for (a, l), (b, _) in zip(r, r[1:]))
    assert parallel.world() == 1 and parallel.l2_grad_scale() == 1.0


def test_grad_sync_bucket_bookkeeping():
    """parallel.GradSync (SURVEY 8e, C1 overlapped with the backward): buckets tile the flat gradient buffer, every
    parameter sits in the bucket(s) its span overlaps, and a bucket's pending count is its member count."""
    m = Model(base_filters=8, groups=2, depth=3)
    m.build((1, 16, 16, 16, 2))
    gs = parallel.GradSync(m, bucket_bytes=1 << 14)
    n = m.flat_grads.numel()
    covered = 0
    for off, ln, members in gs.buckets:
        assert off == covered
        covered += ln
        for pid in members:
            o, l = gs.spans[pid]
            assert o < off + ln and o + l > off
    assert covered == n and len(gs.buckets) > 8
    for p in m.trainable_variables:
        o, l = gs.spans[id(p)]
        nb = len(gs.by_param[id(p)])
        assert nb >= 1 and nb == len([1 for off, ln, _ in gs.buckets if o < off + ln and o + l > off])
    assert sum(l for _, l in gs.spans.values()) == m.n_params


def test_package_synthetic_batch_is_the_oracles():
    """bench.py draws its volumes from the package (bts_amd.data.synthetic_batch); the parity tests draw theirs from the
    oracle's generator: same seeds must give the same tensors"""
    from bts_amd.data import synthetic_batch
    from oracle import torch_ref as R
    for n, crop, latent, seed in ((1, (16, 16, 16), 128, 1234), (2, (8, 16, 8), 4, 77)):
        a = synthetic_batch(n, crop, latent=latent, seed=seed)
        b = R.synthetic_batch(n, crop, latent=latent, seed=seed)
        assert all(torch.equal(u, v) for u, v in zip(a, b))


def _bench(args, env_extra, drop=()):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if not k.startswith('BTS_') and k not in drop}
    env.update(env_extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + args, env=env, cwd=root, capture_output=True,
                          text=True, timeout=300)


def test_bench_refuses_world_size_mismatch_and_overrides():
    """`--gpus N` used to be ignored.  As a rank (WORLD_SIZE set) a mismatch is refused; A/B switches in the
    environment are refused unless explicitly allowed; as a launcher with no GPU it fails loudly instead of falling back"""
    r = _bench(['--gpus', '2'], {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr
    r = _bench(['--gpus', '1'], {'WORLD_SIZE': '4', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=4' in r.stderr
    r = _bench([], {'BTS_WINO': '0'}, drop=('WORLD_SIZE',))
    assert r.returncode != 0 and 'BTS_WINO' in r.stderr and '--allow-overrides' in r.stderr
    if not torch.cuda.is_available():
        r = _bench(['--gpus', '2'], {}, drop=('WORLD_SIZE', 'RANK', 'LOCAL_RANK'))
        assert r.returncode != 0 and 'GPU' in r.stderr


def test_keras_order_weight_list_round_trip(tmp_path):
    """importer keyed by the reference's variable order -- get_weights()/set_weights() in `model.weights` order
    (trainable variables in tracking order, `vae.unproj` last, then `epoch`), Keras layouts, shape errors name the variable"""
    import numpy as np
    m = Model(base_filters=4, groups=2, reduction=2, depth=2)
    m.build((1, 8, 8, 8, 2))
    names = [p.name for p in m.trainable_variables]
    assert names[0] == 'encoder/L0/B0/ptwise_k' and names[-2:] == ['vae/unproj_k', 'vae/unproj_b']
    assert names.index('decoder/out_k') < names.index('vae/down/conv_k') < names.index('vae/proj_k') < names.index('vae/out_k')
    i = names.index('encoder/L0/B0/gn1/gamma')
    assert names[i + 1] == 'encoder/L0/B0/gn1/beta' and names[i - 2:i] == ['encoder/L0/B0/conv1_k', 'encoder/L0/B0/conv1_b']
    g = torch.Generator().manual_seed(2)
    for p in m.trainable_variables:
        p.t.copy_(torch.randn(p.t.shape, generator=g))
    m.epoch.assign(17)
    w = m.get_weights()
    assert len(w) == len(names) + 1 and int(w[-1]) == 17
    assert w[names.index('decoder/L0/up/conv_k')].shape == (3, 3, 3, 4, 16)       # transposed conv: (kd,kh,kw,Cout,Cin)
    np.savez(str(tmp_path / 'keras.npz'), *w)
    m2 = Model(base_filters=4, groups=2, reduction=2, depth=2)
    m2.build((1, 8, 8, 8, 2))
    m2.load_weights_keras_order(str(tmp_path / 'keras.npz'))
    assert all(torch.equal(a.t, b.t) for a, b in zip(m.trainable_variables, m2.trainable_variables))
    assert int(m2.epoch.value().numpy()) == 17
    bad = list(w[:-1])
    bad[3] = bad[3].T.copy()
    with pytest.raises(ValueError, match='se_w2'):
        m2.set_weights(bad)
    with pytest.raises(ValueError):
        m2.set_weights(w[:5])


def test_bench_line_helpers_on_the_committed_captures():
    """bench.py's roofline / kernel_breakdown / step_rooflines helpers against the committed PMC captures (no GPU): every configuration
    of the default line has a traffic table with request-size counters, the dominant kernels resolve to a per-launch byte count, the
    read side is derived from the request sizes (not the blanket 2 x FETCH_SIZE), and the wasted-traffic ratio is a finite number > 1"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for sfx, sym in (('', 'w3_kernel'), ('_bf16_b8', 'lp_s1d_kernel<1,5>'), ('_infer_f16', 'lp_s1d_kernel<1,4>')):
        tab, src = bench._traffic_table(sfx)
        assert tab is not None and tab.get('_meta', {}).get('steps_in_capture'), (sfx, src)
        tr = bench.pmc_traffic(sym, sfx)
        assert tr is not None and tr['bytes'] > 0 and tr['read_side'].startswith('request counters'), (sfx, sym, tr)
        # all requests are 128-byte ones: request counters and 2 x FETCH_SIZE agree (the two come from separate profiler passes, i.e. separate
        # runs).  The bf16 name covers two instantiations, the plain and the fused-shortcut form; the latter's re-reads of the shortcut's rows
        # depend on what its L2 happens to hold, 12-17 % apart between the two passes of one capture -- the plain form is the yardstick
        chk = bench.pmc_traffic('lp_s1d_kernel<1,5,false>', sfx) if sfx == '_bf16_b8' else tr
        if sfx == '_bf16_b8':
            assert tr['variants_in_capture'] == 2 and chk['variants_in_capture'] == 1, (tr, chk)
        assert abs(chk['read_bytes'] - 2.0 * chk['fetch_size_kib_raw'] * 1024.0) <= 0.03 * chk['read_bytes']
    # records of two fake launches -> the contract's objects
    rec = [('lp_s1d_kernel<1,5>', 4.0e11, 0.30), ('lp_s1d_kernel<1,5>', 4.0e11, 0.34), ('lp_k1_kernel', 1.0e9, 0.10)]
    out = bench._roofline_from_records(rec, 1, 1e-3, lambda s: bench.pmc_traffic(s, '_bf16_b8'), 'test')
    r = out['roofline']
    assert r['kernel'] == 'lp_s1d_kernel<1,5>' and r['peak'] == bench.PEAK_F16_MFMA_TFLOPS and 0.4 < r['frac'] < 0.6 and r['traffic'] > 0
    assert out['kernel_breakdown']['lp_k1_kernel']['bound'] == 'hbm' and out['kernel_breakdown']['lp_s1d_kernel<1,5>']['bound'] == 'mfma'
    sr = bench.step_rooflines(52.136, 127.2, 0.0765, 'bf16', '_bf16_b8')
    assert 1.0 < sr['wasted_traffic_ratio'] < 4.0 and 0.2 < sr['mfma_frac'] < 0.35 and sr['read_side'] == ['request counters']


def test_pmc_aggregate_takes_the_last_full_step(tmp_path):
    """scripts/pmc_aggregate.py on a synthetic three-step capture: one-time launches of the first step (weight packs) and a cold first step must
    not leak into the per-step figure -- `steady` holds the launches between the last two Adam dispatches and their own counter means, and
    bench.py's step total is built from it"""
    import csv
    import importlib.util
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cols = ['Dispatch_Id', 'Kernel_Name', 'Counter_Name', 'Counter_Value', 'Start_Timestamp', 'End_Timestamp']
    # step 1: 3 packs (once) + conv (cold: 200 KiB) + adam; steps 2, 3: pack_batch + conv (100 KiB) + adam
    seq = [('lp_pack_kernel<T>(P)', 50.0)] * 3 + [('conv_kernel(P)', 200.0), ('adam_kernel(float*)', 10.0)]
    seq += [('lp_pack_batch_kernel<T>(P)', 30.0), ('conv_kernel(P)', 100.0), ('adam_kernel(float*)', 10.0)] * 2
    for sfx in ('', '_bf16_b8', '_infer_f16'):
        for name, ctrs in (('fetch', ['FETCH_SIZE']), ('write', ['WRITE_SIZE', 'GRBM_GUI_ACTIVE']),
                           ('req', ['TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum', 'TCC_EA0_RDREQ_64B_sum', 'TCC_EA0_RDREQ_128B_sum'])):
            d = tmp_path / ('pmc_T%s_%s' % (sfx, name))
            d.mkdir()
            with open(d / ('%s_counter_collection.csv' % name), 'w', newline='') as f:
                w = csv.DictWriter(f, fieldnames=cols)
                w.writeheader()
                for i, (k, kib) in enumerate(seq):
                    if sfx == '_infer_f16' and 'adam' in k:
                        continue                                # a forward has no optimiser step: no steady segment
                    for c in ctrs:
                        val = {'FETCH_SIZE': kib / 2.0, 'WRITE_SIZE': kib / 4.0, 'GRBM_GUI_ACTIVE': 8000.0, 'TCC_EA0_RDREQ_sum': kib * 8.0,
                               'TCC_EA0_RDREQ_32B_sum': 0.0, 'TCC_EA0_RDREQ_64B_sum': 0.0, 'TCC_EA0_RDREQ_128B_sum': kib * 8.0}[c]
                        w.writerow({'Dispatch_Id': i + 1, 'Kernel_Name': k, 'Counter_Name': c, 'Counter_Value': val,
                                    'Start_Timestamp': 1000 * i, 'End_Timestamp': 1000 * i + 500})
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'pmc_aggregate.py'), str(tmp_path), 'T'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    tab = json.load(open(tmp_path / 'T_pmc_traffic_bf16_b8.json'))
    assert tab['_meta']['steady_step_launches'] == 3
    assert 'steady' not in tab['lp_pack_kernel<T>'] and tab['lp_pack_kernel<T>']['launches'] == 3
    assert tab['conv_kernel']['launches'] == 3 and tab['conv_kernel']['steady']['launches'] == 1
    assert abs(tab['conv_kernel']['FETCH_SIZE_KiB_mean'] - (100 + 50 + 50) / 3.0) < 1e-9 and tab['conv_kernel']['steady']['FETCH_SIZE_KiB_mean'] == 50.0
    assert 'steady_step_launches' not in json.load(open(tmp_path / 'T_pmc_traffic_infer_f16.json'))['_meta']
    # bench.py's step total from such a table: the steady step only; everything else in the capture is "outside the step loop"
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench._traffic_table = lambda suffix='': (tab, 'synthetic')
    sr = bench.step_rooflines(1.0, 1.0, 1.0, 'bf16', '_bf16_b8')
    kib = 1024.0
    steady = (128.0 * (30 + 100 + 10) * 8.0) + (30 + 100 + 10) / 4.0 * kib        # request-counter reads + WRITE_SIZE writes of one step
    assert abs(sr['measured_hbm_gb_per_step'] * 1e9 - steady) < 1e-3 * steady, (sr, steady)
    assert sr['outside_the_step_loop_gb'] > 0
