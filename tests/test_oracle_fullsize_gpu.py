"""-m gpu: the engine against the ORACLE at the sizes the claims are made on (north_star: "Dice within 1e-4 of the reference on a
fixed seed" on 2ch x 128^3, bit-exact argmax label map), not against itself:

  * BASELINE configs[1]: one full train step of the CLI-default model on one 2ch x 128^3 volume (train.py:140-152: forward
    with training=True, Dice + 0.1*MSE + 0.1*KL + L2 loss, Dice metric, gradients, TF-form Adam) -- fp32 engine vs
    oracle/torch_ref.py evaluated in FP64 on the GPU box's host cores (measured there: 121 s on 16 threads, 82 GB), with the
    oracle's own fp32 evaluation (20 s) as the conditioning yardstick, exactly as tests/test_model_gpu.py does at <= 64^3;
  * BASELINE configs[2]'s engine (bf16 storage) at the same size and on the same oracle evaluation, batch 1 (the batch-8 plan
    is held to the fp32 engine in tests/test_lowp_fullsize_gpu.py): what 16-bit storage costs against the REFERENCE arithmetic;
  * BASELINE configs[4]: the 155x190x147 volume zero-padded to 160x192x160 (test.py:164-178), inference=True (model.py:63-68)
    -- fp32 engine and fp16 engine vs the oracle's fp64 forward (79 s).

Synthetic batch seed 1234, injected dropout mask and eps, every gamma / beta / bias randomised (gamma_2 = 0 at init would hide
the conv branch of every block: SURVEY F6).  The oracle evaluations are shared by the tests of this module.

Tolerances (SURVEY 8c / DESIGN 4), fp32 engine: y_pred max-abs <= 1e-4; loss <= 1e-5 relative; macro / micro Dice <= 1e-4; label
map identical outside the margin band the y_pred tolerance implies (|p - 0.5| < 1e-4 or top-2 gap < 1e-4), differing voxels counted
and <= 1e-3 of the volume; every variable's gradient <= 1e-3 of its max-abs,
or 4x what torch-fp32 deviates from fp64 on that variable, or the worst such deviation over all variables, when those are larger;
whole gradient in relative L2 <= max(1e-4, 2x torch-fp32's); parameters after Adam from the engine's own gradient in fp64.
16-bit engines: the bounds their small-size oracle tests use (tests/test_lowp_gpu.py, tests/test_lowp_train_gpu.py), stated in
the tests below."""
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as R  # noqa: E402

CLI = dict(base_filters=32, groups=8, reduction=8, depth=4)
CROP = (128, 128, 128)
# torch-CPU threads for the oracle: measured on the GPU box (EPYC 9575F, 256 hardware threads) for the 128^3 train step --
# 16 threads: fp32 20.5 s / fp64 121 s; 32: 21 / 132; 64: 26 / 150; 128: 46 / 199 (the fp64 convs have no oneDNN path and stop scaling early)
ORACLE_THREADS = int(os.environ.get('BTS_TEST_ORACLE_THREADS', '16'))


def _randomised_params(cfg, crop, seed):
    """oracle ParamSet with every gamma / beta / bias randomised, values fp32-representable (both sides see identical inputs)"""
    P = R.build_params(cfg, crop, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in P:
        if k.endswith('_b'):
            P[k] = torch.randn(P[k].shape, generator=g, dtype=torch.float64) * 0.1
        if k.endswith('_g'):
            P[k] = 1.0 + torch.randn(P[k].shape, generator=g, dtype=torch.float64) * 0.3
    for k in P:
        P[k] = P[k].float().double()
    return P


class _Threads(object):
    def __enter__(self):
        self.saved = torch.get_num_threads()
        torch.set_num_threads(max(1, min(ORACLE_THREADS, os.cpu_count() or 1)))

    def __exit__(self, *a):
        torch.set_num_threads(self.saved)


def _oracle_step(cfg, P, x, y, mask, eps, dtype):
    x, y, mask, eps = x.to(dtype), y.to(dtype), mask.to(dtype), eps.to(dtype)
    leaves = {k: t.clone().to(dtype).requires_grad_(True) for k, t in P.items()}
    PP = R.ParamSet()
    PP.update(leaves)
    PP.l2 = P.l2
    out = R.model(x, PP, cfg, training=True, inference=False, mask=mask, eps=eps)
    loss = R.dice_vae_loss(x, y, *out, cfg['data_format']) + R.l2_regularisation(PP)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return tuple(o.detach() for o in out), loss.detach(), dict(zip(leaves.keys(), (g.detach() for g in grads)))


@pytest.fixture(scope='module')
def train_case():
    """the fp64 oracle step (reference) and the fp32 oracle step (yardstick) at 128^3, evaluated once"""
    cfg = R.default_config(**CLI)
    x, y, mask, eps = R.synthetic_batch(1, CROP, latent=128, seed=1234)
    P = _randomised_params(cfg, CROP, seed=7)
    with _Threads():
        t0 = time.time()
        _, _, g32 = _oracle_step(cfg, P, x, y, mask, eps, torch.float32)
        t1 = time.time()
        out, loss, g64 = _oracle_step(cfg, P, x, y, mask, eps, torch.float64)
        t2 = time.time()
    print('oracle 128^3 train step: fp32 %.1f s, fp64 %.1f s on %d threads' % (t1 - t0, t2 - t1, min(ORACLE_THREADS, os.cpu_count() or 1)))
    macro, micro, labels = R.dice_coefficient(y.double(), out[0], cfg['data_format'])
    return dict(cfg=cfg, x=x, y=y, mask=mask, eps=eps, P=P, out=out, loss=float(loss), g64=g64, g32=g32,
                macro=float(macro), micro=float(micro), labels=labels)


def _margin(yp):
    """the oracle's own decision margin per voxel (util.py:36-44: argmax, then the > 0.5 cut): distance of the winning probability from
    0.5, or of the two largest probabilities from each other, whichever is smaller"""
    top2 = yp.topk(2, dim=-1).values
    return torch.minimum((top2[..., 0] - 0.5).abs(), (top2[..., 0] - top2[..., 1]).abs())


def _label_check(differ, y_engine, yp, what):
    """The POINTWISE label-map criterion (round-5 review, weak 1(i)): a voxel's label may differ from the oracle's only where the oracle's
    own margin there is at most twice the engine's |y_pred - oracle| AT THAT VOXEL (an argmax flip needs the top-2 gap <= |d_i| + |d_j|,
    a threshold flip |p - 0.5| <= |d|), and no more voxels may differ than sit inside the 1e-5 band of SURVEY 8c.  -> (differing, outside
    the pointwise criterion, voxels inside the 1e-5 band)"""
    d = (y_engine.detach().double().cpu() - yp).abs().max(dim=-1).values
    margin = _margin(yp)
    outside = differ & (margin > 2.0 * d)
    n_diff, n_out, n_band5 = int(differ.sum()), int(outside.sum()), int((margin < 1e-5).sum())
    worst = float(margin[differ].max()) if n_diff else 0.0
    print('%s: %d of %d labels differ from the oracle\'s; %d of them outside the pointwise criterion (margin <= 2 |dp| at the voxel); largest '
          'oracle margin among them %.2e; voxels inside the 1e-5 band %d, inside a 1e-4 band %d' %
          (what, n_diff, differ.numel(), n_out, worst, n_band5, int((margin < 1e-4).sum())))
    return n_diff, n_out, n_band5


def _maxerr(a, b):
    return float((a.detach().double().cpu() - b.detach().double()).abs().max())


def test_fp32_train_step_at_128_against_the_fp64_oracle(train_case):
    import bts_amd  # noqa: F401
    from bts_amd.model import Model
    from bts_amd.tape import GradientTape
    from bts_amd.util import DiceCoefficient, DiceVAELoss, ScheduledOptim, reduce_sum
    c = train_case
    model = Model(**CLI)
    model.build((1,) + CROP + (2,))
    assert model.n_params == sum(t.numel() for t in c['P'].values()) == 42174773
    model.set_weights_from(c['P'])
    model.encoder.set_dropout_mask(c['mask'])
    model.vae.set_eps(c['eps'])
    loss_fn, dice_fn = DiceVAELoss(), DiceCoefficient()
    from bts_amd import ops
    ops.profile_enable(True)          # launch records: which conv forms this step took (the 128^3 dispatch, not the small-grid one)
    with GradientTape() as tape:
        y_pred, y_vae, z_mean, z_logvar = model(c['x'], training=True, inference=False)
        loss = loss_fn(c['x'], c['y'], y_pred, y_vae, z_mean, z_logvar)
        loss = loss + reduce_sum(model.losses)
    macro, micro = dice_fn(c['y'], y_pred)
    grads = tape.gradient(loss, model.trainable_variables)
    torch.cuda.synchronize()
    ops.profile_enable(False)
    forms = sorted(set(r[0] for r in ops.profile_records(detail=True)))
    print('conv forms of this step:', forms)
    assert 'w3_kernel' in forms and 'wgw_kernel' in forms, 'the 128^3 step did not take the product conv forms'
    yp_r, yv_r, zm_r, zl_r = c['out']
    e = _maxerr(y_pred.t, yp_r)
    ev = _maxerr(y_vae.t, yv_r)
    print('128^3 vs fp64 oracle: y_pred max |d| %.2e, y_vae max |d| %.2e (|y_vae| max %.2f), loss %.8f vs %.8f, macro %.6f vs %.6f, '
          'micro %.6f vs %.6f' % (e, ev, float(yv_r.abs().max()), float(loss), c['loss'], float(macro), c['macro'], float(micro), c['micro']))
    assert e <= 1e-4, 'y_pred max-abs err %.3e' % e
    assert ev <= 1e-4 * max(1.0, float(yv_r.abs().max())), 'y_vae err %.3e' % ev
    assert _maxerr(z_mean.t, zm_r) <= 1e-4 and _maxerr(z_logvar.t, zl_r) <= 1e-4
    assert abs(float(loss) - c['loss']) <= 1e-5 * max(1.0, abs(c['loss'])), (float(loss), c['loss'])
    # label map: the pointwise criterion (no blanket band), and no more differing voxels than the 1e-5 band of SURVEY 8c holds
    lab = dice_fn.last_labels.cpu().long()
    differ = lab != c['labels'].long()
    n_diff, n_out, n_band5 = _label_check(differ.reshape(yp_r.shape[:-1]), y_pred.t, yp_r, 'label map at 128^3')
    assert n_out == 0, 'argmax label map differs from the oracle at %d voxels whose margin exceeds twice the local y_pred error' % n_out
    assert n_diff <= n_band5, (n_diff, n_band5)
    assert abs(float(macro) - c['macro']) <= 1e-4 and abs(float(micro) - c['micro']) <= 1e-4
    # gradients: the rule of tests/test_model_gpu.py, with the oracle's own fp32 evaluation as the yardstick
    g64, g32 = c['g64'], c['g32']
    gdev = max(_maxerr(g32[k], g64[k]) / (float(g64[k].abs().max()) + 1e-12) for k in g64)
    rows = []
    for p, g in zip(model.trainable_variables, grads):
        assert g is not None, p.name
        gr = g64[model.oracle_name(p)]
        scale = float(gr.abs().max()) + 1e-12
        err = _maxerr(g, gr) / scale
        dev32 = _maxerr(g32[model.oracle_name(p)], gr) / scale
        rows.append((err, p.name, dev32, scale, _maxerr(g, gr)))
    rows.sort(reverse=True)
    for err, name, dev32, scale, _ in rows[:8]:
        print('gradient %-34s rel err %.3e (torch-fp32 deviates %.3e; max-abs %.3e)' % (name, err, dev32, scale))
    print('worst torch-fp32 deviation over all variables %.3e; variables above 1e-3: engine %d, torch-fp32 %d of %d' %
          (gdev, sum(r[0] > 1e-3 for r in rows), sum(r[2] > 1e-3 for r in rows), len(rows)))
    for err, name, dev32, scale, ae in rows:
        assert err <= max(1e-3, 4 * dev32, gdev) or ae <= 1e-9, 'grad %s rel err %.3e (fp32-torch deviates %.3e)' % (name, err, dev32)
    num = sum(float((g.detach().double().cpu() - g64[model.oracle_name(p)]).pow(2).sum())
              for p, g in zip(model.trainable_variables, grads)) ** 0.5
    num32 = sum(float((g32[k].double() - g64[k]).pow(2).sum()) for k in g64) ** 0.5
    den = sum(float(g64[k].pow(2).sum()) for k in g64) ** 0.5
    print('whole-gradient relative L2 error at 128^3: engine %.3e, torch-fp32 %.3e' % (num / den, num32 / den))
    assert num / den <= max(1e-4, 2.0 * num32 / den)
    # one TF-form Adam step: the kernel on the engine's gradient (in fp64), and the parameters next to the oracle's own update
    opt = ScheduledOptim(learning_rate=1e-4)
    opt(epoch=0)
    before = {p.name: p.t.detach().cpu().double().clone() for p in model.trainable_variables}
    opt.apply_gradients(zip(grads, model.trainable_variables), model=model)
    torch.cuda.synchronize()
    flipped = total = 0
    for p, g in zip(model.trainable_variables, grads):
        ge = g.detach().cpu().double()
        exp, _, _ = R.adam_tf_step(before[p.name], ge, torch.zeros_like(ge), torch.zeros_like(ge), 1, 1e-4)
        got = p.t.detach().cpu().double()
        assert float((got - exp).abs().max()) <= 1e-7 + 1e-6 * float(exp.abs().max()), p.name
        # against the ORACLE's update: Adam's first step is lr * g / (|g| + 3e-6) ~ lr * sign(g); only elements whose gradient is
        # within rounding of zero may land elsewhere, and then by at most 2 lr
        gr = g64[model.oracle_name(p)]
        exp_r, _, _ = R.adam_tf_step(before[p.name], gr, torch.zeros_like(gr), torch.zeros_like(gr), 1, 1e-4)
        d = (got - exp_r).abs()
        assert float(d.max()) <= 2.0 * 1e-4 * 1.0001 + 1e-6 * float(exp_r.abs().max()), p.name
        flipped += int((d > 0.5e-4).sum())
        total += d.numel()
    print('parameters after Adam: %d of %d more than lr/2 away from the oracle\'s update (gradient elements within rounding of zero)' % (flipped, total))
    assert flipped <= 1e-2 * total


def _lowp_step_vs_oracle(c, dtype):
    """one step of the 16-bit engine on the 128^3 fixture -> (loss rel, |d macro|, |d micro|, label mismatch rate, gradient rel L2, cosine)"""
    from bts_amd.lowp_train import LowPrecisionTrainer
    from bts_amd.model import Model
    from bts_amd.util import DiceCoefficient, ScheduledOptim
    model = Model(**CLI)
    model.build((1,) + CROP + (2,))
    model.set_weights_from(c['P'])
    model.encoder.set_dropout_mask(c['mask'])
    model.vae.set_eps(c['eps'])
    opt = ScheduledOptim(1e-4)
    opt(epoch=0)
    tr = LowPrecisionTrainer(model, dtype)
    df = DiceCoefficient()
    loss, macro, micro = tr.step(opt, df, c['x'], c['y'])
    torch.cuda.synchronize()
    tr.settle()
    assert tr.skipped_steps == 0, 'the loss-scaled step overflowed'
    g64 = c['g64']
    flat = model.flat_grads
    inv = 1.0 / tr.last_grad_scale           # (fp16: the buffer keeps the loss scale, Adam un-scales as it reads)
    num = den = dot = n2 = 0.0
    for p in model.trainable_variables:
        off = (p._gview.data_ptr() - flat.data_ptr()) // 4
        a = flat[off:off + p._gview.numel()].detach().cpu().double().reshape(-1) * inv
        b = g64[model.oracle_name(p)].reshape(-1)
        num += float((a - b).pow(2).sum())
        den += float(b.pow(2).sum())
        dot += float(torch.dot(a, b))
        n2 += float(a.pow(2).sum())
    rel, cos = (num / den) ** 0.5, dot / (den ** 0.5 * n2 ** 0.5)
    lab = df.last_labels.cpu().long()
    mism = float((lab != c['labels'].long()).float().mean())
    dl = abs(float(loss) - c['loss']) / abs(c['loss'])
    print('%s storage at 128^3 vs fp64 oracle: loss %.6f vs %.6f (rel %.2e), macro Dice %.5f vs %.5f, micro %.5f vs %.5f, label changes '
          '%.4f %%, gradient rel L2 %.3e cosine %.6f' % (dtype, float(loss), c['loss'], dl, float(macro), c['macro'], float(micro), c['micro'],
                                                       100 * mism, rel, cos))
    del tr, model
    torch.cuda.empty_cache()
    return dl, abs(float(macro) - c['macro']), abs(float(micro) - c['micro']), mism, rel, cos


def test_16bit_train_step_at_128_against_the_fp64_oracle(train_case):
    """BASELINE configs[2]'s engine on one 128^3 volume vs the REFERENCE arithmetic: what 16-bit storage moves.  bf16 (the benched type) and
    fp16 (loss-scaled, 11-bit mantissa) side by side on the SAME oracle evaluation, so that bf16's gradient error has a yardstick: the
    regression guards are per type, and the statement that matters is their ratio (DESIGN section 4 / INTEGRATION: choosing compute_dtype)"""
    import bts_amd  # noqa: F401
    c = train_case
    b = _lowp_step_vs_oracle(c, 'bfloat16')
    f = _lowp_step_vs_oracle(c, 'float16')
    print('gradient relative L2 error vs the fp64 oracle at 128^3: bf16 %.3e, fp16 %.3e (ratio %.1f); cosine %.5f / %.5f; label changes %.3f %% / %.3f %%'
          % (b[4], f[4], b[4] / max(f[4], 1e-30), b[5], f[5], 100 * b[3], 100 * f[3]))
    for name, r, lim in (('bfloat16', b, dict(l2=0.2, cos=0.98)), ('float16', f, dict(l2=0.05, cos=0.999))):
        dl, dma, dmi, mism, rel, cos = r
        assert dl <= 5e-3 and dma <= 5e-3 and dmi <= 5e-3 and mism <= 1e-2, (name, r)
        # measured (round 5, bf16): gradient rel L2 0.152, cosine 0.9888, loss rel 6.4e-4, label changes 0.43 % -- one 128^3 volume of an
        # untrained, randomly re-scaled net, 8-bit mantissas on every stored activation and activation gradient (the 32^3 case against the
        # fp32 engine: 0.12 / 0.99, tests/test_lowp_train_gpu.py).  fp16 carries three more mantissa bits: measured (round 6) gradient rel L2
        # 0.0325 (4.7x smaller), cosine 0.99947, loss rel 1.3e-5, label changes 0.061 % -- its bound is the tighter one, and it must not
        # be WORSE than bf16
        assert rel <= lim['l2'] and cos >= lim['cos'], (name, rel, cos)
    assert f[4] <= b[4]


@pytest.fixture(scope='module')
def infer_case():
    cfg = R.default_config(**CLI)
    P = _randomised_params(cfg, CROP, seed=11)      # the weights belong to the training crop (the VAE is tied to it, vae.py:101-111)
    g = torch.Generator().manual_seed(4)
    x = torch.randn((1, 160, 192, 160, 2), generator=g)
    x[:, 155:] = 0                                   # the zero padding of test.py:164-178
    x[:, :, 190:] = 0
    x[:, :, :, 147:] = 0
    with _Threads(), torch.no_grad():
        t0 = time.time()
        yp = R.model(x.double(), P, cfg, training=False, inference=True)[0]
        print('oracle 160x192x160 forward in fp64: %.1f s' % (time.time() - t0))
    return dict(P=P, x=x, yp=yp)


def _engine_model(P):
    from bts_amd.model import Model
    m = Model(**CLI)
    m.build((1,) + CROP + (2,))
    m.set_weights_from(P)
    return m


def test_fp32_full_volume_forward_against_the_fp64_oracle(infer_case):
    import bts_amd  # noqa: F401
    c = infer_case
    m = _engine_model(c['P'])
    y_pred, a, b, d = m(c['x'].cuda(), training=False, inference=True)
    torch.cuda.synchronize()
    assert a is None and b is None and d is None and y_pred.shape == (1, 160, 192, 160, 3)
    yp = c['yp']
    e = _maxerr(y_pred.t, yp)
    yh = y_pred.t.cpu().double()
    arg_d = yh.argmax(-1) != yp.argmax(-1)
    thr_d = (yh.max(-1).values > 0.5) != (yp.max(-1).values > 0.5)
    print('160x192x160 fp32 engine vs fp64 oracle: y_pred max |d| %.2e; argmax differs at %d voxels, threshold at %d' % (e, int(arg_d.sum()), int(thr_d.sum())))
    assert e <= 1e-4
    n_diff, n_out, n_band5 = _label_check(arg_d | thr_d, yh, yp, 'label map at 160x192x160')
    assert n_out == 0 and n_diff <= n_band5, (n_diff, n_out, n_band5)


def test_fp16_full_volume_forward_against_the_fp64_oracle(infer_case):
    """BASELINE configs[4] as benched (fp16 storage): |dp| and the label-map mismatch rate against the reference arithmetic"""
    import bts_amd  # noqa: F401
    from bts_amd import lowp
    c = infer_case
    m = _engine_model(c['P'])
    ylp = lowp.LowPrecisionForward(m, 'float16')(c['x'].cuda())
    torch.cuda.synchronize()
    yp = c['yp']
    yh = ylp.cpu().double()
    d = (yh - yp).abs()

    def labels(p):      # test.py:259-261 form: argmax + 1 where the winning probability clears 0.5, else background
        return torch.where(p.max(-1).values > 0.5, p.argmax(-1) + 1, torch.zeros_like(p.argmax(-1)))
    mism = float((labels(yh) != labels(yp)).float().mean())
    print('160x192x160 fp16 engine vs fp64 oracle: |dp| max %.3e mean %.3e, label changes %.4f %%' % (float(d.max()), float(d.mean()), 100 * mism))
    assert float(d.max()) <= 5e-2 and float(d.mean()) <= 2e-3 and mism <= 5e-3
