"""DiceVAELoss, DiceCoefficient, ScheduledOptim -- drop-in for util.py of the reference (:7-24, :29-57, :60-84),
plus train_step(), the body of the reference's training loop (train.py:140-152)."""
import math

import torch

from . import ops, parallel
from .tape import GradientTape, Tensor, as_tensor, bump_weights_epoch, current_tape


class DiceVAELoss(object):
    """loss = mean_c(1-(2I_c+1)/(P_c+T_c+1)) + 0.1*mean((x-y_vae)^2) + 0.1*mean(mu^2+exp(lv)-lv-1)   (util.py:13-24).
    I, P, T are summed over the batch axis too (util.py:11,18-20); under data parallelism the raw sums are all-reduced
    so every rank evaluates the global-batch loss and its exact gradient."""

    def __init__(self, name='custom_loss', data_format='channels_last', **kwargs):
        if data_format not in ('channels_last', 'channels_first'):
            raise ValueError('unknown data_format %r' % (data_format,))
        self.data_format = data_format
        # util.py:11 -- batch + spatial axes of the public layout; the sums are the same numbers either way
        self.axis = (0, 1, 2, 3) if data_format == 'channels_last' else (0, 2, 3, 4)

    def __call__(self, x, y, y_pred, y_vae, z_mean, z_logvar, sample_weight=None):
        x, y = as_tensor(x, data_format=self.data_format), as_tensor(y, data_format=self.data_format)
        y_pred, y_vae = as_tensor(y_pred, True), as_tensor(y_vae, True)
        pbase = getattr(z_mean, 'base', None)
        if pbase is None or pbase is not getattr(z_logvar, 'base', None):
            raise ValueError('z_mean / z_logvar must be the two halves of the VAE projection returned by Model')
        proj = pbase.t
        c = y_pred.shape[-1]
        sums = ops.loss_sums(y_pred.t, y.t, x.t, y_vae.t, proj)
        parallel.all_reduce_sum(sums)                     # C3: 3*out_ch + 4 doubles, no-op on one rank
        lt, parts = ops.loss_value(sums, c, True)
        loss = Tensor(lt)
        self.last_parts = parts
        tape = current_tape()
        if tape is not None:
            def backward():
                g = loss.grad
                if g is None:
                    return
                dyp = torch.empty(y_pred.shape, dtype=torch.float32, device=g.device)
                dyv = torch.empty(y_vae.shape, dtype=torch.float32, device=g.device)
                dpr = torch.empty_like(proj)
                ops.loss_bwd(y_pred.t, y.t, x.t, y_vae.t, proj, sums, g, dyp, dyv, dpr, through_sigmoid=False)
                for src, d in ((y_pred, dyp), (y_vae, dyv)):
                    if src.requires_grad:
                        buf, acc = src.grad_slot()
                        ops.add_strided(buf, d, acc)
                ops.axpy(pbase.grad_full(), dpr, 1.0)
            tape.record(backward)
        return loss


class DiceCoefficient(object):
    """Hard Dice metric (util.py:35-57).  channels_last reduces axes (0,1,2) of the 5-D tensors, i.e. one Dice cell per
    (last spatial index, class) -- reproduced as is.  Returns (macro, micro) as 1-element Tensors; the
    argmax label map of the last call is kept in .last_labels (uint8, 0 = below threshold, k+1 = class k)."""

    def __init__(self, name='dice_coefficient', data_format='channels_last'):
        if data_format not in ('channels_last', 'channels_first'):
            raise ValueError('unknown data_format %r' % (data_format,))
        self.name = name
        self.data_format = data_format
        self.last_labels = None

    def __call__(self, y_true, y_pred):
        y_true, y_pred = as_tensor(y_true, data_format=self.data_format), as_tensor(y_pred, data_format=self.data_format)
        w, c = y_pred.shape[3], y_pred.shape[4]
        # channels_last reduces axes (0,1,2) only -> one cell per (last spatial index, class);
        # channels_first reduces (0,2,3,4): one cell per class, the intended metric (util.py:36)
        cl = self.data_format == 'channels_last'
        table, labels = ops.dice_metric_sums(y_true.t, y_pred.t, cl, True)
        parallel.all_reduce_sum(table)
        out = ops.dice_metric_value(table, w, c, cl)
        self.last_labels = labels
        return Tensor(out[0:1], requires_grad=False), Tensor(out[1:2], requires_grad=False)


class ScheduledOptim(object):
    """Keras Adam with a per-epoch polynomial schedule (util.py:60-84): optimizer(epoch) sets
    lr = init_lr*(1-epoch/n_epochs)**0.9; apply_gradients() is the TF-form update (epsilon un-corrected),
    fused over the model's flat parameter buffer when the variables are exactly a Model's."""

    def __init__(self, learning_rate=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, name='Adam',
                 n_epochs=300, **kwargs):
        if amsgrad:
            raise NotImplementedError('amsgrad=True is never used by the reference (util.py:65)')
        self.init_lr = float(learning_rate)
        self.learning_rate = float(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.n_epochs = float(n_epochs)
        self.iterations = 0
        self._state = {}

    def __call__(self, epoch):
        self.learning_rate = self.init_lr * ((1.0 - epoch / self.n_epochs) ** 0.9)

    def _lr_t(self):
        t = self.iterations
        return self.learning_rate * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def apply_gradients(self, grads_and_vars, model=None, grad_scale=1.0, skip_flag=None):
        """skip_flag (device int32[1], fused path only): the update is dropped on the device when it is non-zero; the caller
        takes `iterations` back by one when it learns of the skip (lowp_train.LowPrecisionTrainer.settle)"""
        gv = list(grads_and_vars)
        self.iterations += 1
        lr_t = self._lr_t()
        if model is not None and model.flat_params is not None and len(gv) == len(model.trainable_variables):
            key = id(model.flat_params)
            st = self._state.get(key)
            if st is None:
                st = (torch.zeros_like(model.flat_params), torch.zeros_like(model.flat_params))
                self._state[key] = st
            ops.adam_tf_step(model.flat_params, model.flat_grads, st[0], st[1], lr_t, self.beta_1, self.beta_2,
                             self.epsilon, grad_scale, skip=skip_flag)
        else:
            if skip_flag is not None:
                raise ValueError('skip_flag needs the fused update over a Model\'s flat buffers')
            for g, p in gv:
                if g is None:
                    continue
                st = self._state.get(id(p))
                if st is None:
                    st = (torch.zeros(p.t.numel() + 3 & ~3, device=p.t.device), torch.zeros(p.t.numel() + 3 & ~3, device=p.t.device))
                    self._state[id(p)] = st
                n = p.t.numel()
                pc, gc = p.t.reshape(-1).clone(), g.reshape(-1).clone()
                ops.adam_tf_step(pc, gc, st[0][:n], st[1][:n], lr_t, self.beta_1, self.beta_2, self.epsilon, grad_scale)
                p.t.copy_(pc.view(p.t.shape))
        bump_weights_epoch()


def reduce_sum(terms):
    """tf.reduce_sum(model.losses) for a list of 1-element Tensors"""
    tot = 0
    for t in terms:
        tot = tot + t
    return tot


def train_step(model, optimizer, loss_fn, dice_fn, x, y):
    """One iteration of the reference's training loop, train.py:140-152.  Returns (loss, macro_dice, micro_dice)."""
    fence = ops.step_fence('train')          # at most two steps in flight (see ops.step_fence)
    with GradientTape() as tape:
        y_pred, y_vae, z_mean, z_logvar = model(x, training=True, inference=False)     # :143
        loss = loss_fn(x, y, y_pred, y_vae, z_mean, z_logvar)                          # :145
        loss = loss + reduce_sum(model.losses)                                        # :146
    macro_dice, micro_dice = dice_fn(y, y_pred)                                        # :148
    sync = parallel.grad_sync(model)                 # data parallel: all-reduce finished gradient buckets during the backward
    grads = tape.gradient(loss, model.trainable_variables, grad_sync=sync)             # :151
    scale = 1.0 if sync is not None else parallel.all_reduce_gradients(model)          # C1 (no-op on one rank)
    optimizer.apply_gradients(zip(grads, model.trainable_variables), model=model, grad_scale=scale)  # :152
    ops.step_fence_done(fence)
    return loss, macro_dice, micro_dice
