"""How long does the oracle take at the BASELINE sizes on this host, and how much memory?  (Sizing aid for
tests/test_oracle_fullsize_gpu.py; imports oracle/ -- a measurement script, never product.)
usage: python scripts/oracle_fullsize_time.py [threads] [what...]   what in {train32, train64, infer32, infer64}"""
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import torch_ref as R  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
what = sys.argv[2:] or ['train32', 'infer32', 'infer64', 'train64']
torch.set_num_threads(threads)
cfg = R.default_config(base_filters=32, reduction=8)
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def train(dtype):
    x, y, mask, eps = R.synthetic_batch(1, (128, 128, 128), latent=128, seed=1234)
    P = R.build_params(cfg, (128, 128, 128), seed=0)
    for k in P:
        P[k] = P[k].to(dtype)
    t0 = time.time()
    loss, macro, micro, _, _ = R.train_step(P, cfg, x.to(dtype), y.to(dtype), mask.to(dtype), eps.to(dtype), {}, 1e-4, 1)
    return time.time() - t0, float(loss)


def infer(dtype):
    g = torch.Generator().manual_seed(4)
    x = torch.randn((1, 160, 192, 160, 2), generator=g)
    P = R.build_params(cfg, (128, 128, 128), seed=0)
    for k in P:
        P[k] = P[k].to(dtype)
    t0 = time.time()
    with torch.no_grad():
        yp = R.model(x.to(dtype), P, cfg, training=False, inference=True)[0]
    return time.time() - t0, float(yp.mean())


for w in what:
    dt = torch.float64 if w.endswith('64') else torch.float32
    t, v = (train if w.startswith('train') else infer)(dt)
    print('%s: %.1f s on %d threads (of %d cpus), value %.6f, peak RSS so far %.1f GB' % (w, t, threads, os.cpu_count(), v, rss()), flush=True)
