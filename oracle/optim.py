"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the SE-SSD parameter update.

One step over flat float32 vectors, restating in numpy what the reference composes from
  * gradient clipping   det3d/torchie/trainer/hooks/optimizer.py:50-53 -> torch.nn.utils.clip_grad_norm_(max_norm=35, L2)
  * decoupled decay     det3d/solver/fastai_optim.py:155-176  OptimWrapper.step(true_wd): p *= 1 - wd*lr, then opt.step()
  * Adam                torch.optim.Adam(betas=(mom, 0.99)) as built in det3d/torchie/apis/train_sessd.py:169-175
  * EMA teacher         det3d/torchie/trainer/trainer_sessd.py:315-318: alpha = min(1 - 1/(global_step+1), 0.999),
                        theta_T = alpha*theta_T + (1-alpha)*theta_S   (after the optimizer step, :355-357)
Pinned in tests/test_train_cpu.py against torch.optim.Adam / clip_grad_norm_ themselves (the reference does not pin a
torch version; the installed torch's single-tensor Adam is the arithmetic restated here)."""
import math

import numpy as np

F = np.float32


def clip_coef(grad, max_norm):
    """torch.nn.utils.clip_grad_norm_: coef = max_norm / (||g||_2 + 1e-6), clamped to 1. Returns (norm, coef)."""
    norm = math.sqrt(float(np.sum(grad.astype(np.float64) ** 2)))
    if max_norm is None or max_norm <= 0:
        return norm, 1.0
    return norm, min(1.0, max_norm / (norm + 1e-6))


def ema_alpha(global_step):
    return min(1.0 - 1.0 / (global_step + 1), 0.999)


def adam_true_wd_ema_step(p, g, m, v, teacher, lr, wd, beta1, beta2, eps, step, max_norm=None, alpha=None):
    """In-place on float32 arrays p, m, v, teacher (teacher may be None); `step` is 1 for the first update."""
    _, c = clip_coef(g, max_norm)
    g = (g * F(c)).astype(F) if c < 1.0 else g
    p *= F(1.0 - wd * lr)
    m += (g - m) * F(1.0 - beta1)                      # exp_avg.lerp_(grad, 1 - beta1)
    v *= F(beta2)
    v += F(1.0 - beta2) * g * g                        # addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = np.sqrt(v) / F(math.sqrt(bc2)) + F(eps)
    p += F(-(lr / bc1)) * (m / denom)                  # addcdiv_(exp_avg, denom, value=-step_size)
    if teacher is not None:
        teacher *= F(alpha)
        teacher += F(1.0 - alpha) * p
    return p
