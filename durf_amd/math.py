"""Host-side scalar schedules and metrics (internal/math.py:49-56,156-219)."""
import math as _m


def mse_to_psnr(mse):
    """internal/math.py:49-51 (max pixel value 1)."""
    import torch
    if torch.is_tensor(mse):
        return -10. / _m.log(10.) * torch.log(mse)
    return -10. / _m.log(10.) * _m.log(mse)


def psnr_to_mse(psnr):
    """internal/math.py:54-56."""
    return _m.exp(-0.1 * _m.log(10.) * psnr)


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
    """internal/math.py:156-190: log-linear interpolation with a sine warm-up."""
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * _m.sin(
            0.5 * _m.pi * min(max(step / lr_delay_steps, 0), 1))
    else:
        delay_rate = 1.
    t = min(max(step / max_steps, 0), 1)
    return delay_rate * _m.exp(_m.log(lr_init) * (1 - t) + _m.log(lr_final) * t)


def freq_alpha_rate(step, alpha_init, alpha_final, alpha_delay_steps, alpha_max_steps):
    """internal/math.py:193-219: BARF coarse-to-fine alpha."""
    if step < alpha_delay_steps:
        return alpha_init
    if step < alpha_max_steps:
        return (step - alpha_delay_steps) / (alpha_max_steps - alpha_delay_steps) * alpha_final
    return alpha_final
