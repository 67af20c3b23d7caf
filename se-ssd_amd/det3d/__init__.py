"""Host-side mirror of the det3d surface that SE-SSD's inference hot path (and its config file) touches.

Same module paths, class names, constructor kwargs, forward signatures and state_dict layout as the reference
(Vegeta2020/SE-SSD, det3d 1.0.rc0), so `examples/second/configs/config.py` loads unchanged and the released
checkpoint keys map one-to-one -- but every tensor op on the hot path runs in libsessd_hip.so (gfx950 HIP).
Nothing here needs numba, spconv, apex, addict, torchvision or boost."""
__version__ = "1.0.rc0+sessd_hip"
short_version = "1.0.rc0"
