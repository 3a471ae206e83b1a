"""sr-gan-fd_amd -- MI355X-native (gfx950) hot path of MiNeves00/SR-GAN-FD.

Drop-in for the reference's ``model.py`` surface (``sr_gan_fd_amd.model``): same class/factory
names, constructor kwargs and ``state_dict`` keys, with every convolution, resampling, loss and
optimizer step running as hand-written HIP kernels behind the C ABI in ``include/srganfd.h``.
"""
__version__ = "0.1.0"
