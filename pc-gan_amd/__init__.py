"""pcgan_amd -- MI355X-native PC-GAN (wsgan_emb) training path.

Sub-packages
  hip/      ctypes binding of libpcgan_hip.so + autograd wrappers + nn modules
  models/   the reference's plugin surface (define_G/define_D/define_E/define_IP, BaseModel,
            WSGANEmbModel) built on the HIP modules
  options/  the reference's option parser surface
  util/     small host utilities (get_attr_label, str2list, ...)
  csrc/     HIP sources of libpcgan_hip.so
"""
__version__ = '0.1.0'
