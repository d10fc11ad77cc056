"""dindel-tgi_amd — MI355X-native read x candidate-haplotype HMM likelihood path of genome/dindel-tgi.

Product = csrc/ (HIP kernels + C ABI, include/dindel_hmm.h) and host/ (C++ mirror of
DetInDel::computeLikelihoods).  The Python modules here are the harness around the C ABI:
  capi    ctypes binding (raises if libdindel_hmm.so is not built — no CPU fallback)
  batch   numpy packing of windows into the ABI's flat batch
  device  torch-owned device buffers + stream for the device-pointer entry points
  synth   deterministic synthetic windows of the BASELINE shapes
"""
__all__ = ["capi", "batch", "device", "synth"]
