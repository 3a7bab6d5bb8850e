"""hgaprec_amd -- MI355X-native CAVI inner loop of hgaprec (HPF / BPF).

Only what the hot path needs lives here:
  csrc/            HIP kernels + the C-ABI (libhpf_hip.so) and the C++ host side
  capi.py          ctypes binding of include/hpf.h
  hostlib.py       ctypes binding of the C++ host side (reader, MT19937 init, writers)
  synth.py         synthetic sparse ratings (SURVEY.md section 8d generator)
  dist.py          user sharding + the one all-reduce (torch.distributed / RCCL)
The HIP library is mandatory: nothing here falls back to a CPU path.
"""
__all__ = ["capi"]
