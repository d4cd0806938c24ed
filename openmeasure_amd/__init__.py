"""openmeasure_amd -- MI355X-native SPR hot path (fit / optimal_placement / train / predict / reconstruct).

Host classes in `sparse_sensing` (same surface as openmeasure.sparse_sensing.ROM / SPR), HIP kernels behind the
C ABI of include/spr_hip.h in `libspr_hip.so` (built by `make -C openmeasure_amd/csrc`), bound in `_lib`, driven by
`engine.HipEngine`.  Nothing is imported eagerly: importing the package needs neither torch nor a GPU."""

__all__ = ['sparse_sensing', 'engine', 'synth']
